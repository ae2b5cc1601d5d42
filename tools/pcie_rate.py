"""PCIe-inclusive throughput of the host-buffer entry point (cw_run: H2D of the packed piles, kernels, D2H of the results).
Not the headline metric (bench.py times device-resident batches); recorded in DESIGN.md section 6."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import consent_amd as ca  # noqa: E402
from consent_amd.engine import synth_host  # noqa: E402

for depth, msa, n in ((30, 20, 16384), (150, 150, 4096)):
    hb = synth_host(ca.SynthSpec.pacbio(n, depth))
    eng = ca.Engine(ca.Params(9, 4, 8, 2, msa))
    eng.run(hb)  # warm-up (allocations)
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eng.run(hb)
    dt = (time.perf_counter() - t) / reps
    mb = (hb.bases.nbytes + hb.seq_len.nbytes + hb.seq_word_off.nbytes) / 1e6
    print(f"depth {depth}: {n / dt:.0f} windows/s through cw_run (host buffers, {mb:.0f} MB in per batch, {dt * 1e3:.1f} ms/batch)")
    eng.close()
