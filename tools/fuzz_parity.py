"""Randomised differential test on the GPU box: engine vs oracle over random parameters, error profiles, depths and window
lengths.  Prints one line per configuration and a summary; exit code 1 on any difference."""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import consent_amd as ca  # noqa: E402
from consent_amd.engine import synth_host  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t_end = time.time() + seconds
    n_cfg = n_win_total = bad = 0
    while time.time() < t_end:
        k = rng.choice([5, 6, 7, 8, 9, 9, 9, 10, 11, 13])
        solid = rng.choice([1, 2, 3, 4, 4, 6])
        common = rng.choice([2, 4, 8, 8, 12])
        min_anchors = rng.choice([1, 2, 2, 5, 10])
        depth = rng.choice([1, 2, 4, 8, 16, 30, 30, 60, 100, 150])
        max_msa = rng.choice([2, 5, 20, 20, 50, 150])
        wlen = rng.choice([60, 150, 300, 500, 500, 500, 700, 900])
        err = rng.choice([0, 10, 50, 120, 120, 150, 200, 300])
        mix = rng.choice([(10, 60, 30), (30, 30, 40), (100, 0, 0), (0, 100, 0), (0, 0, 100), (34, 33, 33)])
        nw = max(2, min(48, 20000 // ((depth + 1) * max(wlen, 100) // 100)))
        spec = ca.SynthSpec(rng.getrandbits(40), rng.getrandbits(20), nw, depth, wlen, err, mix[0], mix[1], mix[2], (wlen + 60 + wlen // 3) // 16 + 2)
        prm = ca.Params(k, solid, common, min_anchors, max_msa)
        batch = synth_host(spec)
        eng = ca.Engine(prm)
        try:
            got = eng.run(batch)
        finally:
            eng.close()
        exp, _ = oracle_lib.oracle_run(prm, batch, threads=16)
        diff = 0
        for w in range(nw):
            if got.status[w] == ca.WIN_OVERFLOW:
                continue  # a documented capacity, not a wrong answer
            if got.status[w] != exp.status[w] or got.consensus(w) != exp.consensus(w) or not np.array_equal(got.solid_kmers(w), exp.solid_kmers(w)):
                diff += 1
        n_over = int((got.status == ca.WIN_OVERFLOW).sum())
        n_cfg += 1
        n_win_total += nw
        bad += diff
        print(f"k={k} solid={solid} c={common} A={min_anchors} depth={depth} M={max_msa} len={wlen} err={err} mix={mix} windows={nw} overflow={n_over} "
              f"template={int((got.status == ca.WIN_TEMPLATE).sum())} DIFF={diff}", flush=True)
    print(f"{n_cfg} configurations, {n_win_total} windows, {bad} differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
