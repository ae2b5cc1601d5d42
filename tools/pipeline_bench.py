"""Time the end-to-end correction loop (consent_amd.pipeline) on a synthetic long-read data set: 30x coverage of a random genome,
12 % errors (PacBio-like mix), overlaps by construction.  Prints per-stage device time.  GPU box only."""
import io
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import consent_amd as ca  # noqa: E402
from consent_amd import pipeline  # noqa: E402

COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def noisy(rng, seg, rate):
    """numpy version of the test generator: returns the noisy copy and the position map (len(seg)+1 entries)"""
    n = len(seg)
    x = rng.random(n)
    dele = x < rate * 0.3
    ins = (x >= rate * 0.3) & (x < rate * 0.6)
    sub = x < rate
    base = np.where(sub, rng.integers(0, 4, n), seg)
    emit = (~dele).astype(np.int64) + ins.astype(np.int64)
    pos = np.concatenate([[0], np.cumsum(emit)])
    out = np.zeros(pos[-1], np.uint8)
    keep = ~dele
    out[pos[:-1][keep] + ins[keep]] = base[keep]
    out[pos[:-1][ins]] = rng.integers(0, 4, int(ins.sum()))
    return out, pos


def main():
    glen = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rlen = 8000
    rng = np.random.default_rng(7)
    genome = rng.integers(0, 4, glen)
    n_reads = glen * cov // rlen
    reads = []
    for i in range(n_reads):
        ln = int(rng.integers(rlen // 2, rlen * 3 // 2))
        g0 = int(rng.integers(0, glen - ln))
        s, pos = noisy(rng, genome[g0 : g0 + ln], 0.12)
        rev = bool(rng.random() < 0.5)
        reads.append((g0, g0 + ln, pos, rev, s))
    d = tempfile.mkdtemp()
    fa, paf = os.path.join(d, "reads.fa"), os.path.join(d, "ovl.paf")
    lut = np.frombuffer(b"ACGT", np.uint8)
    with open(fa, "w") as f:
        for i, (g0, g1, pos, rev, s) in enumerate(reads):
            a = lut[s]
            if rev:
                a = COMP[a[::-1]]
            f.write(f">r{i}\n{a.tobytes().decode()}\n")
    order = np.argsort([r[0] for r in reads])
    n_lines = 0
    with open(paf, "w") as f:
        for qi in range(n_reads):
            q = reads[qi]
            for ti in order:
                t = reads[ti]
                if ti == qi or t[1] <= q[0] + 500:
                    continue
                if t[0] >= q[1] - 500:
                    break
                a, b = max(q[0], t[0]), min(q[1], t[1])
                if b - a < 500:
                    continue

                def span(r, a, b):
                    s, e = int(r[2][a - r[0]]), int(r[2][b - r[0]])
                    if r[3]:
                        n = len(r[4])
                        s, e = n - e, n - s
                    return s, max(e, s + 1)

                qs, qe = span(q, a, b)
                ts, te = span(t, a, b)
                f.write(f"r{qi}\t{len(q[4])}\t{qs}\t{qe}\t{'+' if q[3] == t[3] else '-'}\tr{ti}\t{len(t[4])}\t{ts}\t{te}\t{int((b - a) * 0.76)}\t{b - a}\t60\n")
                n_lines += 1
    print(f"data set: genome {glen}, {n_reads} reads, {n_lines} overlaps", flush=True)

    # the native driver (cw_run_correction) with its own per-stage clocks
    import json
    import subprocess

    exe = os.path.join(ROOT, "bin", "CONSENT-correction")
    argv = [exe, "-a", paf, "-s", "3", "-S", "150", "-l", "500", "-k", "9", "-c", "8", "-A", "2", "-f", "4", "-m", "50", "-j", os.environ.get("CW_BENCH_GPUS", "1"), "-r", fa, "-M", "150", "-p", "x"]
    for rep in range(2):
        t0 = time.perf_counter()
        out = subprocess.run(argv, capture_output=True, text=True, env=dict(os.environ, CW_DRIVER_STATS="1", CW_DRIVER_TIMING="1"))
        wall = time.perf_counter() - t0
        assert out.returncode == 0, out.stderr[-2000:]
    st = json.loads([l for l in out.stderr.splitlines() if l.startswith("{")][-1])
    lines = out.stdout.split("\n")
    seqs = lines[1::2]
    bases = sum(len(x) for x in seqs)
    up = sum(sum(c.isupper() for c in x) for x in seqs)
    print(f"corrected {len(seqs)} reads, {bases} bases ({up / max(bases, 1):.3f} upper case) in {wall:.2f} s wall (process start to exit)")
    print(json.dumps(st))


if __name__ == "__main__":
    main()
