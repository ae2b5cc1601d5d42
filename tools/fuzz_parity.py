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


WHY = {1: "SETUP", 2: "COUNT", 3: "SOLIDCAP", 4: "TEMPLATE", 5: "MATRIX", 6: "SEGMENTS", 7: "TASKS", 8: "POA", 9: "FIN_LEN", 10: "FIN_SOLID", 11: "FIN_POLISH",
       12: "OUT_CONS", 13: "OUT_SOLID", 14: "ARENA", 15: "ANCHORS"}  # cw_device.h CW_WHY_*
MAX_OVERFLOW_SHARE = float(os.environ.get("CW_FUZZ_CAP_BAR", "0.001"))  # a capacity stop is never a wrong answer, but more than one window in a thousand is a regression
# (CW_FUZZ_CAP_BAR: tools/fuzz_policy.sh raises the bar for the heaviest-bundle build, whose consensuses are longer on chance-anchor piles)


def low_complexity(batch, rng, kind):
    """Rewrites the piles of a synthetic batch in place into a family the random generator never draws: "homopolymer" (runs of one base),
    "tandem" (a short unit repeated with 3 % noise: every k-mer of the template repeats, so few anchors survive), "identical" (every sequence
    a copy of the template: every template k-mer is an anchor and the position matrix is at its largest)."""
    nrng = np.random.default_rng(rng.getrandbits(32))
    wfs, slen, off, bases = batch.win_first_seq, batch.seq_len, batch.seq_word_off, batch.bases
    shifts = (30 - 2 * np.arange(16, dtype=np.uint32))
    for w in range(batch.n_windows):
        s0, s1 = int(wfs[w]), int(wfs[w + 1])
        unit = nrng.integers(0, 4, rng.choice([1, 2, 3, 5, 7, 11]))
        tpl = nrng.integers(0, 4, int(slen[s0:s1].max()) + 64)
        for s in range(s0, s1):
            n = int(slen[s])
            if kind == "homopolymer":
                runs = nrng.integers(3, 40, n // 3 + 2)
                codes = np.repeat((np.cumsum(nrng.integers(1, 4, len(runs))) & 3), runs)[:n]
            elif kind == "tandem":
                codes = unit[(np.arange(n) + (s - s0)) % len(unit)]
                noise = nrng.random(n) < 0.03
                codes = np.where(noise, nrng.integers(0, 4, n), codes)
            else:
                codes = tpl[:n]
            pad = np.zeros((n + 15) // 16 * 16, np.uint32)
            pad[:n] = codes
            o = int(off[s])
            bases[o : o + len(pad) // 16] = (pad.reshape(-1, 16) << shifts[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t_end = time.time() + seconds
    n_cfg = n_win_total = bad = n_over_total = n_win_short_k = n_over_short_k = 0
    why_hist = {}
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
        s_seed, s_first = rng.getrandbits(40), rng.getrandbits(20)
        spec = ca.SynthSpec(s_seed, s_first, nw, depth, wlen, err, mix[0], mix[1], mix[2], (wlen + 60 + wlen // 3) // 16 + 2)
        prm = ca.Params(k, solid, common, min_anchors, max_msa)
        batch = synth_host(spec)
        family = rng.choice(["random"] * 8 + ["homopolymer", "tandem", "identical"])
        if family != "random":
            low_complexity(batch, rng, family)
        eng = ca.Engine(prm)
        try:
            got = eng.run(batch)
            info = eng.win_info(nw)
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
        why_cfg = {}
        for w in range(nw):
            if got.status[w] == ca.WIN_OVERFLOW:
                why_cfg[WHY.get(int(info[w, 15]), "?")] = why_cfg.get(WHY.get(int(info[w, 15]), "?"), 0) + 1
                why_hist[WHY.get(int(info[w, 15]), str(int(info[w, 15])))] = why_hist.get(WHY.get(int(info[w, 15]), str(int(info[w, 15]))), 0) + 1
        n_cfg += 1
        if k < 8:  # with k-mers this short most anchors are chance hits and the segmented consensus comes out several times its template (until round 5
            n_win_short_k += nw  # 5 % of these windows stopped on the finish kernel's 3072-character strings; its second pass holds 32768): counted apart,
            n_over_short_k += n_over  # held to the same bar
        else:
            n_win_total += nw
            n_over_total += n_over
        bad += diff
        print(f"k={k} solid={solid} c={common} A={min_anchors} depth={depth} M={max_msa} len={wlen} err={err} mix={mix} family={family} windows={nw} overflow={n_over} "
              f"template={int((got.status == ca.WIN_TEMPLATE).sum())} DIFF={diff}" + (f" why={why_cfg} spec=({s_seed},{s_first})" if why_cfg else ""), flush=True)
    share = n_over_total / max(1, n_win_total)
    print(f"{n_cfg} configurations, {n_win_total + n_win_short_k} windows, {bad} differences; k >= 8: {n_over_total} of {n_win_total} windows stopped by a capacity ({share:.5f}); "
          f"k < 8: {n_over_short_k} of {n_win_short_k}; by reason: {why_hist}")
    share_short = n_over_short_k / max(1, n_win_short_k)
    if share > MAX_OVERFLOW_SHARE or share_short > float(os.environ.get("CW_FUZZ_CAP_BAR_SHORT_K", "0.002")):
        print(f"FAILED: capacity stops above {MAX_OVERFLOW_SHARE} of the windows (k >= 8: {share:.5f}; k < 8: {share_short:.5f})")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
