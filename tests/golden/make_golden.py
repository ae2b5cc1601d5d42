#!/usr/bin/env python3
"""Generate the committed golden vectors.  Run in the authoring container only (needs /root/reference):

  windows_piles.json   inputs + outputs of the REFERENCE's own getAlignmentWindowsPositions /
                       getAlignmentWindowsSequences (oracle/_ref = the reference TUs compiled unmodified).
  consensus_small.json regression vectors of the restatement (oracle/liboracle.so) for the rows whose reference
                       sources are absent or unbuildable (A3-A10): NOT reference outputs -- they pin the oracle
                       against silent drift and give the GPU tests fixed expectations.
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import consent_amd as ca  # noqa: E402
from consent_amd.engine import synth_host  # noqa: E402
import oracle_lib  # noqa: E402
from test_oracle_ref import rand_overlaps, rand_seq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def windows_piles():
    r = oracle_lib.ref()
    assert r is not None, "build oracle/_ref first (make -C oracle)"
    cases = []
    for seed in range(6):
        rng = random.Random(9000 + seed)
        tpl_len = rng.choice([700, 1300, 2100])
        tpl = rand_seq(rng, tpl_len)
        ovls, targets = rand_overlaps(rng, tpl_len, rng.randrange(3, 12))
        for t in targets:
            assert len(t) < 3000
        wins = oracle_lib.window_positions(r.ref_window_positions, tpl_len, ovls, 2, 500, 50)
        piles = [oracle_lib.window_pile(r.ref_window_pile, ovls, tpl, targets, qb, qe, 9) for (qb, qe) in wins]
        cases.append({"tpl": tpl, "targets": targets, "overlaps": ovls, "min_support": 2, "window_size": 500, "window_overlap": 50, "k": 9, "windows": wins, "piles": piles})
    json.dump({"source": "reference alignmentWindows.cpp via oracle/_ref", "cases": cases}, open(os.path.join(HERE, "windows_piles.json"), "w"))


def mutate(rng, s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate * 0.3:
            continue
        if x < rate * 0.6:
            out.append(rng.choice("ACGT"))
        if x < rate:
            out.append(rng.choice("ACGT"))
        else:
            out.append(c)
    return "".join(out)


def consensus_small():
    rng = random.Random(4242)
    cases = []

    def add(name, pile, prm):
        hb = ca.pack_piles([pile])
        res, _ = oracle_lib.oracle_run(ca.Params(*prm), hb)
        cases.append({"name": name, "params": list(prm), "pile": pile, "status": int(res.status[0]), "consensus": res.consensus(0), "solid": [int(x) for x in res.solid_kmers(0)]})

    truth = rand_seq(rng, 180)
    add("noisy_depth12_k9", [mutate(rng, truth, 0.08) for _ in range(13)], (9, 4, 8, 2, 20))
    add("noisy_depth12_k7_solid2", [mutate(rng, truth, 0.10) for _ in range(13)], (7, 2, 4, 2, 20))
    add("clean_identical", [truth[:150]] * 9, (9, 4, 8, 2, 20))
    add("single_sequence", [truth[:120]], (9, 4, 8, 2, 20))
    add("too_divergent_template_fallback", [rand_seq(rng, 150) for _ in range(8)], (9, 4, 8, 2, 20))
    add("min_anchors_too_high", [mutate(rng, truth, 0.05) for _ in range(10)], (9, 4, 8, 500, 20))
    add("maxmsa_3", [mutate(rng, truth, 0.10) for _ in range(12)], (9, 4, 8, 2, 3))
    add("ragged_short_members", [mutate(rng, truth, 0.06)] + [mutate(rng, truth[a : a + n], 0.06) for a, n in ((0, 60), (30, 25), (100, 80), (5, 9), (20, 8), (0, 170), (90, 90), (10, 150))], (9, 4, 8, 2, 20))
    add("poly_a_low_complexity", ["A" * 120] * 6 + ["A" * 60 + "C" + "A" * 59] * 3, (9, 4, 8, 2, 20))
    add("weak_middle_polish", [truth[:160]] * 7 + [truth[:70] + rand_seq(rng, 12) + truth[82:160]] * 1 + [mutate(rng, truth[:160], 0.2) for _ in range(4)], (9, 4, 8, 2, 20))
    # synthetic generator windows (same generator as bench.py)
    hb = synth_host(ca.SynthSpec.pacbio(3, 12))
    for w in range(3):
        add(f"synth_pacbio_d12_w{w}", hb.pile(w), (9, 4, 8, 2, 20))
    json.dump({"source": "oracle/liboracle.so (restatement regression vectors; reference parity unpinned)", "cases": cases}, open(os.path.join(HERE, "consensus_small.json"), "w"))


if __name__ == "__main__":
    windows_piles()
    consensus_small()
    print("golden vectors written to", HERE)
