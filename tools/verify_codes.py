"""Runs synthetic piles of several depths through a -DCW_POA_VERIFY build (tools/verify_codes.sh) and prints how many members the two
traceback paths were compared on and how many differed; also compares the consensus with the oracle.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import consent_amd as ca
from consent_amd.engine import synth_host
import oracle_lib

bad = 0
for depth, n, msa in ((150, 96, 150), (30, 256, 20), (60, 128, 150), (12, 256, 150), (300, 24, 150)):
    prm = ca.Params(9, 4, 8, 2, msa)
    hb = synth_host(ca.SynthSpec.pacbio(n, depth, first_window=4000 + depth))
    eng = ca.Engine(prm)
    got = eng.run(hb)
    ctr, prof = eng.profile()
    exp, _ = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
    diff = sum(1 for w in range(n) if got.consensus(w) != exp.consensus(w) or int(got.status[w]) != int(exp.status[w]))
    print(f"depth {depth} windows {n}: members compared {int(prof[120])}, differing {int(prof[121])}, windows differing from the oracle {diff}, overflow {int((got.status == ca.WIN_OVERFLOW).sum())}")
    if prof[121]:
        print("  first difference: window", int(prof[122]) >> 32, "member", int(prof[122]) & 0xFFFFFFFF, "cols", int(prof[123]) >> 32, "rows", int(prof[123]) & 0xFFFFFFFF,
              "end row matrix/coded", int(prof[124]) >> 32, int(prof[124]) & 0xFFFFFFFF, "columns differing (first 64) %016x" % int(prof[125]))
    bad += int(prof[121]) + diff
    eng.close()
print("VERIFY", "ok" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
