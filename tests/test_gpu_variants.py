"""GPU: build variants of the engine that must all be the same function.  The library once more with
  -DCW_NO_PAD64     rows of <= 64 columns stored under an execution mask (the default lets lanes beyond the columns write into cells of
                    later rows, which rests on same-wave stores to one address committing in issue order: ADVICE r03),
  -DCW_POA_CODES=0  tiers S / M1 on the matrix path (fill writes the DP matrix, the traceback reads tiles of it back) instead of the
                    recorded-decision path of cw_poa_c.h,
  -DCW_M2_CODES=1   tier M2 on the recorded-decision path too,
  -DCW_POA_LW=1     tier LW (round 6, cw_poa_w.h): the tier-L tasks whose members are wide on average on the four waves of a work-group, the chunks of a wide
                    packed row pipelined by rows (measured: a batch alone on the GPU 7 % faster, the step with batches in flight 1-2 % slower: off by default),
  ... -DCW_POALW_MIN_MEAN=1 -DCW_POALW_MIN_MEMBERS=2   EVERY tier-L task in tier LW (narrow members too: the one-chunk rows stay on wave 0),
  -DCW_POA_GROUP_FILL=0  tiers M2 / L fill every member on its own (round 5's default fills up to four consecutive short members together),
  -DCW_POA_VPROBE=0 the tile traceback without the look down the column inside a long vertical run,
  -DCW_POAQ_REPLAY=0 -DCW_POA_REPLAY=0  every tier aligns every member (round 6's default does not align a member that repeats the one before it: cw_poa_q.h),
  -DCW_Q_CODES=0    tier Q with the DP matrix in LDS and a walk over its values (rounds 3-4, cw_poa_q0.h) instead of recorded decisions,
each compared with the oracle on piles of several depths (all tiers)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"nopad": ["-DCW_NO_PAD64"], "matrix": ["-DCW_POA_CODES=0"], "m2codes": ["-DCW_M2_CODES=1"], "qmatrix": ["-DCW_Q_CODES=0"], "noreplay": ["-DCW_POAQ_REPLAY=0", "-DCW_POA_REPLAY=0"], "qflat": ["-DCW_POAQ_FLAT=1"], "lw": ["-DCW_POA_LW=1"], "lwall": ["-DCW_POA_LW=1", "-DCW_POALW_MIN_MEAN=1", "-DCW_POALW_MIN_MEMBERS=2"], "nogroup": ["-DCW_POA_GROUP_FILL=0", "-DCW_POA_VPROBE=0"]}

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import consent_amd as ca
from consent_amd.engine import synth_host
import oracle_lib
same = True
for depth, n, msa, wlen in ((150, 32, 150, 500), (30, 96, 20, 500), (60, 32, 150, 900), (12, 64, 150, 300)):
    prm = ca.Params(9, 4, 8, 2, msa)
    spec = ca.SynthSpec.pacbio(n, depth, first_window=1200 + depth)
    hb = synth_host(spec)
    eng = ca.Engine(prm)
    got = eng.run(hb)
    eng.close()
    exp, _ = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
    for w in range(n):
        same = same and got.consensus(w) == exp.consensus(w) and int(got.status[w]) == int(exp.status[w])
# few anchors -> segments of hundreds of bases: the wide packed rows of tiers M2 / L (what tier LW spreads over waves)
import random
rng = random.Random(5)
def mutate(s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate * 0.3:
            continue
        if x < rate * 0.6:
            out.append(rng.choice("ACGT"))
        out.append(rng.choice("ACGT") if x < rate else c)
    return "".join(out)
piles = []
for depth, ln, rate in ((8, 900, 0.2), (14, 700, 0.22), (24, 600, 0.2), (40, 520, 0.18)):
    truth = "".join(rng.choice("ACGT") for _ in range(ln))
    piles.append([mutate(truth, rate) for _ in range(depth)])
prm = ca.Params(9, 4, 8, 2, 150)
hb = ca.pack_piles(piles)
eng = ca.Engine(prm)
got = eng.run(hb)
eng.close()
exp, _ = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
for w in range(len(piles)):
    same = same and got.consensus(w) == exp.consensus(w) and int(got.status[w]) == int(exp.status[w])
print("RESULT", int(same))
"""


@pytest.mark.timeout(1500)
def test_build_variants_agree_with_the_oracle(tmp_path):
    from consent_amd import _build

    procs = {}
    for name, flags in VARIANTS.items():
        lib = str(tmp_path / f"libconsent_amd_{name}.so")
        procs[name] = (lib, subprocess.Popen([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *flags, *_build.SRC, "-o", lib]))
    for name, (lib, pr) in procs.items():
        assert pr.wait() == 0, name
    for name, (lib, _) in procs.items():
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=dict(os.environ, CONSENT_AMD_LIB=lib), timeout=900)
        assert out.returncode == 0, (name, out.stderr[-1500:])
        assert [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1] == "RESULT 1", name
