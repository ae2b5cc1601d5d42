"""GPU: build variants of the engine that must all be the same function.  The library once more with
  -DCW_NO_PAD64     rows of <= 64 columns stored under an execution mask (the default lets lanes beyond the columns write into cells of
                    later rows, which rests on same-wave stores to one address committing in issue order: ADVICE r03),
  -DCW_POA_CODES=0  tiers S / M1 on the matrix path (fill writes the DP matrix, the traceback reads tiles of it back) instead of the
                    recorded-decision path of cw_poa_c.h,
  -DCW_M2_CODES=1   tier M2 on the recorded-decision path too,
each compared with the oracle on piles of several depths (all tiers)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"nopad": ["-DCW_NO_PAD64"], "matrix": ["-DCW_POA_CODES=0"], "m2codes": ["-DCW_M2_CODES=1"]}

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
