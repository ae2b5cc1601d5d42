"""GPU: the switches of include/cw_policy.h are live on BOTH sides.  The engine and the oracle are built once more with a non-default
column vote (-DCW_POA_CONS_TIE=1 -DCW_POA_CONS_GAP=1: ties go to the smallest code instead of the template's base, a column whose gap
count equals its best base count is dropped), into a scratch directory; under that policy the HIP engine and the oracle still agree
window by window, and their consensus differs from the default policy's on the same piles -- so neither side ignores the switch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POLICY = ["-DCW_POA_CONS_TIE=1", "-DCW_POA_CONS_GAP=1"]

CHILD = r"""
import hashlib, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import consent_amd as ca
from consent_amd.engine import synth_host
import oracle_lib
prm = ca.Params(9, 4, 8, 2, 150)
h = hashlib.sha256()
same = True
for depth, n in ((150, 24), (30, 64), (12, 48)):
    hb = synth_host(ca.SynthSpec.pacbio(n, depth, first_window=900))
    eng = ca.Engine(prm)
    got = eng.run(hb)
    eng.close()
    exp, _ = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
    for w in range(n):
        same = same and got.consensus(w) == exp.consensus(w) and int(got.status[w]) == int(exp.status[w])
        h.update(got.consensus(w).encode())
print("RESULT", int(same), h.hexdigest())
"""


def run_child(env):
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    _, same, digest = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
    return same == "1", digest


@pytest.mark.timeout(1200)
def test_non_default_consensus_policy_is_honoured_by_engine_and_oracle(tmp_path):
    from consent_amd import _build

    alt_lib = str(tmp_path / "libconsent_amd_policy.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *POLICY, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(POLICY)])
    same_default, digest_default = run_child({})
    same_alt, digest_alt = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_alt            # engine == oracle under either policy
    assert digest_default != digest_alt         # and the policy changes the consensus
    # mixing the two sides must disagree somewhere: the switch is live on each side separately
    mixed1, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    mixed2, _ = run_child({"CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert not mixed1 and not mixed2


@pytest.mark.timeout(1200)
def test_heaviest_bundle_consensus_is_implemented_on_both_sides(tmp_path):
    """-DCW_POA_CONSENSUS=1 (cw_policy.h CW_POA_CONSENSUS_HEAVIEST_BUNDLE, round 4): edge weights kept by the merge, the consensus read back
    along the heaviest in-edges from the best-scoring sink -- in the oracle and in every POA tier of the engine (Q, S, M1, M2, L, G).  The two
    sides agree window by window under that policy, and the consensus is not the column vote's."""
    from consent_amd import _build

    hb = ["-DCW_POA_CONSENSUS=1"]
    alt_lib = str(tmp_path / "libconsent_amd_hb.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *hb, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(hb)])
    same_default, digest_default = run_child({})
    same_hb, digest_hb = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_hb
    assert digest_default != digest_hb


@pytest.mark.timeout(1500)
def test_overlap_alignment_mode_is_implemented_on_both_sides(tmp_path):
    """-DCW_POA_MODE=2 (cw_policy.h CW_POA_MODE_OV, round 4): column 0 of the DP free, the alignment ends in the best cell of a sink's row and
    stops in the first row or column, the bases outside it become insertions -- in the oracle (scalar and AVX2 fills) and in every tier
    and on both POA paths of the engine.  The two sides agree window by window
    under that mode, and the consensus is not the global mode's."""
    from consent_amd import _build

    ov = ["-DCW_POA_MODE=2"]
    alt_lib = str(tmp_path / "libconsent_amd_ov.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *ov, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(ov)])
    same_default, digest_default = run_child({})
    same_ov, digest_ov = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_ov
    assert digest_default != digest_ov
    mixed, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    assert not mixed


@pytest.mark.timeout(1500)
def test_other_alignment_scores_are_honoured_by_engine_and_oracle(tmp_path):
    """-DCW_POA_MATCH=2 -DCW_POA_MISMATCH=-4 -DCW_POA_GAP=-4 (cw_policy.h: the three scores are build-time values on both sides since round 5;
    every derived constant of the engine -- the recorded-decision fill's scaled scores, the packed fills' score pairs, tier Q's -- follows
    them).  The two sides agree window by window under those scores, the consensus is not the default scores', and mixing the sides disagrees."""
    from consent_amd import _build

    sc = ["-DCW_POA_MATCH=2", "-DCW_POA_MISMATCH=-4", "-DCW_POA_GAP=-4"]
    alt_lib = str(tmp_path / "libconsent_amd_sc.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *sc, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(sc)])
    same_default, digest_default = run_child({})
    same_sc, digest_sc = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_sc
    assert digest_default != digest_sc
    mixed, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    assert not mixed


@pytest.mark.timeout(1500)
def test_local_alignment_mode_is_implemented_on_both_sides(tmp_path):
    """-DCW_POA_MODE=1 (cw_policy.h CW_POA_MODE_SW, round 5): first row and column 0, no cell below 0, the alignment ends in the best cell anywhere and
    stops at a cell of value 0; the bases outside it become insertions -- in the oracle and on the engine's matrix paths (under this mode the engine is
    built without recorded decisions: every tier fills a DP matrix and walks its values).  The two sides agree window by window, the consensus is not
    the global mode's, and mixing the sides disagrees."""
    from consent_amd import _build

    sw = ["-DCW_POA_MODE=1"]
    alt_lib = str(tmp_path / "libconsent_amd_sw.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *sw, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(sw)])
    same_default, digest_default = run_child({})
    same_sw, digest_sw = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_sw
    assert digest_default != digest_sw
    mixed, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    assert not mixed


@pytest.mark.timeout(1800)
def test_affine_gap_model_is_implemented_on_both_sides(tmp_path):
    """-DCW_POA_GAP_MODEL=1 (cw_policy.h CW_POA_GAP_MODEL_AFFINE, round 5): a gap of length L costs open + (L - 1) ext -- Gotoh's three layers on the
    graph and a walk back that keeps its layer, in the oracle and in the engine (cw_poa_a.h: every task in the global-memory tier).  The two sides agree
    window by window, the consensus is not the linear model's, mixing the sides disagrees -- and with ext == open the affine build of the ENGINE gives
    the linear model's consensuses bit for bit (the two models coincide there, tie rules included)."""
    from consent_amd import _build

    aff = ["-DCW_POA_GAP_MODEL=1", "-DCW_POA_GAP_OPEN=-8", "-DCW_POA_GAP_EXT=-6"]
    same_as_linear = ["-DCW_POA_GAP_MODEL=1", "-DCW_POA_GAP_OPEN=-8", "-DCW_POA_GAP_EXT=-8"]
    alt_lib, lin_lib = str(tmp_path / "libconsent_amd_affine.so"), str(tmp_path / "libconsent_amd_affine_eq.so")
    p1 = subprocess.Popen([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *aff, *_build.SRC, "-o", alt_lib])
    p2 = subprocess.Popen([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *same_as_linear, *_build.SRC, "-o", lin_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(aff)])
    assert p1.wait() == 0 and p2.wait() == 0
    same_default, digest_default = run_child({})
    same_aff, digest_aff = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_aff
    assert digest_default != digest_aff
    mixed, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    assert not mixed
    same_eq, digest_eq = run_child({"CONSENT_AMD_LIB": lin_lib})  # the affine engine with ext == open against the DEFAULT (linear) oracle
    assert same_eq and digest_eq == digest_default


@pytest.mark.timeout(1500)
def test_chain_tie_largest_successor_is_implemented_on_both_sides(tmp_path):
    """-DCW_CHAIN_TIE=1 (cw_policy.h CW_CHAIN_TIE_LARGEST_SUCCESSOR, round 6): among successors of equal chain length and score the LAST one wins
    (`>=` in the oracle's scan; the successor's index itself instead of its complement in the engine's chain keys, the last lane instead of the first
    in the 32-bit window path).  The two sides agree window by window under it, the consensus differs from the default's on these piles (10 of the 136
    windows in the oracle), and one side alone under the policy disagrees with the other."""
    from consent_amd import _build

    ct = ["-DCW_CHAIN_TIE=1"]
    alt_lib = str(tmp_path / "libconsent_amd_ct.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *ct, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(ct)])
    same_default, digest_default = run_child({})
    same_ct, digest_ct = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_ct
    assert digest_default != digest_ct
    mixed, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    assert not mixed


@pytest.mark.timeout(1800)
def test_missing_anchor_extrapolation_is_implemented_on_both_sides(tmp_path):
    """-DCW_SEG_MISSING_ANCHOR=1 (cw_policy.h CW_SEG_MISSING_ANCHOR_EXTRAPOLATE, round 6): a chain anchor that a sequence lacks is placed where the
    template's spacing puts it, counted from the nearest chain anchor the sequence holds; the segments are then cut as ever.  Oracle: per-sequence
    position table; engine: the chain kernel fills the block's position matrix in place before it cuts the segments (and drops the
    equal-pieces shortcut, which rests on every member holding the left anchor).  The two sides agree window by window, every consensus differs from
    the default policy's (far more members per segment), and one side alone under the policy disagrees with the other."""
    from consent_amd import _build

    ex = ["-DCW_SEG_MISSING_ANCHOR=1"]
    alt_lib = str(tmp_path / "libconsent_amd_ex.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *ex, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(ex)])
    same_default, digest_default = run_child({})
    same_ex, digest_ex = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_ex
    assert digest_default != digest_ex
    mixed, _ = run_child({"CONSENT_AMD_LIB": alt_lib})
    assert not mixed


@pytest.mark.timeout(1500)
def test_alignment_scores_up_to_sixteen(tmp_path):
    """-DCW_POA_MATCH=16 -DCW_POA_MISMATCH=-12 -DCW_POA_GAP=-16 (round 6: the legal range is |score| <= 16, <= 8 through round 5; cw_policy.h "Bounds").
    Every int16 tier keeps its values inside +-29000 by its capacities except tier L, which under such scores hands graphs of more than
    29000 / 16 - 1023 = 789 nodes on to the int32 tier G -- the deep piles of this test have such graphs.  The two sides agree window by window, and
    the consensus is not the default scores'."""
    from consent_amd import _build

    sc = ["-DCW_POA_MATCH=16", "-DCW_POA_MISMATCH=-12", "-DCW_POA_GAP=-16"]
    alt_lib = str(tmp_path / "libconsent_amd_s16.so")
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *sc, *_build.SRC, "-o", alt_lib])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "policy", f"OUT={tmp_path}", "POLICY=" + " ".join(sc)])
    same_default, digest_default = run_child({})
    same_sc, digest_sc = run_child({"CONSENT_AMD_LIB": alt_lib, "CW_ORACLE_LIB": str(tmp_path / "liboracle.so")})
    assert same_default and same_sc
    assert digest_default != digest_sc
