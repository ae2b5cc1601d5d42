"""CPU: hand-checkable known answers for the pieces of the oracle (A4d POA, A5 weightConsensus, A6-A10 polish)."""
import random

import consent_amd as ca
import oracle_lib


def kmer_counts(seqs, k):
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    out = {}
    for s in seqs:
        for i in range(len(s) - k + 1):
            v = 0
            for c in s[i : i + k]:
                v = (v << 2) | code[c]
            out[v] = out.get(v, 0) + 1
    return out


def test_poa_identical_strings_give_the_string():
    assert oracle_lib.oracle_poa(["ACGTTGCA"] * 5) == "ACGTTGCA"
    assert oracle_lib.oracle_poa(["A"]) == "A"


def test_poa_majority_substitution_and_minority_insertion():
    # 3 x ACGT vs 1 x AGGT: column 2 majority C
    assert oracle_lib.oracle_poa(["ACGT", "ACGT", "AGGT", "ACGT"]) == "ACGT"
    # a minority insertion column (1 of 4) is dropped; a majority one (3 of 4) is kept
    assert oracle_lib.oracle_poa(["ACGT", "ACTGT", "ACGT", "ACGT"]) == "ACGT"
    assert oracle_lib.oracle_poa(["ACGT", "ACTGT", "ACTGT", "ACTGT"]) == "ACTGT"
    # deletion carried by the majority removes the column
    assert oracle_lib.oracle_poa(["ACGT", "AGT", "AGT", "AGT"]) == "AGT"


def test_poa_tie_prefers_the_template_base():
    # 2 vs 2 at the third column: the template (first sequence) has G
    assert oracle_lib.oracle_poa(["ACGT", "ACGT", "ACTT", "ACTT"]) == "ACGT"
    assert oracle_lib.oracle_poa(["ACTT", "ACTT", "ACGT", "ACGT"]) == "ACTT"


def test_poa_long_outlier_does_not_hijack_the_consensus():
    rng = random.Random(1)
    core = "ACGTAC"
    junk = "".join(rng.choice("ACGT") for _ in range(120))
    assert oracle_lib.oracle_poa([core, core, junk, core, core]) == core


def test_weight_consensus_closed_form():
    """case[p] = solid(k-mer starting at min(p, L-k)) -- the later k-mer overwrites (correctionMSA.cpp:15-24)."""
    rng = random.Random(3)
    k, solid = 5, 3
    for _ in range(40):
        L = rng.randrange(k, 40)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        counts = {v: rng.randrange(0, 6) for v in kmer_counts([s], k)}
        out = oracle_lib.oracle_weight_polish(s, counts, k, solid, weight=True, polish=False)
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        for p in range(L):
            q = min(p, L - k)
            v = 0
            for c in s[q : q + k]:
                v = (v << 2) | code[c]
            assert out[p] == (s[p] if counts.get(v, 0) >= solid else s[p].lower())


def test_polish_repairs_a_weak_hole():
    """A lower-case hole with one substitution, bordered by solid sequence, is re-linked through the k-mer graph."""
    rng = random.Random(5)
    k, solid = 9, 4
    truth = "".join(rng.choice("ACGT") for _ in range(70))
    counts = {v: 10 for v in kmer_counts([truth], k)}
    bad = truth[:30] + ("A" if truth[30] != "A" else "C") + truth[31:]
    weighted = oracle_lib.oracle_weight_polish(bad, counts, k, solid, weight=True, polish=False)
    assert weighted != weighted.upper()  # the k-mers over the error are weak
    fixed = oracle_lib.oracle_weight_polish(bad, counts, k, solid, weight=True, polish=True)
    assert fixed == truth


def test_polish_extends_weak_head_and_tail():
    rng = random.Random(6)
    k, solid = 9, 4
    truth = "".join(rng.choice("ACGT") for _ in range(60))
    counts = {v: 10 for v in kmer_counts([truth], k)}
    s = truth[:3].lower() + truth[3:57] + truth[57:].lower()
    out = oracle_lib.oracle_weight_polish(s, counts, k, solid, weight=False, polish=True)
    assert out == truth


def test_polish_leaves_all_weak_and_all_solid_strings_alone():
    k, solid = 9, 4
    s = "ACGTTGCATGCATGCAAGTC"
    assert oracle_lib.oracle_weight_polish(s.lower(), {}, k, solid, weight=False, polish=True) == s.lower()
    counts = {v: 9 for v in kmer_counts([s], k)}
    assert oracle_lib.oracle_weight_polish(s, counts, k, solid, weight=True, polish=True) == s


def test_window_consensus_template_fallback_and_single_sequence():
    prm = ca.Params(9, 4, 8, 2, 20)
    rng = random.Random(9)
    tpl = "".join(rng.choice("ACGT") for _ in range(100))
    others = ["".join(rng.choice("ACGT") for _ in range(100)) for _ in range(5)]
    res, _ = oracle_lib.oracle_run(prm, ca.pack_piles([[tpl] + others]))
    assert res.status[0] == ca.WIN_TEMPLATE and res.consensus(0) == tpl  # correctionMSA.cpp:34-36
    res, _ = oracle_lib.oracle_run(prm, ca.pack_piles([[tpl]]))
    assert res.status[0] == ca.WIN_CONSENSUS and res.consensus(0) == tpl.lower()  # every k-mer count is 1 < solid


def test_window_positions_refuses_the_parameters_the_reference_hangs_on():
    """windowOverlap >= windowSize never advances (alignmentWindows.cpp:40-47) and a negative one indexes out of bounds: the library's
    host function returns CW_E_INVALID instead."""
    import ctypes as C

    import numpy as np

    import consent_amd as ca

    lib = ca.load_library()
    ov = np.array([[0, 999, 1, 0, 999, 0]], np.uint32)
    out = np.zeros(64, np.uint32)
    n = C.c_uint32()
    for ws, wo in ((0, 0), (500, 500), (500, 600), (500, -1)):
        assert lib.cw_window_positions(1000, ov.ctypes.data, 1, 1, ws, wo, out.ctypes.data, 32, C.byref(n)) == -1
    assert lib.cw_window_positions(1000, ov.ctypes.data, 1, 1, 500, 50, out.ctypes.data, 32, C.byref(n)) == 0 and n.value == 3


def test_the_baseline_build_of_the_oracle_gives_the_checkers_results(tmp_path):
    """bench.py's cpu_baseline leg times a second build of the restatement (oracle/Makefile simd_pair / native_simd: -DCWO_FAST -DCWO_SIMD --
    direct-addressed k-mer tables, the chain scan's early stop, AVX2 row-vectorised POA fill).  It must be the same function: status,
    consensus and solid set of every window, over depths, k, thresholds and low-complexity piles."""
    import hashlib
    import os
    import subprocess
    import sys

    import pytest

    if "avx2" not in open("/proc/cpuinfo").read():
        pytest.skip("no AVX2 on this host")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "simd_pair", f"OUT={tmp_path}"])
    child = r"""
import hashlib, os, random, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import numpy as np
import consent_amd as ca
from consent_amd.engine import synth_host
import oracle_lib, fuzz_parity
h = hashlib.sha256()
rng = random.Random(5)
for depth, n, msa, k, solid, common, fam in ((150, 6, 150, 9, 4, 8, None), (30, 24, 20, 9, 4, 8, None), (12, 24, 150, 7, 2, 4, None), (60, 8, 50, 11, 3, 12, None),
                                              (40, 8, 150, 13, 4, 8, None), (30, 8, 150, 9, 4, 8, "tandem"), (30, 8, 150, 9, 4, 8, "identical"), (20, 8, 150, 8, 1, 2, "homopolymer")):
    prm = ca.Params(k, solid, common, 2, msa)
    hb = synth_host(ca.SynthSpec.pacbio(n, depth, first_window=300 + depth))
    if fam:
        fuzz_parity.low_complexity(hb, rng, fam)
    res, st = oracle_lib.oracle_run(prm, hb, threads=2)
    for w in range(n):
        h.update(res.consensus(w).encode()); h.update(bytes([int(res.status[w])])); h.update(res.solid_kmers(w).tobytes())
    h.update(str(int(st["dp_cells"])).encode())
print("DIGEST", h.hexdigest())
"""
    def run(env):
        out = subprocess.run([sys.executable, "-c", child, root], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert out.returncode == 0, out.stderr[-1500:]
        return [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1]

    d_default = run({})
    assert d_default == run({"CW_ORACLE_LIB": str(tmp_path / "liboracle_simd.so")})
    # the overlap alignment mode (cw_policy.h CW_POA_MODE_OV) in both fills: same function again, and not the global mode's
    ov, ovs = tmp_path / "ov", tmp_path / "ovs"
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "policy", f"OUT={ov}", "POLICY=-DCW_POA_MODE=2"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "policy", f"OUT={ovs}", "POLICY=-DCW_POA_MODE=2 -mavx2 -DCWO_SIMD -DCWO_FAST"])
    d_ov = run({"CW_ORACLE_LIB": str(ov / "liboracle.so")})
    assert d_ov == run({"CW_ORACLE_LIB": str(ovs / "liboracle.so")}) and d_ov != d_default
    # the affine gap model (cw_policy.h CW_POA_GAP_MODEL_AFFINE, round 5): with ext == open it IS the linear model -- cell values, tie rules, consensuses --
    # and with ext != open it is not
    aeq, aff = tmp_path / "aeq", tmp_path / "aff"
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "policy", f"OUT={aeq}", "POLICY=-DCW_POA_GAP_MODEL=1 -DCW_POA_GAP_OPEN=-8 -DCW_POA_GAP_EXT=-8"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "policy", f"OUT={aff}", "POLICY=-DCW_POA_GAP_MODEL=1 -DCW_POA_GAP_OPEN=-8 -DCW_POA_GAP_EXT=-6"])
    assert run({"CW_ORACLE_LIB": str(aeq / "liboracle.so")}) == d_default
    assert run({"CW_ORACLE_LIB": str(aff / "liboracle.so")}) != d_default
