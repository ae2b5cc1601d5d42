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
