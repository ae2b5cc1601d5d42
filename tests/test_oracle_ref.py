"""Pin the oracle's host-feeder rows (A1 windows, A2 piles, alphabet helpers) against the reference's OWN code:
oracle/_ref is built from /root/reference/src/{alignmentWindows,alignmentPiles,utils,reverseComplement}.cpp unmodified.
On a box without oracle/_ref (it is not in git) these tests skip; the committed golden vectors still run."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_lib


def _need_ref():
    r = oracle_lib.ref()
    if r is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return r


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def rand_overlaps(rng, tpl_len, n, with_targets=False):
    ovls, targets = [], []
    for t in range(n):
        qs = rng.randrange(0, max(1, tpl_len - 50))
        qe = rng.randrange(qs + 20, min(tpl_len, qs + 20 + rng.randrange(30, 2500)) + 1) - 1
        qe = min(qe, tpl_len - 1)
        tl = (qe - qs + 1) + rng.randrange(0, 400)
        ts = rng.randrange(0, tl - (qe - qs + 1) + 1)
        te = min(tl - 1, ts + (qe - qs) + rng.randrange(-15, 16))
        te = max(te, ts + 5)
        ovls.append([tpl_len, qs, qe, rng.randrange(2), tl, ts, te, t])
        targets.append(rand_seq(rng, tl))
    return ovls, targets


@pytest.mark.parametrize("seed", range(12))
def test_window_positions_match_reference(seed):
    r = _need_ref()
    rng = random.Random(seed)
    tpl_len = rng.choice([300, 499, 500, 501, 1200, 2000, 5000, 9000])
    ovls, _ = rand_overlaps(rng, tpl_len, rng.randrange(1, 40))
    for min_support in (1, 3, 4):
        got = oracle_lib.window_positions(oracle_lib.oracle().cwo_window_positions, tpl_len, ovls, min_support, 500, 50)
        exp = oracle_lib.window_positions(r.ref_window_positions, tpl_len, ovls, min_support, 500, 50)
        assert got == exp


def test_window_positions_known_answer():
    """SURVEY 'long-context' probe: a fully covered 2000-bp template."""
    ovls = [[2000, 0, 1999, 0, 2000, 0, 1999, 0]] * 3
    got = oracle_lib.window_positions(oracle_lib.oracle().cwo_window_positions, 2000, ovls, 3, 500, 50)
    assert got == [(0, 499), (450, 949), (900, 1399), (1350, 1849), (1500, 1999)]


@pytest.mark.parametrize("seed", range(12))
def test_window_piles_match_reference(seed):
    r = _need_ref()
    rng = random.Random(100 + seed)
    tpl_len = rng.choice([600, 1500, 3000])
    tpl = rand_seq(rng, tpl_len)
    ovls, targets = rand_overlaps(rng, tpl_len, rng.randrange(2, 25))
    wins = oracle_lib.window_positions(r.ref_window_positions, tpl_len, ovls, 1, 500, 50)
    wins += [(0, 499), (tpl_len - 500, tpl_len - 1), (37, 536)]
    for (qb, qe) in wins:
        for k in (9, 15):
            got = oracle_lib.window_pile(oracle_lib.oracle().cwo_window_pile, ovls, tpl, targets, qb, qe, k)
            exp = oracle_lib.window_pile(r.ref_window_pile, ovls, tpl, targets, qb, qe, k)
            assert got == exp


def test_window_beyond_template_is_empty():
    r = _need_ref()
    ovls = [[400, 0, 399, 0, 400, 0, 399, 0]]
    tpl = "A" * 400
    assert oracle_lib.window_pile(oracle_lib.oracle().cwo_window_pile, ovls, tpl, ["C" * 400], 0, 499, 9) == []
    assert oracle_lib.window_pile(r.ref_window_pile, ovls, tpl, ["C" * 400], 0, 499, 9) == []


def test_alphabet_matches_reference_pack_and_revcomp():
    r = _need_ref()
    import consent_amd as ca

    rng = random.Random(7)
    for _ in range(50):
        s = "".join(rng.choice("ACGTNacgtn") for _ in range(rng.randrange(1, 80)))
        up = s.upper()  # the reference upper-cases at index time (utils.cpp:189)
        out = np.zeros(len(s) + 1, np.uint8)
        n = r.ref_pack_unpack(up.encode(), len(up), C.c_void_p(out.ctypes.data))
        ref_decoded = out[:n].tobytes().decode()
        assert ca.pack_piles([[s]]).pile(0) == [ref_decoded]
        acgt = "".join(c for c in up if c in "ACGT") or "A"
        n = r.ref_revcomp(acgt.encode(), len(acgt), C.c_void_p(out.ctypes.data))
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        assert out[:n].tobytes().decode() == "".join(comp[c] for c in reversed(acgt))


def test_reference_pile_sort_order_on_ties(tmp_path):
    """getNextReadPile sorts by resMatches with std::sort on reverse iterators: for <=16 overlaps ties keep PAF order."""
    r = _need_ref()
    paf = tmp_path / "t.paf"
    rows = [("t1", 100), ("t2", 200), ("t3", 100), ("t4", 200), ("t5", 100)]
    with open(paf, "w") as f:
        for name, m in rows:
            f.write(f"q\t1000\t0\t900\t+\t{name}\t1000\t0\t900\t{m}\t900\t60\n")
    pile = np.zeros(16, np.uint32)
    rm = np.zeros(16, np.uint32)
    names = np.zeros(16 * 16, np.uint8)
    n = r.ref_paf_pile_order(str(paf).encode(), 150, C.c_void_p(pile.ctypes.data), C.c_void_p(rm.ctypes.data), C.c_void_p(names.ctypes.data), 16, 16)
    assert n == 5
    got = [names[i * 16 : (i + 1) * 16].tobytes().split(b"\0")[0].decode() for i in range(n)]
    assert got == ["t2", "t4", "t1", "t3", "t5"]


@pytest.mark.parametrize("seed", range(8))
def test_product_window_positions_match_reference(seed):
    """cw_window_positions (the library's host implementation of A1) against the reference's own getAlignmentWindowsPositions."""
    from consent_amd.engine import window_positions as product_window_positions

    r = _need_ref()
    rng = random.Random(500 + seed)
    tpl_len = rng.choice([300, 500, 501, 1700, 4000, 8000])
    ovls, _ = rand_overlaps(rng, tpl_len, rng.randrange(1, 40))
    rows = np.array([[o[1], o[2], 0, o[5], o[6], o[3]] for o in ovls], np.uint32)
    for min_support, wsize, wover in ((1, 500, 50), (3, 500, 50), (2, 200, 0), (4, 300, 120)):
        exp = oracle_lib.window_positions(r.ref_window_positions, tpl_len, ovls, min_support, wsize, wover)
        assert product_window_positions(tpl_len, rows, min_support, wsize, wover) == exp
