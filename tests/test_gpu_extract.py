"""GPU: device-side pile extraction (cw_extract_piles_device) against the oracle's window_pile, which tests/test_oracle_ref.py
pins to the reference's own alignmentWindows.cpp -- and against that reference build directly when oracle/_ref is present."""
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from test_oracle_ref import rand_overlaps, rand_seq

pytestmark = pytest.mark.gpu


def build_case(seed, n_templates=3, n_ovl=(2, 40)):
    rng = random.Random(seed)
    reads, ovl_rows, jobs, expect_inputs = [], [], [], []
    for _ in range(n_templates):
        tpl_len = rng.choice([600, 1500, 3000])
        tpl = rand_seq(rng, tpl_len)
        tpl_id = len(reads)
        reads.append(tpl)
        ovls, targets = rand_overlaps(rng, tpl_len, rng.randrange(*n_ovl))
        first = len(ovl_rows)
        fixed = []
        for o, t in zip(ovls, targets):
            tid = len(reads)
            reads.append(t)
            # cw_overlap: q_start q_end t_read t_start t_end strand ; oracle row: q_len q_start q_end strand t_len t_start t_end t_id
            ovl_rows.append([o[1], o[2], tid, o[5], o[6], o[3]])
            fixed.append([o[0], o[1], o[2], o[3], len(t), o[5], o[6], len(fixed)])
        wins = oracle_lib.window_positions(oracle_lib.oracle().cwo_window_positions, tpl_len, fixed, 1, 500, 50)
        wins += [(0, 499), (tpl_len - 500, tpl_len - 1), (tpl_len - 300, tpl_len + 199), (13, 512)]
        for (qb, qe) in wins:
            jobs.append([tpl_id, qb, qe, first, len(ovls)])
            expect_inputs.append((fixed, tpl, targets, qb, qe))
    return reads, np.array(ovl_rows, np.uint32), np.array(jobs, np.uint32), expect_inputs


@pytest.mark.parametrize("seed,k", [(1, 9), (2, 9), (3, 15), (4, 7)])
def test_device_piles_match_reference_semantics(seed, k):
    reads, ovl, jobs, exp_in = build_case(seed)
    eng = ca.Engine(ca.Params(9, 4, 8, 2, 20))
    got = eng.extract_piles(ca.pack_piles([reads]), ovl, jobs, k)
    assert got.n_windows == len(jobs)
    r = oracle_lib.ref()
    for w, (fixed, tpl, targets, qb, qe) in enumerate(exp_in):
        exp = oracle_lib.window_pile(oracle_lib.oracle().cwo_window_pile, fixed, tpl, targets, qb, qe, k)
        assert got.pile(w) == exp, f"window {w} [{qb},{qe}]"
        if r is not None:
            assert got.pile(w) == oracle_lib.window_pile(r.ref_window_pile, fixed, tpl, targets, qb, qe, k)
    eng.close()


def test_extracted_batch_feeds_the_engine():
    """piles cut on the device go straight into cw_run: same consensus as packing the same piles on the host"""
    reads, ovl, jobs, _ = build_case(7, n_templates=2)
    prm = ca.Params(9, 4, 8, 2, 20)
    eng = ca.Engine(prm)
    hb = eng.extract_piles(ca.pack_piles([reads]), ovl, jobs, 9)
    keep = [w for w in range(hb.n_windows) if hb.win_first_seq[w + 1] > hb.win_first_seq[w]]
    piles = [hb.pile(w) for w in keep]
    a = eng.run(ca.pack_piles(piles))
    exp, _ = oracle_lib.oracle_run(prm, ca.pack_piles(piles))
    for i in range(len(piles)):
        assert a.consensus(i) == exp.consensus(i) and int(a.status[i]) == int(exp.status[i])
    full = eng.run(hb)  # including the empty windows (beyond the template)
    for i, w in enumerate(keep):
        assert full.consensus(w) == a.consensus(i)
    for w in range(hb.n_windows):
        if w not in keep:
            assert int(full.cons_len[w]) == 0
    eng.close()


@pytest.mark.parametrize("seed,n_ovl", [(11, (65, 130)), (12, (150, 400)), (13, (2500, 3000))])
def test_piles_of_many_overlaps(seed, n_ovl):
    """more overlaps than lanes: the count kernel's 64-at-a-time loop, depth-150 piles, and polishing's "every overlap of the contig per
    window" (CONSENT-polish:43 maxSupport=20000) -- against the reference's own alignmentWindows.cpp (oracle/_ref) when present"""
    reads, ovl, jobs, exp_in = build_case(seed, n_templates=2, n_ovl=n_ovl)
    eng = ca.Engine(ca.Params(9, 4, 8, 2, 20))
    got = eng.extract_piles(ca.pack_piles([reads]), ovl, jobs, 9)
    r = oracle_lib.ref()
    deep = 0
    for w, (fixed, tpl, targets, qb, qe) in enumerate(exp_in):
        fn = r.ref_window_pile if r is not None else oracle_lib.oracle().cwo_window_pile
        exp = oracle_lib.window_pile(fn, fixed, tpl, targets, qb, qe, 9)
        assert got.pile(w) == exp, f"window {w} [{qb},{qe}]"
        deep = max(deep, len(exp))
    assert all(len(f) > 64 for f, *_ in exp_in)  # more overlaps than lanes in every job
    assert deep > 64 or n_ovl[0] < 150   # and, from the second case on, piles deeper than one wave
    eng.close()


def test_a_piece_beyond_the_format_limit_is_an_error_not_a_shorter_pile():
    """-l 70000: a window (and its overlap pieces) longer than the batch format's 65535 bases is CW_E_CAPACITY, never dropped silently"""
    import ctypes as C

    rng = random.Random(5)
    tpl = rand_seq(rng, 70100)
    tgt = rand_seq(rng, 70100)
    reads = ca.pack_piles([[tpl, tgt]])
    ovl = np.array([[0, 70099, 1, 0, 70099, 0]], np.uint32)
    eng = ca.Engine(ca.Params(9, 4, 8, 2, 20))
    with pytest.raises(ca.EngineError, match="capacity"):
        eng.extract_piles(reads, ovl, np.array([[0, 0, 69999, 0, 1]], np.uint32), 9)
    ok = eng.extract_piles(reads, ovl, np.array([[0, 100, 599, 0, 1]], np.uint32), 9)  # the same reads, an ordinary window
    assert ok.pile(0) == [tpl[100:600], tgt[100:600]]
    with pytest.raises(ca.EngineError, match="invalid"):  # an overlap naming a read outside the set
        eng.extract_piles(reads, np.array([[0, 70099, 7, 0, 70099, 0]], np.uint32), np.array([[0, 100, 599, 0, 1]], np.uint32), 9)
    eng.close()
