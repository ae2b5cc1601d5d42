"""GPU: device-side pile extraction (cw_extract_piles_device) against the oracle's window_pile, which tests/test_oracle_ref.py
pins to the reference's own alignmentWindows.cpp -- and against that reference build directly when oracle/_ref is present."""
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from test_oracle_ref import rand_overlaps, rand_seq

pytestmark = pytest.mark.gpu


def build_case(seed, n_templates=3):
    rng = random.Random(seed)
    reads, ovl_rows, jobs, expect_inputs = [], [], [], []
    for _ in range(n_templates):
        tpl_len = rng.choice([600, 1500, 3000])
        tpl = rand_seq(rng, tpl_len)
        tpl_id = len(reads)
        reads.append(tpl)
        ovls, targets = rand_overlaps(rng, tpl_len, rng.randrange(2, 40))
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
