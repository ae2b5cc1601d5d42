"""CPU: read re-assembly restatement: local alignment known answers, trimRead/dropRead pinned against the
reference's utils.cpp (oracle/_ref), and an end-to-end sanity run: windows -> piles -> consensus -> stitch -> trim."""
import ctypes as C
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from test_oracle_ref import rand_seq


def ssw(query, ref):
    out = np.zeros(7, np.int32)
    oracle_lib.oracle().cwo_ssw(query.encode(), len(query), ref.encode(), len(ref), C.c_void_p(out.ctypes.data))
    return dict(zip(("score", "ref_begin", "ref_end", "query_begin", "query_end", "ins", "del"), (int(x) for x in out)))


def test_ssw_exact_substring():
    ref = "TTTTACGTACGGTCAAGGTTTT"
    r = ssw("ACGTACGGTCAAGG", ref)
    assert r["score"] == 28 and r["ref_begin"] == 4 and r["ref_end"] == 17 and r["query_begin"] == 0 and r["query_end"] == 13
    assert r["ins"] == 0 and r["del"] == 0


def test_ssw_case_insensitive_and_clipping():
    r = ssw("GGGGacgtacggtcaaggCCCC", "ttttACGTACGGTCAAGGtttt")
    assert r["score"] == 28 and (r["query_begin"], r["query_end"]) == (4, 17) and (r["ref_begin"], r["ref_end"]) == (4, 17)


def test_ssw_affine_gap_cost_and_indel_totals():
    a = "ACGTTGCATGCCAGTACGGATCCATGCAAGT"
    # two bases deleted from the query: a gap of length 2 costs 3 + 1 = 4 -> 2*29 - 4 = 54
    q = a[:15] + a[17:]
    r = ssw(q, a)
    assert r["score"] == 2 * len(q) - 4 and r["del"] == 2 and r["ins"] == 0
    # two bases inserted in the query
    q = a[:15] + "TT" + a[15:]
    r = ssw(q, a)
    assert r["score"] == 2 * len(a) - 4 and r["ins"] == 2 and r["del"] == 0


def test_ssw_first_best_end_wins():
    # the motif occurs twice in the reference: the first column reaching the best score is reported
    r = ssw("ACGTACG", "TTACGTACGTTTTACGTACGTT")
    assert r["score"] == 14 and r["ref_end"] == 8 and r["ref_begin"] == 2


def test_trim_and_drop_match_reference():
    r = oracle_lib.ref()
    if r is None:
        pytest.skip("oracle/_ref not built")
    rng = random.Random(4)
    o = oracle_lib.oracle()
    for _ in range(300):
        n = rng.randrange(1, 120)
        s = "".join(rng.choice("ACGTacgtacgtacgt") for _ in range(n))
        if not any(c.isupper() for c in s):
            s = s[: n // 2] + "A" + s[n // 2 + 1 :]  # the reference is undefined without an upper-case base
        for mer in (1, 2, 3):
            buf1, buf2 = np.zeros(n + 8, np.uint8), np.zeros(n + 8, np.uint8)
            if max(len(x) for x in "".join(c if c.isupper() else " " for c in s).split()) < mer:
                continue  # no upper-case run of that length: undefined in the reference
            l1 = r.ref_trim_read(s.encode(), n, mer, C.c_void_p(buf1.ctypes.data))
            l2 = o.cwo_trim_read(s.encode(), n, mer, C.c_void_p(buf2.ctypes.data))
            assert l1 == l2 and buf1[:l1].tobytes() == buf2[:l2].tobytes(), (s, mer)
        assert r.ref_drop_read(s.encode(), n) == o.cwo_drop_read(s.encode(), n)


def stitch(seq, cons, tpls, solid, pos, do_trim=True, k=9, wsize=500, wover=50):
    o = oracle_lib.oracle()
    n = len(cons)
    cl = np.array([len(c) for c in cons], np.uint32)
    tl = np.array([len(t) for t in tpls], np.uint32)
    sl = np.array([len(s) for s in solid], np.uint32)
    sol = np.concatenate([np.asarray(s, np.uint32) for s in solid] + [np.zeros(1, np.uint32)])
    pp = np.array(pos, np.uint32).reshape(-1)
    out = np.zeros(2 * len(seq) + 4096, np.uint8)
    st = np.zeros(2 * len(seq) + 4096, np.uint8)
    ol, sln = C.c_uint32(), C.c_uint32()
    rc = o.cwo_stitch(seq.encode(), len(seq), n, "".join(cons).encode(), C.c_void_p(cl.ctypes.data), "".join(tpls).encode(), C.c_void_p(tl.ctypes.data),
                      C.c_void_p(sol.ctypes.data), C.c_void_p(sl.ctypes.data), C.c_void_p(pp.ctypes.data), wsize, wover, k, int(do_trim),
                      C.c_void_p(out.ctypes.data), len(out), C.byref(ol), C.c_void_p(st.ctypes.data), C.byref(sln))
    assert rc == 0
    return out[: ol.value].tobytes().decode(), st[: sln.value].tobytes().decode()


def noisy(rng, s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate * 0.3:
            continue
        if x < rate * 0.6:
            out.append(rng.choice("ACGT"))
        out.append(rng.choice("ACGT") if x < rate else c)
    return "".join(out)


def identity(a, b):
    """cheap global identity by banded edit distance (tests only)"""
    n, m = len(a), len(b)
    prev = list(range(m + 1))
    for i in range(1, n + 1):
        cur = [i] + [0] * m
        lo, hi = max(1, i - 60), min(m, i + 60)
        for j in range(1, m + 1):
            if j < lo or j > hi:
                cur[j] = 10 ** 9
                continue
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return 1.0 - prev[m] / max(n, m)


def test_end_to_end_correction_of_one_read_improves_identity():
    rng = random.Random(21)
    genome = rand_seq(rng, 2600)
    read = noisy(rng, genome[100:2300], 0.12)
    # overlaps: 24 other reads covering the template end to end (coordinates by construction, as a PAF would give them)
    targets, rows = [], []
    for t in range(24):
        targets.append(noisy(rng, genome[100:2300], 0.12))
        rows.append([len(read), 0, len(read) - 1, 0, len(targets[-1]), 0, len(targets[-1]) - 1, t])
    wins = oracle_lib.window_positions(oracle_lib.oracle().cwo_window_positions, len(read), rows, 3, 500, 50)
    piles = []
    for (qb, qe) in wins:
        # proportional coordinates instead of true alignments are good enough for a sanity run
        sc = [[r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]] for r in rows]
        piles.append(oracle_lib.window_pile(oracle_lib.oracle().cwo_window_pile, sc, read, targets, qb, qe, 9))
    prm = ca.Params(9, 4, 8, 2, 150)
    res, _ = oracle_lib.oracle_run(prm, ca.pack_piles(piles))
    cons = [res.consensus(w) for w in range(len(piles))]
    solid = [res.solid_kmers(w) for w in range(len(piles))]
    final, stitched = stitch(read, cons, [p[0] for p in piles], solid, wins)
    assert len(stitched) > 0.9 * len(read)
    assert final and final == final.strip()
    raw_id = identity(read, genome[100:2300])
    cor_id = identity(final.upper(), genome[100:2300])
    assert cor_id > raw_id + 0.05, (raw_id, cor_id)
    up = sum(c.isupper() for c in stitched) / len(stitched)
    assert up > 0.8
