"""GPU parity tests proper: the HIP engine, called through the C ABI, against the CPU oracle and the golden vectors.
Bit-exact: consensus bytes (including case), per-window status, ascending solid k-mer set."""
import json
import os
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from consent_amd.engine import synth_host

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CS = json.load(open(os.path.join(HERE, "golden", "consensus_small.json")))


@pytest.fixture(scope="module")
def engines():
    cache = {}  # at most four engines alive: an engine holds ~10 GB of POA slabs whatever the batch size, and the module uses two dozen parameter sets

    def get(*prm):
        if prm in cache:
            cache[prm] = cache.pop(prm)  # most recently used last
        else:
            while len(cache) >= 4:
                cache.pop(next(iter(cache))).close()
            cache[prm] = ca.Engine(ca.Params(*prm))
        return cache[prm]

    yield get
    for e in cache.values():
        e.close()


def assert_same(got, exp, n, what=""):
    for w in range(n):
        assert int(got.status[w]) == int(exp.status[w]), f"{what} window {w}: status {got.status[w]} != {exp.status[w]}"
        assert got.consensus(w) == exp.consensus(w), f"{what} window {w}: consensus differs"
        assert np.array_equal(got.solid_kmers(w), exp.solid_kmers(w)), f"{what} window {w}: solid set differs"


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate * 0.3:
            continue
        if x < rate * 0.6:
            out.append(rng.choice("ACGT"))
        out.append(rng.choice("ACGT") if x < rate else c)
    return "".join(out)


def test_loaded_library_is_the_hip_engine(engines):
    e = engines(9, 4, 8, 2, 20)
    assert b"gfx950" in e.lib.cw_version()


@pytest.mark.parametrize("i", range(len(CS["cases"])))
def test_golden_vectors(engines, i):
    c = CS["cases"][i]
    got = engines(*c["params"]).run(ca.pack_piles([c["pile"]]))
    assert int(got.status[0]) == c["status"]
    assert got.consensus(0) == c["consensus"]
    assert [int(x) for x in got.solid_kmers(0)] == c["solid"]


@pytest.mark.parametrize("depth,max_msa,n", [(30, 20, 96), (150, 150, 24), (8, 20, 64), (60, 10, 32)])
def test_synthetic_pacbio_matches_oracle(engines, depth, max_msa, n):
    prm = (9, 4, 8, 2, max_msa)
    hb = synth_host(ca.SynthSpec.pacbio(n, depth))
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=os.cpu_count() or 1)
    assert_same(got, exp, n, f"pacbio d{depth}")


def test_large_k_on_deep_piles_uses_the_hashed_count_path(engines):
    """k > 9: exact counts in the partitioned LDS hash table (several passes at depth 150)."""
    prm = (12, 4, 8, 2, 150)
    hb = synth_host(ca.SynthSpec.pacbio(12, 150))
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=os.cpu_count() or 1)
    assert_same(got, exp, 12, "k=12 d150")


def test_synthetic_ont_profile_matches_oracle(engines):
    prm = (9, 4, 8, 2, 50)
    hb = synth_host(ca.SynthSpec.ont(48, 40))
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=os.cpu_count() or 1)
    assert_same(got, exp, 48, "ont")


@pytest.mark.parametrize("k,solid,common,min_anchors", [(5, 2, 3, 2), (7, 3, 6, 3), (8, 4, 8, 2), (9, 1, 1, 2), (9, 16, 8, 2), (9, 4, 0, 2), (6, 4, 8, 40),
                                                        (10, 4, 8, 2), (11, 3, 6, 2), (13, 2, 4, 2), (16, 2, 3, 2)])
def test_other_parameters(engines, k, solid, common, min_anchors):
    prm = (k, solid, common, min_anchors, 12)
    hb = synth_host(ca.SynthSpec.pacbio(24, 18, window_len=200))
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=os.cpu_count() or 1)
    assert_same(got, exp, 24, f"k={k}")


def test_edge_piles(engines):
    rng = random.Random(11)
    truth = rand_seq(rng, 300)
    piles = [
        [truth[:200]],                                                  # template only
        [truth[:200]] * 12,                                             # clean, every segment identical
        [truth[:5]],                                                    # template shorter than k
        [truth[:200], "ACG", "A", truth[50:58], truth[:9]],             # members shorter than k (kept by the ABI, ignored by k-mers)
        [rand_seq(rng, 200) for _ in range(10)],                        # unrelated sequences -> template fallback
        ["A" * 150] * 8,                                                # homopolymer: every k-mer repeated -> no anchors
        ["AC" * 80] * 5 + ["AC" * 70 + "G" + "AC" * 9] * 2,             # dinucleotide repeat
        [mutate(rng, truth[:250], 0.02) for _ in range(40)],            # high identity, deep
        [mutate(rng, truth[:250], 0.25) for _ in range(25)],            # very noisy
        [truth[:250]] + [mutate(rng, truth[a : a + 60], 0.05) for a in range(0, 190, 7)],   # short tiles only
        [truth[:180] + truth[60:240]] + [mutate(rng, truth[:180] + truth[60:240], 0.05) for _ in range(14)],  # tandem repeat inside the window
    ]
    prm = (9, 4, 8, 2, 20)
    hb = ca.pack_piles(piles)
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb)
    assert_same(got, exp, len(piles), "edge")
    assert int(got.status[4]) == ca.WIN_TEMPLATE and got.consensus(4) == piles[4][0]


def test_high_identity_deep_piles_use_the_global_anchor_matrix(engines):
    """Nearly every template k-mer is an anchor and the pile is deep: the anchor position matrix outgrows LDS."""
    rng = random.Random(17)
    truth = rand_seq(rng, 520)
    piles = [[truth[:500]] + [mutate(rng, truth[:500], 0.004) for _ in range(150)],
             [truth[10:510]] * 151,
             [mutate(rng, truth[:500], 0.01) for _ in range(120)]]
    prm = (9, 4, 8, 2, 150)
    hb = ca.pack_piles(piles)
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=3)
    assert_same(got, exp, len(piles), "high identity")
    assert all(int(x) == ca.WIN_CONSENSUS for x in got.status[:3])


@pytest.mark.parametrize("occ", [254, 255, 256, 257, 300, 511, 512, 513, 1030])
def test_byte_counters_hand_over_exactly_at_256_occurrences(engines, occ):
    """Phase A of the index kernel counts k-mers in byte counters with fire-and-forget adds (round 5) and decides afterwards whether a counter
    overflowed: the bytes of the table must add up to the number of k-mers counted.  A marker 9-mer occurs `occ` times in the pile (two or
    more copies in some sequences): up to 255 the bytes are the counts, from 256 on the window takes the nibble path -- the solid set and the
    consensus are the oracle's on both sides of the edge, and when two keys of one table word overflow together."""
    rng = random.Random(occ)
    marker, marker2 = "ACGTTGCAA", "ACGTTGCAC"  # neighbours in the key space: the same 32-bit word of the byte table
    truth = rand_seq(rng, 420)
    depth = 130
    per = [occ // depth + (1 if s < occ % depth else 0) for s in range(depth)]
    pile = []
    for s in range(depth):
        seq = mutate(rng, truth, 0.1)
        cuts = sorted(rng.sample(range(20, len(seq) - 20), per[s])) if per[s] else []
        out, last = [], 0
        for c in cuts:
            out.append(seq[last:c])
            out.append(marker if occ != 1030 or rng.random() < 0.5 else marker2)
            last = c
        out.append(seq[last:])
        pile.append("".join(out))
    prm = (9, 4, 8, 2, 150)
    hb = ca.pack_piles([pile, [mutate(rng, truth, 0.1) for _ in range(40)]])
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=4)
    assert_same(got, exp, 2, f"{occ} occurrences")


def test_out_of_order_anchors_in_most_of_a_deep_pile(engines):
    """Round 2's chain scoring: sequences with an out-of-order anchor stay in the presence bits except at those anchors, whose pairs come
    from correction rows.  Here two unique stretches of the window are exchanged in most sequences (150 dirty sequences: masks of three
    words, a handful of rows), in a few (masks of one word), and shifted as a block (dozens of out-of-order anchors per sequence: more
    rows than fit LDS, read from the block); plus 1000-base windows at depth 110, whose hit positions do not pack into the hit list."""
    rng = random.Random(29)
    truth = rand_seq(rng, 1100)
    t500 = truth[:500]

    def swapped(s, a, b, n):
        return s[:a] + s[b : b + n] + s[a + n : b] + s[a : a + n] + s[b + n :]

    def block_moved(s, a, b, n):  # the stretch s[a:a+n] taken out and put in again at b
        rest = s[:a] + s[a + n :]
        return rest[:b] + s[a : a + n] + rest[b:]

    piles = [
        [t500] + [mutate(rng, swapped(t500, 100, 330, 24), 0.03) for _ in range(150)],
        [t500] + [mutate(rng, t500, 0.06) for _ in range(90)] + [mutate(rng, swapped(t500, 60, 400, 30), 0.04) for _ in range(12)],
        [t500] + [mutate(rng, t500, 0.05) for _ in range(70)] + [mutate(rng, block_moved(t500, 80, 300, 90), 0.03) for _ in range(40)],
        [truth[:1000]] + [mutate(rng, truth[:1000], 0.08) for _ in range(110)],
    ]
    prm = (9, 4, 8, 2, 150)
    hb = ca.pack_piles(piles)
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=4)
    assert_same(got, exp, len(piles), "out-of-order anchors")


def test_pile_layout_in_memory_does_not_matter(engines):
    """The index kernel stages a pile in LDS when its sequences lie front to back in `bases`; any other layout (here: sequences stored
    in reverse order, with gaps) takes the global-memory path and must give the same results."""
    rng = random.Random(41)
    piles = []
    for _ in range(6):
        truth = rand_seq(rng, 520)
        piles.append([truth[:500]] + [mutate(rng, truth[rng.randrange(0, 15) : 500 + rng.randrange(0, 20)], 0.12) for _ in range(24)])
    prm = (9, 4, 8, 2, 20)
    hb = ca.pack_piles(piles)
    n = len(hb.seq_len)
    words = [(int(l) + 15) // 16 for l in hb.seq_len]
    new_off = np.zeros(n, np.uint64)
    pos = 3
    for s in reversed(range(n)):  # last sequence first, 5 unused words between sequences
        new_off[s] = pos
        pos += words[s] + 5
    bases = np.full(pos + 1, 0xDEADBEEF, np.uint32)
    for s in range(n):
        o = int(hb.seq_word_off[s])
        bases[int(new_off[s]) : int(new_off[s]) + words[s]] = hb.bases[o : o + words[s]]
    from consent_amd.engine import HostBatch

    shuffled = HostBatch(hb.win_first_seq, hb.seq_len, new_off, bases)
    a = engines(*prm).run(hb)
    b = engines(*prm).run(shuffled)
    assert_same(b, a, len(piles), "layout")
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=3)
    assert_same(a, exp, len(piles), "layout/oracle")


def test_very_deep_piles_score_chains_without_presence_bitsets(engines):
    """More than 2048 sequences in a pile: no presence bitsets, the chain kernel compares positions row against row, and the
    segmentation loops over the pile in several 64-sequence chunks."""
    rng = random.Random(23)
    truth = rand_seq(rng, 140)
    # low error rates: at this depth a noisier pile has more than 2048 k-mers seen >= 15 times, a documented capacity of the counter table
    piles = [[truth[:120]] + [mutate(rng, truth[:120], 0.03) for _ in range(2100)],
             [truth[10:130]] + [mutate(rng, truth[10:130], 0.02) for _ in range(2300)]]
    prm = (7, 4, 8, 2, 20)
    hb = ca.pack_piles(piles)
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=2)
    assert_same(got, exp, len(piles), "very deep")
    assert all(int(x) == ca.WIN_CONSENSUS for x in got.status[:2])


def test_long_and_outlier_segments_exercise_all_tiers(engines):
    """Few anchors -> segments of hundreds of bases; graphs that outgrow the LDS tiers must give identical results."""
    rng = random.Random(13)
    truth = rand_seq(rng, 700)
    piles = []
    for depth, rate in ((6, 0.18), (10, 0.22), (16, 0.2)):
        piles.append([mutate(rng, truth[:600], rate) for _ in range(depth)])
    # one wildly longer member in the middle of the pile
    piles.append([mutate(rng, truth[:300], 0.1) for _ in range(6)] + [truth[:150] + rand_seq(rng, 380) + truth[150:300]] + [mutate(rng, truth[:300], 0.1) for _ in range(6)])
    prm = (9, 4, 8, 2, 30)
    hb = ca.pack_piles(piles)
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb)
    assert_same(got, exp, len(piles), "tiers")


def test_graph_of_more_than_2048_nodes(engines):
    """Members that share their two ends and nothing in between: one segment of several hundred unrelated bases per member, a graph of a few
    thousand nodes.  Beyond 2048 nodes was a window-level capacity through round 4 (tier G's slab); tier G holds 4096 now."""
    rng = random.Random(29)
    head, tail = rand_seq(rng, 60), rand_seq(rng, 60)
    piles = [[head + rand_seq(rng, rng.randrange(520, 640)) + tail for _ in range(depth)] for depth in (6, 8)]
    prm = (9, 2, 8, 2, 150)
    hb = ca.pack_piles(piles)
    got = engines(*prm).run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb)
    assert not (got.status == ca.WIN_OVERFLOW).any()
    assert_same(got, exp, len(piles), "big graph")
    prof = engines(*prm).profile()[0]
    assert int(prof[22]) >= 1, prof[18:24]  # tasks handed to tier G (n_over[4])


def test_tier_h_two_tasks_per_wave(aids, monkeypatch):
    """Tier H (cw_poa_q.h, round 5): segments whose longest member has up to 63 bases, two tasks per wave on 32-lane halves, recorded decisions.
    Windows with few anchors give such segments; deep piles make nodes with many predecessors (the ordinal of a code covers three, the rest
    is decided from kept rows); CW_TIER_H=2 sends it what tiers S / M1 / M2 would take of them, 1 only what tier S would not,
    0 (the default: measured no faster than tier S) nothing -- the consensus never depends on the tier."""
    rng = random.Random(29)
    piles = []
    for depth, rate, k_gap in ((24, 0.14, 40), (60, 0.16, 55), (150, 0.12, 48), (12, 0.2, 60)):
        truth = rand_seq(rng, 520)
        tpl = list(mutate(rng, truth[:500], 0.05))
        # anchors only every k_gap bases: in between, the template is scrambled so that no 9-mer of it is shared
        for a in range(0, 500, k_gap):
            for x in range(a + 14, min(a + k_gap - 2, len(tpl)), 3):
                tpl[x] = rng.choice("ACGT")
        piles.append(["".join(tpl)] + [mutate(rng, truth[:500], rate) for _ in range(depth)])
    prm = (9, 4, 8, 2, 150)
    hb = ca.pack_piles(piles)
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=os.cpu_count() or 1)
    e = ca.Engine(ca.Params(*prm))  # of the test-aid build (fixture `aids`): the CW_TIER_H switch exists there only
    monkeypatch.setenv("CW_TIER_H", "2")
    got = e.run(hb)
    ctr, _ = e.profile()
    assert int(ctr[11]) > 0, ctr[6:12]  # tasks routed to tier H
    assert_same(got, exp, len(piles), "tier H")
    monkeypatch.setenv("CW_LW", "0")  # (list 5 is tier H's only while tier LW is off: with tier H off, tier LW counts its tasks there -- round 6)
    for mode in ("1", "0"):
        monkeypatch.setenv("CW_TIER_H", mode)
        got = e.run(hb)
        ctr2, _ = e.profile()
        assert (int(ctr2[11]) <= int(ctr[11])) if mode == "1" else int(ctr2[11]) == 0
        assert_same(got, exp, len(piles), f"CW_TIER_H={mode}")
    monkeypatch.delenv("CW_LW")
    monkeypatch.delenv("CW_TIER_H")
    hb2 = synth_host(ca.SynthSpec.pacbio(96, 150, first_window=7000))
    exp2, _ = oracle_lib.oracle_run(ca.Params(*prm), hb2, threads=os.cpu_count() or 1)
    monkeypatch.setenv("CW_TIER_H", "2")
    monkeypatch.setenv("CW_H_MIN_LEN", "8")  # nearly everything that is not tier Q's
    assert_same(e.run(hb2), exp2, 96, "depth 150, tier H for every task it can hold")
    e.close()


def test_empty_batch_and_capacity_overflow(engines):
    e = engines(9, 4, 8, 2, 20)
    lib = e.lib
    import ctypes as C

    from consent_amd.engine import Batch, Result, _result_struct, alloc_results

    b = Batch(0, 0, 0, None, None, None, None)
    r = Result(None, None, None, None, None, None, None)
    assert lib.cw_run(e.handle, C.byref(b), C.byref(r)) == -1  # result arrays are mandatory
    hb = synth_host(ca.SynthSpec.pacbio(3, 10))
    res = alloc_results(hb, True, 4, 9)
    res.cons_off[:] = np.array([0, 10, 2000, 4000], np.uint64)  # window 0 gets 10 bytes only
    rc = lib.cw_run(e.handle, C.byref(hb.c_struct()), C.byref(_result_struct(res)))
    assert rc == -4
    assert int(res.status[0]) == ca.WIN_OVERFLOW and int(res.cons_len[0]) == 0
    exp, _ = oracle_lib.oracle_run(ca.Params(9, 4, 8, 2, 20), hb)
    for w in (1, 2):
        assert res.consensus(w) == exp.consensus(w)


def test_results_do_not_depend_on_batch_composition(engines):
    """Window results are a function of the window only: alone, in a batch, shuffled, or split across two runs."""
    prm = (9, 4, 8, 2, 20)
    e = engines(*prm)
    hb = synth_host(ca.SynthSpec.pacbio(40, 20))
    full = e.run(hb)
    piles = [hb.pile(w) for w in range(40)]
    order = list(range(40))
    random.Random(3).shuffle(order)
    shuf = e.run(ca.pack_piles([piles[i] for i in order]))
    for pos, w in enumerate(order):
        assert shuf.consensus(pos) == full.consensus(w)
        assert np.array_equal(shuf.solid_kmers(pos), full.solid_kmers(w))
    a = e.run(ca.pack_piles(piles[:13]))
    b = e.run(ca.pack_piles(piles[13:]))
    for w in range(40):
        part, idx = (a, w) if w < 13 else (b, w - 13)
        assert part.consensus(idx) == full.consensus(w)
    again = e.run(hb)
    for w in range(40):
        assert again.consensus(w) == full.consensus(w)


def test_mirror_of_the_reference_operator(engines):
    """computeConsensusReadCorrection-shaped call (reference correctionMSA.h:8)."""
    hb = synth_host(ca.SynthSpec.pacbio(2, 12))
    piles = [hb.pile(0), hb.pile(1)]
    out = ca.compute_consensus_read_correction("read1", piles, (0, 499), 3, 9, 8, 2, 4, 500, 20, "")
    exp, _ = oracle_lib.oracle_run(ca.Params(9, 4, 8, 2, 20), hb)
    for w, (cons, solid) in enumerate(out):
        assert cons == exp.consensus(w) and np.array_equal(solid, exp.solid_kmers(w))


def test_full_size_batch_properties(engines):
    """BASELINE configs[1] at its bench size (16384 windows, depth 30): properties that need no oracle run --
    every window finishes, consensus is near the truth length, and a checksum over the whole batch is reproducible
    and equals the checksum of two half-batches; a 256-window sample is compared with the oracle."""
    import zlib

    prm = (9, 4, 8, 2, 20)
    e = engines(*prm)
    n = 16384
    hb = synth_host(ca.SynthSpec.pacbio(n, 30))
    r1 = e.run(hb, want_solid=False)
    assert int((r1.status == ca.WIN_OVERFLOW).sum()) == 0
    lens = r1.cons_len.astype(np.int64)
    assert 450 < np.median(lens) < 550

    def digest(res, count):
        crc = 0
        for w in range(count):
            crc = zlib.crc32(res.consensus(w).encode(), crc)
        return crc

    r2 = e.run(hb, want_solid=False)
    assert digest(r1, n) == digest(r2, n)
    ha = synth_host(ca.SynthSpec.pacbio(n // 2, 30))
    hb2 = synth_host(ca.SynthSpec.pacbio(n // 2, 30, first_window=n // 2))
    ra, rb = e.run(ha, want_solid=False), e.run(hb2, want_solid=False)
    crc = 0
    for res in (ra, rb):
        for w in range(n // 2):
            crc = zlib.crc32(res.consensus(w).encode(), crc)
    assert crc == digest(r1, n)
    sample = synth_host(ca.SynthSpec.pacbio(256, 30, first_window=9000))
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), sample, threads=os.cpu_count() or 1)
    for w in range(256):
        assert r1.consensus(9000 + w) == exp.consensus(w)


def test_full_size_batch_properties_depth_150(engines):
    """BASELINE configs[2], the headline configuration, at its bench size (16384 windows, depth 150, maxMSA 150): every window finishes,
    the consensus is near the truth length, the batch checksum is reproducible and equals the checksum of its two halves run on their own,
    and a 256-window sample out of the middle of the batch is compared with the oracle."""
    import zlib

    prm = (9, 4, 8, 2, 150)
    e = engines(*prm)
    n = 16384
    hb = synth_host(ca.SynthSpec.pacbio(n, 150))
    r1 = e.run(hb, want_solid=False)
    assert int((r1.status == ca.WIN_OVERFLOW).sum()) == 0
    assert 450 < np.median(r1.cons_len.astype(np.int64)) < 550

    def digest(res, count, crc=0):
        for w in range(count):
            crc = zlib.crc32(res.consensus(w).encode(), crc)
        return crc

    whole = digest(r1, n)
    sample = synth_host(ca.SynthSpec.pacbio(256, 150, first_window=7000))
    exp, _ = oracle_lib.oracle_run(ca.Params(*prm), sample, threads=os.cpu_count() or 1)
    for w in range(256):
        assert r1.consensus(7000 + w) == exp.consensus(w), f"window {7000 + w} differs from the oracle"
    del r1
    assert digest(e.run(hb, want_solid=False), n) == whole
    del hb
    crc = 0
    for first in (0, n // 2):
        half = e.run(synth_host(ca.SynthSpec.pacbio(n // 2, 150, first_window=first)), want_solid=False)
        crc = digest(half, n // 2, crc)
    assert crc == whole


def test_running_out_of_task_slots_flags_windows_and_never_runs_stale_tasks(aids, monkeypatch):
    """The chain kernel's task / member / list capacities are heuristics (64 tasks and 2048 members per window).  With the capacity shrunk
    (CW_TASK_CAP, CW_MEMBER_CAP: test aids read when a batch is planned) the windows that do not fit come back as overflow, every other window
    is what the oracle says, and a second, different batch on the same engine -- whose reserved-but-unwritten slots hold the first batch's
    records -- behaves the same way."""
    prm = (9, 4, 8, 2, 20)
    e = ca.Engine(ca.Params(*prm))  # of the test-aid build (fixture `aids`)
    for first, depth in ((0, 30), (500, 12)):
        hb = synth_host(ca.SynthSpec.pacbio(48, depth, first_window=first))
        exp, _ = oracle_lib.oracle_run(ca.Params(*prm), hb, threads=os.cpu_count() or 1)
        full = e.run(hb)
        assert_same(full, exp, 48, "uncapped")
        ctr, _ = e.profile()  # what the uncapped run needed: tasks, members
        assert int(ctr[0]) > 8 and int(ctr[1]) > 8, ctr[:2]
        for env, val in (("CW_TASK_CAP", str(int(ctr[0]) // 2)), ("CW_MEMBER_CAP", str(int(ctr[1]) // 2))):  # half of it: some windows fit, some do not
            monkeypatch.setenv(env, val)
            got = e.run(hb)  # capacity overflow is a per-window status here (cw_run returns CW_E_CAPACITY, which Engine.run lets through)
            monkeypatch.delenv(env)
            n_over = int((got.status == ca.WIN_OVERFLOW).sum())
            assert 0 < n_over < 48, (env, n_over)
            for w in range(48):
                if int(got.status[w]) != ca.WIN_OVERFLOW:
                    assert got.consensus(w) == exp.consensus(w), f"{env}: window {w} differs"
        assert_same(e.run(hb), exp, 48, "after the capped runs")
    e.close()


def test_a_batch_that_outgrows_its_task_capacities_is_run_again_with_a_larger_plan(aids, monkeypatch):
    """Task, member and arena slots are sized per batch from its window count (64 tasks, 2048 members a window); a batch of few, heavy windows
    can need more.  cw_run notices that windows stopped on exactly these capacities (CW_WHY_TASKS), plans x4 and runs the batch again
    (cw_engine.cpp grow_if_that_helps; the native driver does the same through cw_run_device_sync).  Exercised with the first plan divided by
    64 (CW_PLAN_DIV, a test aid): the result is the oracle's, no window is left stopped, and the engine keeps the larger plan."""
    prm = ca.Params(9, 4, 8, 2, 20)
    hb = synth_host(ca.SynthSpec.pacbio(48, 30, first_window=8100))
    exp, _ = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
    monkeypatch.setenv("CW_PLAN_DIV", "64")
    e = ca.Engine(prm)
    try:
        got = e.run(hb)
        ctr, _ = e.profile()
        assert int(ctr[1]) > (2048 * 48 + 4096) // 64 + 1, ctr[:2]  # more members than the first plan held
        assert_same(got, exp, 48, "after the plan grew")
        assert_same(e.run(hb), exp, 48, "second batch on the grown engine")
    finally:
        e.close()


def test_cpp_adapter_runs_on_the_gpu(tmp_path):
    """examples/operator_demo.cpp through include/consent_amd_adapter.hpp: the C++ host side of the boundary."""
    import subprocess

    root = os.path.dirname(HERE)
    exe = tmp_path / "operator_demo"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "operator_demo.cpp"), "-L", os.path.join(root, "consent_amd"),
                           "-lconsent_amd", "-Wl,-rpath," + os.path.join(root, "consent_amd"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "status: consensus" in out.stdout, out.stdout + out.stderr


def test_templates_of_up_to_2048_kmers_after_cw_configure():
    """Round 6 (`-l 1500` must run: the reference takes any window size, src/main.cpp:46-47).  The index kernel holds templates of up to 2048 k-mers (1024
    through round 5); an engine told about them (cw_configure: scratch plan, the chain kernel's 32 KB instance) corrects windows of 1500 and 2000 bases
    exactly as the oracle does -- synthetic piles at three depths, and a near-identical pile in which almost every template k-mer is an anchor (more than
    1240 of them: what the default chain instance cannot hold) -- and a template beyond 2048 k-mers is a reported capacity (CW_WHY_TEMPLATE)."""
    rng = random.Random(61)
    # k = 11 for the long windows: with k = 9 a 1500-base sequence holds a chance copy of one template k-mer in 170, whose misplaced hits make pieces of
    # more than the POA tiers' 1023 bases -- a reported capacity here, and the reason nobody runs k = 9 on such windows
    for k, wlen, depth, n, err in ((11, 1500, 30, 24, 60), (11, 2000, 12, 16, 50), (11, 1500, 90, 8, 80), (9, 1030, 60, 8, 120)):
        prm = ca.Params(k, 4, 8, 2, 150)
        eng = ca.Engine(prm)
        eng.configure(wlen)
        spec = ca.SynthSpec(0xC0115E17, 4000 + wlen, n, depth, wlen, err, 10, 60, 30, (wlen + 60) // 16 + 2)
        hb = synth_host(spec)
        exp, st = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
        got = eng.run(hb)
        assert_same(got, exp, n, f"k {k}, window length {wlen}, depth {depth}")
        assert int((np.asarray(got.status[:n]) == 0).sum()) == n, got.status[:n]  # consensus windows, not template fallbacks or stops
        if wlen == 2000:
            assert st["tpl_anchors"] // n > 1024  # more anchors than one round of the index kernel's candidate scan, or the default chain instance, holds
        eng.close()
    prm = ca.Params(9, 4, 8, 2, 150)
    eng = ca.Engine(prm)
    eng.configure(2056)
    truth = rand_seq(rng, 1800)
    piles = [[truth] + [mutate(rng, truth, 0.01) for _ in range(12)], [rand_seq(rng, 2056)] + [rand_seq(rng, 2056) for _ in range(3)]]
    hb = ca.pack_piles(piles)
    exp, _ = oracle_lib.oracle_run(prm, hb, threads=2)
    got = eng.run(hb)
    assert_same(got, exp, 2, "near-identical pile of 1800 bases")
    # without cw_configure the same windows never give a wrong answer: they are corrected or stop on a capacity
    plain = ca.Engine(prm)
    got2 = plain.run(hb)
    for w in range(2):
        assert int(got2.status[w]) == 2 or (int(got2.status[w]) == int(exp.status[w]) and got2.consensus(w) == exp.consensus(w))
    # beyond the index kernel's 2048 k-mers: a capacity, and cw_configure says so up front
    hb3 = ca.pack_piles([[rand_seq(rng, 2100)] * 4])
    got3 = eng.run(hb3)
    assert int(got3.status[0]) == 2
    with pytest.raises(ca.EngineError):
        eng.configure(2100)
    eng.close(); plain.close()
