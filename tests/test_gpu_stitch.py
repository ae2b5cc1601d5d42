"""GPU: cw_stitch_device (alignConsensus + trimRead + dropRead on the device) against the oracle's restatement,
fed with the same window consensuses and solid sets.  Bit-exact strings and statuses."""
import os
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from test_oracle_ref import rand_seq
from test_oracle_stitch import noisy, stitch

pytestmark = pytest.mark.gpu

K, SOLID = 9, 4


def make_reads(seed, n_reads, depth, lo=900, hi=3200, rate=0.12):
    """n_reads templates over one genome, each with `depth` covering reads (coordinates by construction)."""
    rng = random.Random(seed)
    out = []
    for _ in range(n_reads):
        glen = rng.randrange(lo, hi)
        genome = rand_seq(rng, glen)
        read = noisy(rng, genome, rate)
        targets, rows = [], []
        for t in range(depth):
            a = rng.randrange(0, max(1, glen // 5)) if rng.random() < 0.5 else 0
            b = glen - (rng.randrange(0, max(1, glen // 5)) if rng.random() < 0.5 else 0)
            tg = noisy(rng, genome[a:b], rate)
            targets.append(tg)
            qs = int(a / glen * len(read))
            qe = min(len(read) - 1, int(b / glen * len(read)))
            rows.append([len(read), qs, qe, 0, len(tg), 0, len(tg) - 1, t])
        out.append((read, targets, rows))
    return out


def build(reads_spec, window_size=500, window_overlap=50, min_support=3):
    """windows + piles of every read with the (reference-pinned) oracle feeders"""
    o = oracle_lib.oracle()
    piles, pos, jobs, reads = [], [], [], []
    for ri, (read, targets, rows) in enumerate(reads_spec):
        wins = oracle_lib.window_positions(o.cwo_window_positions, len(read), rows, min_support, window_size, window_overlap)
        jobs.append((ri, len(piles), len(wins)))
        for (qb, qe) in wins:
            piles.append(oracle_lib.window_pile(o.cwo_window_pile, rows, read, targets, qb, qe, K))
            pos.append((qb, qe))
        reads.append(read)
    return reads, jobs, pos, piles


def oracle_stitch_all(reads, jobs, pos, piles, res, do_trim, window_size=500, window_overlap=50, k=K):
    out = []
    for (ri, w0, wn) in jobs:
        cons = [res.consensus(w) for w in range(w0, w0 + wn)]
        solid = [res.solid_kmers(w) for w in range(w0, w0 + wn)]
        tpls = [piles[w][0] if piles[w] else "" for w in range(w0, w0 + wn)]
        final, _ = stitch(reads[ri], cons, tpls, solid, pos[w0 : w0 + wn], do_trim=do_trim, k=k, wsize=window_size, wover=window_overlap)
        out.append(final)
    return out


def run_case(spec, do_trim=True, mutate=None, window_size=500, window_overlap=50, prm=None):
    reads, jobs, pos, piles = build(spec, window_size, window_overlap)
    prm = prm or ca.Params(K, SOLID, 8, 2, 150)
    eng = ca.Engine(prm)
    try:
        batch = ca.pack_piles(piles)
        res = eng.run(batch, want_solid=True)
        assert not (res.status == ca.WIN_OVERFLOW).any()
        if mutate:
            mutate(res, piles)
        got = eng.stitch(ca.pack_piles([reads]), np.array(jobs, np.uint32), np.array(pos, np.uint32), batch, res, window_size, window_overlap, do_trim)
    finally:
        eng.close()
    want = oracle_stitch_all(reads, jobs, pos, piles, res, do_trim, window_size, window_overlap, prm.k)
    n_up = 0
    for i, ((g, st), w) in enumerate(zip(got, want)):
        assert st in (0, 1), (i, st)
        assert g == w, (i, len(g), len(w), first_diff(g, w))
        n_up += sum(c.isupper() for c in g)
    return got, n_up


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i, a[max(0, i - 20) : i + 20], b[max(0, i - 20) : i + 20]
    return min(len(a), len(b)), "", ""


def test_stitch_matches_oracle_pacbio_like():
    got, n_up = run_case(make_reads(101, 6, 20))
    assert n_up > 0 and all(g for g, _ in got)


def test_stitch_without_trimming_keeps_the_uncorrected_ends():
    got, _ = run_case(make_reads(102, 4, 14), do_trim=False)
    assert any(g[0].islower() or g[-1].islower() for g, _ in got)


def test_stitch_with_sparse_coverage_and_dropped_reads():
    # depth 3 with partial overlaps: few windows, long lower-case stretches, some reads dropped or empty
    spec = make_reads(103, 8, 3, lo=700, hi=2500)
    got, _ = run_case(spec)
    spec0 = [(spec[0][0], [], [])]  # a read without any overlap: no window at all
    got0, _ = run_case(spec0)
    assert got0[0][0] == ""


def test_stitch_overlap_reconciliation_paths():
    """Damage the head of every second consensus: overlapping windows now disagree, the previous window has more solid k-mers,
    and the banded traceback decides where to cut (correctionAlignment.cpp:101-119)."""
    rng = random.Random(7)

    def mutate(res, piles):
        for w in range(1, len(piles), 2):
            o, n = int(res.cons_off[w]), int(res.cons_len[w])
            if n < 120:
                continue
            s = bytearray(res.cons[o : o + n].tobytes())
            for _ in range(6):
                p = rng.randrange(5, 70)
                s[p] = ord(rng.choice("ACGT"))
            # one deletion and one insertion inside the overlap, length preserved
            p = rng.randrange(10, 40)
            del s[p]
            q = rng.randrange(40, 70)
            s.insert(q, ord(rng.choice("ACGT")))
            res.cons[o : o + n] = np.frombuffer(bytes(s), np.uint8)

    run_case(make_reads(104, 6, 18), mutate=mutate)


def test_stitch_short_consensus_falls_back_to_the_template_without_writing():
    def mutate(res, piles):
        for w in range(0, len(piles), 3):
            res.cons_len[w] = 5  # < merSize: correctionAlignment.cpp:75-77 aligns the template and :125 skips the replace

    run_case(make_reads(105, 4, 16), mutate=mutate)


def test_stitch_other_window_geometry_and_k():
    run_case(make_reads(106, 4, 16), window_size=300, window_overlap=30, prm=ca.Params(7, 4, 8, 2, 150))
    run_case(make_reads(107, 3, 16, rate=0.15), window_size=700, window_overlap=80, prm=ca.Params(11, 3, 6, 2, 150))


def test_stitch_long_consensuses_use_the_wide_sweeps():
    """Consensuses of 600-1024, of more than 1024 and of more than 1536 bases (junk around the true window sequence): the sweeps of 6, 8, 12 and 16
    registers per slot (the short last windows of the reads and the other tests cover 1-5)."""
    rng = random.Random(9)

    rep = 0

    def mutate(res, piles):
        for w in range(len(piles)):
            o, n = int(res.cons_off[w]), int(res.cons_len[w])
            cap = int(res.cons_off[w + 1]) - o
            extra = [0, 150, 600, 700, 1150, 250][w % 6]
            if n + extra > cap or extra == 0:
                continue
            s = res.cons[o : o + n].tobytes()
            left = [extra // 2, extra, 0][rep % 3]  # the true part in the middle, at the far end, at the start
            junk_l = "".join(rng.choice("ACGT") for _ in range(left)).encode()
            junk_r = "".join(rng.choice("ACGT") for _ in range(extra - left)).encode()
            t = junk_l + s + junk_r
            res.cons[o : o + len(t)] = np.frombuffer(t, np.uint8)
            res.cons_len[w] = len(t)

    for rep in range(6):
        run_case(make_reads(109 + rep, 5, 14), mutate=mutate)
    assert rep == 5


def test_stitch_consensuses_beyond_2048_take_the_last_launch():
    """A window whose consensus is longer than the 2048 characters the re-assembly kernel holds in LDS and registers (chance anchors with k < 8 make
    them; here junk is put around the true window sequence, up to 9000 characters, in result slots as large as k = 7 gets them) marks its read, and
    the last launch -- every buffer and the sweep's state in global memory -- does the read again: status 0 and the oracle's string, not
    CW_READ_CAPACITY.  The other reads of the launch are untouched by it."""
    rng = random.Random(19)
    rep = 0

    def mutate(res, piles):
        hit = 0
        for w in range(len(piles)):
            o, n = int(res.cons_off[w]), int(res.cons_len[w])
            cap = int(res.cons_off[w + 1]) - o
            extra = [0, 700, 2300, 0, 1300, 4100, 0, 9000][(w + rep) % 8]  # (700 / 1300: in a marked read these take the memory-state sweep as well)
            if n == 0 or extra == 0 or n + extra > cap:
                continue
            s = res.cons[o : o + n].tobytes()
            left = [extra // 2, extra, 0][(w + rep) % 3]
            t = "".join(rng.choice("ACGT") for _ in range(left)).encode() + s + "".join(rng.choice("ACGT") for _ in range(extra - left)).encode()
            res.cons[o : o + len(t)] = np.frombuffer(t, np.uint8)
            res.cons_len[w] = len(t)
            hit += 1
        assert hit > 0

    for rep in range(3):
        got, n_up = run_case(make_reads(140 + rep, 5, 12), mutate=mutate, prm=ca.Params(7, SOLID, 8, 2, 150))
        assert n_up > 0


def test_stitch_many_reads_in_one_launch():
    got, n_up = run_case(make_reads(108, 40, 10, lo=600, hi=1800))
    assert len(got) == 40 and n_up > 0


def test_stitch_capacity_path_reports_status_and_does_not_disturb_other_reads(aids, monkeypatch):
    """Shrink the banded-traceback scratch: reads that need it report CW_READ_CAPACITY (2) with an empty result, the others
    are still bit-exact."""
    spec = make_reads(102, 4, 14)
    reads, jobs, pos, piles = build(spec)
    prm = ca.Params(K, SOLID, 8, 2, 150)
    eng = ca.Engine(prm)
    try:
        batch = ca.pack_piles(piles)
        res = eng.run(batch, want_solid=True)
        full = eng.stitch(ca.pack_piles([reads]), np.array(jobs, np.uint32), np.array(pos, np.uint32), batch, res, 500, 50, False)
        monkeypatch.setenv("CW_STITCH_DIR_BYTES", "64")
        small = eng.stitch(ca.pack_piles([reads]), np.array(jobs, np.uint32), np.array(pos, np.uint32), batch, res, 500, 50, False)
    finally:
        eng.close()
    assert all(st == 0 for _, st in full)
    assert any(st == 2 for _, st in small)
    for (g, st), (f, _) in zip(small, full):
        assert (st == 2 and g == "") or (st == 0 and g == f)


def test_stitch_narrow_kernel_hands_long_consensuses_to_the_wide_one(aids, monkeypatch):
    """CW_STITCH_NARROW=1: the five-chunk kernel (consensus and slice <= 640) takes every read first; a read with a longer consensus in any window is
    marked and redone by the wide kernel in a second launch -- same strings either way (the oracle's restatement decides)."""
    monkeypatch.setenv("CW_STITCH_NARROW", "1")
    test_stitch_long_consensuses_use_the_wide_sweeps()
    got, n_up = run_case(make_reads(118, 40, 10, lo=600, hi=1800))
    assert len(got) == 40 and n_up > 0


def test_stitch_several_waves_per_read(aids, monkeypatch):
    """CW_STITCH_SYS=1: one read per work-group of five waves, every sweep shared between them as a pipeline over the query's chunks (one chunk
    per wave up to 640 positions, two up to 1280; longer consensuses are handed to the wide kernel) -- same strings as the oracle's restatement in
    every case the one-wave kernels are tested on."""
    monkeypatch.setenv("CW_STITCH_SYS", "1")
    test_stitch_matches_oracle_pacbio_like()
    test_stitch_without_trimming_keeps_the_uncorrected_ends()
    test_stitch_with_sparse_coverage_and_dropped_reads()
    test_stitch_overlap_reconciliation_paths()
    test_stitch_short_consensus_falls_back_to_the_template_without_writing()
    test_stitch_long_consensuses_use_the_wide_sweeps()
    test_stitch_many_reads_in_one_launch()
    got, n_up = run_case(make_reads(118, 40, 10, lo=600, hi=1800))
    assert len(got) == 40 and n_up > 0



@pytest.mark.timeout(1200)
def test_stitch_chunked_sweep_build_variant(tmp_path):
    """-DCW_ST_STRIPED=0 -DCW_ST_BAND_PAR=0 -DCW_ST_FENCE_AGENT=1: the re-assembly kernel of rounds 1-4 (the chunked sweep with a prefix-max ladder per
    chunk of 128 positions, the banded traceback's rows cell by cell on one lane, agent-scope ordering points) instead of round 5's; the same
    strings in every case of this file that runs on the product library."""
    import subprocess
    import sys

    from consent_amd import _build

    lib = str(tmp_path / "libconsent_amd_chunked.so")
    subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCW_ST_STRIPED=0", "-DCW_ST_BAND_PAR=0", "-DCW_ST_FENCE_AGENT=1", *_build.SRC, "-o", lib], check=True)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not aids and not variant and not narrow and not several_waves and not capacity",
                          "-p", "no:cacheprovider"], capture_output=True, text=True, env=dict(os.environ, CONSENT_AMD_LIB=lib), timeout=1000)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
