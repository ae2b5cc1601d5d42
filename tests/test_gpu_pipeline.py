"""GPU: the whole read-correction loop through the library (host feeders -> device extraction -> consensus -> re-assembly,
consent_amd/pipeline.py) against the same loop assembled from the oracle's pieces.  FASTA text must be identical."""
import io
import os
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from consent_amd.pipeline import correct_reads
from test_oracle_ref import rand_seq
from test_oracle_stitch import stitch

pytestmark = pytest.mark.gpu

COMP = str.maketrans("ACGT", "TGCA")


def noisy_map(rng, s, rate, mix=(0.3, 0.3)):
    """noisy copy of s and, for every position of s, the position of the copy it ends up at; mix = (deletion, insertion) shares of the
    errors, the rest are substitutions: (0.3, 0.3) is the tests' default, (0.4, 0.3) the ONT-like 30:30:40 sub:ins:del of SURVEY 8d,
    (0.3, 0.6) the PacBio-like 10:60:30"""
    out, pos = [], []
    d, i = mix
    for c in s:
        pos.append(len(out))
        x = rng.random()
        if x < rate * d:
            continue  # deletion
        if x < rate * (d + i):
            out.append(rng.choice("ACGT"))  # insertion before the base
        out.append(rng.choice("ACGT") if x < rate else c)
    pos.append(len(out))
    return "".join(out), pos


def make_dataset(tmp_path, seed, n_reads=36, glen=7000, rate=0.1, mix=(0.3, 0.3), read_len=(1200, 3200)):
    rng = random.Random(seed)
    genome = rand_seq(rng, glen)
    reads = []
    for i in range(n_reads):
        ln = rng.randrange(*read_len)
        g0 = rng.randrange(0, glen - ln)
        fwd, pos = noisy_map(rng, genome[g0 : g0 + ln], rate, mix)
        rev = rng.random() < 0.4
        seq = fwd[::-1].translate(COMP) if rev else fwd
        reads.append(dict(name=f"r{i}", g0=g0, g1=g0 + ln, pos=pos, rev=rev, seq=seq))
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for r in reads:
            f.write(f">{r['name']} len={len(r['seq'])}\n{r['seq']}\n")

    def span(r, a, b):
        """coordinates of genome [a,b) on read r's own forward strand: (start, end_exclusive)"""
        s, e = r["pos"][a - r["g0"]], r["pos"][b - r["g0"]]
        if r["rev"]:
            n = len(r["seq"])
            s, e = n - e, n - s
        return s, max(e, s + 1)

    paf = tmp_path / "ovl.paf"
    with open(paf, "w") as f:
        for q in reads:
            for t in reads:
                if t is q:
                    continue
                a, b = max(q["g0"], t["g0"]), min(q["g1"], t["g1"])
                if b - a < 400:
                    continue
                qs, qe = span(q, a, b)
                ts, te = span(t, a, b)
                strand = "+" if q["rev"] == t["rev"] else "-"
                matches = int((b - a) * (1 - 2 * rate)) + rng.randrange(0, 3)
                f.write("\t".join(str(x) for x in [q["name"], len(q["seq"]), qs, qe, strand, t["name"], len(t["seq"]), ts, te, matches, b - a, 60]) + "\n")
    return str(fa), str(paf)


def oracle_pipeline(fa, paf, *, min_support, max_support, window_size, mer_size, common_kmers, min_anchors, solid_thresh, window_overlap, max_msa, do_trim=True):
    """CONSENT-correction.cpp:19-58 with the oracle's restatements (parsing through the reference-pinned host feeders)"""
    o = oracle_lib.oracle()
    ix = ca.ReadIndex(fa)
    seqs = [ix.sequence(i) for i in range(len(ix.names))]
    prm = ca.Params(mer_size, solid_thresh, common_kmers, min_anchors, max_msa)
    out = []
    for tpl, tpl_len, ov, _ in ca.PafReader(paf, ix, max_support):
        rows = [[tpl_len, int(r[0]), int(r[1]), int(r[5]), int(ix.seq_len[int(r[2])]), int(r[3]), int(r[4]), i] for i, r in enumerate(ov)]
        targets = [seqs[int(r[2])] for r in ov]
        wins = oracle_lib.window_positions(o.cwo_window_positions, tpl_len, rows, min_support, window_size, window_overlap)
        if not wins:
            continue
        piles = [oracle_lib.window_pile(o.cwo_window_pile, rows, seqs[tpl], targets, qb, qe, mer_size) for qb, qe in wins]
        res, _ = oracle_lib.oracle_run(prm, ca.pack_piles(piles))
        cons = [res.consensus(w) for w in range(len(piles))]
        solid = [res.solid_kmers(w) for w in range(len(piles))]
        final, _ = stitch(seqs[tpl], cons, [p[0] if p else "" for p in piles], solid, wins, do_trim=do_trim, k=mer_size, wsize=window_size, wover=window_overlap)
        if final:
            out.append((ix.names[tpl], final))
    return out


PRM = dict(min_support=3, max_support=1000, window_size=500, mer_size=9, common_kmers=8, min_anchors=10, solid_thresh=4, window_overlap=50, max_msa=150)


def test_fasta_identical_to_the_oracle_pipeline(tmp_path):
    fa, paf = make_dataset(tmp_path, 31)
    buf = io.StringIO()
    got = correct_reads(fa, paf, buf, **PRM)
    want = oracle_pipeline(fa, paf, **PRM)
    assert len(want) > 10
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, g), (_, w) in zip(got, want):
        assert g == w, n
    assert buf.getvalue() == "".join(f">{n}\n{s}\n" for n, s in want)
    up = sum(c.isupper() for _, s in got for c in s) / sum(len(s) for _, s in got)
    assert up > 0.9


def test_batching_does_not_change_the_output(tmp_path):
    fa, paf = make_dataset(tmp_path, 32, n_reads=24)
    a = correct_reads(fa, paf, None, windows_per_batch=1, **PRM)
    b = correct_reads(fa, paf, None, windows_per_batch=100000, **PRM)
    assert a == b and len(a) > 5


def test_other_parameters_and_max_support_cut(tmp_path):
    fa, paf = make_dataset(tmp_path, 33, n_reads=30, rate=0.13)
    prm = dict(PRM, max_support=6, mer_size=8, min_anchors=2, max_msa=20, window_size=400, window_overlap=40)
    got = correct_reads(fa, paf, None, **prm)
    want = oracle_pipeline(fa, paf, **prm)
    assert got == want and len(got) > 5


def test_capacity_stops_the_run_or_leaves_the_read_out(tmp_path, aids, monkeypatch, capfd):
    """With the banded-traceback scratch shrunk some reads cannot be re-assembled: the default raises, "skip" leaves exactly those
    out (named on stderr) and every other read is still identical."""
    fa, paf = make_dataset(tmp_path, 35, n_reads=24)
    prm = dict(PRM, do_trim=False)
    full = correct_reads(fa, paf, None, **prm)
    monkeypatch.setenv("CW_STITCH_DIR_BYTES", "64")
    with pytest.raises(ca.EngineError, match="capacity"):
        correct_reads(fa, paf, None, **prm)
    part = correct_reads(fa, paf, None, on_capacity="skip", **prm)
    err = capfd.readouterr().err  # the native driver writes to the process's stderr
    assert 0 < len(part) < len(full)
    fd = dict(full)
    for n, s in part:
        assert fd[n] == s
    for n in set(fd) - set(n for n, _ in part):
        assert n in err


def test_polishing_mode_contigs_as_templates(tmp_path):
    """CONSENT-polishing: the templates come from a second file (-R), nothing is trimmed or dropped (CONSENT-polishing.cpp:112-116)."""
    rng = random.Random(34)
    glen = 5200
    genome = rand_seq(rng, glen)
    reads = []
    for i in range(40):
        ln = rng.randrange(900, 2400)
        g0 = rng.randrange(0, glen - ln)
        fwd, pos = noisy_map(rng, genome[g0 : g0 + ln], 0.1)
        rev = rng.random() < 0.5
        reads.append(dict(name=f"r{i}", g0=g0, g1=g0 + ln, pos=pos, rev=rev, seq=fwd[::-1].translate(COMP) if rev else fwd))
    contigs = []
    for c, (a, b) in enumerate([(0, 2600), (2500, glen)]):
        fwd, pos = noisy_map(rng, genome[a:b], 0.03)
        contigs.append(dict(name=f"ctg{c}", g0=a, g1=b, pos=pos, rev=False, seq=fwd))
    fa, proof, paf = tmp_path / "reads.fa", tmp_path / "contigs.fa", tmp_path / "ovl.paf"
    open(fa, "w").write("".join(f">{r['name']}\n{r['seq']}\n" for r in reads))
    open(proof, "w").write("".join(f">{c['name']}\n{c['seq']}\n" for c in contigs))

    def span(r, a, b):
        s, e = r["pos"][a - r["g0"]], r["pos"][b - r["g0"]]
        if r["rev"]:
            n = len(r["seq"])
            s, e = n - e, n - s
        return s, max(e, s + 1)

    with open(paf, "w") as f:
        for q in contigs:
            for t in reads:
                a, b = max(q["g0"], t["g0"]), min(q["g1"], t["g1"])
                if b - a < 300:
                    continue
                qs, qe = span(q, a, b)
                ts, te = span(t, a, b)
                f.write("\t".join(str(x) for x in [q["name"], len(q["seq"]), qs, qe, "-" if t["rev"] else "+", t["name"], len(t["seq"]), ts, te, b - a - 50, b - a, 60]) + "\n")
    got = correct_reads(str(fa), str(paf), None, proof_path=str(proof), **PRM)
    # the same loop from the oracle's pieces, untrimmed
    o = oracle_lib.oracle()
    ix = ca.ReadIndex(str(fa), str(proof))
    seqs = [ix.sequence(i) for i in range(len(ix.names))]
    prm = ca.Params(PRM["mer_size"], PRM["solid_thresh"], PRM["common_kmers"], PRM["min_anchors"], PRM["max_msa"])
    want = []
    for tpl, tpl_len, ov, _ in ca.PafReader(str(paf), ix, PRM["max_support"]):
        rows = [[tpl_len, int(r[0]), int(r[1]), int(r[5]), int(ix.seq_len[int(r[2])]), int(r[3]), int(r[4]), i] for i, r in enumerate(ov)]
        targets = [seqs[int(r[2])] for r in ov]
        wins = oracle_lib.window_positions(o.cwo_window_positions, tpl_len, rows, PRM["min_support"], 500, 50)
        piles = [oracle_lib.window_pile(o.cwo_window_pile, rows, seqs[tpl], targets, qb, qe, 9) for qb, qe in wins]
        res, _ = oracle_lib.oracle_run(prm, ca.pack_piles(piles))
        final, _ = stitch(seqs[tpl], [res.consensus(w) for w in range(len(piles))], [p[0] if p else "" for p in piles],
                          [res.solid_kmers(w) for w in range(len(piles))], wins, do_trim=False, k=9, wsize=500, wover=50)
        want.append((ix.names[tpl], final))
    assert got == want and len(got) == 2
    assert all(len(s) > 2000 for _, s in got)


def test_reads_without_windows_are_not_emitted_even_untrimmed(tmp_path):
    """processRead returns an empty result when getAlignmentWindowsPositions finds nothing (CONSENT-correction.cpp:22-25): with
    trimming off such a read must not come back as its own lower-cased copy."""
    fa, paf = make_dataset(tmp_path, 377473700, n_reads=24, glen=7000, rate=0.16)
    prm = dict(min_support=3, max_support=20, window_size=500, mer_size=8, common_kmers=8, min_anchors=2, solid_thresh=2, window_overlap=20, max_msa=10)
    got = correct_reads(fa, paf, None, do_trim=False, windows_per_batch=64, **prm)
    want = oracle_pipeline(fa, paf, do_trim=False, **prm)
    assert got == want and len(got) == 23


def test_short_random_run_of_the_pipeline_fuzzer(monkeypatch, capsys):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_pipeline

    monkeypatch.setattr(sys, "argv", ["fuzz_pipeline.py", "25", "99"])
    assert fuzz_pipeline.main() == 0
    assert "0 differences" in capsys.readouterr().out
