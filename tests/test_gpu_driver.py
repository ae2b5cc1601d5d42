"""GPU: the drop-in at the wrapper level.  bin/CONSENT-correction and bin/CONSENT-polishing are run with the EXACT argument lists of the
reference's wrappers (CONSENT-correct:202, CONSENT-polish:197) and their stdout is compared with the same loop assembled from the
oracle's pieces; the piles are spread over several engines and the FASTA must not change; the shapes the unit tests do not reach
(depth >= 150 end to end, polishing piles of thousands of overlaps, an ONT-profile read set) run through the same path."""
import os
import random
import subprocess

import pytest

import consent_amd as ca
import oracle_lib
from consent_amd.pipeline import correct_reads
from test_gpu_pipeline import COMP, make_dataset, noisy_map, oracle_pipeline
from test_oracle_ref import rand_seq
from test_oracle_stitch import stitch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def fasta(pairs):
    return "".join(f">{n}\n{s}\n" for n, s in pairs)


def run_bin(exe, argv, env=None):
    out = subprocess.run([os.path.join(BIN, exe)] + [str(x) for x in argv], capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout, out.stderr


def test_consent_correction_with_the_wrappers_argv(tmp_path):
    """CONSENT-correct:202 -- `-a $aln -s 3 -S 150 -l 500 -k 9 -c 8 -A 2 -f 4 -m 50 -j $nproc -r $reads -M 150 -p $LRSCf`, wrapper defaults :42-50"""
    fa, paf = make_dataset(tmp_path, 41, n_reads=30)
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 64, "-r", fa, "-M", 150, "-p", str(tmp_path)]
    got, err = run_bin("CONSENT-correction", argv)
    want = oracle_pipeline(fa, paf, min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    assert len(want) > 10
    assert got == fasta(want)  # stdout is pure FASTA, in PAF order
    # the Python face of the same driver takes the same command line
    out = subprocess.run(["python", "-m", "consent_amd.pipeline"] + [str(x) for x in argv], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0 and out.stdout == got, out.stderr[-1500:]


def test_windows_of_1500_bases_run_end_to_end(tmp_path):
    """`-l 1500` (round 6: the reference takes any window size, src/main.cpp:46-47; through round 5 a template beyond 1032 bases stopped its window).  The
    driver tells its engines the window size (cw_configure), the index kernel holds 2048 template k-mers, the re-assembly slices of 1600 bases; with
    k = 12, as anybody would choose for such windows, the FASTA equals the loop assembled from the oracle's pieces."""
    fa, paf = make_dataset(tmp_path, 43, n_reads=40, glen=12000, rate=0.08, read_len=(3000, 7000))
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 1500, "-k", 12, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 1, "-r", fa, "-M", 150, "-p", str(tmp_path)]
    got, err = run_bin("CONSENT-correction", argv)
    want = oracle_pipeline(fa, paf, min_support=3, max_support=150, window_size=1500, mer_size=12, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    assert len(want) > 10
    assert got == fasta(want)
    assert sum(c.isupper() for _, s_ in want for c in s_) > 20000  # windows were corrected, not passed through


def make_polishing_dataset(tmp_path, seed, glen=5200, n_reads=40, rate=0.1, mix=(0.3, 0.3), read_len=(900, 2400), cuts=((0, 2600), (2500, 5200))):
    rng = random.Random(seed)
    genome = rand_seq(rng, glen)
    reads = []
    for i in range(n_reads):
        ln = rng.randrange(*read_len)
        g0 = rng.randrange(0, glen - ln)
        fwd, pos = noisy_map(rng, genome[g0 : g0 + ln], rate, mix)
        rev = rng.random() < 0.5
        reads.append(dict(name=f"r{i}", g0=g0, g1=g0 + ln, pos=pos, rev=rev, seq=fwd[::-1].translate(COMP) if rev else fwd))
    contigs = []
    for c, (a, b) in enumerate(cuts):
        fwd, pos = noisy_map(rng, genome[a:b], 0.03)
        contigs.append(dict(name=f"ctg{c}", g0=a, g1=b, pos=pos, rev=False, seq=fwd))
    fa, ctg, paf = tmp_path / "reads.fa", tmp_path / "contigs.fa", tmp_path / "ovl.paf"
    open(fa, "w").write("".join(f">{r['name']}\n{r['seq']}\n" for r in reads))
    open(ctg, "w").write("".join(f">{c['name']}\n{c['seq']}\n" for c in contigs))

    def span(r, a, b):
        s, e = r["pos"][a - r["g0"]], r["pos"][b - r["g0"]]
        if r["rev"]:
            n = len(r["seq"])
            s, e = n - e, n - s
        return s, max(e, s + 1)

    with open(paf, "w") as f:  # what reformatPAF leaves: the contig is the query, sorted by contig (CONSENT-polish:192-193)
        for q in contigs:
            for t in reads:
                a, b = max(q["g0"], t["g0"]), min(q["g1"], t["g1"])
                if b - a < 300:
                    continue
                qs, qe = span(q, a, b)
                ts, te = span(t, a, b)
                f.write("\t".join(str(x) for x in [q["name"], len(q["seq"]), qs, qe, "-" if t["rev"] else "+", t["name"], len(t["seq"]), ts, te, b - a - 50, b - a, 60]) + "\n")
    return str(ctg), str(fa), str(paf)


def oracle_polish(ctg, fa, paf, *, min_support, max_support, window_size, mer_size, common_kmers, min_anchors, solid_thresh, window_overlap, max_msa):
    o = oracle_lib.oracle()
    ix = ca.ReadIndex(ctg, fa)
    seqs = [ix.sequence(i) for i in range(len(ix.names))]
    prm = ca.Params(mer_size, solid_thresh, common_kmers, min_anchors, max_msa)
    want = []
    for tpl, tpl_len, ov, _ in ca.PafReader(paf, ix, max_support):
        rows = [[tpl_len, int(r[0]), int(r[1]), int(r[5]), int(ix.seq_len[int(r[2])]), int(r[3]), int(r[4]), i] for i, r in enumerate(ov)]
        targets = [seqs[int(r[2])] for r in ov]
        wins = oracle_lib.window_positions(o.cwo_window_positions, tpl_len, rows, min_support, window_size, window_overlap)
        if not wins:
            continue
        piles = [oracle_lib.window_pile(o.cwo_window_pile, rows, seqs[tpl], targets, qb, qe, mer_size) for qb, qe in wins]
        res, _ = oracle_lib.oracle_run(prm, ca.pack_piles(piles), threads=os.cpu_count() or 1)
        final, _ = stitch(seqs[tpl], [res.consensus(w) for w in range(len(piles))], [p[0] if p else "" for p in piles],
                          [res.solid_kmers(w) for w in range(len(piles))], wins, do_trim=False, k=mer_size, wsize=window_size, wover=window_overlap)
        if final:
            want.append((ix.names[tpl], final))
    return want


def test_consent_polishing_with_the_wrappers_argv(tmp_path):
    """CONSENT-polish:197 -- `-a $aln -s 1 -S 20000 -l 500 -k 9 -c 8 -A 2 -f 4 -m 50 -j $nproc -r $contigs -R $reads -M 150 -p $LRSCf`
    (:42-43 minSupport=1, maxSupport=20000): every overlap of the contig is looked at for every window, nothing is trimmed"""
    ctg, fa, paf = make_polishing_dataset(tmp_path, 42)
    argv = ["-a", paf, "-s", 1, "-S", 20000, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", ctg, "-R", fa, "-M", 150, "-p", str(tmp_path)]
    got, _ = run_bin("CONSENT-polishing", argv)
    want = oracle_polish(ctg, fa, paf, min_support=1, max_support=20000, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    assert len(want) == 2 and got == fasta(want)
    assert all(len(s) > 2000 for _, s in want)


def test_flags_without_a_case_in_the_reference_switch_fail_with_the_usage_text():
    """-d -e -w -n are in the getopt string (main.cpp:29) but fall into `default` (:77): usage + EXIT_FAILURE, there and here"""
    for flag in ("-d", "-e", "-w", "-n", "-x"):
        out = subprocess.run([os.path.join(BIN, "CONSENT-correction"), flag, "1"], capture_output=True, text=True)
        assert out.returncode == 1 and "Usage:" in out.stderr and out.stdout == ""
    out = subprocess.run([os.path.join(BIN, "CONSENT-polishing")], capture_output=True, text=True)
    assert out.returncode == 1 and "Usage:" in out.stderr


def test_two_and_three_engines_write_the_same_fasta_as_one(tmp_path):
    """piles sharded over several engines by the job queue (here: engines on the one GPU of the box) -- byte-equal FASTA, same order
    (CONSENT-correction.cpp:100-103); jobs of ~2 piles so that every engine gets work"""
    fa, paf = make_dataset(tmp_path, 43, n_reads=40)
    prm = dict(min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    one = correct_reads(fa, paf, None, devices=[0], windows_per_batch=8, **prm)
    two = correct_reads(fa, paf, None, devices=[0, 0], windows_per_batch=8, **prm)
    three = correct_reads(fa, paf, None, devices=[0, 0, 0], windows_per_batch=3, **prm)
    assert len(one) > 15 and one == two == three
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", fa, "-M", 150, "-p", "x"]
    got, err = run_bin("CONSENT-correction", argv, env={"CW_DEVICES": "0,0", "CW_DRIVER_STATS": "1"})
    assert got == fasta(one)
    assert '"workers": 2' in err  # counters on stderr, stdout stays pure FASTA


def test_eight_logical_devices_write_the_same_fasta_as_one_engine(tmp_path):
    """What an 8-GPU node runs, on the one GPU of the box: with the test aid CW_VIRTUAL_DEVICES=8 (the -DCW_TEST_AIDS build) the
    driver sees eight devices -- `-j 8` gives it eight read-set uploads, its own choice of workers per device (here 2 x 8 = 16 engines), the job
    size and queue of eight devices -- and the FASTA is byte for byte the one-engine FASTA, in PAF order (CONSENT-correction.cpp:77-119)."""
    fa, paf = make_dataset(tmp_path, 45, n_reads=60)
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", fa, "-M", 150, "-p", "x"]
    one, _ = run_bin("CONSENT-correction", argv, env={"CW_DEVICES": "0"})
    aids = {"LD_LIBRARY_PATH": os.path.join(ROOT, "consent_amd", "aids"), "CW_VIRTUAL_DEVICES": "8", "CW_WORKERS_PER_DEVICE": "2", "CW_JOB_WINDOWS": "6", "CW_DRIVER_STATS": "1"}
    eight, err = run_bin("CONSENT-correction", argv, env=aids)
    assert len(one) > 20000 and eight == one
    assert '"workers": 16' in err and '"workers_per_device": 2' in err
    assert all(f'"device": {d},' in err for d in range(8))  # every logical device has a worker in the statistics


def test_polishing_a_contig_sharded_over_engines(tmp_path):
    ctg, fa, paf = make_polishing_dataset(tmp_path, 44)
    prm = dict(min_support=1, max_support=20000, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    one = correct_reads(ctg, paf, None, proof_path=fa, polishing=True, devices=[0], **prm)
    two = correct_reads(ctg, paf, None, proof_path=fa, polishing=True, devices=[0, 0], windows_per_batch=1, **prm)
    assert one == two and len(one) == 2


@pytest.mark.timeout(900)
def test_depth_150_end_to_end(tmp_path):
    """a 160x read set: piles are cut at maxSupport=150 (alignmentPiles.cpp:43-47), every window is a full-depth pile through extraction,
    all POA tiers and the re-assembly"""
    fa, paf = make_dataset(tmp_path, 45, n_reads=230, glen=2600, rate=0.12, mix=(0.3, 0.6), read_len=(1500, 2100))
    prm = dict(min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    ix = ca.ReadIndex(fa)
    deep = sum(1 for _, _, ov, _ in ca.PafReader(paf, ix, 150) if len(ov) == 150)
    assert deep > 100
    got = correct_reads(fa, paf, None, **prm)
    sub_paf = tmp_path / "first.paf"  # the oracle pipeline on the first 12 reads (a CPU minute); the GPU ran all 230
    names = [f"r{i}" for i in range(12)]
    with open(sub_paf, "w") as f:
        for line in open(paf):
            if line.split("\t", 1)[0] in names:
                f.write(line)
    want = oracle_pipeline(fa, str(sub_paf), **prm)
    gd = dict(got)
    assert len(want) >= 8
    for n, s in want:
        assert gd[n] == s, n


def test_polishing_piles_of_thousands_of_overlaps(tmp_path):
    """polishing with maxSupport=20000: one short contig under 2600 reads -- every window job walks all 2600 overlaps (alignmentWindows.cpp:105),
    piles are ~900 deep (bounded by coverage, not by 151: SURVEY 8 intro)"""
    ctg, fa, paf = make_polishing_dataset(tmp_path, 46, glen=1800, n_reads=2600, rate=0.1, read_len=(500, 900), cuts=((0, 1800),))
    prm = dict(min_support=1, max_support=20000, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    ix = ca.ReadIndex(ctg, fa)
    assert max(len(ov) for _, _, ov, _ in ca.PafReader(paf, ix, 20000)) > 2000
    got = correct_reads(ctg, paf, None, proof_path=fa, polishing=True, **prm)
    want = oracle_polish(ctg, fa, paf, **prm)
    assert got == want and len(got) == 1


def test_ont_profile_read_set_end_to_end(tmp_path):
    """config 4 at toy scale: 30x ONT-like reads (12 % errors, sub:ins:del 30:30:40), wrapper defaults (--type ONT only changes minimap2's
    preset, CONSENT-correct:184-188), two engines"""
    fa, paf = make_dataset(tmp_path, 47, n_reads=48, glen=4200, rate=0.12, mix=(0.4, 0.3), read_len=(1800, 3200))
    prm = dict(min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    got = correct_reads(fa, paf, None, devices=[0, 0], windows_per_batch=16, **prm)
    want = oracle_pipeline(fa, paf, **prm)
    assert got == want and len(got) > 30


def _sub_paf(paf, dst, names):
    keep = set(names)
    with open(dst, "w") as f:
        for line in open(paf):
            if line.split("\t", 1)[0] in keep:
                f.write(line)
    return str(dst)


@pytest.mark.timeout(1200)
def test_config4_mid_scale_ont_correction_with_an_oracle_sample(tmp_path):
    """BASELINE configs[3] above toy size: a 200 kbp genome under 30x ONT-profile reads (750 reads of ~8 kbp, ~14 000 windows, several jobs
    over two engines) through bin/CONSENT-correction with the wrapper's argv; 20 of the reads are corrected by the oracle pipeline as well
    (a read's record depends on its own pile only) and must come out byte-identical; the producer's helper threads do not change the FASTA."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_bench as pb

    fa, paf, _, n_reads, n_ovl = pb.generate(str(tmp_path), 200000, 30, "ont")
    assert n_reads == 750 and n_ovl > 30000
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", fa, "-M", 150, "-p", "x"]
    got, err = run_bin("CONSENT-correction", argv, env={"CW_DRIVER_STATS": "1", "CW_DEVICES": "0,0"})
    recs = dict(zip((l[1:] for l in got.split("\n")[0::2] if l), got.split("\n")[1::2]))
    assert len(recs) > 700
    import json
    st = json.loads([l for l in err.splitlines() if l.startswith("{")][-1])
    assert st["windows"] > 10000 and st["piles"] == 750
    one, _ = run_bin("CONSENT-correction", argv, env={"CW_PRODUCER_THREADS": "1", "CW_DEVICES": "0"})
    assert one == got
    names = [f"r{i}" for i in range(0, 750, 38)][:20]
    want = oracle_pipeline(fa, _sub_paf(paf, tmp_path / "sample.paf", names), min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2,
                           solid_thresh=4, window_overlap=50, max_msa=150)
    assert len(want) >= 18
    for n, s in want:
        assert recs[n] == s, n


@pytest.mark.timeout(1200)
def test_config5_ten_contig_polishing_with_an_oracle_sample(tmp_path):
    """BASELINE configs[4] above toy size: ten contigs of 20 kbp (3 % errors) polished with 30x reads, CONSENT-polish:197 argv
    (maxSupport 20000: a window looks at every overlap of its contig); two of the contigs are polished by the oracle as well."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_bench as pb

    fa, paf, ctg, n_reads, n_ovl = pb.generate(str(tmp_path), 200000, 30, "pacbio", polish=10)
    argv = ["-a", paf, "-s", 1, "-S", 20000, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", ctg, "-R", fa, "-M", 150, "-p", "x"]
    got, _ = run_bin("CONSENT-polishing", argv, env={"CW_DEVICES": "0,0"})
    recs = dict(zip((l[1:] for l in got.split("\n")[0::2] if l), got.split("\n")[1::2]))
    assert len(recs) == 10 and all(len(s) > 15000 for s in recs.values())
    want = oracle_polish(ctg, fa, _sub_paf(paf, tmp_path / "sample.paf", ["ctg3", "ctg8"]), min_support=1, max_support=20000, window_size=500, mer_size=9, common_kmers=8,
                         min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    assert [n for n, _ in want] == ["ctg3", "ctg8"]
    for n, s in want:
        assert recs[n] == s, n


def _records(fasta_text):
    lines = fasta_text.split("\n")
    return dict(zip((l[1:] for l in lines[0::2] if l), lines[1::2]))


@pytest.mark.timeout(1800)
def test_config5_the_reference_example_assembly_polished_at_full_size(tmp_path):
    """BASELINE configs[4] on the input it names: the reference's example/rawAssembly.fasta (86 contigs, 3 353 228 bp, 10-154 kbp; here as the
    data fixture tests/golden/rawAssembly.2bit.npz, made by tests/golden/make_assembly_fixture.py) polished with 30x simulated reads through
    bin/CONSENT-polishing with the argv of CONSENT-polish:197 (minSupport 1, maxSupport 20000: every window of a contig walks ALL its
    overlaps, ~600 for the longest contig).  Oracle sample: the longest contig (contig12, 154 070 bp, ~340 windows) and two shorter ones,
    polished by the oracle's own loop, must come out byte-identical; one engine and two engines write the same FASTA."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_bench as pb

    names, contigs = pb.load_contigs(os.path.join(ROOT, "tests", "golden", "rawAssembly.2bit.npz"))
    assert len(names) == 86 and sum(len(c) for c in contigs) == 3353228 and max(len(c) for c in contigs) == 154070
    fa, paf, ctg, n_reads, n_ovl = pb.generate_from_contigs(str(tmp_path), names, contigs, 30, "pacbio")
    assert n_reads > 12000 and n_ovl > 12000
    argv = ["-a", paf, "-s", 1, "-S", 20000, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", ctg, "-R", fa, "-M", 150, "-p", "x"]
    got, err = run_bin("CONSENT-polishing", argv, env={"CW_DEVICES": "0,0", "CW_DRIVER_STATS": "1"})
    st = json.loads([l for l in err.splitlines() if l.startswith("{")][-1])
    assert st["piles"] == 86 and st["windows"] > 7000, st
    recs = _records(got)
    assert list(recs) == names  # every contig, in the order of the PAF (CONSENT-polishing.cpp:122-134)
    assert all(abs(len(recs[n]) - len(c)) < 0.03 * len(c) for n, c in zip(names, contigs))
    one, _ = run_bin("CONSENT-polishing", argv, env={"CW_DEVICES": "0"})
    assert one == got
    longest = names[max(range(86), key=lambda i: len(contigs[i]))]
    sample = [longest, names[0], names[40]]
    want = oracle_polish(ctg, fa, _sub_paf(paf, tmp_path / "sample.paf", sample), min_support=1, max_support=20000, window_size=500, mer_size=9, common_kmers=8,
                         min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150)
    assert sorted(n for n, _ in want) == sorted(sample)
    for n, s_ in want:
        assert recs[n] == s_, n
    assert len(recs[longest]) > 150000


@pytest.mark.timeout(2400)
def test_config4_e_coli_scale_ont_correction_at_full_size_with_an_oracle_sample(tmp_path):
    """BASELINE configs[3] as stated: a 4.6 Mbp genome under 30x ONT-profile reads (17 250 reads of ~8 kbp, ~1.1 M overlaps, ~3.2e5 windows)
    through bin/CONSENT-correction with the wrapper's argv (CONSENT-correct:202); 20 reads spread over the set are corrected by the
    oracle pipeline as well and must come out byte-identical."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_bench as pb

    fa, paf, _, n_reads, n_ovl = pb.generate(str(tmp_path), 4600000, 30, "ont")
    assert n_reads == 17250 and n_ovl > 900000
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 8, "-r", fa, "-M", 150, "-p", "x"]
    got, err = run_bin("CONSENT-correction", argv, env={"CW_DRIVER_STATS": "1", "CW_DEVICES": "0,0"})
    st = json.loads([l for l in err.splitlines() if l.startswith("{")][-1])
    assert st["piles"] == n_reads and st["windows"] > 250000, st
    recs = _records(got)
    assert len(recs) > 0.97 * n_reads
    names = [f"r{i}" for i in range(7, n_reads, n_reads // 20)][:20]
    want = oracle_pipeline(fa, _sub_paf(paf, tmp_path / "sample.paf", names), min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2,
                           solid_thresh=4, window_overlap=50, max_msa=150)
    assert len(want) >= 18
    for n, s_ in want:
        assert recs[n] == s_, n


def test_a_read_or_contig_may_span_extraction_slices_and_engine_runs(tmp_path, aids):
    """One engine call takes at most CW_MAX_BATCH_WINDOWS windows and one extraction call a bounded number of (window, overlap) descriptors, but
    a job -- and a single contig inside it -- may be larger: the worker extracts in slices into one batch, corrects in runs over it and
    re-assembles once.  With the two limits shrunk (test aids) reads and contigs span many slices and runs; the FASTA must not change."""
    fa, paf = make_dataset(tmp_path, 48, n_reads=40)
    argv = ["-a", paf, "-s", 3, "-S", 150, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 1, "-r", fa, "-M", 150, "-p", "x"]
    whole, _ = run_bin("CONSENT-correction", argv)
    assert whole.count(">") > 15
    for env in ({"CW_DRIVER_RUN_WINDOWS": "5"}, {"CW_DRIVER_SLICE_DESC": "60"}, {"CW_DRIVER_RUN_WINDOWS": "3", "CW_DRIVER_SLICE_DESC": "25"}):
        got, _ = run_bin("CONSENT-correction", argv, env=env)
        assert got == whole, env
    ctg, fa2, paf2 = make_polishing_dataset(tmp_path, 49)
    argv = ["-a", paf2, "-s", 1, "-S", 20000, "-l", 500, "-k", 9, "-c", 8, "-A", 2, "-f", 4, "-m", 50, "-j", 1, "-r", ctg, "-R", fa2, "-M", 150, "-p", "x"]
    whole, _ = run_bin("CONSENT-polishing", argv)
    got, _ = run_bin("CONSENT-polishing", argv, env={"CW_DRIVER_RUN_WINDOWS": "2", "CW_DRIVER_SLICE_DESC": "100"})
    assert got == whole and whole.count(">") == 2
