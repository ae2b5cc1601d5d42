"""Synthetic end-to-end workloads for the wrapper-level configurations of BASELINE.json (reads.fasta, the genome and minimap2
are absent, so reads AND overlaps are synthesised; the PAF is written from ground-truth coordinates with the 12 columns Overlap.h
reads) and a timed run of the native driver through bin/CONSENT-correction / bin/CONSENT-polishing with the wrappers' own flags.

  config 1  --genome 460000  --cov 10 --profile pacbio                 (example/reads.fasta scale: 10x sim PacBio)
  config 4  --genome 4600000 --cov 30 --profile ont                    (E. coli 30x, CONSENT-correct --type ONT)
  config 5  --contigs tests/golden/rawAssembly.2bit.npz --cov 30       (CONSENT-polish on the reference's example assembly: 86 contigs, 3.35 Mbp,
                                                                        + 30x reads simulated from it; --polish N --genome G = N synthetic contigs)
GPU box only.  CW_BENCH_GPUS = number of GPUs handed to -j."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = np.zeros(256, np.uint8)
for a_, b_ in zip(b"ACGT", b"TGCA"):
    COMP[a_] = b_
MIX = {"pacbio": (0.3, 0.6), "ont": (0.4, 0.3)}  # (deletion, insertion) shares of the errors; the rest are substitutions: 10:60:30 and 30:30:40 sub:ins:del


def noisy(rng, seg, rate, mix=(0.3, 0.3)):
    """numpy version of the test generator: returns the noisy copy and the position map (len(seg)+1 entries)"""
    n = len(seg)
    x = rng.random(n)
    d_, i_ = mix
    dele = x < rate * d_
    ins = (x >= rate * d_) & (x < rate * (d_ + i_))
    sub = x < rate
    base = np.where(sub, rng.integers(0, 4, n), seg)
    emit = (~dele).astype(np.int64) + ins.astype(np.int64)
    pos = np.concatenate([[0], np.cumsum(emit)])
    out = np.zeros(pos[-1], np.uint8)
    keep = ~dele
    out[pos[:-1][keep] + ins[keep]] = base[keep]
    out[pos[:-1][ins]] = rng.integers(0, 4, int(ins.sum()))
    return out, pos


def span(r, a, b):
    """coordinates of genome [a, b) on sequence r's own forward strand"""
    s, e = int(r[2][a - r[0]]), int(r[2][b - r[0]])
    if r[3]:
        n = len(r[4])
        s, e = n - e, n - s
    return s, max(e, s + 1)


def generate(d, glen, cov, profile, polish=0, rlen=8000, seed=7):
    """Writes reads.fa, ovl.paf (and contigs.fa when polishing) under d; returns (reads, paf, contigs or None, n_reads, n_overlaps)."""
    mix = MIX[profile]
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, glen)
    n_reads = glen * cov // rlen
    reads = []
    for i in range(n_reads):
        ln = int(min(glen - 1, max(1000, rng.lognormal(np.log(rlen), 0.35))))
        g0 = int(rng.integers(0, glen - ln))
        s, pos = noisy(rng, genome[g0 : g0 + ln], 0.12, mix)
        reads.append((g0, g0 + ln, pos, bool(rng.random() < 0.5), s))
    os.makedirs(d, exist_ok=True)
    fa, paf, ctg_fa = os.path.join(d, "reads.fa"), os.path.join(d, "ovl.paf"), os.path.join(d, "contigs.fa")
    lut = np.frombuffer(b"ACGT", np.uint8)
    with open(fa, "w") as f:
        for i, (g0, g1, pos, rev, s) in enumerate(reads):
            a = lut[s]
            if rev:
                a = COMP[a[::-1]]
            f.write(f">r{i}\n{a.tobytes().decode()}\n")
    order = np.argsort([r[0] for r in reads])
    starts = np.array([reads[i][0] for i in order])
    n_lines = 0
    if polish:
        cuts = np.linspace(0, glen, polish + 1).astype(int)
        contigs = []
        for c in range(polish):
            s, pos = noisy(rng, genome[cuts[c] : cuts[c + 1]], 0.03, (0.3, 0.3))
            contigs.append((int(cuts[c]), int(cuts[c + 1]), pos, False, s))
        with open(ctg_fa, "w") as f:
            for c, q in enumerate(contigs):
                f.write(f">ctg{c}\n{lut[q[4]].tobytes().decode()}\n")
        with open(paf, "w") as f:  # what `sort -k6,6 | reformatPAF` leaves: the contig is the query, one run of lines per contig
            for c, q in enumerate(contigs):
                for ti in order[np.searchsorted(starts, q[0] - 4 * rlen) :]:
                    t = reads[ti]
                    if t[0] >= q[1] - 300:
                        break
                    a, b = max(q[0], t[0]), min(q[1], t[1])
                    if b - a < 300:
                        continue
                    qs, qe = span(q, a, b)
                    ts, te = span(t, a, b)
                    f.write(f"ctg{c}\t{len(q[4])}\t{qs}\t{qe}\t{'-' if t[3] else '+'}\tr{ti}\t{len(t[4])}\t{ts}\t{te}\t{int((b - a) * 0.8)}\t{b - a}\t60\n")
                    n_lines += 1
    else:
        with open(paf, "w") as f:
            for qi in range(n_reads):
                q = reads[qi]
                for ti in order[np.searchsorted(starts, q[0] - 4 * rlen) :]:
                    t = reads[ti]
                    if ti == qi or t[1] <= q[0] + 500:
                        continue
                    if t[0] >= q[1] - 500:
                        break
                    a, b = max(q[0], t[0]), min(q[1], t[1])
                    if b - a < 500:
                        continue
                    qs, qe = span(q, a, b)
                    ts, te = span(t, a, b)
                    f.write(f"r{qi}\t{len(q[4])}\t{qs}\t{qe}\t{'+' if q[3] == t[3] else '-'}\tr{ti}\t{len(t[4])}\t{ts}\t{te}\t{int((b - a) * 0.76)}\t{b - a}\t60\n")
                    n_lines += 1
    return fa, paf, (ctg_fa if polish else None), n_reads, n_lines


def generate_from_contigs(d, names, contigs, cov, profile, rlen=8000, seed=11, truth_err=0.03):
    """Polishing data for GIVEN contigs (BASELINE configs[4]: example/rawAssembly.fasta, the fixture tests/golden/rawAssembly.2bit.npz):
    every contig is taken as a draft of a truth sequence that differs from it by `truth_err` (truth = noisy copy of the contig, position
    map kept), `cov`x reads of the profile are drawn from the truth, and the PAF is written from the ground-truth coordinates as
    `minimap2 | sort -k6,6 | reformatPAF` leaves it (CONSENT-polish:189-193: the contig is the query, one run of lines per contig).
    Writes contigs.fa, reads.fa, ovl.paf under d; returns (reads, paf, contigs, n_reads, n_overlaps)."""
    mix = MIX[profile]
    rng = np.random.default_rng(seed)
    os.makedirs(d, exist_ok=True)
    fa, paf, ctg_fa = os.path.join(d, "reads.fa"), os.path.join(d, "ovl.paf"), os.path.join(d, "contigs.fa")
    lut = np.frombuffer(b"ACGT", np.uint8)
    n_reads = n_lines = 0
    with open(ctg_fa, "w") as fc, open(fa, "w") as fr, open(paf, "w") as fp:
        for name, ctg in zip(names, contigs):
            fc.write(f">{name}\n{lut[ctg].tobytes().decode()}\n")
            truth, pos_ct = noisy(rng, ctg, truth_err, (0.3, 0.3))  # contig position -> truth position
            tl = len(truth)
            inv = np.minimum(np.searchsorted(pos_ct, np.arange(tl + 1), side="left"), len(ctg))  # truth position -> contig position
            q = (0, tl, inv, False, ctg)
            fl = min(4000, tl // 4)  # the truth continues beyond the draft's ends: reads hang over them and overlap the contig partially
            ext = np.concatenate([rng.integers(0, 4, fl).astype(truth.dtype), truth, rng.integers(0, 4, fl).astype(truth.dtype)])
            lines = []
            for _ in range(max(1, (tl + fl) * cov // rlen)):
                ln = int(min(tl - 1, max(1000, rng.lognormal(np.log(rlen), 0.35))))
                g0 = int(rng.integers(0, len(ext) - ln)) - fl  # truth coordinate of the read's first base (negative: in the left flank)
                s_, pos = noisy(rng, ext[g0 + fl : g0 + fl + ln], 0.12, mix)
                rev = bool(rng.random() < 0.5)
                a_ = lut[s_]
                if rev:
                    a_ = COMP[a_[::-1]]
                rid = f"r{n_reads}"
                fr.write(f">{rid}\n{a_.tobytes().decode()}\n")
                n_reads += 1
                a, b = max(0, g0), min(tl, g0 + ln)
                if b - a < 500:
                    continue
                t = (g0, g0 + ln, pos, rev, s_)
                qs, qe = span(q, a, b)
                ts, te = span(t, a, b)
                lines.append((a, f"{name}\t{len(ctg)}\t{qs}\t{qe}\t{'-' if rev else '+'}\t{rid}\t{len(s_)}\t{ts}\t{te}\t{int((b - a) * 0.8)}\t{b - a}\t60\n"))
            for _, l in lines:
                fp.write(l)
            n_lines += len(lines)
    return fa, paf, ctg_fa, n_reads, n_lines


def load_contigs(path):
    """contigs from the 2-bit fixture (tests/golden/make_assembly_fixture.py) or from a FASTA file: (names, [code arrays])"""
    if path.endswith(".npz"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_assembly_fixture as maf

        return maf.load(path)
    lutc = np.full(256, 3, np.uint8)  # utils.cpp:21-32: everything but A, C, G is T
    for i_, c_ in enumerate(b"ACG"):
        lutc[c_] = i_
        lutc[c_ + 32] = i_
    names, seqs, cur = [], [], []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:].split(" ")[0])
            cur = []
        elif line:
            cur.append(line)
    if names:
        seqs.append("".join(cur))
    return names, [lutc[np.frombuffer(x.encode(), np.uint8)] for x in seqs]


def replicate(fa, paf, copies):
    """`copies` independent copies of a read-correction data set (reads r<i> become c<k>r<i>, piles refer to their own copy): the same
    piles `copies` times over -- a larger job list for the scaling runs without minutes of simulation.  Returns the new (reads, paf)."""
    if copies <= 1:
        return fa, paf
    d = os.path.dirname(fa)
    fa2, paf2 = os.path.join(d, f"reads_x{copies}.fa"), os.path.join(d, f"ovl_x{copies}.paf")
    rd, pf = open(fa, "rb").read(), open(paf, "rb").read()
    with open(fa2, "wb") as f, open(paf2, "wb") as g:
        for k in range(copies):
            tag = b"c%d" % k
            f.write(rd.replace(b">r", b">" + tag + b"r"))
            g.write(tag + pf.replace(b"\tr", b"\t" + tag + b"r").replace(b"\nr", b"\n" + tag + b"r"))
    return fa2, paf2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=400000)
    ap.add_argument("--cov", type=int, default=30)
    ap.add_argument("--profile", choices=sorted(MIX), default="pacbio")
    ap.add_argument("--polish", type=int, default=0, help="number of contigs: polish them with the reads instead of correcting the reads")
    ap.add_argument("--contigs", default="", help="polish THESE contigs (FASTA, or the 2-bit fixture tests/golden/rawAssembly.2bit.npz = BASELINE configs[4]) with reads simulated from them")
    ap.add_argument("--read-len", type=int, default=8000)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--copies", type=int, default=1, help="read correction only: this many independent copies of the data set in one run")
    args = ap.parse_args()
    d = os.environ.get("CW_KEEP_DATA") or tempfile.mkdtemp()
    if args.contigs:
        names_, ctgs_ = load_contigs(args.contigs)
        fa, paf, ctg_fa, n_reads, n_lines = generate_from_contigs(d, names_, ctgs_, args.cov, args.profile, args.read_len)
        args.polish, args.genome = len(names_), int(sum(len(c) for c in ctgs_))
    else:
        fa, paf, ctg_fa, n_reads, n_lines = generate(d, args.genome, args.cov, args.profile, args.polish, args.read_len)
    if not args.polish:
        fa, paf = replicate(fa, paf, args.copies)
    print(f"data set: genome {args.genome}, {n_reads} {args.profile} reads ({os.path.getsize(fa) / 1e6:.0f} MB), {n_lines} overlaps ({os.path.getsize(paf) / 1e6:.0f} MB)"
          + (f", {args.polish} contigs" if args.polish else "") + (f", x{args.copies}" if args.copies > 1 else ""), flush=True)

    gpus = os.environ.get("CW_BENCH_GPUS", "1")
    if args.polish:  # CONSENT-polish:197
        argv = [os.path.join(ROOT, "bin", "CONSENT-polishing"), "-a", paf, "-s", "1", "-S", "20000", "-l", "500", "-k", "9", "-c", "8", "-A", "2", "-f", "4", "-m", "50", "-j", gpus,
                "-r", ctg_fa, "-R", fa, "-M", "150", "-p", "x"]
    else:  # CONSENT-correct:202
        argv = [os.path.join(ROOT, "bin", "CONSENT-correction"), "-a", paf, "-s", "3", "-S", "150", "-l", "500", "-k", "9", "-c", "8", "-A", "2", "-f", "4", "-m", "50", "-j", gpus,
                "-r", fa, "-M", "150", "-p", "x"]
    for rep in range(args.reps):
        t0 = time.perf_counter()
        out = subprocess.run(argv, capture_output=True, text=True, env=dict(os.environ, CW_DRIVER_STATS="1", CW_DRIVER_TIMING=os.environ.get("CW_DRIVER_TIMING", "1"), CW_ON_CAPACITY="skip"))
        wall = time.perf_counter() - t0
        assert out.returncode == 0, out.stderr[-2000:]
    st = json.loads([l for l in out.stderr.splitlines() if l.startswith("{")][-1])
    lines = out.stdout.split("\n")
    seqs = lines[1::2]
    bases = sum(len(x) for x in seqs)
    up = sum(sum(c.isupper() for c in x) for x in seqs[:2000])
    print(f"{'polished' if args.polish else 'corrected'} {len(seqs)} sequences, {bases} bases ({up / max(sum(len(x) for x in seqs[:2000]), 1):.3f} upper case in the first 2000) in {wall:.2f} s wall (process start to exit)")
    for l in out.stderr.splitlines():
        if l.startswith("[job") or l.startswith("[debug"):
            print(l)
    skipped = [l for l in out.stderr.splitlines() if "left out" in l]
    if skipped:
        print(f"{len(skipped)} job(s) left sequences out for an engine capacity: {skipped[0][:300]}")
    print(json.dumps(st))


if __name__ == "__main__":
    main()
