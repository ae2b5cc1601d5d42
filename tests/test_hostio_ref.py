"""CPU: the host feeders of the library (cw_index_reads, cw_paf_next_pile) pinned against the reference's own
indexReads / getNextReadPile / Overlap parser compiled unmodified into oracle/_ref."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from test_oracle_ref import rand_seq


def need_ref():
    r = oracle_lib.ref()
    if r is None:
        pytest.skip("oracle/_ref not built")
    return r


def write_reads(path, rng, n, fastq=False, multiline=False, weird=False):
    names, seqs = [], []
    with open(path, "w") as f:
        for i in range(n):
            name = f"read_{i}"
            alphabet = "ACGTacgtNnRy" if weird else "ACGT"
            s = "".join(rng.choice(alphabet) for _ in range(rng.randrange(20, 400)))
            names.append(name)
            seqs.append(s)
            head = ("@" if fastq else ">") + name + (" some description 123" if i % 3 == 0 else "")
            lines = [s[x : x + 60] for x in range(0, len(s), 60)] if multiline else [s]
            f.write(head + "\n" + "\n".join(lines) + "\n")
            if fastq:
                f.write("+\n" + "\n".join("I" * len(x) for x in lines) + "\n")
    return names, seqs


@pytest.mark.parametrize("fastq,multiline,weird", [(False, False, False), (False, True, True), (True, False, False), (True, True, True)])
def test_index_reads_matches_reference(tmp_path, fastq, multiline, weird):
    r = need_ref()
    rng = random.Random(11 + fastq * 2 + multiline)
    path = str(tmp_path / ("reads.fq" if fastq else "reads.fa"))
    names, seqs = write_reads(path, rng, 25, fastq, multiline, weird)
    ix = ca.ReadIndex(path)
    assert sorted(ix.names) == sorted(names)
    buf = np.zeros(4096, np.uint8)
    ln = C.c_uint32()
    for name in names:
        n_ref = r.ref_index_reads_lookup(path.encode(), name.encode(), C.c_void_p(buf.ctypes.data), len(buf), C.byref(ln))
        assert n_ref == len(names)
        i = ix.find(name)
        assert i >= 0 and ix.seq_len[i] == ln.value
        assert ix.sequence(i) == buf[: ln.value].tobytes().decode()
    assert ix.find("no_such_read") == -1


def test_index_reads_duplicate_name_keeps_the_last_record(tmp_path):
    r = need_ref()
    path = str(tmp_path / "dup.fa")
    open(path, "w").write(">a x\nACGT\n>b\nGGGTTT\n>a\nTTTTTTTTTT\n")
    ix = ca.ReadIndex(path)
    buf = np.zeros(64, np.uint8)
    ln = C.c_uint32()
    assert r.ref_index_reads_lookup(path.encode(), b"a", C.c_void_p(buf.ctypes.data), 64, C.byref(ln)) == 2
    assert len(ix.names) == 2 and ix.sequence(ix.find("a")) == buf[: ln.value].tobytes().decode() == "TTTTTTTTTT"


def write_paf(path, rng, names, lens, n_piles, blank_lines=False):
    rows = []
    with open(path, "w") as f:
        for p in range(n_piles):
            q = rng.randrange(len(names))
            for _ in range(rng.randrange(1, 40)):
                t = rng.randrange(len(names))
                qs = rng.randrange(0, lens[q] - 5)
                qe = rng.randrange(qs + 1, lens[q] + 1)
                ts = rng.randrange(0, lens[t] - 5)
                te = rng.randrange(ts + 1, lens[t] + 1)
                rm = rng.choice([10, 20, 20, 20, 35, 50, rng.randrange(1, 300)])  # many ties: the sort is unstable
                row = [names[q], lens[q], qs, qe, rng.choice("+-"), names[t], lens[t], ts, te, rm, rng.randrange(rm, rm + 50), rng.randrange(0, 61)]
                f.write("\t".join(str(x) for x in row) + "\tcm:i:5\n")
                rows.append(row)
            if blank_lines and p % 3 == 1:
                f.write("\n")
    return rows


@pytest.mark.parametrize("block", [0, 64, 700])
@pytest.mark.parametrize("max_support,blank", [(150, False), (7, False), (150, True), (1, False)])
def test_paf_piles_match_reference(tmp_path, monkeypatch, max_support, blank, block):
    """block > 0: the mapped file is cut into blocks of that many bytes (piles straddle them) parsed by five threads ahead of the reader"""
    r = need_ref()
    if block:
        monkeypatch.setenv("CW_PAF_BLOCK", str(block))
        monkeypatch.setenv("CW_HOST_THREADS", "5")
    rng = random.Random(5 + max_support)
    fa = str(tmp_path / "reads.fa")
    names, seqs = write_reads(fa, rng, 12)
    paf = str(tmp_path / "ovl.paf")
    write_paf(paf, rng, names, [len(s) for s in seqs], 30, blank)
    cap = 4096
    pile_id = np.zeros(cap, np.uint32)
    fields = np.zeros((cap, 8), np.uint32)
    qn = np.zeros((cap, 32), np.uint8)
    tn = np.zeros((cap, 32), np.uint8)
    n = r.ref_paf_piles(paf.encode(), max_support, C.c_void_p(pile_id.ctypes.data), C.c_void_p(fields.ctypes.data), C.c_void_p(qn.ctypes.data),
                        C.c_void_p(tn.ctypes.data), 32, cap)
    assert n > 0
    cstr = lambda row: bytes(row).split(b"\0")[0].decode()
    ix = ca.ReadIndex(fa)
    got = []
    for p, (tpl, tlen, ov, rm) in enumerate(ca.PafReader(paf, ix, max_support)):
        for o, m in zip(ov, rm):
            got.append((p, ix.names[tpl], tlen, int(o[0]), int(o[1]), ix.names[int(o[2])], int(o[3]), int(o[4]), int(o[5]), int(m)))
    want = [(int(pile_id[i]), cstr(qn[i]), int(fields[i][6]), int(fields[i][0]), int(fields[i][1]), cstr(tn[i]), int(fields[i][2]), int(fields[i][3]),
             int(fields[i][4]), int(fields[i][5])) for i in range(n)]
    assert got == want


def test_paf_errors_are_loud(tmp_path):
    fa = str(tmp_path / "r.fa")
    open(fa, "w").write(">a\nACGTACGTAC\n>b\nACGTACGTAC\n")
    ix = ca.ReadIndex(fa)
    bad = str(tmp_path / "bad.paf")
    open(bad, "w").write("a\t10\t0\t5\t+\tzzz\t10\t0\t5\t5\t5\t60\n")
    with pytest.raises(ca.EngineError):
        list(ca.PafReader(bad, ix))
    short = str(tmp_path / "short.paf")
    open(short, "w").write("a\t10\t0\t5\t+\tb\n")
    with pytest.raises(ca.EngineError):
        list(ca.PafReader(short, ix))
    # a malformed line is reported when the reader reaches it, not earlier: the piles before it are delivered (blocks parsed ahead)
    late = str(tmp_path / "late.paf")
    good = "a\t10\t0\t5\t+\tb\t10\t0\t5\t5\t5\t60\n"
    open(late, "w").write(good + good.replace("a\t10", "b\t10", 1).replace("\tb\t", "\ta\t") + "a\tx\n" + good)
    os.environ["CW_PAF_BLOCK"] = "64"
    try:
        it = iter(ca.PafReader(late, ix))
        assert ix.names[next(it)[0]] == "a"
        with pytest.raises(ca.EngineError):  # pile "b" ends at the malformed line: a sequential getline loop fails there too
            next(it)
    finally:
        del os.environ["CW_PAF_BLOCK"]
    empty = str(tmp_path / "empty.paf")
    open(empty, "w").write("")
    assert list(ca.PafReader(empty, ix)) == []
    nonl = str(tmp_path / "nonl.paf")
    open(nonl, "w").write(good.rstrip("\n"))
    assert len(list(ca.PafReader(nonl, ix))) == 1
    with pytest.raises(ca.EngineError):
        ca.ReadIndex(str(tmp_path / "missing.fa"))


@pytest.mark.parametrize("fastq", [False, True])
def test_index_reads_parallel_path_on_a_larger_file(tmp_path, fastq):
    """more than 64 records and more than 1 MB: the records are packed by several threads; every read still equals the reference's,
    including lower case, non-ACGT letters, duplicate names and a '\\r' kept as a base"""
    r = need_ref()
    rng = random.Random(77 + fastq)
    path = str(tmp_path / ("big.fq" if fastq else "big.fa"))
    names, seqs = [], []
    with open(path, "w") as f:
        for i in range(700):
            name = f"r{i % 690}"  # ten names come back: the later record wins
            s = "".join(rng.choice("ACGTacgtNn") for _ in range(rng.randrange(800, 3200)))
            if i == 5:
                s = s[:100] + "\r" + s[100:]
            names.append(name)
            seqs.append(s)
            lines = [s[x : x + 70] for x in range(0, len(s), 70)] if i % 2 else [s]
            f.write(("@" if fastq else ">") + name + (" desc" if i % 5 == 0 else "") + "\n" + "\n".join(lines))
            if fastq:
                f.write("\n+\n" + "\n".join("#" * len(x) for x in lines))
            f.write("\n")  # (the reference itself never returns from a file whose last line lacks its newline)
    assert os.path.getsize(path) > (1 << 20)
    ix = ca.ReadIndex(path)
    assert len(ix.names) == 690
    buf = np.zeros(8192, np.uint8)
    ln = C.c_uint32()
    for name in sorted(set(names), key=lambda x: int(x[1:]))[::7] + ["r5", "r0", "r9", "r689"]:
        n_ref = r.ref_index_reads_lookup(path.encode(), name.encode(), C.c_void_p(buf.ctypes.data), len(buf), C.byref(ln))
        assert n_ref == 690
        i = ix.find(name)
        assert i >= 0 and ix.seq_len[i] == ln.value, name
        assert ix.sequence(i) == buf[: ln.value].tobytes().decode(), name


def test_index_reads_without_a_final_newline_simply_ends(tmp_path):
    """(no reference comparison: the reference spins on such a file) the last record is complete, FASTA and FASTQ"""
    pa, pq = str(tmp_path / "a.fa"), str(tmp_path / "a.fq")
    open(pa, "w").write(">x\nACGT\nGG\n>y z\nTTTA")
    open(pq, "w").write("@x\nACGT\n+\nIIII\n@y\nTTTA\n+\nIIII")
    for p in (pa, pq):
        ix = ca.ReadIndex(p)
        assert ix.names == ["x", "y"] and ix.sequence(1) == "TTTA"
    assert ca.ReadIndex(pa).sequence(0) == "ACGTGG"
