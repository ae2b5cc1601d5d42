"""Turns the reference's example assembly (example/rawAssembly.fasta: 86 contigs, 3 353 228 bp, ACGT only -- the input BASELINE.json
configs[4] and SURVEY 8(d) name for the polishing configuration) into a DATA fixture that travels to the GPU box: contig names, lengths
and the bases packed four per byte (A0 C1 G2 T3, first base in the two high bits), np.savez_compressed.  Run here only (the reference
tree does not exist on the GPU box):

    python tests/golden/make_assembly_fixture.py            # writes tests/golden/rawAssembly.2bit.npz

`load()` is what tests and tools use to get the contigs back as numpy code arrays."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/example/rawAssembly.fasta"
OUT = os.path.join(HERE, "rawAssembly.2bit.npz")


def read_fasta(path):
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
    return names, seqs


def make():
    names, seqs = read_fasta(SRC)
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    lens = np.array([len(s) for s in seqs], np.int64)
    codes = lut[np.frombuffer("".join(seqs).encode(), np.uint8)]
    assert (codes < 4).all(), "the assembly holds only ACGT"
    pad = (-len(codes)) % 4
    c4 = np.concatenate([codes, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    packed = (c4[:, 0] << 6 | c4[:, 1] << 4 | c4[:, 2] << 2 | c4[:, 3]).astype(np.uint8)
    np.savez_compressed(OUT, names=np.array(names), lengths=lens, packed=packed)
    print(f"{len(names)} contigs, {int(lens.sum())} bases ({lens.min()}..{lens.max()}) -> {OUT} ({os.path.getsize(OUT)} bytes)")


def load(path=OUT):
    """-> (names, [codes per contig as uint8 arrays 0..3])"""
    z = np.load(path)
    p = z["packed"]
    codes = np.stack([p >> 6, (p >> 4) & 3, (p >> 2) & 3, p & 3], axis=1).reshape(-1)
    lens = z["lengths"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [str(n) for n in z["names"]], [codes[offs[i] : offs[i + 1]] for i in range(len(lens))]


if __name__ == "__main__":
    make()
    n, c = load()
    names, seqs = read_fasta(SRC)
    assert n == names and all(bytes(np.frombuffer(b"ACGT", np.uint8)[a]).decode() == s for a, s in zip(c, seqs))
    print("round trip ok")
