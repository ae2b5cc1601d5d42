"""CPU: the wrapper plumbing of the library (cw_paf_reformat / cw_paf_explode / cw_paf_merge) against the reference's own
reformatPAF.cpp, explode.cpp and merge.cpp compiled unmodified into oracle/_ref: byte-identical files."""
import glob
import os
import random
import subprocess

import pytest

import consent_amd as ca

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def tool(name):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.skip("oracle/_ref tools not built")
    return p


def make_paf(path, rng, n_reads=15, n_lines=400, shuffle_level=0.3):
    """lines grouped by query, with a share of queries coming back later (what minimap2 does when the index is split)"""
    names = [f"read{i}" for i in range(n_reads)]
    runs = []
    for _ in range(n_lines // 6):
        q = rng.choice(names)
        runs.append([q] * rng.randrange(1, 12))
    lines = []
    for run in runs:
        for q in run:
            t = rng.choice(names)
            f = [q, 1000, rng.randrange(0, 400), rng.randrange(500, 1000), rng.choice("+-"), t, 1200, rng.randrange(0, 400), rng.randrange(500, 1200),
                 rng.randrange(100, 900), rng.randrange(900, 1000), 60, "tp:A:S", f"cm:i:{rng.randrange(99)}"][: rng.choice([12, 13, 14])]
            lines.append("\t".join(str(x) for x in f))
    open(path, "w").write("\n".join(lines) + "\n")
    return names


def test_reformat_matches_reference(tmp_path):
    rng = random.Random(1)
    src = str(tmp_path / "a.paf")
    make_paf(src, rng)
    subprocess.check_call([tool("ref_reformatPAF"), src, str(tmp_path / "ref.paf")])
    ca.paf_reformat(src, str(tmp_path / "got.paf"))
    assert open(tmp_path / "got.paf", "rb").read() == open(tmp_path / "ref.paf", "rb").read()
    # swapping twice gives the original back
    ca.paf_reformat(str(tmp_path / "got.paf"), str(tmp_path / "back.paf"))
    assert open(tmp_path / "back.paf", "rb").read() == open(src, "rb").read()


@pytest.mark.parametrize("seed", [2, 3, 4])
def test_explode_and_merge_match_reference(tmp_path, seed):
    rng = random.Random(seed)
    src = str(tmp_path / "a.paf")
    names = make_paf(src, rng, n_reads=10 + seed, n_lines=300 + 100 * seed)
    subprocess.check_call([tool("ref_explode"), src, str(tmp_path / "refx")])
    got = ca.paf_explode(src, str(tmp_path / "gotx"))
    ref_chunks = sorted(glob.glob(str(tmp_path / "refx_*")), key=lambda p: int(p.rsplit("_", 1)[1]))
    assert len(got) == len(ref_chunks) > 1
    for g, r in zip(got, ref_chunks):
        assert open(g, "rb").read() == open(r, "rb").read()
    # merge: headers in an order of our choosing (the wrapper takes them from the read file), with names that have no overlap at all
    order = names[:] + ["ghost"]
    rng.shuffle(order)
    hdr = str(tmp_path / "headers.txt")
    open(hdr, "w").write("".join(f">{n}\n" for n in order))
    subprocess.check_call([tool("ref_merge"), str(tmp_path / "ref_merged.paf"), hdr] + ref_chunks)
    ca.paf_merge(str(tmp_path / "got_merged.paf"), hdr, got)
    a, b = open(tmp_path / "got_merged.paf", "rb").read(), open(tmp_path / "ref_merged.paf", "rb").read()
    assert a == b and len(a) > 0


@pytest.mark.parametrize("name,text", [
    # (a last line WITHOUT a newline is not a case: the reference program never ends on it -- getline at end-of-file leaves the line in place)
    ("stops_at_empty_line", "a\t1\nb\t2\n\na\t3\nc\t4\n"),           # nothing behind the empty line is read
    ("empty_name_joins_its_neighbours", "a\t1\n\t2\nb\t3\na\t4\nb\t5\n"),  # the reference takes "" for "no previous name"
    ("empty_file", ""),
    ("one_line", "solo\t1\n"),
    ("no_tabs", "x\nx\ny\nx\n"),
    ("crlf", "a\t1\r\na\t2\r\nb\t3\r\n\r\na\t4\r\n"),           # "\r" is a line of one character, not an empty one
    ("name_back_twice", "a\t1\nb\t2\na\t3\nb\t4\na\t5\n"),
])
def test_explode_edge_cases_match_reference(tmp_path, name, text):
    """the one-pass slice writer of cw_paf_explode against the reference program on the inputs where line handling shows"""
    src = str(tmp_path / "e.paf")
    open(src, "wb").write(text.encode())
    subprocess.check_call([tool("ref_explode"), src, str(tmp_path / "refx")])
    got = ca.paf_explode(src, str(tmp_path / "gotx"))
    ref_chunks = sorted(glob.glob(str(tmp_path / "refx_*")), key=lambda p: int(p.rsplit("_", 1)[1]))
    assert len(got) == len(ref_chunks) >= 1, name
    for g, r in zip(got, ref_chunks):
        assert open(g, "rb").read() == open(r, "rb").read(), (name, g)


def test_wrapper_errors(tmp_path):
    with pytest.raises(ca.EngineError):
        ca.paf_reformat(str(tmp_path / "missing.paf"), str(tmp_path / "o.paf"))
    short = str(tmp_path / "short.paf")
    open(short, "w").write("a\tb\tc\n")
    with pytest.raises(ca.EngineError):
        ca.paf_reformat(short, str(tmp_path / "o.paf"))
