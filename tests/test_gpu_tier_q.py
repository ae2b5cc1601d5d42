"""GPU: tiers Q and H (four / two POA tasks per wave, cw_poa_q.h) on the recorded-decision path of round 5, against the oracle.

The piles are built so that every window has ONE variable region between two stretches every sequence shares: the anchor chain runs
through the shared stretches, the region between them is a POA task whose members are at most 31 bases long -- tier Q's -- and what goes
on inside that region is chosen to reach the parts of the kernel the synthetic PacBio piles seldom do:
  * nodes with more than three in-edges (in-edge ordinal 3 = "fourth or later": decided from kept DP rows),
  * predecessors further back than the LDS ring of rows reaches (long insertions and deletions: kept rows read back from the slab),
  * graphs that outgrow the tier's 64 nodes or 184 edges (handed to tier S and redone there),
  * tasks of very different sizes side by side in one wave (the four rows of a wave advance in lock step).
Bit-exact: consensus bytes, status, solid set."""
import random

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib

pytestmark = pytest.mark.gpu
PRM = (9, 4, 8, 2, 150)


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def assert_same(got, exp, n, what=""):
    for w in range(n):
        assert int(got.status[w]) == int(exp.status[w]), f"{what} window {w}: status {got.status[w]} != {exp.status[w]}"
        assert got.consensus(w) == exp.consensus(w), f"{what} window {w}: consensus differs"
        assert np.array_equal(got.solid_kmers(w), exp.solid_kmers(w)), f"{what} window {w}: solid set differs"


def region_pile(rng, depth, mid_len, kind, cap=22):
    """left (shared) + a variable middle + right (shared); the last anchor of `left` starts 9 bases before the middle"""
    left, right = rand_seq(rng, 40), rand_seq(rng, 40)
    mid = rand_seq(rng, mid_len)
    pile = []
    for s in range(depth):
        m = list(mid)
        if kind == "fan":  # many different bases (and gaps) at the same places: nodes with four, five, six in-edges
            for p in rng.sample(range(len(m)), min(len(m), 4)):
                x = rng.random()
                if x < 0.25:
                    m[p] = ""
                elif x < 0.9:
                    m[p] = rng.choice("ACGT")
                else:
                    m[p] = m[p] + rng.choice("ACGT")
        elif kind == "far":  # long insertions and deletions: predecessors ten to twenty ranks back
            x = rng.random()
            if x < 0.35 and len(m) >= 4:
                p = rng.randrange(1, len(m) - 2)
                m[p] = m[p] + rand_seq(rng, rng.randrange(9, 14))
            elif x < 0.6 and len(m) > 14:
                p = rng.randrange(1, len(m) - 11)
                for q in range(p, p + rng.randrange(9, 11)):
                    m[q] = ""
            elif x < 0.8:
                p = rng.randrange(len(m))
                m[p] = rng.choice("ACGT")
        elif kind == "grow":  # every member its own middle: the graph grows by the member, past 64 nodes
            m = list(rand_seq(rng, rng.randrange(max(1, mid_len - 6), mid_len + 1)))
        else:  # "noise": 12 % errors of every kind
            out = []
            for c in m:
                x = rng.random()
                if x < 0.04:
                    continue
                if x < 0.08:
                    out.append(rng.choice("ACGT"))
                out.append(rng.choice("ACGT") if x < 0.12 else c)
            m = out
        ms = "".join(m)[:cap]  # member = 9 bases of the left anchor + the middle: at most 31 (tier Q; cap 54: at most 63, tier H)
        pile.append(left + ms + right)
    return pile


@pytest.fixture(scope="module")
def engine():
    e = ca.Engine(ca.Params(*PRM))
    yield e
    e.close()


@pytest.mark.parametrize("kind", ["noise", "fan", "far", "grow"])
def test_regions_of_tier_q_shapes_match_the_oracle(engine, kind):
    rng = random.Random({"noise": 11, "fan": 12, "far": 13, "grow": 14}[kind])
    piles = []
    for _ in range(96):
        depth = rng.choice((3, 5, 8, 12, 20, 33, 60, 120))
        piles.append(region_pile(rng, depth, rng.randrange(2, 23), kind))
    hb = ca.pack_piles(piles)
    got = engine.run(hb)
    ctr, _ = engine.profile()
    exp, _ = oracle_lib.oracle_run(ca.Params(*PRM), hb, threads=8)
    assert_same(got, exp, len(piles), kind)
    assert int(ctr[6]) >= 40, ctr[6:12]  # the regions were tier Q's tasks (list 0)
    if kind == "grow":
        assert int(ctr[18]) > 0, ctr[18:24]  # some outgrew it and were redone in tier S


@pytest.mark.parametrize("kind", ["noise", "fan", "far", "grow"])
def test_regions_of_tier_h_shapes_match_the_oracle(aids, monkeypatch, kind):
    """the same shapes with middles of up to 54 bases: members of 32 .. 63 bases, tier H's (two tasks per wave, code words in the slab; the
    tier is opt-in -- measured no faster than tier S -- and switched on with the test-aid build's CW_TIER_H)"""
    monkeypatch.setenv("CW_TIER_H", "2")
    engine = ca.Engine(ca.Params(*PRM))
    rng = random.Random({"noise": 31, "fan": 32, "far": 33, "grow": 34}[kind])
    piles = []
    for _ in range(96):
        depth = rng.choice((3, 5, 8, 12, 20, 33, 60))
        piles.append(region_pile(rng, depth, rng.randrange(24, 55), kind, cap=54))
    hb = ca.pack_piles(piles)
    got = engine.run(hb)
    ctr, _ = engine.profile()
    exp, _ = oracle_lib.oracle_run(ca.Params(*PRM), hb, threads=8)
    assert_same(got, exp, len(piles), kind)
    engine.close()
    assert int(ctr[11]) >= 3, ctr[6:12]  # regions were tier H's tasks (list 5; deep piles and long insertions route past it)
    if kind == "grow":
        assert int(ctr[21]) > 0, ctr[18:24]  # some outgrew its 128 nodes / 250 edges and were redone in tier L


def test_unlike_tasks_share_a_wave(engine):
    """a window with a 2-member region beside one with 120 members, tiny middles beside long ones, in one batch"""
    rng = random.Random(21)
    piles = []
    for w in range(128):
        depth = (2, 120, 3, 40)[w % 4]
        piles.append(region_pile(rng, depth, (1, 22, 50, 3)[w % 4], ("noise", "fan", "far", "noise")[(w // 4) % 4], cap=(22, 22, 54, 22)[w % 4]))
    hb = ca.pack_piles(piles)
    got = engine.run(hb)
    exp, _ = oracle_lib.oracle_run(ca.Params(*PRM), hb, threads=8)
    assert_same(got, exp, len(piles), "mixed")


def test_depth_150_batch_routes_most_tasks_to_tier_q(engine):
    """the headline shape: at depth 150 tier Q takes the tasks with members of up to 31 bases -- three quarters of all tasks"""
    from consent_amd.engine import synth_host

    hb = synth_host(ca.SynthSpec.pacbio(48, 150, first_window=4000))
    got = engine.run(hb)
    ctr, _ = engine.profile()
    exp, _ = oracle_lib.oracle_run(ca.Params(*PRM), hb, threads=8)
    assert_same(got, exp, 48, "depth 150")
    assert int(ctr[6]) * 10 >= int(ctr[0]) * 7, (ctr[0], ctr[6:12])
