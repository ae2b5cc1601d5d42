"""CPU: the oracle against the committed golden vectors (tests/golden/, generator script beside them)."""
import json
import os

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
WP = json.load(open(os.path.join(HERE, "golden", "windows_piles.json")))
CS = json.load(open(os.path.join(HERE, "golden", "consensus_small.json")))


@pytest.mark.parametrize("i", range(len(WP["cases"])))
def test_windows_and_piles_match_reference_vectors(i):
    """Expected values were produced by the reference's own alignmentWindows.cpp (oracle/_ref)."""
    c = WP["cases"][i]
    tpl_len = len(c["tpl"])
    wins = oracle_lib.window_positions(oracle_lib.oracle().cwo_window_positions, tpl_len, c["overlaps"], c["min_support"], c["window_size"], c["window_overlap"])
    assert wins == [tuple(w) for w in c["windows"]]
    for (qb, qe), exp in zip(c["windows"], c["piles"]):
        got = oracle_lib.window_pile(oracle_lib.oracle().cwo_window_pile, c["overlaps"], c["tpl"], c["targets"], qb, qe, c["k"])
        assert got == exp


@pytest.mark.parametrize("i", range(len(WP["cases"])))
def test_product_window_positions_match_reference_vectors(i):
    """the library's own host A1 (cw_window_positions) against the committed reference-generated vectors"""
    from consent_amd.engine import window_positions as product_window_positions

    c = WP["cases"][i]
    rows = np.array([[o[1], o[2], 0, o[5], o[6], o[3]] for o in c["overlaps"]], np.uint32)
    got = product_window_positions(len(c["tpl"]), rows, c["min_support"], c["window_size"], c["window_overlap"])
    assert got == [tuple(w) for w in c["windows"]]


@pytest.mark.parametrize("i", range(len(CS["cases"])))
def test_consensus_regression_vectors(i):
    c = CS["cases"][i]
    res, _ = oracle_lib.oracle_run(ca.Params(*c["params"]), ca.pack_piles([c["pile"]]))
    assert int(res.status[0]) == c["status"]
    assert res.consensus(0) == c["consensus"]
    assert [int(x) for x in res.solid_kmers(0)] == c["solid"]


def test_oracle_is_deterministic_and_thread_count_independent():
    from consent_amd.engine import synth_host

    hb = synth_host(ca.SynthSpec.pacbio(6, 10))
    prm = ca.Params(9, 4, 8, 2, 20)
    a, _ = oracle_lib.oracle_run(prm, hb, threads=1)
    b, _ = oracle_lib.oracle_run(prm, hb, threads=4)
    for w in range(6):
        assert a.consensus(w) == b.consensus(w)
        assert np.array_equal(a.solid_kmers(w), b.solid_kmers(w))
