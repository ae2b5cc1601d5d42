"""CPU: the host side of the native driver without a device (CW_DRIVER_DRY=1: the workers take the jobs and drop them).  What is
checked: the producer -- PAF piles -> window positions on helper threads -> jobs in pile order -- counts the same piles, windows,
overlaps and jobs with one helper and with several, the window total is what the oracle's getAlignmentWindowsPositions restatement
says for the same piles, and nothing is written.  (`bench.py --mode driver` uses the same switch for its feeder-ceiling figure.)"""
import json
import os
import subprocess

import consent_amd as ca
import oracle_lib
from test_gpu_pipeline import make_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def dry_run(fa, paf, threads, per_batch_env=None, j=3):
    argv = [os.path.join(BIN, "CONSENT-correction"), "-a", paf, "-s", "3", "-S", "150", "-l", "500", "-k", "9", "-c", "8", "-A", "2", "-f", "4", "-m", "50", "-j", str(j), "-r", fa, "-M", "150", "-p", "x"]
    # the dry run is a test aid: it exists in the -DCW_TEST_AIDS build of the library only (consent_amd/aids/, csrc/cw_env.h)
    env = dict(os.environ, CW_DRIVER_DRY="1", CW_DRIVER_STATS="1", CW_PRODUCER_THREADS=str(threads), LD_LIBRARY_PATH=os.path.join(ROOT, "consent_amd", "aids"))
    out = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    assert out.stdout == ""  # a dry run corrects nothing
    st = json.loads([ln for ln in out.stderr.splitlines() if ln.startswith("{")][-1])
    assert st["dry"] is True and st["producer_threads"] == threads
    return st


def test_dry_run_counts_do_not_depend_on_the_helpers_and_match_the_oracle(tmp_path):
    fa, paf = make_dataset(tmp_path, 5, n_reads=90, glen=16000)
    a = dry_run(fa, paf, 1)
    b = dry_run(fa, paf, 5, j=7)
    for k in ("piles", "windows", "jobs"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["workers"] == 3 * a["workers_per_device"] and b["workers"] == 7 * b["workers_per_device"] and 1 <= a["workers_per_device"] <= 4  # no device touched
    o = oracle_lib.oracle()
    ix = ca.ReadIndex(fa)
    n_win = n_piles = 0
    for tpl, tpl_len, ov, _ in ca.PafReader(paf, ix, 150):
        rows = [[tpl_len, int(r[0]), int(r[1]), int(r[5]), int(ix.seq_len[int(r[2])]), int(r[3]), int(r[4]), i] for i, r in enumerate(ov)]
        n_win += len(oracle_lib.window_positions(o.cwo_window_positions, tpl_len, rows, 3, 500, 50))
        n_piles += 1
    assert n_win > 100
    assert (a["piles"], a["windows"]) == (n_piles, n_win)
