"""GPU tests of the host-buffer entry points of the C ABI: cw_run = cw_submit + cw_wait (results compacted on the device, only the
used bytes cross PCIe), two batches in flight, pinned staging, the batch-size limit, and bench.py's --gpus self-spawn."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import consent_amd as ca
import oracle_lib
from consent_amd.engine import Batch, Result, alloc_results, synth_host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same(a, b, n):
    for w in range(n):
        assert int(a.status[w]) == int(b.status[w]), w
        assert a.consensus(w) == b.consensus(w), w
        assert np.array_equal(a.solid_kmers(w), b.solid_kmers(w)), w


def test_two_batches_in_flight_equal_the_oracle_and_a_third_is_refused():
    prm = ca.Params(9, 4, 8, 2, 20)
    eng = ca.Engine(prm)
    hbs = [synth_host(ca.SynthSpec.pacbio(40, 30, first_window=1000 * i)) for i in range(3)]
    ress = [alloc_results(hb, True, prm.solid, prm.k) for hb in hbs]
    t0, k0 = eng.submit(hbs[0], ress[0])
    t1, k1 = eng.submit(hbs[1], ress[1])
    b2, r2 = hbs[2].c_struct(), None
    from consent_amd.engine import _result_struct
    r2 = _result_struct(ress[2])
    t = C.c_int(-1)
    assert eng.lib.cw_submit(eng.handle, C.byref(b2), C.byref(r2), C.byref(t)) == -1  # two slots only
    eng.wait(t0)
    t2, k2 = eng.submit(hbs[2], ress[2])  # the freed slot is reusable while batch 1 is still in flight
    eng.wait(t1)
    eng.wait(t2)
    assert eng.lib.cw_wait(eng.handle, t2) == -1  # a ticket is waited for once
    for hb, res in zip(hbs, ress):
        exp, _ = oracle_lib.oracle_run(prm, hb, threads=os.cpu_count() or 1)
        same(res, exp, hb.n_windows)
    eng.close()


def test_results_are_scattered_to_the_callers_offsets_and_nothing_else_is_touched():
    """cw_wait writes window w's bytes at cons_off[w] / solid_off[w] and leaves the reserved tail of every slot alone."""
    prm = ca.Params(9, 4, 8, 2, 20)
    eng = ca.Engine(prm)
    hb = synth_host(ca.SynthSpec.pacbio(33, 12))
    res = alloc_results(hb, True, prm.solid, prm.k)
    res.cons[:] = 0xEE
    res.solid[:] = 0xDEADBEEF
    t, keep = eng.submit(hb, res)
    eng.wait(t)
    exp, _ = oracle_lib.oracle_run(prm, hb)
    same(res, exp, 33)
    for w in range(33):
        o, n, e = int(res.cons_off[w]), int(res.cons_len[w]), int(res.cons_off[w + 1])
        assert (res.cons[o + n : e] == 0xEE).all()
        o, n, e = int(res.solid_off[w]), int(res.solid_len[w]), int(res.solid_off[w + 1])
        assert (res.solid[o + n : e] == 0xDEADBEEF).all()
    eng.close()


def test_pinned_host_buffers_from_the_library_work_as_batch_storage():
    prm = ca.Params(9, 4, 8, 2, 20)
    eng = ca.Engine(prm)
    hb = synth_host(ca.SynthSpec.ont(25, 20))
    ptrs = []

    def pinned_copy(a):
        p = C.c_void_p()
        assert eng.lib.cw_host_alloc(C.byref(p), a.nbytes) == 0 and p.value
        C.memmove(p.value, a.ctypes.data, a.nbytes)
        ptrs.append(p)
        return p.value

    b = Batch(hb.n_windows, len(hb.seq_len), len(hb.bases), pinned_copy(hb.win_first_seq), pinned_copy(hb.seq_len), pinned_copy(hb.seq_word_off), pinned_copy(hb.bases))
    res = alloc_results(hb, True, prm.solid, prm.k)
    from consent_amd.engine import _result_struct
    r = _result_struct(res)
    assert eng.lib.cw_run(eng.handle, C.byref(b), C.byref(r)) == 0
    exp, _ = oracle_lib.oracle_run(prm, hb)
    same(res, exp, hb.n_windows)
    for p in ptrs:
        eng.lib.cw_host_free(p)
    eng.close()


def test_a_batch_beyond_the_window_limit_is_refused_before_anything_runs():
    eng = ca.Engine(ca.Params(9, 4, 8, 2, 20))
    one = np.zeros(4, np.uint64)
    b = Batch(131073, 131073, 131073, one.ctypes.data, one.ctypes.data, one.ctypes.data, one.ctypes.data)
    r = Result(one.ctypes.data, one.ctypes.data, one.ctypes.data, one.ctypes.data, None, None, None)
    assert eng.lib.cw_run_device(eng.handle, C.byref(b), C.byref(r), None) == -1  # CW_E_INVALID: 32-bit scratch offsets would wrap
    t = C.c_int(-1)
    assert eng.lib.cw_submit(eng.handle, C.byref(b), C.byref(r), C.byref(t)) == -1
    eng.close()


def test_empty_batch_and_template_only_windows_through_submit_wait():
    prm = ca.Params(9, 4, 8, 2, 20)
    eng = ca.Engine(prm)
    hb = ca.pack_piles([["ACGTACGTTGCAACGTAGCTAGCTAGGATCGATCGAT"], ["ACGT"], ["ACGTACGTTGCAACGTAGCTAGCTAGGATCGATCGAT"] * 5])
    res = eng.run(hb)
    exp, _ = oracle_lib.oracle_run(prm, hb)
    same(res, exp, 3)
    eng.close()


@pytest.mark.timeout(600)
def test_bench_gpus_2_spawns_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` from a plain shell must run TWO ranks (here both on the one GPU of the box, gloo for the barrier)
    and say n_gpus: 2; with a launcher environment it would be one rank of that launch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CW_BENCH_SINGLE_DEVICE="1", CW_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "256", "--workload", "pacbio_d30_msa20"],
                         capture_output=True, text=True, env=env, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["config"]["distinct_windows_per_run"] == 2 * 2 * 256
    assert j["value"] > 0 and "roofline" in j


@pytest.mark.timeout(1500)
def test_bench_gpus_8_runs_eight_ranks_and_the_driver_leg_on_one_box():
    """The command the driver runs once on an 8-GPU node, here with all eight ranks on the one GPU of the box (CW_BENCH_SINGLE_DEVICE, gloo): rank 0's
    native-driver leg first with `-j 8` while the seven other ranks wait in the rendezvous, then 8 x 4 engines side by side (the fourth of each rank with an allocation run of its own, three warm-up steps), one JSON line with
    n_gpus 8, a driver_strong_scaling object without an error, three timed repetitions and their median."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CW_BENCH_SINGLE_DEVICE="1", CW_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--windows", "1024", "--steps", "4", "--warmup", "3"],
                         capture_output=True, text=True, env=env, timeout=1400)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 8 and j["steps"] == 4 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["engines_per_gpu"] == 4 and j["config"]["engine_allocation_runs"] == 1 and j["config"]["distinct_windows_per_run"] == 8 * 4 * 1024
    assert j["reps"] == 3 and len(j["ms_per_step_reps"]) == 3 and j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    d = j["driver_strong_scaling"]
    assert "error" not in d, d
    assert d["n_gpus"] == 8 and d["windows"] > 100000 and d["windows_per_s"] > 0 and d["scaling"] == "strong"
    assert "cpu_baseline" not in j  # rank 0 at N = 1 only


def test_two_engines_take_batches_in_turn_and_cw_poll_tells_which_is_free():
    """Two engines on one GPU, each with a batch in flight on its own streams (what bench.py does): cw_poll never blocks, turns 1 once
    the batch is done, and every result equals the oracle's."""
    import time

    import torch

    dev = torch.device("cuda", 0)
    prm = ca.Params(9, 4, 8, 2, 150)
    engs = [ca.Engine(prm), ca.Engine(prm)]
    assert all(e.idle() for e in engs)  # nothing launched yet
    hbs = [synth_host(ca.SynthSpec.pacbio(96, 60, first_window=5000 * i)) for i in range(4)]
    exp = [oracle_lib.oracle_run(prm, hb, want_solid=True)[0] for hb in hbs]

    def up(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)

    def launch(e, hb):
        res = alloc_results(hb, True, prm.solid, prm.k)
        t_in = (up(hb.win_first_seq, np.int32), up(hb.seq_len, np.int32), up(hb.seq_word_off, np.int64), up(np.concatenate([hb.bases, np.zeros(4, np.uint32)]), np.int32))
        t_out = [torch.zeros(len(res.cons), dtype=torch.uint8, device=dev), up(res.cons_off, np.int64), torch.zeros(hb.n_windows, dtype=torch.int32, device=dev),
                 torch.full((hb.n_windows,), 255, dtype=torch.uint8, device=dev), torch.zeros(len(res.solid), dtype=torch.int32, device=dev), up(res.solid_off, np.int64),
                 torch.zeros(hb.n_windows, dtype=torch.int32, device=dev)]
        torch.cuda.synchronize(dev)  # the engine launches on its own stream
        b = Batch(hb.n_windows, len(hb.seq_len), len(hb.bases), *[t.data_ptr() for t in t_in])
        r = Result(*[t.data_ptr() for t in t_out])
        e.run_device(b, r)

        def fetch():
            res.cons[:] = t_out[0].cpu().numpy()
            res.cons_len[:] = t_out[2].cpu().numpy().view(np.uint32)
            res.status[:] = t_out[3].cpu().numpy()
            res.solid[:] = t_out[4].cpu().numpy().view(np.uint32)
            res.solid_len[:] = t_out[6].cpu().numpy().view(np.uint32)
            return res

        return fetch, (t_in, t_out, b, r)

    got, pending, nxt = [None] * 4, {}, 0
    t_end = time.time() + 120
    while (nxt < 4 or pending) and time.time() < t_end:
        for k, e in enumerate(engs):
            if not e.idle():
                continue
            if k in pending:
                i, (fetch, _keep) = pending.pop(k)
                got[i] = fetch()
            if nxt < 4:
                pending[k] = (nxt, launch(e, hbs[nxt]))
                nxt += 1
    assert not pending and all(g is not None for g in got)
    for i in range(4):
        same(got[i], exp[i], hbs[i].n_windows)
