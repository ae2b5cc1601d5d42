#!/usr/bin/env python3
"""bench.py -- window-correction hot path on synthetic PacBio-profile piles.

Default workload = BASELINE.json configs[2], the configuration the headline metric is quoted on: depth 150, maxMSA 150, 500 bp
windows, k=9 (`--workload pacbio_d30_msa20` = configs[1], the shallow path).

One "step" = one pass of the hot path (setup -> index -> chain -> POA tiers -> finish kernels, cw_run_device) over one batch of
synthetic window piles that is already resident in HBM.  Every step of a run works on its OWN batch of windows (window ids are
disjoint across steps and ranks; all batches are generated on the device before the clock starts and stay resident -- 25 batches of
depth-150 piles are 9 GB of the 288), so `steps x windows_per_step x n_gpus` distinct windows are corrected per run (20 steps =
327 680 >= 2^18).  Rank r of N works on its own windows; there is no collective on the data path (windows are
independent), so scaling is weak.  `python bench.py --gpus N` without a torch.distributed environment re-launches itself as N ranks.
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Must be in the environment before the HIP runtime starts (i.e. before torch is imported): HIP multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4).  Three engines have twelve kernel streams (main + three tier streams each); with four
# queues a stream of the second engine lands behind the first engine's long-running tier-L kernel and the batches do not overlap.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # (round 6: four engines, sixteen kernel streams + the process's own: 24 queues -- 49.1-51.6 ms per step against 52.4-54.7 with three engines on 16)

WORKLOADS = {
    # name: (depth, maxMSA, windows per step per GPU)
    "pacbio_d30_msa20": (30, 20, 16384),
    "pacbio_d150_msa150": (150, 150, 16384),
}
DEFAULT_WORKLOAD = "pacbio_d150_msa150"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAX_RESIDENT_BATCHES = 48  # distinct input batches kept in HBM; runs with more steps than this cycle through them


def algorithmic_bytes(seq_len, n_windows, cons_len, solid_len):
    """sum ceil(len/4) + 4*N + 16/window + consensus bytes + 4*|solid| + 8/window."""
    return int(np.sum((seq_len.astype(np.int64) + 3) // 4) + 4 * len(seq_len) + 16 * n_windows + int(np.sum(cons_len)) + 4 * int(np.sum(solid_len)) + 8 * n_windows)


def effective_cores():
    """CPU parallelism this process can actually use: affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pd = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pd))
        except Exception:
            pass
    return n


def self_spawn(args):
    """`python bench.py --gpus N` outside torch.distributed: become the launcher of N ranks (one per GPU) and relay their output."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def driver_measure(args, dry_reps=3):
    """`--mode driver`: the native driver end to end -- ONE host process (bin/CONSENT-correction = cw_run_correction) feeding
    `--gpus` devices from a PAF + read file, piles cut, corrected and re-assembled on the devices, FASTA written in PAF order.  Strong
    scaling: the read set is fixed (BASELINE configs[3] scale: 4.6 Mbp genome, 30x ONT-profile reads, ground-truth overlaps; `--driver-copies 8`
    = eight independent copies of it in one run).  Besides the timed run it prints the feeder ceiling: the same command with
    CW_DRIVER_DRY=1 (workers drop their jobs; no device touched) = the windows/s the host side alone can hand out."""
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_bench as pb

    d = os.environ.get("CW_KEEP_DATA") or os.path.join(tempfile.gettempdir(), f"cw_driver_data_{args.driver_genome}_{args.driver_cov}")
    t0 = time.perf_counter()
    fa, paf = os.path.join(d, "reads.fa"), os.path.join(d, "ovl.paf")
    if not (os.path.exists(fa) and os.path.exists(paf)):
        fa, paf, _, _, _ = pb.generate(d, args.driver_genome, args.driver_cov, "ont")
    fa, paf = pb.replicate(fa, paf, args.driver_copies)
    gen_s = time.perf_counter() - t0
    out_fa = os.path.join(d, "corrected.fa")
    argv = [os.path.join(ROOT, "bin", "CONSENT-correction"), "-a", paf, "-s", "3", "-S", "150", "-l", "500", "-k", "9", "-c", "8", "-A", "2", "-f", "4", "-m", "50", "-j", str(args.gpus),
            "-r", fa, "-M", "150", "-p", "x"]  # CONSENT-correct:202

    def run(extra_env, sink):
        env = dict(os.environ, CW_DRIVER_STATS="1", CW_ON_CAPACITY="skip", **extra_env)
        t1 = time.perf_counter()
        with open(sink, "wb") as f:
            pr = subprocess.run(argv, stdout=f, stderr=subprocess.PIPE, text=False, env=env)
        wall = time.perf_counter() - t1
        err = pr.stderr.decode(errors="replace")
        if pr.returncode != 0:
            raise SystemExit(f"driver failed ({pr.returncode}): {err[-1500:]}")
        return wall, json.loads([ln for ln in err.splitlines() if ln.startswith("{")][-1])

    # (the dry run is a test aid: only the -DCW_TEST_AIDS build of the library has it -- consent_amd/aids/, csrc/cw_env.h)
    dry_wall, dry = min((run({"CW_DRIVER_DRY": "1", "LD_LIBRARY_PATH": os.path.join(ROOT, "consent_amd", "aids")}, os.devnull) for _ in range(max(1, dry_reps))), key=lambda x: x[0])
    best = None
    for _ in range(max(1, args.driver_reps)):
        wall, st = run({}, out_fa)
        if best is None or wall < best[0]:
            best = (wall, st)
    wall, st = best
    inside = st["ms_total"] * 1e-3
    per_dev = {}
    for w in st["per_device"]:
        q = per_dev.setdefault(w["device"], {"windows": 0, "jobs": 0, "busy_ms_summed_over_workers": 0.0})
        q["windows"] += w["windows"]; q["jobs"] += w["jobs"]; q["busy_ms_summed_over_workers"] += w["ms_extract"] + w["ms_consensus"] + w["ms_stitch"]
    out = {
        "metric": "corrected windows/sec (native driver, end to end: PAF + reads in, FASTA out)",
        "value": st["windows"] / inside, "unit": "windows/s", "n_gpus": args.gpus, "steps": 1, "warmup": 0, "ms_per_step": inside * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8/int16 (2-bit bases, integer DP)", "data": "synthetic",
        "config": {"workload": f"ont_reads_{args.driver_genome}bp_{args.driver_cov}x_x{args.driver_copies}: simulated ONT-profile reads, ground-truth PAF, CONSENT-correct:202 argv", "mode": "driver",
                   "piles": st["piles"], "windows": st["windows"], "jobs": st["jobs"], "records": st["records"], "bases_out": st["bases_out"], "workers": st["workers"]},
        "wall_s_process": wall, "s_inside_cw_run_correction": inside, "ms_index": st["ms_index"], "ms_engines": st["ms_engines"], "ms_producer": st["ms_producer"],
        "ms_paf_parse": st["ms_paf_parse"], "producer_threads": st["producer_threads"],
        "steady_state_windows_per_s": st["windows"] / max(1e-9, (st["ms_total"] - st["ms_index"] - st["ms_engines"]) * 1e-3),
        "per_device": per_dev,
        "feeder_ceiling": {"windows_per_s": dry["windows"] / max(1e-9, dry["ms_producer"] * 1e-3), "ms_producer": dry["ms_producer"], "ms_index": dry["ms_index"], "wall_s_process": dry_wall,
                           "what": "CW_DRIVER_DRY=1: PAF piles -> window positions -> jobs, workers drop the jobs; windows / producer time"},
        "data_generation_s": gen_s,
    }
    return out


def driver_mode(args):
    print(json.dumps(driver_measure(args)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="kernel", choices=["kernel", "driver"], help="kernel: the hot path on resident batches (the contract line); driver: the native driver end to end, one process over --gpus devices")
    ap.add_argument("--driver-genome", type=int, default=4600000)
    ap.add_argument("--driver-cov", type=int, default=30)
    ap.add_argument("--driver-copies", type=int, default=1)
    ap.add_argument("--driver-reps", type=int, default=2)
    ap.add_argument("--driver-leg", type=int, default=1, help="kernel mode: after the timed region rank 0 also runs the native driver (ONE process over the --gpus devices) on the fixed BASELINE configs[3]-scale read set and reports it under `driver_strong_scaling` of the same JSON line (strong scaling: the set does not grow with N); 0 disables")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=int(os.environ.get("CW_BENCH_REPS", "3")),
                    help="timed repetitions of the K-step region inside this process (SURVEY 8d: warm-up, then several timed runs, the MEDIAN reported): every repetition is exactly --steps steps between a barrier + synchronize on both sides, max over ranks; value / ms_per_step are the median repetition's, min and max beside them")
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--windows", type=int, default=0, help="windows per step per GPU (default: per workload)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="windows timed on the CPU oracle (rank 0, N=1); 0 disables")
    # (round 5: three -- four alternating runs each on one box: 53.4 52.8 54.2 52.2 ms per step with two engines and eight queues, 52.0 52.5 52.9 51.7 with three
    # and sixteen; the native driver takes a third worker per device on long runs for the same reason)
    # (round 6: four, with 24 hardware queues -- four alternating runs each on one box: 52.4 54.7 53.0 53.1 ms per step with three engines on 16 queues, 50.6 49.1 49.5 51.6 with
    # four on 24, 51.5 51.9 52.2 50.3 with four on 16; the scratch plan went from 24 to 18 GB an engine in the same round)
    ap.add_argument("--engines", type=int, default=int(os.environ.get("CW_BENCH_ENGINES", "4")),
                    help="engines per GPU taking the steps in turn (each has its own scratch and streams): batch n+1's index/chain kernels fill the CUs that the tail of batch n's POA stage leaves idle")
    ap.add_argument("--alone-steps", type=int, default=3, help="untimed steps after the timed region with ONE batch in flight: per-kernel HIP-event times that are work, not waiting (roofline.launch_ms)")
    ap.add_argument("--pcie-engines", type=int, default=3, help="engines the PCIe-inclusive leg spreads its host batches over (round 6: three, six host batches in flight)")
    ap.add_argument("--pcie-steps", type=int, default=-1, help="batches timed through cw_submit/cw_wait from pinned host memory (rank 0, N=1); 0 disables")
    args = ap.parse_args()

    if args.mode == "driver":
        if int(os.environ.get("RANK", "0")) == 0:  # one process feeds every device
            driver_mode(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # --- strong scaling, same JSON line: the weak-scaling line below is N independent processes; what `north_star` means by
    # "reads sharded across the 8 GPUs" is ONE process feeding N devices from one PAF + read file.  It runs FIRST, while no rank has touched its
    # device yet -- the other ranks wait for rank 0 in the process group's rendezvous below (measured: run after the kernel legs, beside this process's idle HIP context, the same command took 2.2x as long).
    driver_leg = None
    if args.driver_leg:
        if rank == 0:
            try:
                da = argparse.Namespace(**vars(args))
                da.driver_copies, da.driver_reps = 1, 3  # (the best of three: a fresh process's start-up varies by half a second -- runtime, code object, device memory)
                d = driver_measure(da, dry_reps=2)
                driver_leg = {
                    "windows_per_s": d["value"], "s_total": d["s_inside_cw_run_correction"], "wall_s_process": d["wall_s_process"], "n_gpus": args.gpus,
                    "windows": d["config"]["windows"], "jobs": d["config"]["jobs"], "workers": d["config"]["workers"],
                    "per_device_busy": d["per_device"], "ms_index": d["ms_index"], "ms_engines": d["ms_engines"],
                    "steady_state_windows_per_s": d["steady_state_windows_per_s"], "feeder_ceiling": d["feeder_ceiling"],
                    "workload": d["config"]["workload"], "scaling": "strong",
                    "what": "bin/CONSENT-correction (cw_run_correction): one host process, piles cut / corrected / re-assembled on the --gpus devices, FASTA out in PAF order; the set is fixed (4.6 Mbp, 30x), so this value at N = 1, 2, 4, 8 is the strong-scaling curve",
                }
            except (SystemExit, Exception) as exc:  # the contract line must survive a failure of the extra leg
                driver_leg = {"error": str(exc)[-400:]}
            # ... and on EIGHT copies of that set (2.6e6 windows): the set on which every device of an 8-GPU node still gets full-size jobs -- the x1 set is
            # 0.8 s of work for ONE GPU, and a fresh process spends 0.5 s starting the runtime and loading the code object whatever N is (round 6)
            if args.driver_leg >= 1 and isinstance(driver_leg, dict) and "error" not in driver_leg:
                try:
                    da8 = argparse.Namespace(**vars(args))
                    da8.driver_copies, da8.driver_reps = 8, 1
                    d8 = driver_measure(da8, dry_reps=1)
                    driver_leg["x8"] = {"windows_per_s": d8["value"], "s_total": d8["s_inside_cw_run_correction"], "wall_s_process": d8["wall_s_process"], "windows": d8["config"]["windows"],
                                        "jobs": d8["config"]["jobs"], "workers": d8["config"]["workers"], "ms_index": d8["ms_index"], "ms_engines": d8["ms_engines"],
                                        "workload": d8["config"]["workload"], "scaling": "strong"}
                except (SystemExit, Exception) as exc:
                    driver_leg["x8"] = {"error": str(exc)[-400:]}

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if os.environ.get("CW_BENCH_SINGLE_DEVICE"):  # test aid: several ranks on one GPU (gloo only)
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the data path has no collective (north_star: "no RCCL"): the barrier and the one max-reduce of the timing go over gloo;
        # CW_BENCH_BACKEND=nccl (= RCCL on ROCm) remains for boxes where that is preferred
        backend = os.environ.get("CW_BENCH_BACKEND", "gloo")
        import datetime

        # (generous: the other ranks wait here while rank 0 runs its driver leg -- data generation, two dry runs, two full runs over all devices)
        pg_timeout = datetime.timedelta(minutes=int(os.environ.get("CW_BENCH_PG_TIMEOUT_MIN", "60")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
        except Exception as exc:  # keep the run alive on a box where RCCL cannot initialise: the data path has no collective
            print(f"[bench] {backend} init failed ({exc}); falling back to gloo for the barrier", file=sys.stderr)
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        dist.barrier()  # (rank 0 arrives here after its driver leg)
    else:
        backend = None
    torch.cuda.set_device(local_rank)

    import consent_amd as ca
    from consent_amd.engine import Batch, HostBatch, Result, alloc_results, synth_host

    depth, max_msa, n_win = WORKLOADS[args.workload]
    if args.windows > 0:
        n_win = args.windows
    prm = ca.Params(9, 4, 8, 2, max_msa)
    # (an engine allocates its scratch in its first run.  Through round 5 there were no more engines than warm-up steps, which silently measured another configuration when
    #  --warmup was small; now an engine the warm-up steps do not reach gets one untimed allocation run before them -- `config.engine_allocation_runs`)
    engines = [ca.Engine(prm, device=local_rank) for _ in range(max(1, args.engines))]
    for e_ in engines:  # the caller knows its window size, as the native driver does (cw_configure: the scratch plan goes by it -- 492 k-mers per template, not 1024)
        e_.configure(500)
    eng = engines[0]
    lib = eng.lib
    dev = torch.device("cuda", local_rank)

    # --- every step's batch, generated on the device, resident before the clock starts; window ids disjoint by (rank, step) -------
    n_batches = min(args.warmup + args.steps, MAX_RESIDENT_BATCHES)
    ns, nw = C.c_uint32(), C.c_uint64()
    spec0 = ca.SynthSpec.pacbio(n_win, depth)
    assert lib.cw_synth_sizes(C.byref(spec0), C.byref(ns), C.byref(nw)) == 0
    n_seqs, n_words = ns.value, nw.value
    batches, keep = [], []
    for i in range(n_batches):
        spec = ca.SynthSpec.pacbio(n_win, depth, first_window=(rank * (args.warmup + args.steps) + i) * n_win)
        t_wfs = torch.zeros(n_win + 1, dtype=torch.int32, device=dev)
        t_len = torch.zeros(n_seqs, dtype=torch.int32, device=dev)
        t_off = torch.zeros(n_seqs, dtype=torch.int64, device=dev)
        t_bases = torch.zeros(n_words + 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        rc = lib.cw_synth_device(eng.handle, C.byref(spec), t_wfs.data_ptr(), t_len.data_ptr(), t_off.data_ptr(), t_bases.data_ptr(), None)
        assert rc == 0, rc
        keep.append((t_wfs, t_len, t_off, t_bases))
        batches.append(Batch(n_win, n_seqs, n_words, t_wfs.data_ptr(), t_len.data_ptr(), t_off.data_ptr(), t_bases.data_ptr()))
    torch.cuda.synchronize(dev)
    from consent_amd.engine import cons_slot_bytes

    cons_cap = int(cons_slot_bytes(prm.k, spec0.window_len))  # the slot cw_plan_results_device gives every window (include/consent_amd.h CW_CONS_SLOT_BYTES)
    solid_cap = (depth + 1) * (spec0.window_len + 24) // prm.solid + 16
    t_cons = torch.zeros(n_win * cons_cap, dtype=torch.uint8, device=dev)
    t_coff = (torch.arange(n_win + 1, dtype=torch.int64, device=dev) * cons_cap)
    t_clen = torch.zeros(n_win, dtype=torch.int32, device=dev)
    t_stat = torch.zeros(n_win, dtype=torch.uint8, device=dev)
    t_solid = torch.zeros(n_win * solid_cap, dtype=torch.int32, device=dev)
    t_soff = (torch.arange(n_win + 1, dtype=torch.int64, device=dev) * solid_cap)
    t_slen = torch.zeros(n_win, dtype=torch.int32, device=dev)
    r = Result(t_cons.data_ptr(), t_coff.data_ptr(), t_clen.data_ptr(), t_stat.data_ptr(), t_solid.data_ptr(), t_soff.data_ptr(), t_slen.data_ptr())
    rs, keep_r = [r], []
    for _ in range(1, len(engines)):  # batches in flight at the same time need their own result arrays
        tt_ = (torch.zeros_like(t_cons), torch.zeros_like(t_clen), torch.zeros_like(t_stat), torch.zeros_like(t_solid), torch.zeros_like(t_slen))
        keep_r.append(tt_)
        rs.append(Result(tt_[0].data_ptr(), t_coff.data_ptr(), tt_[1].data_ptr(), tt_[2].data_ptr(), tt_[3].data_ptr(), t_soff.data_ptr(), tt_[4].data_ptr()))
    torch.cuda.synchronize(dev)

    ne = len(engines)
    n_alloc_runs = max(0, ne - args.warmup)
    for i in range(args.warmup, ne):  # set-up, not warm-up: the first run of an engine allocates its scratch
        engines[i].run_device(batches[i % n_batches], rs[i])
    for i in range(args.warmup):
        engines[i % ne].run_device(batches[i % n_batches], rs[i % ne])
    torch.cuda.synchronize(dev)
    stage_ms = {}
    n_over = n_tpl = 0
    host_ms = [0.0, 0.0]
    ran = [0] * ne  # steps handed to each engine
    stagger = os.environ.get("CW_BENCH_STAGGER", "0") != "0"  # measured slower (82.9 vs 79.9 ms at depth 150): tier S ends as late as tier L, there is no tail to run under
    e_last = 0

    def timed_region():
        """EXACTLY --steps steps between barrier + synchronize on both sides; seconds, max over ranks."""
        nonlocal e_last
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(args.steps):
            _h0 = time.perf_counter()
            if ne == 1:
                k_ = 0
            else:  # the engine that is free takes the step (an engine still in a long tier-L tail does not hold up the others)
                k_ = None
                while k_ is None:
                    ph = [e_.phase() for e_ in engines]
                    for c_ in range(ne):
                        c2 = (e_last + 1 + c_) % ne
                        # a free engine takes the step; with CW_BENCH_STAGGER=1 only once every other engine is free or in its tail
                        if ph[c2] == 1 and (not stagger or all(ph[o_] in (1, 2) for o_ in range(ne) if o_ != c2)):
                            k_ = c2
                            break
                    else:
                        time.sleep(5e-5)
                if ran[k_]:  # HIP events of that engine's previous step (complete: no wait)
                    for k, v in engines[k_].timings().items():
                        stage_ms.setdefault(k, []).append(v)
            _h2 = time.perf_counter()
            engines[k_].run_device(batches[(args.warmup + i) % n_batches], rs[k_])
            e_last = k_
            ran[k_] += 1
            _h1 = time.perf_counter()
            host_ms[0] += (_h1 - _h2) * 1e3
            if ne == 1:
                for k, v in engines[0].timings().items():  # HIP events on the streams the kernels are launched on (cw_last_timings); waits for the step
                    stage_ms.setdefault(k, []).append(v)
            host_ms[1] += (time.perf_counter() - _h1 + _h2 - _h0) * 1e3
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_

    # SURVEY 8(d)'s protocol: one process, warm-up, then `--reps` timed repetitions of the same K-step region over the same
    # resident batches; the line's value is the MEDIAN repetition (boxes of the pool differ by 10 %, runs on one box by +-1.5 ms: a single run of
    # twenty steps met a 52-ms bar on one box and missed it on the next)
    n_reps = max(1, args.reps)
    rep_dt = [timed_region() for _ in range(n_reps)]
    dt = float(np.median(rep_dt))
    if os.environ.get("CW_PROFILE"):
        print("host ms per step: enqueue", round(host_ms[0] / (args.steps * n_reps), 3), "wait + read stage timings", round(host_ms[1] / (args.steps * n_reps), 3), file=sys.stderr)
    if ne > 1:
        for c_ in range(ne):  # the last step of every engine
            if ran[c_]:
                for k, v in engines[c_].timings().items():
                    stage_ms.setdefault(k, []).append(v)

    # --- kernel times with ONE batch in flight (untimed, after the clock stopped): with two engines a kernel's HIP-event span also holds
    # the wait for CUs the other batch occupies, so "launch duration = work" is only true here.  This is what `roofline` is priced on and
    # what the `--engines 1` rocprofv3 summary under profiles/ must agree with.
    alone_ms = {}
    n_alone = max(0, args.alone_steps)
    for i in range(n_alone):
        engines[0].run_device(batches[(args.warmup + i) % n_batches], rs[0])
        for k, v in engines[0].timings().items():  # waits for the step
            alone_ms.setdefault(k, []).append(v)
    if ne == 1 and not alone_ms:
        alone_ms = stage_ms
    last = (args.warmup + args.steps - 1) % n_batches
    if n_alone:
        last, e_last = (args.warmup + n_alone - 1) % n_batches, 0
    r_last = ([(t_cons, t_clen, t_stat, t_solid, t_slen)] + keep_r)[e_last]  # the result arrays the last step wrote
    status = r_last[2].cpu().numpy()
    n_over = int((status == ca.WIN_OVERFLOW).sum())
    n_tpl = int((status == ca.WIN_TEMPLATE).sum())
    clen = r_last[1].cpu().numpy()
    slen = r_last[4].cpu().numpy()
    seq_len = keep[last][1].cpu().numpy()
    alg_bytes = algorithmic_bytes(seq_len, n_win, clen, slen)
    stage_avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
    alone_avg = {k: float(np.mean(v)) for k, v in alone_ms.items()}
    # The dominant kernel is the one that does the largest share of the work, by the wave-cycles the POA kernels count themselves
    # (cw_debug_profile; the tier kernels run side by side between one fork and one join, so wall time does not tell them apart).
    tier_stage = ["poa", "poa_m1", "poa_m2", "poa_large", "poa_q", "poa_h"]  # wave-cycles per tier kernel (tiers Q and H: as the first task of each wave saw them)
    busy_share = None
    _, prof = eng.profile()
    busy = [float(prof[8 + 5 * t : 13 + 5 * t].sum()) for t in range(4)] + [float(prof[28:33].sum()), float(prof[64:69].sum())]
    if sum(busy) > 0:
        busy_share = {tier_stage[t]: busy[t] / sum(busy) for t in range(6)}
    dom = max(busy_share, key=busy_share.get) if busy_share else (max((k for k in alone_avg if k != "total"), key=alone_avg.get) if alone_avg else None)
    total_windows = n_win * world * args.steps
    value = total_windows / dt

    out = {
        "metric": "corrected windows/sec",
        "value": value,
        "unit": "windows/s",
        "bases_per_sec": value * spec0.window_len,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "reps": n_reps,
        "ms_per_step_reps": [d_ / args.steps * 1e3 for d_ in rep_dt],
        "ms_per_step_min": min(rep_dt) / args.steps * 1e3,
        "ms_per_step_max": max(rep_dt) / args.steps * 1e3,
        "value_min": total_windows / max(rep_dt),
        "value_max": total_windows / min(rep_dt),
        "value_is": "median of the timed repetitions (each exactly `steps` steps, barrier + synchronize on both sides, max over ranks)",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/int16 (2-bit bases, integer DP)",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: synthetic PacBio-profile piles, 500 bp windows, depth {depth}, maxMSA {max_msa}, k=9, solid=4, commonKMers=8, minAnchors=2",
            "windows_per_step_per_gpu": n_win,
            "distinct_windows_per_run": n_win * world * min(args.steps, max(1, n_batches - args.warmup)),
            "sharding": "windows by rank, no collective",
            "engines_per_gpu": ne,
            "engines_requested": args.engines,
            "engine_allocation_runs": n_alloc_runs,
            "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
            "overflow_windows": n_over,
            "template_fallback_windows": n_tpl,
        },
        "stage_ms": stage_avg,
        "stage_ms_one_batch_in_flight": alone_avg,
    }
    stage_kernel = {"index": "cw_index_kernel", "chain": "cw_chain_kernel", "poa": "cw_poa_kernel", "poa_q": "cw_poa_q_kernel", "poa_h": "cw_poa_h_kernel", "poa_m1": "cw_poa_slab_kernel<256", "poa_m2": "cw_poa_slab_kernel<512",
                    "poa_large": "cw_poa_slab_kernel<1536, 4096, 1023, 1, 3, 0, 1, 3>", "poa_lw": "cw_poa_slab_kernel<1536, 4096, 1023, 1, 3, 0, 4, 5>", "poa_overflow": "cw_poa_big_kernel", "finish": "cw_finish_kernel", "setup": "cw_setup_kernel"}
    traffic, traffic_step, traffic_src, prof_whole = None, None, None, None
    prof_json = os.path.join(ROOT, "profiles", f"latest_{args.workload}.json")
    if dom and os.path.exists(prof_json):
        try:  # HBM-side bytes from the separate --pmc passes of the same command with one engine (tools/profile_r03.sh, tools/summarize_r03.py)
            pj = json.load(open(prof_json))
            if pj.get("windows_per_step") == n_win:
                traffic_src = os.path.relpath(prof_json, ROOT)
                for name, k in pj["kernels"].items():
                    if stage_kernel.get(dom, "?") in name and "traffic_bytes_per_launch" in k:
                        traffic = k["traffic_bytes_per_launch"]
                traffic_step = pj.get("traffic_bytes_per_step")
                prof_whole = pj.get("whole_step")
        except Exception:
            pass
    if dom and dom in alone_avg:
        # dominant kernel: algorithmic bytes of the batch / its launch duration with one batch in flight (HIP events on its own stream)
        ach = alg_bytes / (alone_avg[dom] * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm",
            "kernel": dom,
            "kernel_symbol": stage_kernel.get(dom),
            "kernel_chosen_by": "largest share of the POA wave-cycles (counted by the kernels)",
            "poa_tier_share_of_wave_cycles": busy_share,
            "achieved": ach,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_whole_step": traffic_step,
            "traffic_whole_step_over_algorithmic": (traffic_step / alg_bytes) if traffic_step else None,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes_per_window": alg_bytes / n_win,
            "launch_ms": alone_avg[dom],
            "launch_ms_batches_in_flight": stage_avg.get(dom),
            "step_ms_one_batch_in_flight": alone_avg.get("total"),
            "frac_of_whole_step": alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        }

    # --- PCIe-inclusive rate: the same batches through cw_submit / cw_wait from pinned host memory, two batches in flight --------
    if rank == 0 and world == 1 and args.pcie_steps != 0:
        n_p = args.pcie_steps if args.pcie_steps > 0 else max(8, min(args.steps, 16))
        n_host = min(3, n_batches)
        host = []
        for i in range(n_host):  # pinned copies of resident batches (generating them on the host would take minutes)
            hw = [t.cpu().pin_memory() for t in keep[i]]
            host.append((hw, Batch(n_win, n_seqs, n_words, hw[0].data_ptr(), hw[1].data_ptr(), hw[2].data_ptr(), hw[3].data_ptr())))
        coff_h = (np.arange(n_win + 1, dtype=np.uint64) * cons_cap)
        soff_h = (np.arange(n_win + 1, dtype=np.uint64) * solid_cap)
        res_h = []
        ne_p = max(1, min(ne, args.pcie_engines))
        # host batches in flight.  Round 5, one box: two on one engine 2.41e5 windows/s, two over two engines 2.74e5, three over two 2.89e5, three over three 2.72e5
        # (rounds 2-3, with longer steps and the results copied back whole: four over two engines 1.59e5, two over two 1.75e5, two on one 1.87e5)
        # (round 6, one box, 16 batches each: two engines / three in flight 2.58e5 windows/s, two / four 2.83e5, three / four 2.85e5, three / six 3.00e5, four / six 2.96e5)
        depth_p = int(os.environ.get("CW_BENCH_PCIE_DEPTH", str(2 * ne_p)))
        for _ in range(depth_p):
            arrs = (np.zeros(n_win * cons_cap, np.uint8), np.zeros(n_win, np.uint32), np.zeros(n_win, np.uint8), np.zeros(n_win * solid_cap, np.uint32), np.zeros(n_win, np.uint32))
            res_h.append((arrs, Result(arrs[0].ctypes.data, coff_h.ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, soff_h.ctypes.data, arrs[4].ctypes.data)))

        def submit(i):
            t = C.c_int(-1)
            rc_ = lib.cw_submit(engines[i % ne_p].handle, C.byref(host[i % n_host][1]), C.byref(res_h[i % depth_p][1]), C.byref(t))
            assert rc_ == 0, rc_
            return (i % ne_p, t.value)

        def wait(tk):
            assert lib.cw_wait(engines[tk[0]].handle, tk[1]) in (0, -4)

        # warm-up: allocations and pinned staging of EVERY slot the timed loop will use (round 6: one submit per engine left the engines' second slots to be
        # allocated inside the timed region -- and obtaining new device memory can stall a call for a second)
        for _ in range(2):
            warm = [submit(i) for i in range(depth_p)]
            for tk in warm:
                wait(tk)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        flight = []
        for i in range(n_p):
            if len(flight) >= depth_p:
                wait(flight.pop(0))
            flight.append(submit(i))
        while flight:
            wait(flight.pop(0))
        pdt = time.perf_counter() - t1
        in_mb = sum(t.numel() * t.element_size() for t in host[0][0]) / 1e6
        out_mb = (int(clen.sum()) + 4 * int(slen.sum()) + 9 * n_win) / 1e6
        out["value_pcie_inclusive"] = n_win * n_p / pdt
        out["pcie_inclusive"] = {
            "value": n_win * n_p / pdt, "unit": "windows/s", "batches": n_p, "ms_per_batch": pdt / n_p * 1e3,
            "h2d_mb_per_batch": in_mb, "d2h_mb_per_batch": out_mb,
            "engines": ne_p, "batches_in_flight": depth_p, "path": "cw_submit/cw_wait (= cw_run in two halves), inputs in pinned host memory, batches in flight over the engines as stated, results compacted on the device and only the used bytes copied back, then scattered to the caller's (pageable) arrays",
        }
        del host, res_h

    # --- CPU baseline: the oracle (a scalar restatement, "port") on a bounded sample, rank 0, N=1 only ----
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))

        native_dir = f"/tmp/cw_oracle_native_{os.getpid()}"  # built for THIS host's CPU, here, never shipped
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "native_simd", f"OUT={native_dir}"])
        lib_plain, lib_fast = os.path.join(native_dir, "liboracle.so"), os.path.join(native_dir, "liboracle_simd.so")
        os.environ["CW_ORACLE_LIB"] = lib_plain
        import oracle_lib

        cores = effective_cores()
        n_s = args.cpu_sample if args.cpu_sample > 0 else max(cores * (128 if depth > 60 else 1024), 64)
        n_s = min(n_s, n_win)
        hb = synth_host(ca.SynthSpec.pacbio(n_s, depth))
        # one worker PROCESS per core, each running the oracle on a contiguous slice (threads of one process contend in malloc;
        # processes do not); the workers are forked before the clock starts.  Two builds of the same restatement are timed:
        #   "fast"  -DCWO_FAST -DCWO_SIMD (oracle/Makefile native_simd): direct-addressed k-mer tables, the chain scan's exact early stop,
        #           AVX2 row-vectorised POA fill -- what a tuned CPU implementation of the same algorithm does; results identical to
        #           the plain build (tests/test_oracle_units.py).  This is cpu_baseline.value ("simd": true).
        #   "plain" the checker as it is (hash maps, every anchor pair tested, scalar fill) on a quarter of the sample: scalar_value.
        import multiprocessing as mp

        from consent_amd.sharding import shard_range

        def time_build(lib_path, n_take):
            def _work(lo, hi, go, q):
                oracle_lib._O = None
                os.environ["CW_ORACLE_LIB"] = lib_path
                sub = hb.slice(lo, hi)
                oracle_lib.oracle()
                go.wait()
                t_a = time.perf_counter()
                oracle_lib.oracle_run(prm, sub, want_solid=True, threads=1)
                q.put((lo, hi, time.perf_counter() - t_a))

            ctx = mp.get_context("fork")
            go, q = ctx.Event(), ctx.Queue()
            procs = [ctx.Process(target=_work, args=(*shard_range(n_take, r_, cores), go, q)) for r_ in range(cores) if shard_range(n_take, r_, cores)[1] > shard_range(n_take, r_, cores)[0]]
            for pr in procs:
                pr.start()
            time.sleep(0.5)
            t1 = time.perf_counter()
            go.set()
            for _ in procs:
                q.get()
            dt_ = time.perf_counter() - t1
            for pr in procs:
                pr.join()
            return n_take / dt_, len(procs)

        fast_rate, n_procs = time_build(lib_fast, n_s)
        plain_rate, _ = time_build(lib_plain, max(cores, n_s // 4))
        procs = [None] * n_procs
        n_st = min(n_s, 256)  # windows byte-compared with the GPU (and counted for the secondary figures): the first n_st of the timed sample
        exp, ost = oracle_lib.oracle_run(prm, hb.slice(0, n_st), threads=min(cores, 32))
        # secondary, interpretable rates: the path is integer DP, far from the HBM roof by construction
        out["secondary"] = {
            "poa_dp_cells_per_window": ost["dp_cells"] / n_st,
            "poa_alignments_per_window": ost["alignments"] / n_st,
            "poa_gcups": ost["dp_cells"] / n_st * value / 1e9,
            "poa_dp_cells_routed_per_window": ost["dp_cells_routed"] / n_st,
            "poa_alignments_routed_per_window": ost["alignments_routed"] / n_st,
            "poa_gcups_routed": ost["dp_cells_routed"] / n_st * value / 1e9,
            "kmers_per_window": ost["kmers"] / n_st,
            "poa_alignments_per_sec": ost["alignments"] / n_st * value,
            "note": f"cell, alignment and k-mer counts from the oracle on the first {n_st} windows of the workload (they include the segments both sides resolve without a POA: a single piece, or equal pieces no longer than k; the *_routed figures leave those out -- the cells of the alignments that reach one of the engine's POA tiers); GCUPS = cells/window x windows/s",
        }
        if prof_whole:  # from the SQ counter passes of the same command with one engine (profiles/latest_*.json, tools/summarize_r03.py)
            out["secondary"]["valu_issue_frac"] = prof_whole.get("valu_issue_frac")
            out["secondary"]["lds_bw_frac"] = prof_whole.get("lds_busy_frac")
            out["secondary"]["valu_lane_util"] = prof_whole.get("valu_lane_util")  # SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), whole step (pass 3)
            out["secondary"]["counters_source"] = traffic_src
        # parity spot-check of the same windows on the GPU
        got = eng.run(hb.slice(0, n_st))
        same = all(got.consensus(w) == exp.consensus(w) and got.status[w] == exp.status[w] for w in range(n_st))
        out["cpu_baseline"] = {
            "value": fast_rate,
            "unit": "windows/s",
            "cores": cores,
            "kind": "port",
            "simd": True,
            "scalar_value": plain_rate,
            "gpu_over_cpu": value / fast_rate,
            "gpu_compared_windows": n_st,
            "sample": f"first {n_s} windows of the same workload (the first {n_st} of them also byte-compared with the GPU), the oracle's restatement built -O3 -march=native -DCWO_FAST -DCWO_SIMD (direct-addressed k-mer tables, exact early stop of the chain scan, AVX2 row-vectorised POA fill; identical results), {len(procs)} single-thread worker processes (= usable cores: affinity capped by the cgroup CPU quota; the box shows {os.cpu_count()} logical CPUs), consensus stage only; scalar_value = the plain checker build on a quarter of the sample.  Where the plain build's time goes at depth 150: 56 % k-mer hash maps, 25 % the all-pairs chain scan, 8 % POA -- a SIMD fill alone changes nothing, which is why the fast build also replaces the maps and stops the scan early",
            "gpu_identical_on_sample": bool(same),
        }
    if rank == 0 and os.environ.get("CW_PROFILE"):
        ctr, prof = eng.profile()
        names = ["idx.count", "idx.exact", "idx.export", "idx.support", "idx.cand+P", "idx.chain", "idx.segments", "idx.tplhash"]
        for tname in ("S", "M1", "M2", "L", "Q"):
            names += [f"{tname}.{x}" for x in ("meta", "fill", "trace", "merge", "cons")]
        names += ["-"] * (64 - len(names)) + [f"H.{x}" for x in ("meta", "fill", "trace", "merge", "cons")]
        print("tasks", int(ctr[0]), "members", int(ctr[1]), "routed per tier (Q, M1, M2, L, G, H)", ctr[6:12].tolist(), "outgrew into tier (S, M1, M2, L, G, H)", ctr[18:24].tolist(), file=sys.stderr)
        print("phase Mcycles", {n: round(float(v) / 1e6, 2) for n, v in zip(names, prof) if n != "-"}, file=sys.stderr)
        print("chain kernel: windows", int(prof[44]), "mean anchors", float(prof[42]) / max(1, int(prof[44])), "mean dirty sequences", float(prof[43]) / max(1, int(prof[44])),
              "Mcycles: stage-in", round(float(prof[48]) / 1e6, 1), "(part of idx.chain) segment flush", round(float(prof[49]) / 1e6, 1), "(part of idx.segments)", file=sys.stderr)
        print("tier L: chunk-rows", int(prof[46]), "rows", int(prof[47]), "fill cycles per chunk-row", float(prof[8 + 5 * 3 + 1]) / max(1, int(prof[46])), file=sys.stderr)
        print("chain kernel: windows with bad masks", int(prof[52]), "with correction rows", int(prof[50]), "rows", int(prof[51]), "windows on the slow path", int(prof[53]), "Mcycles there", round(float(prof[54]) / 1e6, 1), file=sys.stderr)
        print("index kernel detail, Mcycles:", {n: round(float(prof[i]) / 1e6, 1) for n, i in (("stage+clear", 55), ("count pass", 0), ("export scan", 56), ("export write", 2), ("tplhash", 7), ("support pass", 3),
              ("candidates", 58), ("P fill", 59), ("clean scan", 60), ("dirty+masks", 61), ("presence", 62), ("hand-over", 4))}, file=sys.stderr)
        print("GPU idle between the batch before and this one (wall clock, finish kernel end -> setup kernel start): ms", round(float(prof[57]) * 1e-5, 3), file=sys.stderr)
        for _k, _v in stage_ms.items():
            print(f"per-step {_k} ms (HIP events):", [round(v, 2) for v in _v], file=sys.stderr)
        print("stage ms", {k: round(v, 2) for k, v in eng.timings().items()}, file=sys.stderr)
        print("longest single task, Mcycles", {t: round(float(prof[36 + i]) / 1e6, 3) for i, t in enumerate(("S", "M1", "M2", "L", "G"))}, file=sys.stderr)
        if prof[72:120].any():  # a -DCW_DIAG build of the library (tools/diag_rows.sh): what the fills and tracebacks of the last batch did
            dn = ("rows<=64", "linear", "far>RC", "preds(nonlinear)", "rows_packed", "linear_pk", "far_pk", "tb_trips", "tb_slow", "tb_members", "far>16", "far_pk>16")
            for i, t in enumerate(("S", "M1", "M2", "L")):
                print(f"diag tier {t}:", {n: int(prof[72 + 12 * i + k]) for k, n in enumerate(dn)}, file=sys.stderr)
            dc = ("kind 0 (linear)", "kind 1", "kinds 2-3", "generic", "in-edges of generic rows", "slab loads of generic rows", "rows with a flag", "tb_trips", "members with code words in LDS", "members", "rows", "rows of members <= 31 bases")
            for i, t in enumerate(("S", "M1")):  # the recorded-decision fill (cw_poa_c.h) counts by row kind
                print(f"diag tier {t}, coded fill:", {n: int(prof[72 + 12 * i + k]) for k, n in enumerate(dc)}, file=sys.stderr)
    if driver_leg is not None and rank == 0:
        out["driver_strong_scaling"] = driver_leg
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
