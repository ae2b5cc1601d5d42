"""What the job size costs on ONE GPU, for the multi-GPU model of DESIGN.md: the E. coli-scale ONT set (BASELINE configs[3]:
4.6 Mbp, 30x, ~3.2e5 windows) through cw_run_correction with two workers on device 0 and the job size forced to what N GPUs would get --
32768 windows (the default on one or two GPUs) down to the floor the driver picks for sixteen workers (windows / (4 x 16)).  Prints, per job
size, jobs, seconds inside the driver, the serial part (read index) and windows/s; the model for N GPUs is then
    T(N) = t_index + t_first_job_latency + windows / (N x rate(job size of N)).
GPU box only."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pipeline_bench as pb
from consent_amd.pipeline import run_correction

d = os.environ.get("CW_KEEP_DATA") or tempfile.mkdtemp()
fa, paf, _, n_reads, n_ovl = pb.generate(d, 4600000, 30, "ont")


def run(fa_, paf_, devices, per_job):
    best = None
    for rep in range(2):
        fd = os.open(os.devnull, os.O_WRONLY)
        st = run_correction(fa_, paf_, fd, min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150,
                            nb_threads=1, devices=devices, windows_per_batch=per_job)
        os.close(fd)
        if best is None or st.ms_total < best.ms_total:
            best = st
    return best


if os.environ.get("JSM_MODE") == "parts":
    # What ONE of N GPUs sees of this set, with the driver's own choices (workers per device, windows per job): a set of 1/N of the genome.
    # T(N) = index of the whole set + the 1/N set's time after its index.
    full = run(fa, paf, None, 0)
    print(json.dumps({"set": "x1", "windows": int(full.windows), "jobs": int(full.jobs), "s_total": full.ms_total / 1e3, "s_index": full.ms_index / 1e3}), flush=True)
    for n in (2, 4, 8):
        dn = os.path.join(d, "part%d" % n)
        os.makedirs(dn, exist_ok=True)
        fa_n, paf_n, _, _, _ = pb.generate(dn, 4600000 // n, 30, "ont")
        st = run(fa_n, paf_n, None, 0)
        t_n = full.ms_index / 1e3 + (st.ms_total - st.ms_index) / 1e3
        print(json.dumps({"set": "x1 / %d" % n, "windows": int(st.windows), "jobs": int(st.jobs), "s_total": st.ms_total / 1e3, "s_index": st.ms_index / 1e3,
                          "modelled_T_N": t_n, "speed_up_over_one_gpu": full.ms_total / 1e3 / t_n}), flush=True)
    sys.exit(0)
if os.environ.get("JSM_MODE") == "sweep1":
    # Round 6: the whole x1 set on one GPU under (workers per device, windows per job): is the driver's choice (two workers, jobs of 32 768) still the best
    # one now that an engine's plan is a quarter smaller and the bench runs four batches at a time?
    for workers in (2, 3, 4):
        for per_job in (0, 32768, 24576, 16384, 12288, 8192):
            if per_job == 0 and workers != 2:
                continue
            if per_job == 0:
                os.environ.pop("CW_WORKERS_PER_DEVICE", None)
            else:
                os.environ["CW_WORKERS_PER_DEVICE"] = str(workers)
            st = run(fa, paf, None, per_job)
            print(json.dumps({"set": "x1", "workers": "driver's own" if per_job == 0 else workers, "windows_per_job": per_job or "driver's own", "jobs": int(st.jobs), "s_total": round(st.ms_total / 1e3, 4),
                              "s_after_index": round((st.ms_total - st.ms_index) / 1e3, 4), "windows_per_s": round(st.windows / (st.ms_total / 1e3))}), flush=True)
    os.environ.pop("CW_WORKERS_PER_DEVICE", None)
    sys.exit(0)
if os.environ.get("JSM_MODE") == "parts8":
    # The x8 set (eight copies of the x1 set: 2.6e6 windows, full-size jobs on every device up to N = 8): what one of N GPUs sees is 8 / N copies.
    # T(N) = read index of the whole x8 set + the part's time after its own index.
    res = {}
    for copies in (8, 4, 2, 1):
        fa_c, paf_c = pb.replicate(fa, paf, copies)
        st = run(fa_c, paf_c, None, 0)
        res[copies] = st
        t_n = res[8].ms_index / 1e3 + (st.ms_total - st.ms_index) / 1e3
        print(json.dumps({"set": "x8 / %d" % (8 // copies), "copies_per_device": copies, "windows": int(st.windows), "jobs": int(st.jobs), "s_total": round(st.ms_total / 1e3, 4),
                          "s_index": round(st.ms_index / 1e3, 4), "modelled_T_N": round(t_n, 4), "speed_up_over_one_gpu": round(res[8].ms_total / 1e3 / t_n, 3)}), flush=True)
    sys.exit(0)
if os.environ.get("JSM_MODE") == "sweep":
    # Round 6: what one of N GPUs sees of the x1 set (a set of 1/N of the genome), under every (workers per device, windows per job) the driver could
    # choose -- the table its own choice (cw_driver.cpp: workers by windows per device, eight jobs per device) is checked against.
    full = run(fa, paf, None, 0)
    print(json.dumps({"set": "x1", "windows": int(full.windows), "jobs": int(full.jobs), "s_total": full.ms_total / 1e3, "s_index": full.ms_index / 1e3}), flush=True)
    for n in [int(x) for x in os.environ.get("JSM_PARTS", "8,4").split(",")]:
        dn = os.path.join(d, "part%d" % n)
        os.makedirs(dn, exist_ok=True)
        fa_n, paf_n, _, _, _ = pb.generate(dn, 4600000 // n, 30, "ont")
        os.environ.pop("CW_WORKERS_PER_DEVICE", None)
        st = run(fa_n, paf_n, None, 0)
        t_n = full.ms_index / 1e3 + (st.ms_total - st.ms_index) / 1e3
        print(json.dumps({"set": "x1 / %d" % n, "choice": "driver's own", "windows": int(st.windows), "jobs": int(st.jobs), "s_after_index": (st.ms_total - st.ms_index) / 1e3,
                          "modelled_T_N": t_n, "speed_up_over_one_gpu": full.ms_total / 1e3 / t_n}), flush=True)
        for workers in (2, 3, 4, 6, 8):
            for jobs_per_dev in (2, 3, 4, 6, 8, 12):
                per_job = max(1024, int(st.windows) // jobs_per_dev + 1)
                if jobs_per_dev < workers // 2:
                    continue
                os.environ["CW_WORKERS_PER_DEVICE"] = str(workers)
                s2 = run(fa_n, paf_n, None, per_job)
                t2 = full.ms_index / 1e3 + (s2.ms_total - s2.ms_index) / 1e3
                print(json.dumps({"set": "x1 / %d" % n, "workers": workers, "windows_per_job": per_job, "jobs": int(s2.jobs), "s_after_index": round((s2.ms_total - s2.ms_index) / 1e3, 4),
                                  "modelled_T_N": round(t2, 4), "speed_up_over_one_gpu": round(full.ms_total / 1e3 / t2, 3)}), flush=True)
        os.environ.pop("CW_WORKERS_PER_DEVICE", None)
    sys.exit(0)
rows = []
n_workers = int(os.environ.get("JSM_WORKERS", "2"))  # workers (engines) on device 0
sizes = [int(x) for x in os.environ.get("JSM_SIZES", "32768,20000,10000,5000,4096").split(",")]
for per_job in sizes:
    best = None
    for rep in range(2):
        fd = os.open(os.devnull, os.O_WRONLY)
        t0 = time.perf_counter()
        st = run_correction(fa, paf, fd, min_support=3, max_support=150, window_size=500, mer_size=9, common_kmers=8, min_anchors=2, solid_thresh=4, window_overlap=50, max_msa=150,
                            nb_threads=1, devices=[0] * n_workers, windows_per_batch=per_job)
        wall = time.perf_counter() - t0
        os.close(fd)
        if best is None or st.ms_total < best[0]:
            best = (st.ms_total, st.ms_index, int(st.windows), int(st.jobs), wall)
    ms_total, ms_index, windows, jobs, wall = best
    rows.append({"workers": n_workers, "windows_per_job": per_job, "jobs": jobs, "windows": windows, "s_total": ms_total / 1e3, "s_index": ms_index / 1e3, "windows_per_s_after_index": windows / ((ms_total - ms_index) / 1e3)})
    print(json.dumps(rows[-1]), flush=True)
base = rows[0]
print("model: T(N) = s_index + windows / (N x rate(job size)) with the job size the driver picks for 2N workers (windows / (8N), floor 4096, cap 32768)")
for n in (1, 2, 4, 8):
    want = max(4096, min(32768, base["windows"] // (8 * n) + 1))
    r = min(rows, key=lambda x: abs(x["windows_per_job"] - want))
    t = base["s_index"] + base["windows"] / (n * r["windows_per_s_after_index"])
    print(f"N={n}: job size {want} (measured at {r['windows_per_job']}), modelled {t:.3f} s, speed-up over N=1 {((base['s_index'] + base['windows'] / rows[0]['windows_per_s_after_index']) / t):.2f}x")
