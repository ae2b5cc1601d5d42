#!/usr/bin/env python3
"""Condense the rocprofv3 databases written by tools/profile_r03.sh into profiles/<tag>.txt + profiles/<tag>.json.

usage: summarize_r03.py gpurun_out/prof_<tag>_<workload> <tag> <windows_per_step>

What is kept (every figure of bench.py's `roofline` / `secondary` objects can be recomputed from this file):
  * kernel time with ONE engine (a kernel's time = its work; trace_e1) and with TWO engines (throughput configuration; trace_e2);
  * FETCH_SIZE / WRITE_SIZE per kernel (own passes, one engine) and traffic = 2 x FETCH + WRITE (MI355X_MICROARCH.md, HBM section),
    per kernel and summed over the kernels of one step;
  * two passes of SQ counters per kernel (one engine) and what follows from them: VALU issue share, LDS busy share, bank conflicts.
"""
import json
import os
import sqlite3
import sys

OURS = ("cw_", "void cw_")
CLOCK_GHZ = 2.4       # MI355X peak engine clock (MI355X_MICROARCH.md); SQ_BUSY_CYCLES gives the measured one per kernel
N_SIMD, N_CU = 1024, 256


def short(name):
    return name.split("(")[0].replace("void ", "")


def per_kernel(db_path, what="avg"):
    out = {}
    if not os.path.exists(db_path):
        return out
    d = sqlite3.connect(db_path)
    for name, ctr, cnt, avg in d.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"):
        if name.startswith(OURS):
            out.setdefault(name, {})[ctr] = avg
            out[name]["launches"] = cnt
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    windows = int(sys.argv[3]) if len(sys.argv) > 3 else None
    lines, summary = [], {"kernels": {}, "source": src, "windows_per_step": windows}
    cmd = "python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pcie-steps 0 --workload <workload>"
    step_ms = {}
    for eng in ("e1", "e2"):
        p = os.path.join(src, f"trace_{eng}", "trace_results.db")
        if not os.path.exists(p):
            continue
        db = sqlite3.connect(p)
        lines.append(f"# rocprofv3 --kernel-trace --stats -- {cmd} --engines {eng[1]}   [{tag}]")
        lines.append(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
        for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if not name.startswith(OURS):
                continue
            lines.append(f"{name[:90]:90s} {calls:6d} {total:12.1f} {avg:12.1f} {pct:7.2f}")
            k = summary["kernels"].setdefault(name, {})
            k[f"calls_{eng}"] = calls
            k[f"avg_us_{eng}"] = avg
        # bench.py prints one JSON line: the step time of the profiled run itself
        log = os.path.join(src, f"bench_trace_{eng}.log")
        if os.path.exists(log):
            for ln in open(log):
                if ln.startswith("{") and '"ms_per_step"' in ln:
                    step_ms[eng] = json.loads(ln)["ms_per_step"]
                    lines.append(f"# bench.py under this trace: ms_per_step {step_ms[eng]:.2f}")
        lines.append("")
    summary["ms_per_step_under_trace"] = step_ms

    fetch = per_kernel(os.path.join(src, "pmc_fetch", "fetch_results.db"))
    write = per_kernel(os.path.join(src, "pmc_write", "write_results.db"))
    if fetch or write:
        lines.append("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (own passes, --engines 1); KiB per launch, averaged over the launches")
        lines.append("# traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM);")
        lines.append("# other widths and WRITE_SIZE are uncalibrated, Infinity-Cache hits are counted: an upper estimate of the HBM bytes")
        lines.append(f"{'kernel':90s} {'FETCH KiB':>14s} {'WRITE KiB':>14s} {'traffic MB':>12s}")
        tot = 0.0
        for name in sorted(set(fetch) | set(write)):
            f_, w_ = fetch.get(name, {}).get("FETCH_SIZE"), write.get(name, {}).get("WRITE_SIZE")
            k = summary["kernels"].setdefault(name, {})
            if f_ is not None:
                k["FETCH_SIZE_KiB"] = f_
            if w_ is not None:
                k["WRITE_SIZE_KiB"] = w_
            if f_ is not None and w_ is not None:
                k["traffic_bytes_per_launch"] = (2 * f_ + w_) * 1024
                tot += k["traffic_bytes_per_launch"]
                lines.append(f"{name[:90]:90s} {f_:14.1f} {w_:14.1f} {k['traffic_bytes_per_launch'] / 1e6:12.1f}")
        summary["traffic_bytes_per_step"] = tot
        lines.append(f"# whole step (sum over the kernels above, one launch each): {tot / 1e9:.3f} GB")
        lines.append("")

    sq = per_kernel(os.path.join(src, "pmc_sq1", "sq1_results.db"))
    sq2 = per_kernel(os.path.join(src, "pmc_sq2", "sq2_results.db"))
    for name, v in sq2.items():
        sq.setdefault(name, {}).update(v)
    if sq:
        c1 = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
        c2 = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"]
        for title, cols in (("pass 1", c1), ("pass 2", c2)):
            lines.append(f"# rocprofv3 --pmc {' '.join(cols)} --kernel-trace ({title}, own pass, --engines 1); per launch, averaged.  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles")
            lines.append(f"{'kernel':58s} " + " ".join(f"{c.replace('SQ_', '')[:16]:>16s}" for c in cols))
            for name in sorted(sq, key=lambda n: -sq[n].get("SQ_WAVE_CYCLES", 0)):
                lines.append(f"{short(name)[:58]:58s} " + " ".join(f"{sq[name].get(c, float('nan')):16.4g}" for c in cols))
            lines.append("")
        lines.append("# derived per kernel: share of the kernel's wave-cycles in which a wave issued VALU / LDS / anything, waited at s_waitcnt/barrier, or stalled on issue;")
        lines.append("# LDS array busy share = LDS_IDX_ACTIVE / (BUSY_CYCLES summed over the SQs ~ CU-cycles); bank-conflict share of the LDS-array cycles")
        lines.append(f"{'kernel':58s} {'valu/wave':>10s} {'lds/wave':>10s} {'active':>10s} {'wait':>10s} {'stall':>10s} {'conflict':>10s} {'share_wc':>10s}")
        tot_wc = sum(v.get("SQ_WAVE_CYCLES", 0) for v in sq.values())
        for name in sorted(sq, key=lambda n: -sq[n].get("SQ_WAVE_CYCLES", 0)):
            v = sq[name]
            wc = v.get("SQ_WAVE_CYCLES") or float("nan")
            k = summary["kernels"].setdefault(name, {})
            d = {
                "valu_active_per_wave_cycle": v.get("SQ_ACTIVE_INST_VALU", float("nan")) / wc,
                "lds_active_per_wave_cycle": v.get("SQ_ACTIVE_INST_LDS", float("nan")) / wc,
                "active_per_wave_cycle": v.get("SQ_ACTIVE_INST_ANY", float("nan")) / wc,
                "wait_per_wave_cycle": v.get("SQ_WAIT_ANY", float("nan")) / wc,
                "issue_stall_per_wave_cycle": v.get("SQ_WAIT_INST_ANY", float("nan")) / wc,
                "lds_bank_conflict_share": v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0.0,
                "share_of_wave_cycles": v.get("SQ_WAVE_CYCLES", 0) / tot_wc if tot_wc else 0.0,
            }
            k["sq"] = {c: v[c] for c in v if c.startswith("SQ_")}
            k["sq_derived"] = d
            lines.append(f"{short(name)[:58]:58s} " + " ".join(f"{d[x]:10.3f}" for x in ("valu_active_per_wave_cycle", "lds_active_per_wave_cycle", "active_per_wave_cycle", "wait_per_wave_cycle",
                                                                                          "issue_stall_per_wave_cycle", "lds_bank_conflict_share", "share_of_wave_cycles")))
        # alone on the GPU (the counter passes serialise the kernels): mean resident waves per SIMD and VALU issue rate over the kernel's own duration
        # (SQ_BUSY_CYCLES is summed over the 32 shader engines: / 32 = the kernel's duration in cycles)
        lines.append("")
        lines.append("# per kernel ALONE on the GPU (rocprofv3 counter mode serialises the kernels): duration = SQ_BUSY_CYCLES / 32 shader engines; mean resident waves per SIMD")
        lines.append("# = wave-cycles / (duration x 1024 SIMDs); VALU issue = INSTS_VALU x 4 cycles / (duration x 1024 SIMDs)")
        lines.append(f"{'kernel':58s} {'Mcycles':>10s} {'waves/SIMD':>10s} {'VALU issue':>10s}")
        for name in sorted(sq, key=lambda n: -sq[n].get("SQ_WAVE_CYCLES", 0)):
            v = sq[name]
            if not v.get("SQ_BUSY_CYCLES"):
                continue
            dur = v["SQ_BUSY_CYCLES"] / 32.0
            a = {"alone_mcycles": dur / 1e6, "alone_waves_per_simd": v.get("SQ_WAVE_CYCLES", 0) * 4 / (dur * N_SIMD), "alone_valu_issue": v.get("SQ_INSTS_VALU", 0) * 4 / (dur * N_SIMD)}
            summary["kernels"].setdefault(name, {}).setdefault("sq_derived", {}).update(a)
            lines.append(f"{short(name)[:58]:58s} {a['alone_mcycles']:10.2f} {a['alone_waves_per_simd']:10.2f} {a['alone_valu_issue']:10.3f}")
        # whole step: VALU issue slots used / available, LDS array cycles used / available, over the step time of the one-engine trace
        ms = step_ms.get("e1")
        if ms:
            valu = sum(v.get("SQ_INSTS_VALU", 0) for v in sq.values())
            lds_c = sum(v.get("SQ_LDS_IDX_ACTIVE", 0) for v in sq.values())
            cyc = ms * 1e-3 * CLOCK_GHZ * 1e9
            summary["whole_step"] = {
                "ms_per_step_e1": ms,
                "valu_issue_frac": valu * 4 / (N_SIMD * cyc),  # a wave64 VALU instruction occupies its SIMD for 4 cycles
                "lds_busy_frac": lds_c / (N_CU * cyc),         # LDS-array cycles / CU-cycles; x 256 B/clk = bytes against the ~150 TB/s aggregate
                "insts_valu": valu, "lds_idx_active": lds_c, "clock_ghz_assumed": CLOCK_GHZ,
            }
            lines.append("")
            lines.append(f"# whole step ({ms:.2f} ms, one engine, {CLOCK_GHZ} GHz assumed): VALU issue {summary['whole_step']['valu_issue_frac']:.3f} of 1024 SIMDs x 1 wave-instruction / 4 cycles;"
                         f" LDS array busy {summary['whole_step']['lds_busy_frac']:.3f} of 256 CUs")
    sq3 = per_kernel(os.path.join(src, "pmc_sq3", "sq3_results.db"))
    if sq3:
        lines.append("")
        lines.append("# rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace (pass 3, own pass, --engines 1); per launch, averaged")
        lines.append("# lane utilisation of the VALU instructions = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): 1.0 = every lane of every VALU instruction active")
        lines.append(f"{'kernel':58s} {'THREAD_CYC_VALU':>16s} {'ACTIVE_INST_VALU':>16s} {'INSTS_VALU':>16s} {'lane util':>10s}")
        t_thr = t_act = 0.0
        for name in sorted(sq3, key=lambda n: -sq3[n].get("SQ_WAVE_CYCLES", 0)):
            v = sq3[name]
            thr, act = v.get("SQ_THREAD_CYCLES_VALU", 0.0), v.get("SQ_ACTIVE_INST_VALU", 0.0)
            util = thr / (64.0 * act) if act else float("nan")
            t_thr += thr; t_act += act
            summary["kernels"].setdefault(name, {}).setdefault("sq_derived", {})["valu_lane_util"] = util
            lines.append(f"{short(name)[:58]:58s} {thr:16.4g} {act:16.4g} {v.get('SQ_INSTS_VALU', float('nan')):16.4g} {util:10.3f}")
        if t_act:
            summary.setdefault("whole_step", {})["valu_lane_util"] = t_thr / (64.0 * t_act)
            lines.append(f"# whole step: lane utilisation {t_thr / (64.0 * t_act):.3f}")
    os.makedirs("profiles", exist_ok=True)
    open(os.path.join("profiles", f"{tag}.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join("profiles", f"{tag}.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
