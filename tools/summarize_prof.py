#!/usr/bin/env python3
"""Condense the rocprofv3 databases written by tools/profile_round.sh into the text + JSON summary kept under profiles/."""
import json
import os
import sqlite3
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    windows = int(sys.argv[3]) if len(sys.argv) > 3 else None
    out_txt = os.path.join("profiles", f"{tag}.txt")
    out_json = os.path.join("profiles", f"{tag}.json")
    lines, summary = [], {"kernels": {}, "source": src, "windows_per_step": windows}
    db = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
    cmd = os.environ.get("CW_PROF_CMD", "python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pcie-steps 0 --workload ...")
    lines.append(f"# rocprofv3 --kernel-trace --stats -- {cmd}   [{tag}]")
    lines.append(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if not name.startswith(("cw_", "void cw_", "synth_kernel")):
            continue
        lines.append(f"{name[:90]:90s} {calls:6d} {total:12.1f} {avg:12.1f} {pct:7.2f}")
        summary["kernels"][name] = {"calls": calls, "avg_us": avg, "pct": pct}
    for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        p = os.path.join(src, sub, sub.replace("pmc_", "") + "_results.db")
        if not os.path.exists(p):
            continue
        d = sqlite3.connect(p)
        lines.append("")
        lines.append(f"# rocprofv3 --pmc {counter} --kernel-trace (own pass); value per launch as reported (KiB), averaged over launches")
        for name, cnt, avg in d.execute("select kernel_name,count(*),avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
            if not name.startswith(("cw_", "void cw_", "synth_kernel")):
                continue
            lines.append(f"{name[:90]:90s} launches {cnt:4d}  {counter} {avg:14.1f} KiB/launch")
            summary["kernels"].setdefault(name, {})[counter + "_KiB"] = avg
    lines.append("")
    lines.append("# Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced streaming reads by 2x on gfx950;")
    lines.append("# other access widths and WRITE_SIZE are uncalibrated.  traffic(bytes) = (2*FETCH_SIZE + WRITE_SIZE) * 1024 is an upper estimate;")
    lines.append("# Infinity-Cache hits are counted, so HBM proper is at most this.")
    for name, k in summary["kernels"].items():
        if "FETCH_SIZE_KiB" in k and "WRITE_SIZE_KiB" in k:
            k["traffic_bytes_per_launch"] = (2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024
    os.makedirs("profiles", exist_ok=True)
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(out_json, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
