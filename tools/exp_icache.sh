#!/bin/bash
# Is the POA stage short of instruction fetch?  Two PMC passes over the bench step (one engine): the instruction cache's requests /
# hits / misses and the SQs' fetch counters, per kernel.  GPU box only; prints a table.
set -u
WL=${1:-pacbio_d150_msa150}
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache_$WL
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
BASE="python bench.py --steps 4 --warmup 2 --cpu-sample 0 --pcie-steps 0 --workload $WL --engines 1"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES --kernel-trace -d "$OUT/p1" -o p1 -- $BASE > "$OUT/p1.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --kernel-trace -d "$OUT/p2" -o p2 -- $BASE > "$OUT/p2.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d "$OUT/p3" -o p3 -- $BASE > "$OUT/p3.log" 2>&1
tail -1 "$OUT"/p*.log | cut -c1-200
python - "$OUT" <<'P'
import sqlite3, sys, os, glob
for p in ("p1", "p2", "p3"):
    dbs = glob.glob(os.path.join(sys.argv[1], p, "**", "*.db"), recursive=True)
    if not dbs:
        print(p, "no db"); continue
    d = sqlite3.connect(dbs[0])
    rows = {}
    for name, ctr, cnt, avg in d.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"):
        if name.startswith(("cw_", "void cw_")):
            rows.setdefault(name.split("(")[0].replace("void ", "")[:48], {})[ctr] = avg
    cols = sorted({c for r in rows.values() for c in r})
    print("%-48s" % "kernel" + "".join("%22s" % c[-21:] for c in cols))
    for k, r in sorted(rows.items(), key=lambda kv: -max(kv[1].values())):
        print("%-48s" % k + "".join("%22.4g" % r.get(c, 0) for c in cols))
P
