#!/bin/bash
# Round-3 evidence for one workload on the GPU box: rocprofv3 kernel trace + stats with ONE engine (a kernel's
# time = its work) and with TWO engines (the throughput configuration), FETCH_SIZE / WRITE_SIZE in their own passes, and two passes
# of SQ counters (eight slots each).  PMC is never combined with other trace domains.  Output under gpurun_out/prof_<tag>_<workload>/.
set -u
TAG=${1:-r03}
WL=${2:-pacbio_d150_msa150}
WHAT=${3:-all} # all | trace | pmc
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BASE="python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pcie-steps 0 --driver-leg 0 --workload $WL"
if [ "$WHAT" != pmc ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_e1" -o trace -- $BASE --engines 1 > "$OUT/bench_trace_e1.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_e2" -o trace -- $BASE --engines 2 > "$OUT/bench_trace_e2.log" 2>&1
fi
if [ "$WHAT" != trace ]; then
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- $BASE --engines 1 > "$OUT/bench_fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- $BASE --engines 1 > "$OUT/bench_write.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace \
      -d "$OUT/pmc_sq1" -o sq1 -- $BASE --engines 1 > "$OUT/bench_sq1.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace \
      -d "$OUT/pmc_sq2" -o sq2 -- $BASE --engines 1 > "$OUT/bench_sq2.log" 2>&1
  # round 5: how full the lanes of the VALU instructions are -- thread-cycles against instruction-cycles
  timeout 600 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace \
      -d "$OUT/pmc_sq3" -o sq3 -- $BASE --engines 1 > "$OUT/bench_sq3.log" 2>&1
fi
find "$OUT" -name "*.db" | head
tail -2 "$OUT"/bench_*.log
du -sh "$OUT"
