#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE
# in their own passes (PMC never combined with other trace domains).  Output under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
WL=${2:-pacbio_d30_msa20}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pcie-steps 0 --workload $WL"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/bench_trace.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/bench_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/bench_write.log" 2>&1
find "$OUT" -name "*.csv" | head -40
find "$OUT" -name "*stats*.csv" -exec sh -c 'echo "== $1"; head -20 "$1"' _ {} \;
du -sh "$OUT"
