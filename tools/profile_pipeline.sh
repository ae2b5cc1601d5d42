#!/bin/bash
# rocprofv3 kernel trace of the native driver on a synthetic read set (tools/pipeline_bench.py writes the data and prints the command):
# per-kernel time of extraction, consensus and re-assembly end to end.  Output under gpurun_out/prof_<tag>_pipeline/.
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_pipeline
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CW_KEEP_DATA=$OUT/data python tools/pipeline_bench.py --genome 1500000 --cov 30 --profile pacbio --reps 1 > "$OUT/bench.log" 2>&1
D=$OUT/data
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- bin/CONSENT-correction -a $D/ovl.paf -s 3 -S 150 -l 500 -k 9 -c 8 -A 2 -f 4 -m 50 -j 1 -r $D/reads.fa -M 150 -p x > /dev/null 2> "$OUT/run.log"
rm -rf "$D"
tail -3 "$OUT/bench.log"
