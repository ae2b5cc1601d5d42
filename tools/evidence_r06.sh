#!/bin/bash
# Round-6 evidence on the GPU box, everything into gpurun_out/ev_r06/ (copied into profiles/ afterwards): verify build, bench lines of both
# workloads (with the CPU baseline legs), rocprofv3 trace + PMC passes summarized, driver lines (x1, x8), job-size model, pipeline trace,
# fuzz totals.  Usage: tools/evidence_r06.sh [fuzz_seconds]
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
FZ=${1:-300}
EV=gpurun_out/ev_r06
mkdir -p $EV
bash tools/verify_codes.sh > $EV/verify.txt 2>&1
python bench.py --steps 20 --warmup 5 > $EV/bench_r06_pacbio_d150_msa150.json 2> $EV/bench_d150.err
python bench.py --steps 20 --warmup 5 --workload pacbio_d30_msa20 > $EV/bench_r06_pacbio_d30_msa20.json 2> $EV/bench_d30.err
python bench.py --steps 10 --warmup 3 --engines 1 --cpu-sample 0 --pcie-steps 0 --driver-leg 0 > $EV/bench_r06_pacbio_d150_msa150_one_engine.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --engines 1 --windows 2048 --cpu-sample 0 --pcie-steps 0 --driver-leg 0 > $EV/bench_r06_pacbio_d150_msa150_one_engine_2048.json 2>/dev/null
for wl in pacbio_d150_msa150 pacbio_d30_msa20; do
  W=16384
  bash tools/profile_r03.sh r06 $wl all > $EV/prof_$wl.log 2>&1
  python tools/summarize_r03.py gpurun_out/prof_r06_$wl r06_$wl $W > /dev/null 2>&1
  cp profiles/r06_$wl.txt profiles/r06_$wl.json $EV/ 2>/dev/null
  cp profiles/r06_$wl.json $EV/latest_$wl.json 2>/dev/null
  rm -rf gpurun_out/prof_r06_$wl
done
# bench once more with the fresh counters beside it (roofline.traffic comes from profiles/latest_*.json)
cp $EV/latest_pacbio_d150_msa150.json profiles/latest_pacbio_d150_msa150.json; cp $EV/latest_pacbio_d30_msa20.json profiles/latest_pacbio_d30_msa20.json
python bench.py --steps 20 --warmup 5 > $EV/bench_r06_pacbio_d150_msa150.json 2> $EV/bench_d150.err
python bench.py --steps 20 --warmup 5 --workload pacbio_d30_msa20 > $EV/bench_r06_pacbio_d30_msa20.json 2> $EV/bench_d30.err
export CW_KEEP_DATA=/tmp/cw_drv_data
python bench.py --mode driver --gpus 1 --driver-copies 1 > $EV/driver_r06_x1.json 2> $EV/driver_x1.err
python bench.py --mode driver --gpus 1 --driver-copies 8 > $EV/driver_r06_x8.json 2> $EV/driver_x8.err
python tools/job_size_model.py > $EV/r06_job_size_model.txt 2>&1
JSM_MODE=parts python tools/job_size_model.py > $EV/r06_job_size_parts.txt 2>/dev/null
JSM_MODE=parts8 python tools/job_size_model.py > $EV/r06_job_size_parts_x8.txt 2>/dev/null
JSM_MODE=sweep1 python tools/job_size_model.py > $EV/r06_job_size_sweep_x1.txt 2>/dev/null
python tools/codeobj_audit.py > $EV/r06_codeobj.txt 2>&1
python tools/plan_sizes.py > $EV/r06_plan_sizes.txt 2>&1
CW_TASK_TRACE=1 python tools/task_trace.py > $EV/r06_task_trace_16384.txt 2>&1
CW_TRACE_WINDOWS=2048 CW_TASK_TRACE=1 python tools/task_trace.py > $EV/r06_task_trace_2048.txt 2>&1
unset CW_KEEP_DATA
bash tools/profile_pipeline.sh r06 > $EV/pipeline.log 2>&1
python - <<'PY' > $EV/r06_pipeline.txt 2>&1
import glob, sqlite3
dbs = glob.glob("gpurun_out/prof_r06_pipeline/trace/**/*.db", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- bin/CONSENT-correction -a ovl.paf -s 3 -S 150 -l 500 -k 9 -c 8 -A 2 -f 4 -m 50 -j 1 -r reads.fa -M 150 -p x   [r06_pipeline]")
print("# data: tools/pipeline_bench.py --genome 1500000 --cov 30 --profile pacbio (ground-truth PAF); two workers on one GPU")
for db in dbs[:1]:
    d = sqlite3.connect(db)
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>6s}")
    for n, c, t, a, pct in d.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{n[:90]:90s} {c:6d} {t:12.1f} {a:12.1f} {pct:6.2f}")
PY
rm -rf gpurun_out/prof_r06_pipeline
python tools/fuzz_parity.py $FZ 20260930 > $EV/fuzz_parity.txt 2>&1
python tools/fuzz_pipeline.py $FZ 20260930 > $EV/fuzz_pipeline.txt 2>&1
bash tools/fuzz_policy.sh "-DCW_POA_MODE=2" $((FZ / 2)) > $EV/fuzz_policy_ov.txt 2>&1
bash tools/fuzz_policy.sh "-DCW_POA_CONSENSUS=1" $((FZ / 2)) > $EV/fuzz_policy_hb.txt 2>&1
bash tools/fuzz_policy.sh "-DCW_POA_MODE=1" $((FZ / 2)) > $EV/fuzz_policy_sw.txt 2>&1
bash tools/fuzz_policy.sh "-DCW_POA_MATCH=2 -DCW_POA_MISMATCH=-4 -DCW_POA_GAP=-4" $((FZ / 2)) > $EV/fuzz_policy_scores.txt 2>&1
bash tools/fuzz_policy.sh "-DCW_CHAIN_TIE=1" $((FZ / 2)) > $EV/fuzz_policy_chain_tie.txt 2>&1
CW_FUZZ_CAP_BAR=0.05 CW_FUZZ_CAP_BAR_SHORT_K=0.2 bash tools/fuzz_policy.sh "-DCW_SEG_MISSING_ANCHOR=1" $((FZ / 2)) > $EV/fuzz_policy_extrapolate.txt 2>&1
CW_FUZZ_CAP_BAR=0.05 CW_FUZZ_CAP_BAR_SHORT_K=0.2 bash tools/fuzz_policy.sh "-DCW_POA_MATCH=16 -DCW_POA_MISMATCH=-12 -DCW_POA_GAP=-16" $((FZ / 2)) > $EV/fuzz_policy_scores16.txt 2>&1
CW_FUZZ_CAP_BAR=0.05 CW_FUZZ_CAP_BAR_SHORT_K=0.2 bash tools/fuzz_policy.sh "-DCW_POA_GAP_MODEL=1 -DCW_POA_GAP_OPEN=-8 -DCW_POA_GAP_EXT=-6" $((FZ / 2)) > $EV/fuzz_policy_affine.txt 2>&1
tail -2 $EV/verify.txt; tail -1 $EV/fuzz_parity.txt; tail -2 $EV/fuzz_pipeline.txt; tail -1 $EV/fuzz_policy_ov.txt; tail -1 $EV/fuzz_policy_hb.txt; tail -1 $EV/fuzz_policy_sw.txt; tail -1 $EV/fuzz_policy_scores.txt; tail -1 $EV/fuzz_policy_chain_tie.txt; tail -1 $EV/fuzz_policy_extrapolate.txt; tail -1 $EV/fuzz_policy_scores16.txt; tail -1 $EV/fuzz_policy_affine.txt
python -c "
import json
for f in ('bench_r06_pacbio_d150_msa150','bench_r06_pacbio_d30_msa20','bench_r06_pacbio_d150_msa150_one_engine','driver_r06_x1','driver_r06_x8'):
    try:
        d=json.load(open('$EV/'+f+'.json')); print(f, round(d['value']), round(d['ms_per_step'],2), d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('scalar_value'))
    except Exception as e: print(f, 'ERR', e)
"
