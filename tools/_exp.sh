cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for wl in pacbio_d30_msa20 pacbio_d150_msa150; do
  timeout 200 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --workload $wl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['kernel'])"
done
timeout 300 python tools/pipeline_bench.py 2>&1 | tail -4
