cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for wl in pacbio_d30_msa20 pacbio_d150_msa150; do
  CW_PROFILE=1 timeout 200 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --workload $wl 2>&1 | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['stage_ms'])
    else: print(l[:260])"
done
