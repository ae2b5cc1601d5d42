#!/bin/bash
# Work-group sizes of tiers M1 / M2 (waves per work-group, and with them the LDS a work-group takes: what packs into a CU's 160 KB beside the
# other tiers) -- aids builds consent_amd/aids/libconsent_amd_{base,m2w2,m1w2,both}.so, work-groups per CU by environment.  GPU box only.
run() { # name lib env...
  local n=$1 l=$2; shift 2
  env "$@" CONSENT_AMD_LIB=$PWD/consent_amd/aids/libconsent_amd_$l.so python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$n', round(d['ms_per_step'],2))"
}
for r in 1 2; do
  run base base X=1
  run m2w2x6 m2w2 CW_WGS_M2=6
  run m2w2x5 m2w2 CW_WGS_M2=5
  run m1w2x10 m1w2 CW_WGS_M1=10
  run m1w2x8 m1w2 CW_WGS_M1=8
  run both both CW_WGS_M1=10 CW_WGS_M2=6
done
