#!/bin/bash
# Work-groups per CU of the tiers (CW_WGS_S / _M1 / _M2 / _L of the aids build) on the current LDS sizes.  GPU box only.
run() { local n=$1 l=$2; shift 2; env "$@" CONSENT_AMD_LIB=$PWD/$l python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload ${WL:-pacbio_d150_msa150} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$n', round(d['ms_per_step'],2))"; }
A=consent_amd/aids/libconsent_amd.so
for r in 1 2 3; do
  run base $A X=1
  run S5 $A CW_WGS_S=5
  run S6 $A CW_WGS_S=6
  run S5M2x5 $A CW_WGS_S=5 CW_WGS_M2=5
  run S5M1x6 $A CW_WGS_S=5 CW_WGS_M1=6
done
