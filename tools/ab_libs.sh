#!/bin/bash
# Same-box A/B of two builds of the library (boxes of the pool differ by up to 10 %: only runs on ONE box compare): alternates
# bench.py between the two libraries, ROUNDS times each, and prints ms per step (two engines) and the one-engine stage total.
#   tools/ab_libs.sh consent_amd/aids/libconsent_amd_A.so consent_amd/libconsent_amd.so [rounds] [workload]
A=$1; B=$2; R=${3:-3}; WL=${4:-pacbio_d150_msa150}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    CONSENT_AMD_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pcie-steps 0 --driver-leg 0 --workload $WL 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$L', round(d['ms_per_step'],2), 'one engine', round(d['stage_ms_one_batch_in_flight']['total'],2), 'Q', round(d['stage_ms_one_batch_in_flight']['poa_q'],2), 'H', round(d['stage_ms_one_batch_in_flight'].get('poa_h',0),2), 'S', round(d['stage_ms_one_batch_in_flight']['poa'],2), 'M1', round(d['stage_ms_one_batch_in_flight']['poa_m1'],2), 'index', round(d['stage_ms_one_batch_in_flight']['index'],2), 'finish', round(d['stage_ms_one_batch_in_flight']['finish'],2), 'M2', round(d['stage_ms_one_batch_in_flight']['poa_m2'],2), 'L', round(d['stage_ms_one_batch_in_flight']['poa_large'],2))"
  done
done
