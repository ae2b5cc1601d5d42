#!/bin/bash
# Launch-shape sweep on the GPU box: engines per GPU x tier-S work-groups per CU x tier-S routing bound; prints ms per step.
cd "$(dirname "$0")/.."
export CONSENT_AMD_LIB=${CONSENT_AMD_LIB:-$PWD/consent_amd/aids/libconsent_amd.so} # the experiment knobs exist in the test-aid build only (csrc/cw_env.h)
W=${1:-pacbio_d150_msa150}
for e in 1 2; do for s in 3 4 6; do for nodes in 96 112 128; do
  r=$(CW_WGS_S=$s CW_S_ROUTE_NODES=$nodes python bench.py --steps 6 --warmup 2 --engines $e --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],2))")
  echo "engines $e wgs_s $s route_nodes $nodes: $r ms"
done; done; done
