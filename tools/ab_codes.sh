#!/bin/bash
# A/B of the recorded-decision path (cw_poa_c.h) against the matrix path: the library once more with -DCW_POA_CODES=0 into /tmp, then
# one-engine bench steps of both with CW_PROFILE=1 (phase cycle totals per tier on stderr).  GPU box only.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
W=${1:-pacbio_d150_msa150}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_POA_CODES=0 $SRC -o /tmp/libconsent_amd_nocodes.so
for v in codes nocodes; do
  if [ $v = nocodes ]; then export CONSENT_AMD_LIB=/tmp/libconsent_amd_nocodes.so; else unset CONSENT_AMD_LIB; fi
  CW_PROFILE=1 python bench.py --steps 3 --warmup 2 --engines 1 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W > gpurun_out/ab_${v}_$W.json 2> gpurun_out/ab_${v}_$W.txt
  echo "== $v"; grep "phase Mcycles\|^stage ms\|longest" gpurun_out/ab_${v}_$W.txt | cut -c1-1800
done
