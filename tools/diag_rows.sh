#!/bin/bash
# Diagnostic build (-DCW_DIAG: the POA fills and tracebacks count rows, linear rows, far predecessor reads, traceback trips and slow
# steps per tier) into gpurun_out/, then one-engine bench steps with CW_PROFILE=1 so that bench.py prints the counts.  GPU box only.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_DIAG $SRC -o /tmp/libconsent_amd_diag.so
CONSENT_AMD_LIB=/tmp/libconsent_amd_diag.so CW_PROFILE=1 python bench.py --steps 2 --warmup 1 --engines 1 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload ${1:-pacbio_d150_msa150} > gpurun_out/diag_${1:-pacbio_d150_msa150}.json 2> gpurun_out/diag_${1:-pacbio_d150_msa150}.txt
tail -25 gpurun_out/diag_${1:-pacbio_d150_msa150}.txt
