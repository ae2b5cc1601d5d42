#!/bin/bash
# tier L / M2 launch-shape sweep + M2 with and without the recorded-decision path.  GPU box only.
cd "$(dirname "$0")/.."
export CONSENT_AMD_LIB=${CONSENT_AMD_LIB:-$PWD/consent_amd/aids/libconsent_amd.so} # the experiment knobs exist in the test-aid build only (csrc/cw_env.h)
W=${1:-pacbio_d150_msa150}
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_M2_CODES=1 $SRC -o /tmp/libconsent_amd_m2c.so
run() { python bench.py --steps 6 --warmup 2 --engines 2 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],2))"; }
echo "M2 matrix path (default): $(run) ms"
echo "M2 coded: $(CONSENT_AMD_LIB=/tmp/libconsent_amd_m2c.so run) ms"
for l in 2 3 4; do for m2 in 3 4; do for m1 in 4 5; do
  echo "wgs_l $l wgs_m2 $m2 wgs_m1 $m1: $(CW_WGS_L=$l CW_WGS_M2=$m2 CW_WGS_M1=$m1 run) ms"
done; done; done
