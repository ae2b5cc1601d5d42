#!/bin/bash
cd "$(dirname "$0")/.."
export CONSENT_AMD_LIB=${CONSENT_AMD_LIB:-$PWD/consent_amd/aids/libconsent_amd.so} # the experiment knobs exist in the test-aid build only (csrc/cw_env.h)
W=${1:-pacbio_d150_msa150}
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_M2_PRIO=3 $SRC -o /tmp/libconsent_amd_p3.so &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_M2_PRIO=2 $SRC -o /tmp/libconsent_amd_p2.so &
wait
run() { python bench.py --steps 8 --warmup 3 --engines $1 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],2))"; }
for e in 2 1; do
echo "engines $e default: $(run $e)"
echo "engines $e M2 prio 2: $(CONSENT_AMD_LIB=/tmp/libconsent_amd_p2.so run $e)"
echo "engines $e M2 prio 3: $(CONSENT_AMD_LIB=/tmp/libconsent_amd_p3.so run $e)"
for m2 in 4 6 8; do for l in 2 4 6; do
  echo "engines $e wgs_m2 $m2 wgs_l $l: $(CW_WGS_M2=$m2 CW_WGS_L=$l run $e)   prio3: $(CONSENT_AMD_LIB=/tmp/libconsent_amd_p3.so CW_WGS_M2=$m2 CW_WGS_L=$l run $e)"
done; done
done
