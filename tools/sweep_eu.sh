#!/bin/bash
cd "$(dirname "$0")/.."
export CONSENT_AMD_LIB=${CONSENT_AMD_LIB:-$PWD/consent_amd/aids/libconsent_amd.so} # the experiment knobs exist in the test-aid build only (csrc/cw_env.h)
W=${1:-pacbio_d150_msa150}
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_S_EU=6 -DCW_M1_EU=6 $SRC -o /tmp/lib_66.so &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_S_EU=8 -DCW_M1_EU=6 $SRC -o /tmp/lib_86.so &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_S_EU=6 -DCW_M1_EU=5 $SRC -o /tmp/lib_65.so &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_S_EU=4 -DCW_M1_EU=4 $SRC -o /tmp/lib_44.so &
wait
run() { python bench.py --steps 8 --warmup 3 --engines 2 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],2))"; }
echo "default (5,5): $(run) $(run)"
for v in 44 65 66 86; do for s in 4 6; do
echo "S/M1 waves per SIMD $v wgs_s $s: $(CONSENT_AMD_LIB=/tmp/lib_$v.so CW_WGS_S=$s run) $(CONSENT_AMD_LIB=/tmp/lib_$v.so CW_WGS_S=$s CW_WGS_M1=6 run)"
done; done
