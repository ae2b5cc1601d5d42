#!/bin/bash
# A/B of one compile-time flag on the GPU box: tools/ab_flag.sh "-DCW_S_EDGES_LDS=0" [workload]  -- two-engine bench steps of the default
# build and of the build with the flag, and the one-engine phase cycle totals of both.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
F="$1"; W=${2:-pacbio_d150_msa150}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $F $SRC -o /tmp/libconsent_amd_flag.so || exit 1
for v in default flag; do
  if [ $v = flag ]; then export CONSENT_AMD_LIB=/tmp/libconsent_amd_flag.so; else unset CONSENT_AMD_LIB; fi
  a=$(python bench.py --steps 8 --warmup 3 --engines 2 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],2))")
  CW_PROFILE=1 python bench.py --steps 3 --warmup 2 --engines 1 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W > /dev/null 2> gpurun_out/abflag_$v.txt
  echo "== $v ($F): two engines $a ms/step"; grep "phase Mcycles" gpurun_out/abflag_$v.txt | grep -o "'[SMLQ][12]*\.[a-z]*': [0-9.]*" | tr '\n' ' '; echo; grep "^stage ms" gpurun_out/abflag_$v.txt
done
