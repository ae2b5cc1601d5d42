#!/bin/bash
# Experiment: the marginal cost of one scalar / one vector instruction in the row loop of cw_poa_c.h's fill.  Three more builds of the
# library (+16 scalar, +16 vector, +16 of each per row) and one-engine bench steps with CW_PROFILE=1.  GPU box only.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
W=${1:-pacbio_d150_msa150}
for v in base salu valu both; do
  case $v in base) F="";; salu) F="-DCW_EXP_SALU=16";; valu) F="-DCW_EXP_VALU=16";; both) F="-DCW_EXP_SALU=16 -DCW_EXP_VALU=16";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $F $SRC -o /tmp/libconsent_amd_exp_$v.so &
done
wait
for v in base salu valu both; do
  CONSENT_AMD_LIB=/tmp/libconsent_amd_exp_$v.so CW_PROFILE=1 python bench.py --steps 3 --warmup 2 --engines 1 --cpu-sample 0 --pcie-steps 0 --alone-steps 0 --workload $W > gpurun_out/exp_${v}.json 2> gpurun_out/exp_${v}.txt
  echo "== $v"; grep "phase Mcycles" gpurun_out/exp_${v}.txt | grep -o "'S.fill': [0-9.]*\|'M1.fill': [0-9.]*\|'M2.fill': [0-9.]*\|'S.trace': [0-9.]*" | tr '\n' ' '; grep "^stage ms" gpurun_out/exp_${v}.txt
done
