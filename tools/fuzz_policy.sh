#!/bin/bash
# The randomised differential tester under a non-default policy of include/cw_policy.h: both sides built with the given -D flags into a
# scratch directory, then tools/fuzz_parity.py for the given number of seconds.  GPU box only.
#   tools/fuzz_policy.sh "-DCW_POA_MODE=2" 120
set -e
cd "$(dirname "$0")/.."
FLAGS=${1:--DCW_POA_MODE=2}
SECS=${2:-120}
OUT=$(mktemp -d)
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAGS $SRC -o $OUT/libconsent_amd.so
make -s -C oracle policy OUT=$OUT POLICY="$FLAGS"
# (the capacity bar of the default policy does not carry over: the heaviest bundle's consensuses are longer and pass the finish kernel's 3072 characters more often)
CW_FUZZ_CAP_BAR=${CW_FUZZ_CAP_BAR:-0.002} CONSENT_AMD_LIB=$OUT/libconsent_amd.so CW_ORACLE_LIB=$OUT/liboracle.so python tools/fuzz_parity.py $SECS 20260930 2>&1 | tail -3
