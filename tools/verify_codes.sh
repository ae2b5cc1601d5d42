#!/bin/bash
# -DCW_POA_VERIFY build into /tmp: every member of <= 63 bases in tiers S / M1 / M2 is aligned twice, by the matrix path and by the
# recorded-decision path of cw_poa_c.h, and the two tracebacks are compared on the device (counters in BatchCounters::prof[120..125]).
# GPU box only.
set -e
cd "$(dirname "$0")/.."
SRC="consent_amd/csrc/cw_engine.cpp consent_amd/csrc/cw_synth.cpp consent_amd/csrc/cw_hostio.cpp consent_amd/csrc/cw_driver.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCW_POA_VERIFY $SRC -o /tmp/libconsent_amd_verify.so
CONSENT_AMD_LIB=/tmp/libconsent_amd_verify.so python tools/verify_codes.py "$@"
