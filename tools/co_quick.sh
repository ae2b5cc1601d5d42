#!/bin/bash
# The device code of the engine with extra -D flags, compiled alone (no host pass: ~40 s) and audited: registers, spills, scratch.
#   tools/co_quick.sh [-DCW_S_EU=4 ...]   (the object is left at /tmp/cw_quick.co)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only --no-gpu-bundle-output "$@" -c consent_amd/csrc/cw_engine.cpp -o ${CO_OUT:-/tmp/cw_quick.co} && \
  python tools/codeobj_audit.py ${CO_OUT:-/tmp/cw_quick.co} ${AUDIT_ARGS}
