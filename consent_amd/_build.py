"""Build libconsent_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("cw_engine.cpp", "cw_synth.cpp", "cw_hostio.cpp")]
HDR = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith(".h")]
HDR += [os.path.join(HERE, "..", "include", f) for f in ("consent_amd.h", "cw_policy.h")]
OUT = os.path.join(HERE, "libconsent_amd.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SRC + HDR)


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *SRC, "-o", OUT]
    if verbose:
        print("[consent_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
