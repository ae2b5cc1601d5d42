"""Build libconsent_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("cw_engine.cpp", "cw_synth.cpp", "cw_hostio.cpp", "cw_driver.cpp")]
HDR = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith(".h")]
HDR += [os.path.join(HERE, "..", "include", f) for f in ("consent_amd.h", "cw_policy.h")]
OUT = os.path.join(HERE, "libconsent_amd.so")
# the same sources with -DCW_TEST_AIDS (csrc/cw_env.h): test aids, experiment knobs and the opt-in kernels exist only here; same file name in its
# own directory so that the executables of bin/ run against it with LD_LIBRARY_PATH (tests, tools/, bench.py --mode driver's dry run)
AIDS_DIR = os.path.join(HERE, "aids")
AIDS_OUT = os.path.join(AIDS_DIR, "libconsent_amd.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(OUT) or not os.path.exists(AIDS_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(AIDS_OUT))
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SRC + HDR)


BIN = os.path.normpath(os.path.join(HERE, "..", "bin"))
CLI = os.path.join(HERE, "cli")
# (executable, source, extra flags): the wrappers' five programs (reference Makefile:5), plain g++ over the C ABI
BINS = [
    ("CONSENT-correction", "consent_main.cpp", ["-DCW_DRIVER_POLISHING=0"]),
    ("CONSENT-polishing", "consent_main.cpp", ["-DCW_DRIVER_POLISHING=1"]),
    ("explode", "paf_tools.cpp", ["-DCW_TOOL=1"]),
    ("merge", "paf_tools.cpp", ["-DCW_TOOL=2"]),
    ("reformatPAF", "paf_tools.cpp", ["-DCW_TOOL=3"]),
]


def bins_stale():
    t_lib = os.path.getmtime(OUT) if os.path.exists(OUT) else 0
    for exe, src, _ in BINS:
        p = os.path.join(BIN, exe)
        if not os.path.exists(p) or os.path.getmtime(p) < max(os.path.getmtime(os.path.join(CLI, src)), os.path.getmtime(HDR[-2])):
            return True
        if os.path.getmtime(p) < t_lib - 86400 * 365:
            return True
    return False


def build_bins(verbose=True):
    os.makedirs(BIN, exist_ok=True)
    for exe, src, flags in BINS:
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", *flags, "-I", os.path.join(HERE, "..", "include"), os.path.join(CLI, src), "-L", HERE, "-lconsent_amd",
               "-Wl,-rpath,$ORIGIN/../consent_amd", "-o", os.path.join(BIN, exe)]
        if verbose:
            print("[consent_amd] " + " ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)


def build(force=False, verbose=True):
    if force or stale():
        os.makedirs(AIDS_DIR, exist_ok=True)
        cmds = [[hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *SRC, "-o", OUT],
                [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCW_TEST_AIDS", *SRC, "-o", AIDS_OUT]]
        procs = []
        for cmd in cmds:  # side by side: each takes most of a minute
            if verbose:
                print("[consent_amd] " + " ".join(cmd), file=sys.stderr)
            procs.append(subprocess.Popen(cmd))
        for cmd, pr in zip(cmds, procs):
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        force = True
    if force or bins_stale():
        build_bins(verbose)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
