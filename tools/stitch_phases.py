"""Where a window's time goes in the re-assembly kernel (GPU box): builds the library once more with -DCW_TEST_AIDS -DCW_ST_PROF=1 (the debug
trace then holds the shader clocks of a window's phases), runs cw_stitch_device on synthetic reads of several kbp and prints the totals.
Usage: python tools/stitch_phases.py [n_reads] [extra -D flags ...]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    from consent_amd import _build

    lib = os.path.join(tempfile.mkdtemp(), "libconsent_amd_stprof.so")
    subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCW_TEST_AIDS", "-DCW_ST_PROF=1", *sys.argv[2:], *_build.SRC, "-o", lib], check=True)
    os.environ["CONSENT_AMD_LIB"] = lib
    os.environ["CW_STITCH_TRACE"] = "1"
    import consent_amd as ca
    import test_gpu_stitch as ts

    spec = ts.make_reads(7, n_reads, 14, lo=4000, hi=9000)
    reads, jobs, pos, piles = ts.build(spec)
    eng = ca.Engine(ca.Params(ts.K, ts.SOLID, 8, 2, 150))
    try:
        batch = ca.pack_piles(piles)
        res = eng.run(batch, want_solid=True)
        eng.stitch(ca.pack_piles([reads]), np.array(jobs, np.uint32), np.array(pos, np.uint32), batch, res, 500, 50, True)
        tr = np.zeros((len(piles), 8), np.uint32)
        eng.lib.cw_debug_stitch_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        rc = eng.lib.cw_debug_stitch_trace(eng.handle, len(piles), tr.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
    finally:
        eng.close()
    ok = tr[:, 0] != 0xFFFFFFFF
    t = tr[ok].astype(np.float64)
    names = ["load consensus + slice", "alignment (two sweeps)", "shift down", "overlap: compare + solid counts", "overlap: second alignment + banded traceback", "replace + keep"]
    tot = t[:, :6].sum()
    print(f"{int(ok.sum())} windows of {len(piles)}, mean consensus {t[:, 6].mean():.0f}, mean slice {t[:, 7].mean():.0f}; clocks per window {tot / ok.sum():.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:48s} {t[:, k].sum() / tot * 100:5.1f} %   mean {t[:, k].mean():9.0f}   windows with it {(t[:, k] > 0).sum()}")


if __name__ == "__main__":
    main()
