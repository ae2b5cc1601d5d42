#!/usr/bin/env python3
"""The engine's scratch plan for a batch of given dimensions, component by component (host arithmetic: cw_debug_plan, no device needed).
  python tools/plan_sizes.py [windows depth [window_len [cus]]] ...   default: the bench's batch, a driver job at depth 30, a small job"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("CONSENT_AMD_LIB") or os.path.join(ROOT, "consent_amd", "libconsent_amd.so"))
lib.cw_debug_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
NAMES = ["total", "window records", "solid table", "segments", "arena", "tasks + members", "tier lists", "slabs S", "slabs M1", "slabs M2", "slabs L", "slabs G",
         "tier Q / H rows", "anchor blocks", "matrix + count fallbacks"]


def plan(windows, depth, wlen=500, cus=256, tmax=1024):
    n_seqs = windows * (depth + 1)
    n_words = n_seqs * ((wlen + 15) // 16 + 1)
    out = (C.c_uint64 * 15)()
    assert lib.cw_debug_plan(9, 4, windows, n_seqs, n_words, cus, 1, tmax, out) == 0
    return list(out)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a)] if a else [(16384, 150), (32768, 30), (10240, 30), (2048, 30), (24, 30)]
    for c in cases:
        v = plan(*c)
        print(f"windows {c[0]}, depth {c[1]}: total {v[0] / 1e9:.2f} GB")
        print("   " + ", ".join(f"{n} {x / 1e6:.0f} MB" for n, x in zip(NAMES[1:], v[1:])) + f"; the rest {(v[0] - sum(v[1:])) / 1e6:.0f} MB")
