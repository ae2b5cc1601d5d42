"""ctypes access to the CPU oracle (oracle/liboracle.so) and to the reference-built checker (oracle/_ref).
Test infrastructure only."""
import ctypes as C
import os

import numpy as np

from consent_amd.engine import Batch, Params, Result, WindowResults, alloc_results, _ptr, _result_struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_O = None
_R = None


def oracle():
    global _O
    if _O is None:
        _O = C.CDLL(os.environ.get("CW_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so"))
        _O.cwo_run.argtypes = [C.POINTER(Params), C.POINTER(Batch), C.POINTER(Result), C.c_void_p, C.c_int]
    return _O


def ref():
    """The reference's own alignmentWindows/alignmentPiles/utils/reverseComplement TUs (None if not built)."""
    global _R
    p = os.path.join(ROOT, "oracle", "_ref", "libconsent_ref.so")
    if _R is None and os.path.exists(p):
        _R = C.CDLL(p)
    return _R


STAT_NAMES = ["kmers", "tpl_anchors", "chain_len", "pair_tests", "segments", "poa_segments", "alignments", "dp_cells", "max_nodes", "max_seg_len", "link_calls", "nbr_calls", "alignments_routed", "dp_cells_routed"]


def oracle_run(params, batch, want_solid=True, threads=1):
    res = alloc_results(batch, want_solid, params.solid, params.k)
    b = batch.c_struct()
    r = _result_struct(res)
    stats = np.zeros(len(STAT_NAMES), np.uint64)
    rc = oracle().cwo_run(C.byref(params), C.byref(b), C.byref(r), _ptr(stats), threads)
    assert rc in (0, -4), rc
    return res, dict(zip(STAT_NAMES, (int(x) for x in stats)))


def oracle_poa(seqs):
    arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    lens = np.array([len(s) for s in seqs], np.uint32)
    out = np.zeros(4 * max(lens.max(), 1) + 64, np.uint8)
    n = C.c_uint32()
    rc = oracle().cwo_poa(arr, _ptr(lens), len(seqs), _ptr(out), len(out), C.byref(n))
    assert rc == 0
    return out[: n.value].tobytes().decode()


def oracle_weight_polish(cons, counts, k, solid, weight=True, polish=True):
    keys = np.array(list(counts.keys()), np.uint64)
    cnts = np.array(list(counts.values()), np.uint32)
    out = np.zeros(4 * len(cons) + 256, np.uint8)
    n = C.c_uint32()
    rc = oracle().cwo_weight_polish(cons.encode(), len(cons), _ptr(keys), _ptr(cnts), len(keys), k, solid, int(weight), int(polish), _ptr(out), len(out), C.byref(n))
    assert rc == 0
    return out[: n.value].tobytes().decode()


def _ovl_array(ovls):
    return np.ascontiguousarray(np.array(ovls, np.uint32).reshape(-1, 8))


def window_positions(lib_fn, tpl_len, ovls, min_support, window_size, window_overlap):
    o = _ovl_array(ovls)
    out = np.zeros(2 * (tpl_len // max(1, window_size - window_overlap) + 8), np.uint32)
    n = lib_fn(tpl_len, _ptr(o), len(o), min_support, window_size, window_overlap, _ptr(out), len(out) // 2)
    assert n >= 0
    return [tuple(int(x) for x in out[2 * i : 2 * i + 2]) for i in range(n)]


def window_pile(lib_fn, ovls, tpl, targets, q_beg, q_end, k):
    o = _ovl_array(ovls)
    tg = (C.c_char_p * len(targets))(*[t.encode() for t in targets])
    tl = np.array([len(t) for t in targets], np.uint32)
    out = np.zeros((len(ovls) + 1) * (q_end - q_beg + 200) * 2 + 1024, np.uint8)
    lens = np.zeros(len(ovls) + 2, np.uint32)
    n = lib_fn(_ptr(o), len(o), tpl.encode(), len(tpl), tg, _ptr(tl), len(targets), q_beg, q_end, k, _ptr(out), len(out), _ptr(lens), len(lens))
    assert n >= 0, n
    res, off = [], 0
    for i in range(n):
        res.append(out[off : off + lens[i]].tobytes().decode())
        off += int(lens[i])
    return res
