"""ctypes binding of the C ABI (include/consent_amd.h) + the host-side mirror of CONSENT's operator.

Mirrors ``computeConsensusReadCorrection`` / ``computeConsensusAssemblyPolishing``
(reference src/correctionMSA.h:8,10): same argument names and meaning, batched over windows.
"""
import ctypes as C
import os

import numpy as np

WIN_CONSENSUS, WIN_TEMPLATE, WIN_OVERFLOW = 0, 1, 2
_HERE = os.path.dirname(os.path.abspath(__file__))


class EngineError(RuntimeError):
    pass


class Params(C.Structure):
    """cw_params: merSize, solidThresh, commonKMers, minAnchors, maxMSA (reference main.cpp:17-26 names)."""

    _fields_ = [("k", C.c_uint32), ("solid", C.c_uint32), ("common_kmers", C.c_uint32), ("min_anchors", C.c_uint32), ("max_msa", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [
        ("n_windows", C.c_uint32),
        ("n_seqs", C.c_uint32),
        ("n_words", C.c_uint64),
        ("win_first_seq", C.c_void_p),
        ("seq_len", C.c_void_p),
        ("seq_word_off", C.c_void_p),
        ("bases", C.c_void_p),
    ]


class Result(C.Structure):
    _fields_ = [
        ("cons", C.c_void_p),
        ("cons_off", C.c_void_p),
        ("cons_len", C.c_void_p),
        ("win_status", C.c_void_p),
        ("solid", C.c_void_p),
        ("solid_off", C.c_void_p),
        ("solid_len", C.c_void_p),
    ]


class ReadSet(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("read_len", C.c_void_p), ("read_word_off", C.c_void_p), ("bases", C.c_void_p)]


class SynthSpec(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("first_window", C.c_uint64),
        ("n_windows", C.c_uint32),
        ("depth", C.c_uint32),
        ("window_len", C.c_uint32),
        ("err_permille", C.c_uint32),
        ("sub_w", C.c_uint32),
        ("ins_w", C.c_uint32),
        ("del_w", C.c_uint32),
        ("seq_stride_words", C.c_uint32),
    ]

    @classmethod
    def pacbio(cls, n_windows, depth, first_window=0, seed=0xC0115E17, window_len=500):
        return cls(seed, first_window, n_windows, depth, window_len, 120, 10, 60, 30, (window_len + 60) // 16 + 2)

    @classmethod
    def ont(cls, n_windows, depth, first_window=0, seed=0xC0115E17, window_len=500):
        return cls(seed, first_window, n_windows, depth, window_len, 120, 30, 30, 40, (window_len + 60) // 16 + 2)


def lib_path():
    """The in-tree library; CONSENT_AMD_LIB names another build of the same sources (tests/test_gpu_policy.py: a non-default policy)."""
    return os.environ.get("CONSENT_AMD_LIB") or os.path.join(_HERE, "libconsent_amd.so")


_LIB = None
_LIBS = {}  # path -> loaded library
AIDS_LIB = os.path.join(_HERE, "aids", "libconsent_amd.so")  # the -DCW_TEST_AIDS build of the same sources (csrc/cw_env.h): tests and tools only


def use_library(path=None):
    """Make `path` (None: the default, lib_path()) the library every later Engine / helper of this process uses; returns the previous choice.
    What the tests that need a test aid do for their duration (tests/conftest.py `aids`): both builds can be loaded side by side."""
    global _LIB
    prev = _LIB
    _LIB = None
    if path is not None:
        _LIB = _load(path)
    return prev


def load_library():
    """Load libconsent_amd.so.  Fails loudly: there is no fallback implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    _LIB = _load(lib_path())
    return _LIB


def _load(p):
    if p in _LIBS:
        return _LIBS[p]
    try:  # PyTorch-ROCm ships its own HIP runtime: let it initialise first so both sides share one runtime and one context
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    if not os.path.exists(p):
        raise EngineError(f"{p} is missing: build it with `python -m consent_amd._build` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(p)
    lib.cw_version.restype = C.c_char_p
    lib.cw_strerror.restype = C.c_char_p
    lib.cw_strerror.argtypes = [C.c_int]
    lib.cw_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
    lib.cw_destroy.argtypes = [C.c_void_p]
    lib.cw_destroy.restype = None
    lib.cw_run.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Result)]
    lib.cw_run_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Result), C.c_void_p]
    lib.cw_submit.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Result), C.POINTER(C.c_int)]
    lib.cw_wait.argtypes = [C.c_void_p, C.c_int]
    lib.cw_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.cw_host_free.argtypes = [C.c_void_p]
    lib.cw_host_free.restype = None
    lib.cw_poll.argtypes = [C.c_void_p]
    lib.cw_poll.restype = C.c_int
    lib.cw_configure.argtypes = [C.c_void_p, C.c_uint32]
    lib.cw_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]
    lib.cw_debug_win_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.cw_debug_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.cw_extract_piles_device.argtypes = [C.c_void_p, C.POINTER(ReadSet), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p]
    lib.cw_stitch_device.argtypes = [C.c_void_p, C.POINTER(ReadSet), C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Batch), C.POINTER(Result), C.c_uint32, C.c_uint32,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cw_index_reads.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.cw_index_reads_append.argtypes = [C.c_void_p, C.c_char_p]
    lib.cw_read_index_free.argtypes = [C.c_void_p]
    lib.cw_read_index_free.restype = None
    lib.cw_read_index_count.argtypes = [C.c_void_p]
    lib.cw_read_index_count.restype = C.c_uint32
    lib.cw_read_index_view.argtypes = [C.c_void_p, C.POINTER(ReadSet), C.POINTER(C.c_uint64)]
    lib.cw_read_index_find.argtypes = [C.c_void_p, C.c_char_p]
    lib.cw_read_index_find.restype = C.c_int32
    lib.cw_read_index_name.argtypes = [C.c_void_p, C.c_uint32]
    lib.cw_read_index_name.restype = C.c_char_p
    lib.cw_paf_open.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.cw_paf_next_pile.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.cw_paf_close.argtypes = [C.c_void_p]
    lib.cw_paf_close.restype = None
    lib.cw_paf_reformat.argtypes = [C.c_char_p, C.c_char_p]
    lib.cw_paf_explode.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32)]
    lib.cw_paf_merge.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32]
    lib.cw_window_positions.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.cw_pack_sequence.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.cw_pack_sequence.restype = C.c_int64
    lib.cw_synth_sizes.argtypes = [C.POINTER(SynthSpec), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.cw_synth_host.argtypes = [C.POINTER(SynthSpec), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cw_synth_device.argtypes = [C.c_void_p, C.POINTER(SynthSpec), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _LIBS[p] = lib
    return lib


def _check(lib, rc, what, allow_capacity=False):
    if rc == 0 or (allow_capacity and rc == -4):
        return rc
    raise EngineError(f"{what}: {lib.cw_strerror(rc).decode()} ({rc})")


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class HostBatch:
    """Host arrays laid out as cw_batch."""

    def __init__(self, win_first_seq, seq_len, seq_word_off, bases):
        self.win_first_seq = np.ascontiguousarray(win_first_seq, np.uint32)
        self.seq_len = np.ascontiguousarray(seq_len, np.uint32)
        self.seq_word_off = np.ascontiguousarray(seq_word_off, np.uint64)
        self.bases = np.ascontiguousarray(bases, np.uint32)

    @property
    def n_windows(self):
        return len(self.win_first_seq) - 1

    def c_struct(self):
        return Batch(self.n_windows, len(self.seq_len), len(self.bases), _ptr(self.win_first_seq), _ptr(self.seq_len), _ptr(self.seq_word_off), _ptr(self.bases))

    def pile(self, w):
        """Decode window w back to a list of ASCII strings (test helper)."""
        out = []
        for s in range(int(self.win_first_seq[w]), int(self.win_first_seq[w + 1])):
            n = int(self.seq_len[s])
            o = int(self.seq_word_off[s])
            words = self.bases[o : o + (n + 15) // 16]
            codes = ((words[:, None] >> (30 - 2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).reshape(-1)[:n]
            out.append(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode())
        return out

    def slice(self, w0, w1):
        s0, s1 = int(self.win_first_seq[w0]), int(self.win_first_seq[w1])
        return HostBatch(self.win_first_seq[w0 : w1 + 1] - s0, self.seq_len[s0:s1], self.seq_word_off[s0:s1], self.bases)


def pack_piles(piles):
    """List of piles (each a list of ACGT strings, template first) -> HostBatch (reference 2-bit alphabet, utils.cpp:21-32)."""
    lib = load_library()
    n_seqs = sum(len(p) for p in piles)
    wfs = np.zeros(len(piles) + 1, np.uint32)
    lens = np.zeros(n_seqs, np.uint32)
    offs = np.zeros(n_seqs, np.uint64)
    total_words = sum((len(s) + 15) // 16 for p in piles for s in p)
    bases = np.zeros(max(total_words, 1), np.uint32)
    si = 0
    wo = 0
    for w, p in enumerate(piles):
        wfs[w] = si
        for s in p:
            raw = s.encode()
            lens[si] = len(raw)
            offs[si] = wo
            got = lib.cw_pack_sequence(raw, len(raw), C.c_void_p(bases.ctypes.data + 4 * wo), total_words - wo)
            if got < 0:
                raise EngineError("cw_pack_sequence failed")
            wo += got
            si += 1
    wfs[len(piles)] = si
    return HostBatch(wfs, lens, offs, bases)


def window_positions(tpl_len, overlaps, min_support, window_size, window_overlap):
    """cw_window_positions: `overlaps` is an (n,6) uint32 array of cw_overlap rows; returns [(beg, end), ...]."""
    lib = load_library()
    ov = np.ascontiguousarray(overlaps, np.uint32).reshape(-1, 6)
    n = C.c_uint32()
    cap = tpl_len // max(1, window_size - window_overlap) + 8
    out = np.zeros(2 * cap, np.uint32)
    _check(lib, lib.cw_window_positions(tpl_len, _ptr(ov) if len(ov) else None, len(ov), min_support, window_size, window_overlap, _ptr(out), cap, C.byref(n)), "cw_window_positions")
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value)]


def synth_host(spec):
    lib = load_library()
    ns, nw = C.c_uint32(), C.c_uint64()
    _check(lib, lib.cw_synth_sizes(C.byref(spec), C.byref(ns), C.byref(nw)), "cw_synth_sizes")
    wfs = np.zeros(spec.n_windows + 1, np.uint32)
    lens = np.zeros(ns.value, np.uint32)
    offs = np.zeros(ns.value, np.uint64)
    bases = np.zeros(nw.value, np.uint32)
    _check(lib, lib.cw_synth_host(C.byref(spec), _ptr(wfs), _ptr(lens), _ptr(offs), _ptr(bases)), "cw_synth_host")
    return HostBatch(wfs, lens, offs, bases)


def paf_reformat(in_path, out_path):
    """reformatPAF (src/reformatPAF.cpp): swap the query and target columns of a PAF file."""
    lib = load_library()
    _check(lib, lib.cw_paf_reformat(os.fsencode(in_path), os.fsencode(out_path)), "cw_paf_reformat")


def paf_explode(in_path, out_prefix):
    """explode (src/explode.cpp): returns the list of chunk files written (out_prefix_1 ...)."""
    lib = load_library()
    n = C.c_uint32()
    _check(lib, lib.cw_paf_explode(os.fsencode(in_path), os.fsencode(out_prefix), C.byref(n)), "cw_paf_explode")
    return [f"{out_prefix}_{i}" for i in range(1, n.value + 1)]


def paf_merge(out_path, headers_path, in_paths):
    """merge (src/merge.cpp): gather every read's lines from the chunks, in the order of the header file."""
    lib = load_library()
    arr = (C.c_char_p * max(len(in_paths), 1))(*[os.fsencode(p) for p in in_paths])
    _check(lib, lib.cw_paf_merge(os.fsencode(out_path), os.fsencode(headers_path), arr, len(in_paths)), "cw_paf_merge")


class ReadIndex:
    """indexReads (utils.cpp:166-205) through the library's host feeder: names, lengths and the 2-bit read set."""

    def __init__(self, path, *more_paths):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.cw_index_reads(os.fsencode(path), C.byref(h)), f"cw_index_reads({path})")
        self.handle = h
        for p in more_paths:
            _check(self.lib, self.lib.cw_index_reads_append(h, os.fsencode(p)), f"cw_index_reads_append({p})")
        n = self.lib.cw_read_index_count(h)
        view, nw = ReadSet(), C.c_uint64()
        _check(self.lib, self.lib.cw_read_index_view(h, C.byref(view), C.byref(nw)), "cw_read_index_view")
        self.names = [self.lib.cw_read_index_name(h, i).decode() for i in range(n)]
        as_np = lambda ptr, ct, cnt: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(max(cnt, 1),))[:cnt].copy()
        self.seq_len = as_np(view.read_len, C.c_uint32, n)
        self.seq_word_off = as_np(view.read_word_off, C.c_uint64, n)
        self.bases = as_np(view.bases, C.c_uint32, nw.value)

    def find(self, name):
        return self.lib.cw_read_index_find(self.handle, name.encode())

    def sequence(self, i):
        o, n = int(self.seq_word_off[i]), int(self.seq_len[i])
        w = self.bases[o : o + (n + 15) // 16]
        codes = ((w[:, None] >> (30 - 2 * np.arange(16, dtype=np.uint32))) & 3).reshape(-1)[:n]
        return np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()

    def close(self):
        if self.handle:
            self.lib.cw_read_index_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PafReader:
    """getNextReadPile (alignmentPiles.cpp:22-58): iterate over (template id, template length, overlaps (n,6), resMatches (n,))."""

    def __init__(self, path, index, max_support=150):
        self.lib = load_library()
        self.index = index
        self.max_support = max_support
        h = C.c_void_p()
        _check(self.lib, self.lib.cw_paf_open(os.fsencode(path), index.handle, max_support, C.byref(h)), f"cw_paf_open({path})")
        self.handle = h

    def __iter__(self):
        ov = np.zeros((max(self.max_support, 1), 6), np.uint32)
        rm = np.zeros(max(self.max_support, 1), np.uint32)
        tpl, tlen, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
        while True:
            _check(self.lib, self.lib.cw_paf_next_pile(self.handle, C.byref(tpl), C.byref(tlen), _ptr(ov), _ptr(rm), len(ov), C.byref(n)), "cw_paf_next_pile")
            if n.value == 0:
                return
            yield tpl.value, tlen.value, ov[: n.value].copy(), rm[: n.value].copy()

    def close(self):
        if self.handle:
            self.lib.cw_paf_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WindowResults:
    def __init__(self, cons, cons_off, cons_len, status, solid=None, solid_off=None, solid_len=None):
        self.cons, self.cons_off, self.cons_len, self.status = cons, cons_off, cons_len, status
        self.solid, self.solid_off, self.solid_len = solid, solid_off, solid_len

    def consensus(self, w):
        o = int(self.cons_off[w])
        return self.cons[o : o + int(self.cons_len[w])].tobytes().decode()

    def solid_kmers(self, w):
        o = int(self.solid_off[w])
        return self.solid[o : o + int(self.solid_len[w])]


def cons_slot_bytes(k, tpl_len):
    """CW_CONS_SLOT_BYTES of include/consent_amd.h: the consensus slot cw_plan_results_device gives a window (numpy-friendly)."""
    tpl = np.asarray(tpl_len, np.int64)
    if k >= 9:
        return np.clip(4 * tpl + 1024, 3072, 32768)
    if k == 8:
        return np.minimum(16 * tpl + 1024, 32768)
    return np.full_like(tpl, 32768)


def alloc_results(batch, want_solid=True, solid_thresh=4, k=9):
    W = batch.n_windows
    wfs = batch.win_first_seq.astype(np.int64)
    tpl = batch.seq_len[wfs[:-1]].astype(np.int64)
    cap = cons_slot_bytes(k, tpl)  # as cw_plan_results_device sizes it (cw_pack.h)
    cons_off = np.zeros(W + 1, np.uint64)
    cons_off[1:] = np.cumsum(cap)
    cons = np.zeros(int(cons_off[-1]), np.uint8)
    cons_len = np.zeros(W, np.uint32)
    status = np.full(W, 255, np.uint8)
    if not want_solid:
        return WindowResults(cons, cons_off, cons_len, status)
    csum = np.concatenate([[0], np.cumsum(batch.seq_len.astype(np.int64))])
    per_win = csum[wfs[1:]] - csum[wfs[:-1]]
    scap = per_win // max(1, solid_thresh) + 16
    solid_off = np.zeros(W + 1, np.uint64)
    solid_off[1:] = np.cumsum(scap)
    solid = np.zeros(int(solid_off[-1]), np.uint32)
    solid_len = np.zeros(W, np.uint32)
    return WindowResults(cons, cons_off, cons_len, status, solid, solid_off, solid_len)


def _result_struct(r):
    return Result(
        _ptr(r.cons), _ptr(r.cons_off), _ptr(r.cons_len), _ptr(r.status),
        _ptr(r.solid) if r.solid is not None else None,
        _ptr(r.solid_off) if r.solid is not None else None,
        _ptr(r.solid_len) if r.solid is not None else None,
    )


class Engine:
    """One engine per process per GPU (cw_create / cw_destroy)."""

    def __init__(self, params, device=0):
        self.lib = load_library()
        self.params = params
        self.device = device
        self.handle = C.c_void_p()
        _check(self.lib, self.lib.cw_create(C.byref(params), device, C.byref(self.handle)), "cw_create")

    def close(self):
        if self.handle:
            self.lib.cw_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, batch, want_solid=True):
        """Host buffers in, host buffers out (cw_run).  Windows that overflow a capacity carry WIN_OVERFLOW."""
        res = alloc_results(batch, want_solid, self.params.solid, self.params.k)
        b = batch.c_struct()
        r = _result_struct(res)
        _check(self.lib, self.lib.cw_run(self.handle, C.byref(b), C.byref(r)), "cw_run", allow_capacity=True)
        return res

    def submit(self, batch, res):
        """cw_submit: enqueue a host batch (HostBatch + WindowResults from alloc_results); returns (ticket, keep-alive).  Up to two in flight."""
        b = batch.c_struct()
        r = _result_struct(res)
        t = C.c_int(-1)
        _check(self.lib, self.lib.cw_submit(self.handle, C.byref(b), C.byref(r), C.byref(t)), "cw_submit")
        return t.value, (batch, res, b, r)

    def wait(self, ticket):
        """cw_wait: blocks until the batch of `ticket` is done and its WindowResults are filled."""
        _check(self.lib, self.lib.cw_wait(self.handle, ticket), "cw_wait", allow_capacity=True)

    def run_device(self, batch_struct, result_struct, stream=None):
        _check(self.lib, self.lib.cw_run_device(self.handle, C.byref(batch_struct), C.byref(result_struct), stream), "cw_run_device")

    def extract_piles(self, reads, overlaps, jobs, k):
        """Device-side getAlignmentWindowsSequences (cw_extract_piles_device).  `reads` is a HostBatch-like packing of the read
        set (one "window" holding every read), `overlaps` an (n,6) uint32 array (q_start, q_end, t_read, t_start, t_end, strand;
        ends inclusive), `jobs` an (m,5) uint32 array (tpl_read, q_beg, q_end, ovl_first, ovl_count).  Returns a HostBatch."""
        import torch

        dev = torch.device("cuda", self.device)

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)

        t_len, t_off, t_bases = up(reads.seq_len, np.int32), up(reads.seq_word_off, np.int64), up(np.concatenate([reads.bases, np.zeros(1, np.uint32)]), np.int32)
        ov = np.ascontiguousarray(overlaps, np.uint32).reshape(-1, 6)
        jb = np.ascontiguousarray(jobs, np.uint32).reshape(-1, 5)
        t_ov = up(ov if len(ov) else np.zeros((1, 6), np.uint32), np.int32)
        t_jb = up(jb, np.int32)
        rs = ReadSet(len(reads.seq_len), t_len.data_ptr(), t_off.data_ptr(), t_bases.data_ptr())
        ns, nw = C.c_uint32(), C.c_uint64()
        rc = self.lib.cw_extract_piles_device(self.handle, C.byref(rs), t_ov.data_ptr(), len(ov), t_jb.data_ptr(), len(jb), k, None, None, None, None, 0, 0, C.byref(ns), C.byref(nw), None)
        if rc not in (0, -4):
            _check(self.lib, rc, "cw_extract_piles_device(size)")  # -4 here only says "now you know the sizes"; a real capacity error repeats below
        o_wfs = torch.zeros(len(jb) + 1, dtype=torch.int32, device=dev)
        o_len = torch.zeros(max(ns.value, 1), dtype=torch.int32, device=dev)
        o_off = torch.zeros(max(ns.value, 1), dtype=torch.int64, device=dev)
        o_bases = torch.zeros(max(nw.value, 1) + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)  # the engine launches on its own stream: torch's fills above must have landed
        _check(self.lib, self.lib.cw_extract_piles_device(self.handle, C.byref(rs), t_ov.data_ptr(), len(ov), t_jb.data_ptr(), len(jb), k, o_wfs.data_ptr(), o_len.data_ptr(),
                                                          o_off.data_ptr(), o_bases.data_ptr(), ns.value, nw.value, C.byref(ns), C.byref(nw), None), "cw_extract_piles_device")
        torch.cuda.synchronize(dev)
        return HostBatch(o_wfs.cpu().numpy().view(np.uint32), o_len.cpu().numpy().view(np.uint32)[: ns.value], o_off.cpu().numpy().view(np.uint64)[: ns.value],
                         o_bases.cpu().numpy().view(np.uint32)[: max(nw.value, 1)])

    def stitch(self, reads, jobs, win_pos, batch, results, window_size=500, window_overlap=50, do_trim=True):
        """Device-side alignConsensus + trimRead + dropRead (cw_stitch_device).  `reads`: HostBatch-like packing of the read set;
        `jobs`: (n,3) uint32 (read, win_first, win_count); `win_pos`: (W,2) uint32 window (beg,end); `batch`/`results`: the piles
        and what the engine returned for them.  Returns [(corrected_read, status)]."""
        import torch

        dev = torch.device("cuda", self.device)

        def up(a, dt):
            a = np.ascontiguousarray(a)
            if a.size == 0:
                a = np.zeros(1, a.dtype)
            return torch.from_numpy(a.view(dt)).to(dev)

        jb = np.ascontiguousarray(jobs, np.uint32).reshape(-1, 3)
        t_len, t_off, t_bases = up(reads.seq_len, np.int32), up(reads.seq_word_off, np.int64), up(np.concatenate([reads.bases, np.zeros(1, np.uint32)]), np.int32)
        rs = ReadSet(len(reads.seq_len), t_len.data_ptr(), t_off.data_ptr(), t_bases.data_ptr())
        b_wfs, b_len, b_off, b_bases = up(batch.win_first_seq, np.int32), up(batch.seq_len, np.int32), up(batch.seq_word_off, np.int64), up(np.concatenate([batch.bases, np.zeros(1, np.uint32)]), np.int32)
        bs = Batch(batch.n_windows, len(batch.seq_len), len(batch.bases), b_wfs.data_ptr(), b_len.data_ptr(), b_off.data_ptr(), b_bases.data_ptr())
        r = results
        r_cons, r_coff, r_clen, r_st = up(r.cons, np.uint8), up(r.cons_off, np.int64), up(r.cons_len, np.int32), up(r.status, np.uint8)
        r_sol, r_soff, r_slen = up(r.solid, np.int32), up(r.solid_off, np.int64), up(r.solid_len, np.int32)
        rstruct = Result(r_cons.data_ptr(), r_coff.data_ptr(), r_clen.data_ptr(), r_st.data_ptr(), r_sol.data_ptr(), r_soff.data_ptr(), r_slen.data_ptr())
        t_jb, t_pos = up(jb, np.int32), up(np.ascontiguousarray(win_pos, np.uint32).reshape(-1), np.int32)
        cap = 2 * reads.seq_len[jb[:, 0]].astype(np.int64) + 1024
        out_off = np.zeros(len(jb) + 1, np.uint64)
        out_off[1:] = np.cumsum(cap)
        t_ooff = up(out_off, np.int64)
        t_out = torch.zeros(int(out_off[-1]) + 1, dtype=torch.uint8, device=dev)
        t_olen = torch.zeros(len(jb) + 1, dtype=torch.int32, device=dev)
        t_ost = torch.full((len(jb) + 1,), 255, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)  # the engine launches on its own stream: torch's fills above must have landed
        _check(self.lib, self.lib.cw_stitch_device(self.handle, C.byref(rs), t_jb.data_ptr(), len(jb), t_pos.data_ptr(), C.byref(bs), C.byref(rstruct), window_size, window_overlap,
                                                   int(bool(do_trim)), t_out.data_ptr(), t_ooff.data_ptr(), t_olen.data_ptr(), t_ost.data_ptr(), None), "cw_stitch_device")
        torch.cuda.synchronize(dev)
        out, olen, ost = t_out.cpu().numpy(), t_olen.cpu().numpy(), t_ost.cpu().numpy()
        return [(out[int(out_off[i]) : int(out_off[i]) + int(olen[i])].tobytes().decode("latin-1"), int(ost[i])) for i in range(len(jb))]

    def configure(self, max_template_len):
        """cw_configure: the longest template (window) this engine will see, before its first run -- templates beyond 1024 + k - 1 bases need a larger
        scratch plan and the chain kernel's long instance (up to 2048 + k - 1 bases)."""
        _check(self.lib, self.lib.cw_configure(self.handle, int(max_template_len)), "cw_configure")

    def idle(self):
        """cw_poll: True when the last run_device on this engine has completed (never blocks)."""
        rc = self.lib.cw_poll(self.handle)
        if rc < 0:
            _check(self.lib, rc, "cw_poll")
        return rc == 1

    def phase(self):
        """cw_poll as it is: 1 done / nothing launched, 2 running but in its tail, 0 running."""
        rc = self.lib.cw_poll(self.handle)
        if rc < 0:
            _check(self.lib, rc, "cw_poll")
        return rc

    def timings(self):
        ms = (C.c_float * 16)()
        names = (C.c_char_p * 16)()
        n = C.c_int()
        _check(self.lib, self.lib.cw_last_timings(self.handle, ms, names, 16, C.byref(n)), "cw_last_timings")
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    def profile(self):
        c = np.zeros(30, np.uint32)  # n_tasks, n_members, 3 cursors, any_overflow, then n_tier / next_tier / n_over / next_over for the six tiers
        p = np.zeros(128, np.uint64)  # phase cycle totals (cw_device.h BatchCounters::prof)
        _check(self.lib, self.lib.cw_debug_profile(self.handle, _ptr(c), len(c), _ptr(p), len(p), None, None), "cw_debug_profile")
        return c, p

    def win_info(self, n_windows):
        a = np.zeros((n_windows, 16), np.uint32)
        _check(self.lib, self.lib.cw_debug_win_info(self.handle, n_windows, _ptr(a)), "cw_debug_win_info")
        return a


def _operator(piles, merSize, commonKMers, minAnchors, solidThresh, maxMSA, device):
    eng = Engine(Params(merSize, solidThresh, commonKMers, minAnchors, maxMSA), device)
    try:
        res = eng.run(pack_piles(piles), want_solid=True)
    finally:
        eng.close()
    out = []
    for w in range(len(piles)):
        if res.status[w] == WIN_OVERFLOW:
            raise EngineError(f"window {w}: capacity overflow")
        out.append((res.consensus(w), res.solid_kmers(w).copy()))
    return out


def compute_consensus_read_correction(readId, piles, pilesPos, minSupport, merSize, commonKMers, minAnchors, solidThresh, windowSize, maxMSA, path="", device=0):
    """Batched mirror of computeConsensusReadCorrection (reference src/correctionMSA.cpp:29-49).

    ``piles`` is a list of window piles (template first).  As in the reference body, ``readId``, ``pilesPos``,
    ``minSupport``, ``windowSize`` and ``path`` do not influence the result.  Returns, per window,
    ``(consensus, solid_kmers)``: the case-annotated consensus and the ascending k-mers whose pile-wide count
    is >= solidThresh (what callers read from the reference's merCounts map).
    """
    return _operator(piles, merSize, commonKMers, minAnchors, solidThresh, maxMSA, device)


def compute_consensus_assembly_polishing(id, readId, piles, pilesPos, minSupport, merSize, commonKMers, minAnchors, solidThresh, windowSize, maxMSA, path="", nbThreads=1, device=0):
    """Batched mirror of computeConsensusAssemblyPolishing (reference src/correctionMSA.cpp:51-71); same body."""
    return _operator(piles, merSize, commonKMers, minAnchors, solidThresh, maxMSA, device)
