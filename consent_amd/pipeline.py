"""Read correction end to end on one GPU: the loop of CONSENT-correction.cpp (runCorrection :62-135, processRead :19-58) with every
stage behind the library's C ABI -- host feeders (cw_index_reads, cw_paf_next_pile, cw_window_positions), then device-resident
cw_extract_piles_device -> cw_run_device -> cw_stitch_device.  Piles are batched (one window per call cannot feed a GPU); the
output order is the PAF order, as the reference's futures ring keeps it (:96-133).

This module only marshals buffers; it computes nothing itself and needs the HIP library and a GPU.
"""
import ctypes as C
import os
import sys

import numpy as np

from .engine import Batch, Engine, EngineError, PafReader, Params, ReadIndex, ReadSet, Result, _check, window_positions


_DEBUG = os.environ.get("CW_PIPE_DEBUG", "")


class _DeviceReads:
    def __init__(self, index, dev):
        import torch

        self.len = torch.from_numpy(index.seq_len.view(np.int32)).to(dev)
        self.off = torch.from_numpy(index.seq_word_off.view(np.int64)).to(dev)
        self.bases = torch.from_numpy(np.concatenate([index.bases, np.zeros(1, np.uint32)]).view(np.int32)).to(dev)
        self.struct = ReadSet(len(index.seq_len), self.len.data_ptr(), self.off.data_ptr(), self.bases.data_ptr())


def correct_reads(reads_path, paf_path, out=None, *, min_support=3, max_support=1000, window_size=500, mer_size=9, common_kmers=8,
                  min_anchors=10, solid_thresh=4, window_overlap=50, max_msa=150, do_trim=True, proof_path=None, windows_per_batch=32768,
                  device=0, on_capacity="raise"):
    """Corrects every read that has a pile in `paf_path`; writes FASTA (">name\\nsequence\\n", upper case = corrected) to `out`
    (a text file object; None = collect) and returns the list of (name, sequence) in PAF order.  Reads whose corrected
    sequence is empty -- no window, or dropped by the 10 % rule -- are skipped, as CONSENT-correction.cpp:101-103 does.
    With `proof_path` (assembly polishing, or correction against proof reads) that file is indexed into the same read set and the
    result is neither trimmed nor dropped (CONSENT-correction.cpp:69-73, CONSENT-polishing.cpp:112-116).
    `windows_per_batch`: piles are collected until they hold this many windows, then extracted, corrected and re-assembled in one go;
    re-assembly runs one wave per read, so a batch should hold a few thousand reads (32768 windows of 500 bases at depth 30 are ≈ 0.3 GB).
    `on_capacity`: a read whose re-assembly (or one of whose windows) exceeded a documented capacity of the engine either stops the run
    ("raise", the default: nothing is silently different from the reference) or is left out of the output and reported on stderr ("skip")."""
    import torch

    dev = torch.device("cuda", device)
    index = ReadIndex(reads_path, *([proof_path] if proof_path else []))
    if proof_path:
        do_trim = False
    eng = Engine(Params(mer_size, solid_thresh, common_kmers, min_anchors, max_msa), device)
    lib = eng.lib
    reads_dev = _DeviceReads(index, dev)
    results = []

    def up(a, dt):
        a = np.ascontiguousarray(a)
        if a.size == 0:
            a = np.zeros(1, a.dtype)
        return torch.from_numpy(a.view(dt)).to(dev)

    def flush(piles):
        """piles: list of (tpl, overlaps (n,6), windows [(beg,end)...]) -> corrected strings, one per pile"""
        if not piles:
            return
        ov_rows, jobs, win_pos, stitch_jobs, win_len, win_depth = [], [], [], [], [], []
        ov_base = 0
        for tpl, ov, wins in piles:
            stitch_jobs.append((tpl, len(jobs), len(wins)))
            for (qb, qe) in wins:
                jobs.append((tpl, qb, qe, ov_base, len(ov)))
                win_pos.append((qb, qe))
                win_len.append(qe - qb + 1)
                win_depth.append(len(ov) + 1)
            ov_rows.append(ov)
            ov_base += len(ov)
        n_win = len(jobs)
        out_strings = [""] * len(piles)
        if n_win:
            ov = np.concatenate(ov_rows).astype(np.uint32).reshape(-1, 6)
            jb = np.array(jobs, np.uint32).reshape(-1, 5)
            t_ov, t_jb = up(ov, np.int32), up(jb, np.int32)
            ns, nw = C.c_uint32(), C.c_uint64()
            rc = lib.cw_extract_piles_device(eng.handle, C.byref(reads_dev.struct), t_ov.data_ptr(), len(ov), t_jb.data_ptr(), n_win, mer_size, None, None, None, None,
                                             0, 0, C.byref(ns), C.byref(nw), None)
            if rc not in (0, -4):
                _check(lib, rc, "cw_extract_piles_device(size)")
            b_wfs = torch.zeros(n_win + 1, dtype=torch.int32, device=dev)
            b_len = torch.zeros(max(ns.value, 1), dtype=torch.int32, device=dev)
            b_off = torch.zeros(max(ns.value, 1), dtype=torch.int64, device=dev)
            b_bases = torch.zeros(max(nw.value, 1) + 1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            _check(lib, lib.cw_extract_piles_device(eng.handle, C.byref(reads_dev.struct), t_ov.data_ptr(), len(ov), t_jb.data_ptr(), n_win, mer_size, b_wfs.data_ptr(),
                                                     b_len.data_ptr(), b_off.data_ptr(), b_bases.data_ptr(), ns.value, nw.value, C.byref(ns), C.byref(nw), None),
                   "cw_extract_piles_device")
            if "e" in _DEBUG:
                torch.cuda.synchronize()
                print(f"[pipeline] extraction done: {n_win} windows, jobs {jb.tolist()[:4]}", file=sys.stderr, flush=True)
            batch = Batch(n_win, ns.value, nw.value, b_wfs.data_ptr(), b_len.data_ptr(), b_off.data_ptr(), b_bases.data_ptr())
            wl, wd = np.array(win_len, np.int64), np.array(win_depth, np.int64)
            cons_off = np.zeros(n_win + 1, np.uint64)
            cons_off[1:] = np.cumsum(3 * wl + 256)
            solid_off = np.zeros(n_win + 1, np.uint64)
            solid_off[1:] = np.cumsum(wl * wd // max(1, solid_thresh) + 16)
            r_cons = torch.zeros(int(cons_off[-1]) + 1, dtype=torch.uint8, device=dev)
            r_clen = torch.zeros(n_win, dtype=torch.int32, device=dev)
            r_st = torch.full((n_win,), 255, dtype=torch.uint8, device=dev)
            r_sol = torch.zeros(int(solid_off[-1]) + 1, dtype=torch.int32, device=dev)
            r_slen = torch.zeros(n_win, dtype=torch.int32, device=dev)
            t_coff, t_soff = up(cons_off, np.int64), up(solid_off, np.int64)
            res = Result(r_cons.data_ptr(), t_coff.data_ptr(), r_clen.data_ptr(), r_st.data_ptr(), r_sol.data_ptr(), t_soff.data_ptr(), r_slen.data_ptr())
            torch.cuda.synchronize()
            eng.run_device(batch, res)
            if "c" in _DEBUG:
                torch.cuda.synchronize()
                print(f"[pipeline] consensus done: {n_win} windows, {ns.value} sequences", file=sys.stderr, flush=True)
            sj = np.array(stitch_jobs, np.uint32).reshape(-1, 3)
            cap = 2 * index.seq_len[sj[:, 0]].astype(np.int64) + 1024
            out_off = np.zeros(len(sj) + 1, np.uint64)
            out_off[1:] = np.cumsum(cap)
            t_sj, t_pos, t_ooff = up(sj, np.int32), up(np.array(win_pos, np.uint32).reshape(-1), np.int32), up(out_off, np.int64)
            t_out = torch.zeros(int(out_off[-1]) + 1, dtype=torch.uint8, device=dev)
            t_olen = torch.zeros(len(sj), dtype=torch.int32, device=dev)
            t_ost = torch.full((len(sj),), 255, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            _check(lib, lib.cw_stitch_device(eng.handle, C.byref(reads_dev.struct), t_sj.data_ptr(), len(sj), t_pos.data_ptr(), C.byref(batch), C.byref(res), window_size,
                                             window_overlap, int(bool(do_trim)), t_out.data_ptr(), t_ooff.data_ptr(), t_olen.data_ptr(), t_ost.data_ptr(), None),
                   "cw_stitch_device")
            torch.cuda.synchronize()
            h_out, h_len, h_st, h_wst = t_out.cpu().numpy(), t_olen.cpu().numpy(), t_ost.cpu().numpy(), r_st.cpu().numpy()
            over_reads = set(int(i) for i in np.nonzero(h_st == 2)[0])
            for i in range(len(sj)):  # a window without a consensus taints its read
                if (h_wst[int(sj[i, 1]) : int(sj[i, 1]) + int(sj[i, 2])] == 2).any():
                    over_reads.add(i)
            if over_reads:
                names = ", ".join(index.names[int(sj[i, 0])] for i in sorted(over_reads))
                if on_capacity != "skip":
                    raise EngineError(f"capacity exceeded: {int((h_wst == 2).sum())} windows, {int((h_st == 2).sum())} reads ({names})")
                print(f"[consent_amd] left out (engine capacity): {names}", file=sys.stderr)
            for i in range(len(sj)):
                if i not in over_reads:
                    out_strings[i] = h_out[int(out_off[i]) : int(out_off[i]) + int(h_len[i])].tobytes().decode()
        for (tpl, _, _), s in zip(piles, out_strings):
            if s:
                results.append((index.names[tpl], s))
                if out is not None:
                    out.write(f">{index.names[tpl]}\n{s}\n")

    try:
        pending, n_pending_windows = [], 0
        reader = PafReader(paf_path, index, max_support)
        for tpl, tpl_len, ov, _rm in reader:
            if tpl_len != int(index.seq_len[tpl]):
                raise EngineError(f"PAF states length {tpl_len} for {index.names[tpl]}, the read file has {int(index.seq_len[tpl])}")
            wins = window_positions(tpl_len, ov, min_support, window_size, window_overlap)
            if not wins:
                continue  # processRead returns (readId, "") before anything else (CONSENT-correction.cpp:22-25): no output for this read
            pending.append((tpl, ov, wins))
            n_pending_windows += len(wins)
            if n_pending_windows >= windows_per_batch:
                flush(pending)
                pending, n_pending_windows = [], 0
        flush(pending)
        reader.close()
    finally:
        eng.close()
        index.close()
    return results


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="CONSENT read correction on one MI355X (flags as in CONSENT's main.cpp)")
    ap.add_argument("-a", dest="paf", required=True)
    ap.add_argument("-r", dest="reads", required=True)
    ap.add_argument("-s", dest="min_support", type=int, default=3)
    ap.add_argument("-S", dest="max_support", type=int, default=1000)
    ap.add_argument("-l", dest="window_size", type=int, default=500)
    ap.add_argument("-k", dest="mer_size", type=int, default=9)
    ap.add_argument("-c", dest="common_kmers", type=int, default=8)
    ap.add_argument("-A", dest="min_anchors", type=int, default=10)
    ap.add_argument("-f", dest="solid_thresh", type=int, default=4)
    ap.add_argument("-m", dest="window_overlap", type=int, default=50)
    ap.add_argument("-M", dest="max_msa", type=int, default=150)
    ap.add_argument("-R", dest="proof", default=None, help="proof reads / contigs: indexed too, no trimming")
    a = ap.parse_args(argv)
    correct_reads(a.reads, a.paf, sys.stdout, min_support=a.min_support, max_support=a.max_support, window_size=a.window_size, mer_size=a.mer_size,
                  common_kmers=a.common_kmers, min_anchors=a.min_anchors, solid_thresh=a.solid_thresh, window_overlap=a.window_overlap, max_msa=a.max_msa,
                  proof_path=a.proof)


if __name__ == "__main__":
    main()
