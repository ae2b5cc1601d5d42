/*
 * cw_extract.h -- device-side pile extraction: getAlignmentWindowsSequences
 * (alignmentWindows.cpp:87-149) evaluated on the GPU from the 2-bit read set and the overlap tuples, so that neither the
 * decoded reads (alignmentPiles.cpp:5-20) nor the piles ever cross PCIe.
 *
 * Pass 1 (cw_extract_count_kernel): one wave per window, one lane per overlap -> piece descriptor (source read, first
 *        source base, length, strand) by the reference's coordinate-shift rules (three boundary cases :119-131, keep
 *        iff length >= merSize :141); ordered member index by ballot prefix; per-window sequence and word counts.
 * Scan  (cw_extract_scan_kernel): exclusive prefix sums over windows.
 * Pass 2 (cw_extract_fill_kernel): one wave per window; each lane assembles one 16-base output word per step
 *        (reverse-complement by reading the source backwards with 3-code).
 */
#ifndef CW_EXTRACT_H
#define CW_EXTRACT_H

#include "cw_device.h"

struct ExtractDesc { /* one per (window, overlap) */
    uint32_t src_read;
    uint32_t src_pos;  /* '+': first source base; '-': source base of the piece's FIRST output base (walks downwards) */
    uint32_t len;      /* 0 = not a member */
    uint32_t member;   /* index inside the window's pile (template = 0) */
};

struct ExtractArgs {
    cw_read_set reads;
    const cw_overlap* ovl;
    const cw_window_job* jobs;
    uint32_t n_jobs;
    uint32_t k;
    ExtractDesc* desc;        /* [sum of ovl_count]; window w's block starts at desc_off[w] */
    const uint64_t* desc_off; /* [n_jobs + 1] */
    uint32_t* win_seqs;       /* [n_jobs + 1] counts, then exclusive offsets */
    uint64_t* win_words;      /* [n_jobs + 1] */
    /* outputs laid out as cw_batch */
    uint32_t* win_first_seq;
    uint32_t* seq_len;
    uint64_t* seq_word_off;
    uint32_t* bases;
    uint32_t seq_cap;
    uint64_t word_cap;
    uint32_t* status; /* [0] = 1 when an output capacity was exceeded; [1] = 1 when a piece is longer than the batch format's 65535
                         bases (reachable through -l: reported, never dropped); [2] = 1 when an overlap names a read outside the set */
};

/* piece of one overlap for window [q_beg, q_end] -- alignmentWindows.cpp:106-141 on plain integers */
__device__ __forceinline__ void cw_extract_piece(const cw_overlap& al, uint32_t t_len_u, uint32_t q_beg, uint32_t end, uint32_t k,
                                                 uint32_t* src_pos, uint32_t* out_len, bool* too_long) {
    *out_len = 0;
    uint32_t t_beg = al.t_start, t_end = al.t_end;
    uint32_t length = end - q_beg + 1;
    uint32_t shift = q_beg > al.q_start ? q_beg - al.q_start : 0;                       /* :110-114 */
    const bool spans = (al.q_start <= q_beg && al.q_end > q_beg) || (end <= al.q_end && al.q_start < end);
    if (!(spans && al.t_start + shift <= al.t_end)) return;                             /* :117 */
    const int t_len = (int)t_len_u;
    if (q_beg < al.q_start && al.q_end < end) {                                         /* :119-123 */
        shift = 0;
        t_beg = (uint32_t)max(0, (int)al.t_start - ((int)al.q_start - (int)q_beg));
        t_end = (uint32_t)min(t_len - 1, (int)al.t_end + ((int)end - (int)al.q_end));
        length = t_end - t_beg + 1;
    } else if (q_beg < al.q_start) {                                                    /* :124-127 */
        shift = 0;
        t_beg = (uint32_t)max(0, (int)al.t_start - ((int)al.q_start - (int)q_beg));
        length = (uint32_t)min((int)length, min(t_len - 1, (int)t_beg + (int)length - 1) - (int)t_beg + 1);
    } else if (al.q_end < end) {                                                        /* :128-130 */
        t_end = (uint32_t)min(t_len - 1, (int)al.t_end + ((int)end - (int)al.q_end));
        length = (uint32_t)min((int)length, (int)t_end - max(0, (int)t_end - (int)length + 1) + 1);
    }
    /* :133 substr(tBeg, tEnd-tBeg+1) clamps at the end of the read; counts are size_t conversions of unsigned ints */
    if (t_beg > t_len_u) return;
    const uint64_t want = (uint64_t)(uint32_t)(t_end - t_beg + 1);
    const uint64_t s_len = min(want, (uint64_t)t_len_u - t_beg);
    if ((uint64_t)shift > s_len) return;                                                /* :138 would throw */
    const uint64_t p_len = min((uint64_t)length, s_len - shift);
    if (p_len < k) return;                                                              /* :141 */
    if (p_len > 65535u) { *too_long = true; return; }                                   /* the batch format's length limit: reported by the caller */
    *out_len = (uint32_t)p_len;
    /* '+': piece[y] = T[t_beg + shift + y];  '-': S = revcomp(T[t_beg, t_beg+s_len)), piece[y] = comp(T[t_beg + s_len - 1 - shift - y]) */
    *src_pos = al.strand ? (uint32_t)(t_beg + s_len - 1 - shift) : t_beg + shift;
}

__global__ void __launch_bounds__(256) cw_extract_count_kernel(ExtractArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.n_jobs) return;
    const cw_window_job jb = a.jobs[w];
    const uint32_t tpl_len = a.reads.read_len[jb.tpl_read];
    const uint32_t length = jb.q_end - jb.q_beg + 1;
    const bool has_tpl = (uint64_t)jb.q_beg + length - 1 < tpl_len;                     /* :95-97: else an empty pile */
    if (has_tpl && length > 65535u) { if (lane == 0) a.status[1] = 1; }
    uint32_t n_mem = has_tpl ? 1u : 0u;
    uint64_t words = has_tpl ? (length + 15) / 16 : 0;
    ExtractDesc* dd = a.desc + a.desc_off[w];
    for (uint32_t o0 = 0; o0 < jb.ovl_count; o0 += 64) {
        const uint32_t o = o0 + lane;
        uint32_t len = 0, pos = 0, src = 0;
        if (o < jb.ovl_count && has_tpl) {
            const cw_overlap al = a.ovl[jb.ovl_first + o];
            src = al.t_read;
            if (src >= a.reads.n_reads) { a.status[2] = 1; src = 0; }
            else {
                bool too_long = false;
                cw_extract_piece(al, a.reads.read_len[src], jb.q_beg, jb.q_end, a.k, &pos, &len, &too_long);
                if (too_long) a.status[1] = 1;
            }
        }
        const unsigned long long bal = __ballot(len > 0);
        const uint32_t idx = n_mem + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (o < jb.ovl_count) { ExtractDesc d; d.src_read = src; d.src_pos = pos; d.len = len; d.member = idx; dd[o] = d; }
        uint32_t wsum = len ? (len + 15) / 16 : 0;
        for (int off = 32; off > 0; off >>= 1) wsum += __shfl_xor((int)wsum, off);
        words += wsum;
        n_mem += (uint32_t)__popcll(bal);
    }
    if (lane == 0) { a.win_seqs[w] = n_mem; a.win_words[w] = words; }
}

__global__ void __launch_bounds__(1024) cw_extract_scan_kernel(ExtractArgs a) {
    __shared__ uint32_t ps[1024];
    __shared__ unsigned long long pw[1024];
    __shared__ unsigned long long run_w;
    __shared__ uint32_t run_s;
    const int tid = threadIdx.x;
    if (tid == 0) { run_s = 0; run_w = 0; }
    __syncthreads();
    for (uint32_t w0 = 0; w0 < a.n_jobs; w0 += 1024) {
        const uint32_t w = w0 + tid;
        const uint32_t s = w < a.n_jobs ? a.win_seqs[w] : 0;
        const unsigned long long x = w < a.n_jobs ? a.win_words[w] : 0;
        ps[tid] = s; pw[tid] = x;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t vs = 0; unsigned long long vw = 0;
            if (tid >= o) { vs = ps[tid - o]; vw = pw[tid - o]; }
            __syncthreads();
            ps[tid] += vs; pw[tid] += vw;
            __syncthreads();
        }
        if (w < a.n_jobs) { a.win_seqs[w] = run_s + ps[tid] - s; a.win_words[w] = run_w + pw[tid] - x; }
        __syncthreads();
        if (tid == 0) { run_s += ps[1023]; run_w += pw[1023]; }
        __syncthreads();
    }
    if (tid == 0) {
        a.win_seqs[a.n_jobs] = run_s; a.win_words[a.n_jobs] = run_w;
        if (run_s > a.seq_cap || run_w > a.word_cap) a.status[0] = 1;
    }
}

__global__ void __launch_bounds__(256) cw_extract_fill_kernel(ExtractArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w > a.n_jobs || a.status[0] || a.status[1] || a.status[2]) return;
    if (w == a.n_jobs) { if (lane == 0) a.win_first_seq[w] = a.win_seqs[w]; return; }
    const cw_window_job jb = a.jobs[w];
    const uint32_t s_base = a.win_seqs[w];
    const uint32_t n_mem = a.win_seqs[w + 1] - s_base;
    if (lane == 0) a.win_first_seq[w] = s_base;
    if (n_mem == 0) return;
    uint64_t wo = a.win_words[w];
    const ExtractDesc* dd = a.desc + a.desc_off[w];
    /* member m: template for m == 0, else the m-th kept overlap; walk the overlaps once, carrying the word offset */
    for (int32_t o = -1; o < (int32_t)jb.ovl_count; ++o) {
        uint32_t src, pos, len, strand = 0, member;
        if (o < 0) { src = jb.tpl_read; pos = jb.q_beg; len = jb.q_end - jb.q_beg + 1; member = 0; }
        else {
            const ExtractDesc d = dd[o];
            if (d.len == 0) continue;
            src = d.src_read; pos = d.src_pos; len = d.len; member = d.member;
            strand = a.ovl[jb.ovl_first + o].strand;
        }
        const uint32_t* words = a.reads.bases + a.reads.read_word_off[src];
        const uint32_t nw = (len + 15) / 16;
        if (lane == 0) { a.seq_len[s_base + member] = len; a.seq_word_off[s_base + member] = wo; }
        for (uint32_t x = lane; x < nw; x += 64) {
            uint32_t out = 0;
            const uint32_t nb = min(16u, len - x * 16);
            for (uint32_t y = 0; y < nb; ++y) {
                const uint32_t q = x * 16 + y;
                const uint32_t code = strand ? 3u - cw_base_at(words, pos - q) : cw_base_at(words, pos + q);
                out |= code << (30 - 2 * y);
            }
            a.bases[wo + x] = out;
        }
        wo += nw;
    }
}

#endif
