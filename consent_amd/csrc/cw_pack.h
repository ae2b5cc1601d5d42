/*
 * cw_pack.h -- result compaction for host batches (cw_submit / cw_wait).
 *
 * The caller of cw_run reserves a capacity per window (cw_result.cons_off / solid_off); the windows use a fraction of it
 * (a consensus is ~550 of 1756 reserved bytes, a solid set a few percent of its worst case).  Copying the capacities back
 * over PCIe made the copy as long as the kernels.  These two kernels pack what was produced front to back, so that the
 * D2H transfer is the used bytes only; the host scatters them to the caller's offsets.
 *   cw_pack_scan_kernel : exclusive prefix sums of cons_len / solid_len over the windows (one work-group)
 *   cw_pack_copy_kernel : one wave per window copies its consensus bytes and solid k-mers to the packed arrays
 * HBM-bound by construction (each produced byte read once, written once).
 */
#ifndef CW_PACK_H
#define CW_PACK_H

#include "cw_device.h"

struct PackArgs {
    uint32_t n_windows;
    const char* cons;
    const uint64_t* cons_off;
    const uint32_t* cons_len;
    const uint8_t* win_status;
    const uint32_t* solid; /* may be NULL */
    const uint64_t* solid_off;
    const uint32_t* solid_len;
    char* pc;          /* packed consensus bytes */
    uint32_t* ps;      /* packed solid k-mers    */
    uint64_t* pc_off;  /* [n_windows + 1]        */
    uint64_t* ps_off;  /* [n_windows + 1]        */
    uint64_t* totals;  /* [0] consensus bytes, [1] solid k-mers, [2] windows stopped on the batch's task / member capacities (CW_WHY_TASKS, low half) and on their arena slices (CW_WHY_ARENA, high half) */
    const WinInfo* win; /* the batch's per-window records in the engine's scratch (read for [2] only) */
    const uint32_t* used; /* the batch's counters n_tasks, n_members (BatchCounters): copied to totals[3], [4] -- what cw_run's scale decay goes by */
};

__global__ void __launch_bounds__(1024) cw_pack_scan_kernel(PackArgs a) {
    __shared__ unsigned long long pa[1024], pb[1024];
    __shared__ unsigned long long run[2];
    __shared__ unsigned int why_tasks, why_arena;
    const int tid = threadIdx.x;
    if (tid < 2) run[tid] = 0;
    if (tid == 0) { why_tasks = 0; why_arena = 0; }
    __syncthreads();
    for (uint32_t w0 = 0; w0 < a.n_windows; w0 += 1024) {
        const uint32_t w = w0 + tid;
        const bool live = w < a.n_windows && a.win_status[w] != CW_WIN_OVERFLOW;
        if (w < a.n_windows && !live && a.win[w].status == CW_WIN_OVERFLOW && a.win[w].pad_ == CW_WHY_TASKS) atomicAdd(&why_tasks, 1u);
        if (w < a.n_windows && !live && a.win[w].status == CW_WIN_OVERFLOW && a.win[w].pad_ == CW_WHY_ARENA) atomicAdd(&why_arena, 1u);
        const unsigned long long c = live ? a.cons_len[w] : 0, s = (live && a.solid) ? a.solid_len[w] : 0;
        pa[tid] = c; pb[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            unsigned long long va = 0, vb = 0;
            if (tid >= o) { va = pa[tid - o]; vb = pb[tid - o]; }
            __syncthreads();
            pa[tid] += va; pb[tid] += vb;
            __syncthreads();
        }
        if (w < a.n_windows) { a.pc_off[w] = run[0] + pa[tid] - c; a.ps_off[w] = run[1] + pb[tid] - s; }
        __syncthreads();
        if (tid == 0) { run[0] += pa[1023]; run[1] += pb[1023]; }
        __syncthreads();
    }
    if (tid == 0) { a.pc_off[a.n_windows] = run[0]; a.ps_off[a.n_windows] = run[1]; a.totals[0] = run[0]; a.totals[1] = run[1]; a.totals[2] = (uint64_t)why_tasks | ((uint64_t)why_arena << 32); a.totals[3] = a.used[0]; a.totals[4] = a.used[1]; }
}

__global__ void __launch_bounds__(256) cw_pack_copy_kernel(PackArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.n_windows) return;
    const uint64_t c0 = a.pc_off[w], c1 = a.pc_off[w + 1];
    const char* src = a.cons + a.cons_off[w];
    for (uint64_t i = lane; i < c1 - c0; i += 64) a.pc[c0 + i] = src[i];
    if (a.solid) {
        const uint64_t s0 = a.ps_off[w], s1 = a.ps_off[w + 1];
        const uint32_t* ss = a.solid + a.solid_off[w];
        for (uint64_t i = lane; i < s1 - s0; i += 64) a.ps[s0 + i] = ss[i];
    }
}

/* ---- result-slot planning for device-resident batches (cw_plan_results_device) ----------------------------------------------
 * cons slot = 3 x template + 256 bytes (the polish can lengthen a consensus; the engine reports an overflow beyond it);
 * solid slot = (k-mers in the pile) / solidThresh + 16 entries (a k-mer needs solidThresh occurrences to be solid). */
struct PlanArgs {
    uint32_t n_windows, k, solid;
    const uint32_t* win_first_seq;
    const uint32_t* seq_len;
    uint64_t* cons_off;  /* [n_windows + 1] */
    uint64_t* solid_off; /* [n_windows + 1] */
    uint64_t* totals;    /* [2] */
};

__global__ void __launch_bounds__(256) cw_plan_need_kernel(PlanArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.n_windows) return;
    const uint32_t s0 = a.win_first_seq[w], s1 = a.win_first_seq[w + 1];
    unsigned long long nk = 0;
    for (uint32_t s = s0 + lane; s < s1; s += 64) { const uint32_t l = a.seq_len[s]; nk += l >= a.k ? l - a.k + 1 : 0; }
    for (int o = 32; o > 0; o >>= 1) nk += __shfl_xor(nk, o);
    if (lane == 0) {
        /* the slot rule of include/consent_amd.h (CW_CONS_SLOT_BYTES): real device memory, so it follows k (round 5 gave every window the finish
           kernel's 32768 characters: 1 GiB per job of 32768 windows and worker, for a case the default k never met) */
        const uint32_t tpl = s1 > s0 ? a.seq_len[s0] : 0u;
        a.cons_off[w] = (unsigned long long)CW_CONS_SLOT_BYTES(a.k, tpl);
        a.solid_off[w] = nk / a.solid + 16ull;
    }
}

__global__ void __launch_bounds__(1024) cw_plan_scan_kernel(PlanArgs a) {
    __shared__ unsigned long long pa[1024], pb[1024];
    __shared__ unsigned long long run[2];
    const int tid = threadIdx.x;
    if (tid < 2) run[tid] = 0;
    __syncthreads();
    for (uint32_t w0 = 0; w0 < a.n_windows; w0 += 1024) {
        const uint32_t w = w0 + tid;
        const unsigned long long c = w < a.n_windows ? a.cons_off[w] : 0, s = w < a.n_windows ? a.solid_off[w] : 0;
        pa[tid] = c; pb[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            unsigned long long va = 0, vb = 0;
            if (tid >= o) { va = pa[tid - o]; vb = pb[tid - o]; }
            __syncthreads();
            pa[tid] += va; pb[tid] += vb;
            __syncthreads();
        }
        if (w < a.n_windows) { a.cons_off[w] = run[0] + pa[tid] - c; a.solid_off[w] = run[1] + pb[tid] - s; }
        __syncthreads();
        if (tid == 0) { run[0] += pa[1023]; run[1] += pb[1023]; }
        __syncthreads();
    }
    if (tid == 0) { a.cons_off[a.n_windows] = run[0]; a.solid_off[a.n_windows] = run[1]; a.totals[0] = run[0]; a.totals[1] = run[1]; }
}

#endif
