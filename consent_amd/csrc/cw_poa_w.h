/*
 * cw_poa_w.h -- round 5: ONE tier-L task on the waves of a work-group (A4d; VERDICT r02-r04 "the chunks of a wide row on several waves").
 *
 * A tier-L task is ~50 members of 300-500 bases against a graph of ~1000 nodes: on one wave its fill walks four to eight 128-column
 * chunks per DP row, one after the other, and the task runs for 17-25 ms whatever else is resident -- the floor of every small engine
 * run and of every driver job.  Here the chunks of a row belong to the CW_POAL_MW waves of the work-group, pipelined by rows:
 *
 *   wave w fills columns [128 K w, 128 K (w + 1)) of every row (K = 1 or 2 chunks of 128 packed columns), one row behind wave w - 1;
 *   what crosses a chunk boundary is one word per row in LDS: the last column of the row (the diagonal of the neighbour's first
 *   column in a later row) and the running maximum of the horizontal recurrence (the neighbour's scan carry), published by the
 *   producer together with a row counter the consumer polls -- no barrier in the row loop;
 *   every wave stores and reads back its own columns of the matrix in the slab (a lane reads what the same lane wrote), and writes
 *   the direction words of its own chunks.
 *
 * Wave 0 runs the task exactly as before (poa_run: metadata, end cell, traceback, merge) and posts a FILL command in LDS when a
 * member is wider than one chunk; waves 1.. sit in a service loop (poa_mw_serve), take the command, fill their chunks, sign off.
 * The arithmetic is poa_fill_pk's: the same cells, the same direction words, the same results (forced-tier and fuzz tests).
 */
#ifndef CW_POA_W_H
#define CW_POA_W_H

/* CW_POAL_MW (cw_poa.h): waves of such a work-group, 4.  Round 5 ran EVERY tier-L task this way (one four-wave work-group per CU in place of two one-wave
   ones) and measured it bit-identical and not faster: tier L's kernel 45 -> 72 ms, its fill's wave-cycles unchanged, because most of tier L's rows are
   NARROW -- a -DCW_DIAG batch counts 25.3 M rows of members of at most 63 bases against 0.5 M rows of wider ones: the ragged first and last segments
   of a window are one long member that makes the graph, then dozens of short pieces aligned against all of it.  Round 6 (tier "LW"): only the tasks
   whose MEMBERS are wide on average come here (cw_chain.h routes them to their own list, a second instance of the tier-L kernel with four waves
   takes it on its own stream): they are few -- and they are the stragglers: one wave issues a vector instruction every four cycles at best, a
   25-member task of 400-base members against 1000 nodes ran for 26 ms on it while the rest of tier L was done in 15
   (tools/task_trace.py), and in the native driver, whose piles have more of them, tier L's kernel was the longest of a job (27.6 ms). */
#define CW_MW_CMD_FILL 1u
#define CW_MW_CMD_EXIT 2u

struct PoaComm {
    uint32_t seq;            /* command number: wave 0 increments it after the parameters are written */
    uint32_t cmd;
    int n, cols, hs, nw, k;  /* rows, columns, row stride, waves that take part, chunks per wave */
    uint32_t use_dirs;
    uint32_t slab;           /* the slab wave 0 claimed for the work-group */
    uint32_t done[CW_POAL_MW];  /* command number each helper has finished */
    uint32_t ready[CW_POAL_MW]; /* rows wave w has published, + 1 (0 = not even the virtual start row) */
    uint32_t edge[CW_POAL_MW - 1 > 0 ? CW_POAL_MW - 1 : 1][CW_POAL_NC + 2]; /* per boundary and DP row: last column of the producer's chunk << 16 | its scan carry (biased) */
};
#define CW_POA_COMM_BYTES ((sizeof(PoaComm) + 15) / 16 * 16)

typedef __attribute__((address_space(3))) uint32_t* cww_l32;
__device__ __forceinline__ uint32_t cww_load(const uint32_t* p) { return __hip_atomic_load((cww_l32)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void cww_store(uint32_t* p, uint32_t v) { __hip_atomic_store((cww_l32)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
/* wait until *p >= need (rows are published in order); `have` caches what was last seen */
__device__ __forceinline__ void cww_wait_rows(const uint32_t* p, uint32_t need, uint32_t& have) {
    while (have < need) {
        have = (uint32_t)__builtin_amdgcn_readfirstlane((int)cww_load(p));
        if (have < need) __builtin_amdgcn_s_sleep(1);
    }
    /* no fence: one wave's LDS operations are carried out in order, so the producer's boundary word (written before its row counter) is in LDS when the
       counter is seen, and this wave's read of the word follows its read of the counter.  A workgroup-scope fence would also wait for this wave's own
       matrix stores of the row before (vmcnt) -- in every row: measured, the four-wave fill was no faster than one wave */
    asm volatile("" ::: "memory");
}

/* Wave w's share of the packed fill (cf. poa_fill_pk): K chunks of 128 columns starting at chunk w * K.  nw waves take part. */
template <int K, bool DIRS>
__device__ __forceinline__ void poa_fill_pk_w(const PoaMem<int16_t>& M, PoaComm* C, const int n, const int cols, const int hs, const int lane, const bool use_dirs_,
                                              const int w, const int nw) {
    const bool use_dirs = DIRS && use_dirs_;
    const int G = CW_POA_GAP;
    const int GPK = pk_make(G, G);
    const int nch = (cols + 127) >> 7;
    const int c0 = w * K; /* first global chunk of this wave */
    constexpr int RC = K <= 2 ? 3 : 2;
    int rc_[RC][K], jg[K], qpk[K];
    int* Hw = (int*)M.H;
    const uint32_t* e_in = w > 0 ? C->edge[w - 1] : nullptr;
    uint32_t* e_out = w + 1 < nw ? C->edge[w] : nullptr;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const int j0 = (c0 + c) * 128 + 2 * lane, j1 = j0 + 1;
        jg[c] = pk_make(j0 * G, j1 * G);
#pragma unroll
        for (int k = 0; k < RC; ++k) rc_[k][c] = jg[c]; /* row 0 */
        const int q0 = (j0 >= 1 && j0 < cols) ? (int)M.sq[j0 - 1] : -1, q1 = (j1 < cols) ? (int)M.sq[j1 - 1] : -1;
        qpk[c] = (q0 >= 0 ? 1 << q0 : 0) | (q1 >= 0 ? 1 << (16 + q1) : 0);
        if (j0 < cols) Hw[j0 >> 1] = jg[c]; /* this wave's part of row 0: read back by the same lanes when a source node is far up */
    }
    if (e_out) { /* row 0 of the boundary: its last column; the carry of a row 0 is never asked for */
        if (lane == 0) { cww_store(&e_out[0], (uint32_t)(((c0 + K) * 128 - 1) * G) << 16); asm volatile("" ::: "memory"); cww_store(&C->ready[w], 1u); }
    }
    uint32_t have = 0;
    uint32_t meta_n = M.rmeta[0];
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta_n);
        if (r + 1 < n) meta_n = M.rmeta[r + 1];
        const int base = (int)(meta & 3u), np = CW_RM_NP(meta), off = CW_RM_X(meta), pr0 = off;
        if (e_in) cww_wait_rows(&C->ready[w - 1], (uint32_t)i, have); /* every row above this one is published by the neighbour */
        int v[K], dgv[K], upv[K], srow[K];
#pragma unroll
        for (int c = 0; c < K; ++c) { v[c] = CW_NEGPK; dgv[c] = CW_NEGPK; upv[c] = CW_NEGPK; srow[c] = pk_score(qpk[c], base); }
        for (int q = 0; q < np; ++q) {
            const int prow = (np == 1) ? pr0 : __builtin_amdgcn_readfirstlane((int)M.plist[off + q]);
            int up[K];
            const int dist = i - prow;
            /* the neighbour's last column of that row: the pair whose high half is the cell left of this wave's first column */
            int carry_in = e_in ? (int)(cww_load(&e_in[prow]) & 0xFFFF0000u) : CW_NEGPK;
            if (dist <= RC) {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    up[c] = rc_[0][c];
#pragma unroll
                    for (int k = 1; k < RC; ++k) up[c] = (dist == k + 1) ? rc_[k][c] : up[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const int j0 = (c0 + c) * 128 + 2 * lane;
                    up[c] = (j0 < cols) ? Hw[(prow * hs + j0) >> 1] : CW_NEGPK;
                }
#pragma unroll
                for (int c = 0; c < K; ++c) asm volatile("" : "+v"(up[c])); /* see poa_fill_pk: keeps the wait for this load inside the branch */
            }
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const int sh = CW_DPP(carry_in, up[c], 0x138, 0xF);
                carry_in = cw_lane_value(up[c], 63);
                const int dg = __builtin_amdgcn_alignbit(up[c], sh, 16);
                dgv[c] = pk_add(dg, srow[c]); upv[c] = pk_add(up[c], GPK);
                v[c] = pk_max(v[c], pk_max(dgv[c], upv[c]));
            }
        }
        if (CW_POA_OV && c0 == 0) v[0] = lane == 0 ? (int)((unsigned)v[0] & 0xFFFF0000u) : v[0];
        unsigned carry = 0u;
        if (e_in) { /* the neighbour's running maximum of this row */
            cww_wait_rows(&C->ready[w - 1], (uint32_t)i + 1u, have);
            carry = cww_load(&e_in[i]) & 0xFFFFu;
        }
        int nv_last = 0;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            int wv = pk_sub(v[c], jg[c]);
            wv = pk_max(wv, (wv << 16) | 0x8AD0);
            const unsigned inc = cw_wave_scan_max_u32(((unsigned)wv >> 16) ^ 0x8000u);
            unsigned ex = (unsigned)CW_DPP(0, (int)inc, 0x138, 0xF);
            if (c0 + c > 0) ex = max(ex, carry);
            wv = pk_max(wv, pk_splat_lo((int)(ex ^ 0x8000u)));
            carry = max(carry, (unsigned)cw_lane_value((int)inc, 63));
            const int nv = pk_add(wv, jg[c]);
#pragma unroll
            for (int k = RC - 1; k > 0; --k) rc_[k][c] = rc_[k - 1][c];
            rc_[0][c] = nv;
            nv_last = nv;
            const int j0 = (c0 + c) * 128 + 2 * lane;
            if (j0 < cols) Hw[(i * hs + j0) >> 1] = nv;
            if (use_dirs && c0 + c < nch) {
                unsigned long long e0, e1, o0, o1;
                if (np == 1) {
                    const bool de = j0 > 0 && (short)nv == (short)dgv[c], ue = (short)nv == (short)upv[c];
                    const bool dd = (short)((unsigned)nv >> 16) == (short)((unsigned)dgv[c] >> 16), uo = (short)((unsigned)nv >> 16) == (short)((unsigned)upv[c] >> 16);
                    const unsigned long long bde = __ballot(de), bue = __ballot(ue), bdo = __ballot(dd), buo = __ballot(uo);
                    e0 = ~bde & bue; e1 = ~bde & ~bue; o0 = ~bdo & buo; o1 = ~bdo & ~buo;
                } else {
                    e0 = e1 = o0 = o1 = ~0ull;
                }
                if (lane == 0) {
                    unsigned long long* d = M.dirs + (size_t)(r * nch + c0 + c) * 4;
                    d[0] = e0; d[1] = e1; d[2] = o0; d[3] = o1;
                }
            }
        }
        if (e_out) { /* publish: the last pair's high half (column 128 (c0 + K) - 1) and the running maximum, then the row count */
            const uint32_t word = ((uint32_t)cw_lane_value(nv_last, 63) & 0xFFFF0000u) | (carry & 0xFFFFu);
            if (lane == 0) { cww_store(&e_out[i], word); asm volatile("" ::: "memory"); cww_store(&C->ready[w], (uint32_t)i + 1u); } /* (in order: see cww_wait_rows) */
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* matrix rows and direction words are read by wave 0 next */
    cw_wave_sync();
}

/* wave 0: run the fill of a wide packed member on all waves of the work-group; returns when every wave has finished */
template <bool DIRS>
__device__ __forceinline__ void poa_fill_mw(const PoaMem<int16_t>& M, const int n, const int cols, const int hs, const int lane, const bool use_dirs) {
    PoaComm* C = M.comm;
    const int nch = (cols + 127) >> 7;
    const int k = nch > CW_POAL_MW ? 2 : 1;
    const int nw = (nch + k - 1) / k;
    if (lane == 0) {
        C->cmd = CW_MW_CMD_FILL; C->n = n; C->cols = cols; C->hs = hs; C->nw = nw; C->k = k; C->use_dirs = use_dirs ? 1u : 0u;
        for (int x = 0; x < CW_POAL_MW; ++x) C->ready[x] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const uint32_t seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)cww_load(&C->seq)) + 1u;
    if (lane == 0) cww_store(&C->seq, seq);
    if (k == 1) poa_fill_pk_w<1, DIRS>(M, C, n, cols, hs, lane, use_dirs, 0, nw);
    else poa_fill_pk_w<2, DIRS>(M, C, n, cols, hs, lane, use_dirs, 0, nw);
    for (int x = 1; x < nw; ++x) {
        uint32_t have = 0;
        while (have != seq) { have = (uint32_t)__builtin_amdgcn_readfirstlane((int)cww_load(&C->done[x])); if (have != seq) __builtin_amdgcn_s_sleep(2); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* waves 1 ..: take FILL commands until wave 0 says EXIT */
template <bool DIRS>
__device__ __forceinline__ void poa_mw_serve(const PoaMem<int16_t>& M, const int lane, const int w) {
    PoaComm* C = M.comm;
    uint32_t last = 0;
    for (;;) {
        uint32_t seq = last;
        while (seq == last) { seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)cww_load(&C->seq)); if (seq == last) __builtin_amdgcn_s_sleep(8); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        last = seq;
        const uint32_t cmd = (uint32_t)__builtin_amdgcn_readfirstlane((int)C->cmd);
        if (cmd == CW_MW_CMD_EXIT) return;
        const int n = __builtin_amdgcn_readfirstlane(C->n), cols = __builtin_amdgcn_readfirstlane(C->cols), hs = __builtin_amdgcn_readfirstlane(C->hs),
                  nw = __builtin_amdgcn_readfirstlane(C->nw), k = __builtin_amdgcn_readfirstlane(C->k);
        const bool use_dirs = __builtin_amdgcn_readfirstlane((int)C->use_dirs) != 0;
        if (w < nw) {
            if (k == 1) poa_fill_pk_w<1, DIRS>(M, C, n, cols, hs, lane, use_dirs, w, nw);
            else poa_fill_pk_w<2, DIRS>(M, C, n, cols, hs, lane, use_dirs, w, nw);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) cww_store(&C->done[w], seq);
    }
}

#endif
