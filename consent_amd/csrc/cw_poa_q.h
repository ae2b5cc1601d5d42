/*
 * cw_poa_q.h -- tier Q of the partial-order alignment (A4d): FOUR tasks per wavefront, one per 16-lane row.
 *
 * Most POA tasks are small: a deep pile is chained densely, so a segment between two anchors is 10-30 bases and its graph a few
 * dozen nodes.  One wave per task (tier S) leaves half to three quarters of the lanes of every DP row idle and spends the same
 * instructions on a 14-column row as on a 64-column one.  Here a task owns a 16-lane DPP row: two DP columns per lane in packed
 * int16 (members up to 31 bases), a four-step row_shr prefix max (the row_bcast steps of the 64-lane ladder fall away).  The four
 * tasks of a wave run the same instruction stream under their own predicates -- every "wave-uniform" quantity of cw_poa.h (graph
 * size, row, path position) is a per-lane value that is equal inside a row -- so one instruction advances four alignments.
 *
 * Round 5: the fill RECORDS THE TRACEBACK'S DECISIONS (what cw_poa_c.h does for the one-task-per-wave tiers, here on packed halves)
 * and the DP matrix is gone from LDS:
 *   values     rows are kept as W[i][j] = H[i][j] - j * gap, scaled by 4, the two low bits of every stored value set.  A horizontal
 *              move costs nothing in that form (the horizontal recurrence is a plain prefix max), and a candidate that came through
 *              in-edge q of a node carries 3 - min(q, 3) in its low bits: on equal values the earlier in-edge is the larger key.
 *   decision   a cell is at least every candidate, so a candidate explains it iff it is the largest of its kind and equal to it;
 *              four bits per cell (in-edge | move << 2; 8 = horizontal), derived with packed 16-bit arithmetic for both columns
 *              of a lane at once; four rows of a lane's two columns make one 32-bit word in LDS (16 bytes per row and task).
 *   row store  the last three rows in registers, the last CW_POAQ_RING rows in an LDS ring; the rare row a later row needs from
 *              further back, and the rows round a node with more than three in-edges, also go to the task's slab in global memory
 *              (a bit in the row word, set by the rank metadata pass).
 *   traceback  a walk over code words, no value is looked at: the sixteen lanes look down the diagonal (cell (i - t, j - t) in lane
 *              t) and a whole run of diagonal moves along a chain of the graph is one step; any other move is one word and one row
 *              word away.  In-edge 3 means "fourth or later": decided from the kept values, among the in-edges 3.. only.
 *   bytes      64 nodes, 184 edges, 32 columns: every node id, rank, edge id and DP row of this tier fits a byte, and the graph arrays are
 *              byte arrays (PoaQ; the one-task-per-wave tiers keep 16-bit ids).  A task is 3.3 KB of LDS where the matrix version took 4.7 KB
 *              for 40 nodes: twelve waves per CU instead of eight -- and the tier is bound by the latency its resident waves can hide.
 * What that buys: 64 nodes / 184 edges (tasks of tier S with members of up to 31 bases come here: 43 % of tier S's DP work at depth
 * 150), half again as many tasks resident, and a traceback step of one or two LDS round trips instead of seven.
 *
 * Same policy, same arithmetic, same results as poa_run (cw_poa.h; include/cw_policy.h): the tests compare every tier with the
 * oracle.  A task that outgrows a capacity is handed to tier S (which runs after this kernel on the same stream).
 * -DCW_Q_CODES=0 builds the matrix version of rounds 3-4 (cw_poa_q0.h; tests/test_gpu_variants.py).
 */
#ifndef CW_POA_Q_H
#define CW_POA_Q_H

#include "cw_poa.h"

#ifndef CW_Q_CODES
#define CW_Q_CODES 1
#endif

/* ---- 16-lane row primitives ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ unsigned q_ballot(bool p) { return (unsigned)(__ballot(p) >> (threadIdx.x & 48u)) & 0xFFFFu; }
__device__ __forceinline__ int q_bcast(int v, int src) { return __shfl(v, (int)(threadIdx.x & 48u) + src); }
__device__ __forceinline__ int q_scan_add(int v) {
    v += CW_DPP(0, v, 0x111, 0xF);
    v += CW_DPP(0, v, 0x112, 0xF);
    v += CW_DPP(0, v, 0x114, 0xF);
    v += CW_DPP(0, v, 0x118, 0xF);
    return v;
}
__device__ __forceinline__ unsigned q_scan_max_u32(unsigned v) {
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x111, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x112, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x114, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x118, 0xF));
    return v;
}

#if !CW_Q_CODES
#include "cw_poa_q0.h"
#else

#define CW_NONE8 0xFFu
typedef unsigned short cw_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pku_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(cw_u2, a), __builtin_bit_cast(cw_u2, b))); }
__device__ __forceinline__ int pku_min(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(cw_u2, a), __builtin_bit_cast(cw_u2, b))); }
__device__ __forceinline__ int pku_shl1(int a) { return __builtin_bit_cast(int, (cw_u2)(__builtin_bit_cast(cw_u2, a) << (cw_u2)(unsigned short)1)); }

/* One shape of the several-tasks-per-wave tiers: a task owns GW lanes (16: tier Q, four tasks per wave; 32: tier H, two), two DP columns per lane.
   CG: the code words live in the task's global slab instead of LDS (tier H: 4 KB per task would halve the waves a CU holds). */
template <int GW_, int NC_, int EC_, int RING_, bool CG_>
struct PoaQT {
    static constexpr int GW = GW_, NC = NC_, EC = EC_, RING = RING_, LC = 2 * GW_ - 1;
    static constexpr bool CG = CG_;
    static constexpr int CODE_WORDS = (NC_ + 3) / 4 * GW_;
    static constexpr int GFLAG_WORDS = NC_ / 32 + 2;
    static constexpr int GRAPH_BYTES = (4 * NC_ + (3 + (CW_CONS_HEAVIEST_BUNDLE ? 1 : 0)) * EC_ + 14 * NC_ + 4 * (LC + 1) + 15) / 16 * 16; /* row words, then bytes (poaq_carve) */
    static constexpr int TASK_BYTES = (GRAPH_BYTES + RING_ * GW_ * 4 + (CG_ ? 0 : CODE_WORDS * 4) + GFLAG_WORDS * 4 + 15) / 16 * 16;
    static constexpr int KEEP_WORDS = (NC_ + 1) * GW_;                          /* kept rows, GW words each */
    static constexpr int SLAB_BYTES = (KEEP_WORDS + (CG_ ? CODE_WORDS : 0)) * 4; /* per task, global */
    static_assert(NC_ < 255 && EC_ < 255 && LC < 255, "these tiers keep node ids, DP rows, edge ids and sequence positions in bytes");
    static_assert((CG_ ? RING_ * GW_ : CODE_WORDS) >= NC_ + 1, "the merge's rank histogram borrows the code words (or, when they are global, the row ring)");
    static_assert((RING_ & (RING_ - 1)) == 0 && RING_ >= 4, "the ring is a power of two and reaches beyond the three rows kept in registers");
};

#ifndef CW_POAQ_NC
#define CW_POAQ_NC 64
#endif
#ifndef CW_POAQ_EC
#define CW_POAQ_EC 184 /* (with 64 nodes and the ring: 3392 bytes a task, 40 704 a three-wave work-group -- four of them are a CU's 160 KB to the byte) */
#endif
#ifndef CW_POAQ_RING
#define CW_POAQ_RING 8 /* rows of the LDS ring: a predecessor further back comes from the task's slab */
#endif
typedef PoaQT<16, CW_POAQ_NC, CW_POAQ_EC, CW_POAQ_RING, false> PoaQ16;
#define CW_POAQ_LC 31
#define CW_POAQ_TASK_BYTES (PoaQ16::TASK_BYTES)
#define CW_POAQ_SLAB_BYTES (PoaQ16::SLAB_BYTES)
#ifndef CW_POAQ_WAVES
#define CW_POAQ_WAVES (CW_CONS_HEAVIEST_BUNDLE ? 10 : 12) /* at most, per CU: 48 tasks (40 with the edge weights of the heaviest-bundle policy: 3576 bytes a task) */
#endif
static_assert(PoaQ16::TASK_BYTES * 4 * CW_POAQ_WAVES <= 163840, "tier Q: CW_POAQ_WAVES waves of four tasks fit a CU's LDS");
static_assert(4 * CW_POA_SMAX * (CW_POAQ_NC + 64) <= CW_POA_I16_BOUND, "include/cw_policy.h \"Bounds\": tier Q's recorded decisions (values x 4 in 16-bit halves)");
#ifndef CW_POAQ_REPLAY
#define CW_POAQ_REPLAY 1 /* 0: every member is aligned, as through round 5 (tests/test_gpu_variants.py) */
#endif
#ifndef CW_POAQ_ROUTE_NODES
#define CW_POAQ_ROUTE_NODES 60 /* tasks expected to stay below this many nodes come here (cw_chain.h; the estimate overshoots by ~10 %) */
#endif

/* tier H (round 5; replaces the opt-in kernel of round 3): two tasks per wave on 32-lane halves, members of up to 63 bases, 128 nodes */
#ifndef CW_POAH_NC
#define CW_POAH_NC 128
#endif
#ifndef CW_POAH_EC
#define CW_POAH_EC 250
#endif
typedef PoaQT<32, CW_POAH_NC, CW_POAH_EC, 8, true> PoaQ32;
#define CW_POAH_LC 63
#define CW_POAH_TASK_BYTES (PoaQ32::TASK_BYTES)
#define CW_POAH_SLAB_BYTES (PoaQ32::SLAB_BYTES)
#ifndef CW_POAH_WAVES
#define CW_POAH_WAVES 18 /* at most, per CU: 36 tasks */
#endif
#ifndef CW_POAH_ROUTE_NODES
#define CW_POAH_ROUTE_NODES 112 /* (the depth-aware estimate of cw_chain.h) */
#endif
#define CW_POAH_MIN_LEN 1

/* ---- GW-lane group primitives ----------------------------------------------------------------------------------------------- */
template <class T> __device__ __forceinline__ unsigned g_ballot(bool p) {
    if constexpr (T::GW == 16) return (unsigned)(__ballot(p) >> (threadIdx.x & 48u)) & 0xFFFFu;
    else return (unsigned)(__ballot(p) >> (threadIdx.x & 32u));
}
template <class T> __device__ __forceinline__ int g_bcast(int v, int src) { return __shfl(v, (int)(threadIdx.x & (unsigned)(64 - T::GW)) + src); }
template <class T> __device__ __forceinline__ int g_scan_add(int v) {
    v += CW_DPP(0, v, 0x111, 0xF); v += CW_DPP(0, v, 0x112, 0xF); v += CW_DPP(0, v, 0x114, 0xF); v += CW_DPP(0, v, 0x118, 0xF);
    if constexpr (T::GW == 32) v += CW_DPP(0, v, 0x142, 0xA); /* lane 15 of the group's first row into its second row */
    return v;
}
template <class T> __device__ __forceinline__ unsigned g_scan_max_u32(unsigned v) {
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x111, 0xF)); v = max(v, (unsigned)CW_DPP(0, (int)v, 0x112, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x114, 0xF)); v = max(v, (unsigned)CW_DPP(0, (int)v, 0x118, 0xF));
    if constexpr (T::GW == 32) v = max(v, (unsigned)CW_DPP(0, (int)v, 0x142, 0xA));
    return v;
}
/* lane gl receives v of lane gl - 1 of its group; lane 0 of the group receives 0 */
template <class T> __device__ __forceinline__ int g_shr1(int v, int gl) {
    if constexpr (T::GW == 16) return CW_DPP(0, v, 0x111, 0xF);
    else { const int s_ = CW_DPP(0, v, 0x138, 0xF); return gl == 0 ? 0 : s_; } /* (wave_shr:1 would hand lane 32 the other task's last lane) */
}

template <class T>
struct PoaQ { /* one task's arrays: LDS, except `keep` (and `codes` when T::CG) */
    uint32_t* rmeta;  /* rank -> the row word (CW_RM_WORD) */
    uint8_t* plist;   /* predecessor DP rows in in-edge order (CSR) */
    uint8_t* efrom;   /* edge -> source node */
    uint8_t* enext;   /* edge -> next in-edge of the same target, CW_NONE8 at the end */
    uint8_t* ew;      /* edge -> sequences whose path uses it (heaviest-bundle policy only, else NULL) */
    uint8_t* rpred0;  /* rank -> DP row of its first predecessor (0 = virtual start) */
    uint8_t* ncov;    /* node -> sequences through it (a task of more than 255 members goes to the next tier) */
    uint8_t* nal;     /* node -> 3 aligned node ids */
    uint8_t* in_head; /* node -> first in-edge, CW_NONE8 if none */
    uint8_t* in_tail;
    uint8_t* indeg;
    uint8_t* r2n;     /* rank -> node */
    uint8_t* n2r;
    uint8_t* rtmp;
    uint8_t* nbase;
    uint8_t* nalc;
    uint8_t* has_out;
    uint8_t* seqrank; /* sequence position -> rank of the node it is aligned to, CW_NONE8 for an insertion */
    uint8_t* pcur;
    uint8_t* pat;
    uint8_t* sq;      /* current member, base codes */
    uint32_t* ring;   /* T::RING rows x GW words */
    uint32_t* codes;  /* one word per four rows and lane (LDS, or global when T::CG) */
    uint32_t* hist;   /* the merge's rank histogram: borrows the code words, or the ring */
    uint32_t* gflag;  /* one bit per rank: the row is also kept in the slab */
    int* keep;        /* global: kept rows, GW words each, row i at keep + GW i */
};
template <class T>
__device__ __forceinline__ PoaQ<T> poaq_carve(uint8_t* p, uint8_t* slab) {
    PoaQ<T> M;
    const uint32_t nc = T::NC, ec = T::EC, lc1 = T::LC + 1;
    uint8_t* const base = p;
    M.rmeta = (uint32_t*)p; p += 4 * nc;
    M.plist = p; p += ec; M.efrom = p; p += ec; M.enext = p; p += ec;
    M.ew = nullptr;
    if (CW_CONS_HEAVIEST_BUNDLE) { M.ew = p; p += ec; }
    M.rpred0 = p; p += nc; M.ncov = p; p += nc; M.nal = p; p += 3 * nc; M.in_head = p; p += nc; M.in_tail = p; p += nc; M.indeg = p; p += nc;
    M.r2n = p; p += nc; M.n2r = p; p += nc; M.rtmp = p; p += nc; M.nbase = p; p += nc; M.nalc = p; p += nc; M.has_out = p; p += nc;
    M.seqrank = p; p += lc1; M.pcur = p; p += lc1; M.pat = p; p += lc1; M.sq = p; p += lc1;
    uint32_t* extra = (uint32_t*)(base + T::GRAPH_BYTES);
    M.ring = extra; extra += T::RING * T::GW;
    M.keep = (int*)slab;
    if constexpr (T::CG) { M.codes = (uint32_t*)slab + T::KEEP_WORDS; M.hist = M.ring; }
    else { M.codes = extra; extra += T::CODE_WORDS; M.hist = M.codes; }
    M.gflag = extra;
    return M;
}
typedef __attribute__((address_space(1))) uint32_t* cwq_g32;
template <class T> __device__ __forceinline__ void poaq_code_store(const PoaQ<T>& M, int idx, uint32_t v) {
    if constexpr (T::CG) ((cwq_g32)M.codes)[idx] = v; else ((cwc_l32)M.codes)[idx] = v;
}
template <class T> __device__ __forceinline__ uint32_t poaq_code_load(const PoaQ<T>& M, int idx) {
    if constexpr (T::CG) return ((cwq_g32)M.codes)[idx]; else return ((cwc_l32)M.codes)[idx];
}

#if CW_CONS_HEAVIEST_BUNDLE
/* cw_policy.h CW_POA_CONSENSUS_HEAVIEST_BUNDLE on one lane (cf. poa_consensus_hb): scores by node in the row words, the chosen source in rpred0 */
template <class T>
__device__ __forceinline__ uint32_t poaq_consensus_hb(const PoaQ<T>& M, const int n, const PoaTask& t, const DevScratch& sc) {
    uint32_t* score = M.rmeta;
    uint8_t* pred = M.rpred0;
    int end = -1;
    uint32_t end_score = 0;
    for (int r = 0; r < n; ++r) {
        const int v = M.r2n[r];
        int p = -1;
        uint32_t wb = 0, ps = 0;
        for (uint32_t e = M.in_head[v]; e != CW_NONE8; e = M.enext[e]) {
            const int u = M.efrom[e];
            const uint32_t w = M.ew[e], su = score[u];
            if (p < 0 || wb < w || (wb == w && ps <= su)) { wb = w; p = u; ps = su; }
        }
        const uint32_t s_ = p < 0 ? 0u : wb + ps;
        score[v] = s_;
        pred[v] = p < 0 ? (uint8_t)CW_NONE8 : (uint8_t)p;
        if (!M.has_out[v] && (end < 0 || end_score < s_)) { end = v; end_score = s_; }
    }
    uint32_t len = 0;
    for (int v = end; v >= 0; v = pred[v] == CW_NONE8 ? -1 : (int)pred[v]) ++len;
    if (len <= t.out_cap) {
        uint32_t k = len;
        for (int v = end; v >= 0; v = pred[v] == CW_NONE8 ? -1 : (int)pred[v]) sc.arena[t.out_off + --k] = CW_ACGT(M.nbase[v]);
    }
    return len;
}
#endif

/* DP fill of one member against the graph, recording the decisions.  Lane gl of the group owns columns 2gl (low half) and 2gl + 1.
   Returns the DP row of the end cell | its column << 16 (the best sink of the last column, lowest rank on ties; overlap mode: the
   best cell of a sink's row, lowest rank then lowest column). */
template <class T>
__device__ __forceinline__ int poaq_fill_c(const PoaQ<T>& M, const int n, const int cols, const int gl) {
    typedef __attribute__((address_space(1))) int* gint;
    constexpr int GW = T::GW;
    const int G4 = 4 * CW_POA_GAP, MS4 = 4 * (CW_POA_MATCH - CW_POA_GAP), XS4 = 4 * (CW_POA_MISMATCH - CW_POA_GAP);
    const int G4PK = pk_make(G4, G4);
    const int ROW0 = 0x00030003; /* the virtual start row: W = 0 everywhere */
    const int L = cols - 1;
    const int j0 = 2 * gl, j1 = j0 + 1;
    const int q0 = (j0 >= 1 && j0 < cols) ? (int)M.sq[j0 - 1] : -1, q1 = (j1 < cols) ? (int)M.sq[j1 - 1] : -1;
    const int qpk = (q0 >= 0 ? 1 << q0 : 0) | (q1 >= 0 ? 1 << (16 + q1) : 0); /* bit b of a half: the column's base is b */
    const int xs_pk = pk_make(gl == 0 ? CW_NEG16 : XS4, XS4);                 /* column 0 has no diagonal */
    const cw_s2 ms_pk = (cw_s2)(short)(MS4 - XS4);
    gint keep = (gint)M.keep;
    int rc0 = ROW0, rc1 = ROW0, rc2 = ROW0; /* rows i-1, i-2, i-3 */
    int bs0 = (int)0x80000000, bi0 = 0, bs1 = (int)0x80000000, bi1 = 0; /* best sink cell of this lane's even / odd column */
    uint32_t acc = 0u;
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        const uint32_t meta = M.rmeta[r];
        const int np = CW_RM_NP(meta), x = CW_RM_X(meta);
        const int t_ = (int)(((unsigned)qpk >> (meta & 3u)) & 0x00010001u);
        const int srow = __builtin_bit_cast(int, (cw_s2)(__builtin_bit_cast(cw_s2, t_) * ms_pk + __builtin_bit_cast(cw_s2, xs_pk)));
        int dm, vm;
        if (CW_RM_LIN(meta)) { /* one in-edge, from the row before */
            dm = __builtin_amdgcn_alignbit(rc0, g_shr1<T>(rc0, gl), 16); vm = rc0;
        } else {
            dm = CW_NEGPK; vm = CW_NEGPK;
            for (int q = 0; q < np; ++q) {
                const int prow = np == 1 ? x : (int)M.plist[x + q];
                const int dist = i - prow;
                int up;
                if (prow == 0) up = ROW0;
                else if (dist <= 3) up = dist == 1 ? rc0 : dist == 2 ? rc1 : rc2;
                else if (dist <= T::RING) up = (int)M.ring[(prow & (T::RING - 1)) * GW + gl];
                else up = keep[prow * GW + gl];
                const int qq = q < 3 ? q : 3;
                up = pk_sub(up, pk_make(qq, qq));
                dm = pk_max(dm, __builtin_amdgcn_alignbit(up, g_shr1<T>(up, gl), 16)); /* (col 2gl - 1, col 2gl) of the predecessor row; group lane 0: nothing (column 0 takes no diagonal) */
                vm = pk_max(vm, up);
            }
        }
        const int kD = pk_add(dm, srow), kV = pk_add(vm, G4PK);
        int v = pk_max(kD, kV);
        if (CW_POA_OV && gl == 0) v = (int)(((unsigned)v & 0xFFFF0000u) | 3u); /* overlap mode (cw_policy.h): column 0 is free */
        int w = pk_max(v, (v << 16) | (CW_NEG16 & 0xFFFF));                   /* the odd column sees the even one of its lane */
        const unsigned inc = g_scan_max_u32<T>(((unsigned)w >> 16) ^ 0x8000u); /* the lanes' running maxima as biased unsigned numbers */
        const unsigned ex = (unsigned)g_shr1<T>((int)inc, gl);                 /* exclusive; 0 = nothing to the left */
        w = pk_max(w, pk_splat_lo((int)(ex ^ 0x8000u)));
        const int nv = w | 0x00030003;
        /* the codes of both columns: in-edge q (candidate ^ cell, the cell's low bits being 3) for a diagonal move, 4 + q for a vertical one, 8 for a horizontal one */
        const int tD = kD ^ nv, tV = kV ^ nv;
        const int fD = pku_max(tD, pku_shl1(tD & (int)0xFFFCFFFC));                    /* q if the upper bits agree, >= 8 otherwise */
        const int fV = pk_add(pku_min(pku_max(tV, pku_shl1(tV & (int)0xFFFCFFFC)), 0x00080008), 0x00040004);
        const int code = pku_min(pku_min(fD, fV), 0x00080008);
        acc |= (uint32_t)code << ((r & 3) * 4);
        if ((r & 3) == 3) { poaq_code_store<T>(M, (r >> 2) * GW + gl, acc); acc = 0u; }
        M.ring[(i & (T::RING - 1)) * GW + gl] = (uint32_t)nv;
        if (meta & 24u) {
            if (meta & 16u) keep[i * GW + gl] = nv;
            if (CW_RM_SINK(meta)) {
                if (CW_POA_OV) { /* H form: columns of one row are compared */
                    const int h0 = ((int)(short)(nv & 0xFFFF) >> 2) + j0 * CW_POA_GAP, h1 = (nv >> 18) + j1 * CW_POA_GAP;
                    if (j0 >= 1 && j0 <= L && h0 > bs0) { bs0 = h0; bi0 = i; }
                    if (j1 <= L && h1 > bs1) { bs1 = h1; bi1 = i; }
                } else {
                    const int h0 = (int)(short)(nv & 0xFFFF), h1 = nv >> 16;
                    if (h0 > bs0) { bs0 = h0; bi0 = i; } /* ranks ascend: the lowest rank keeps a tie */
                    if (h1 > bs1) { bs1 = h1; bi1 = i; }
                }
            }
        }
        rc2 = rc1; rc1 = rc0; rc0 = nv;
    }
    if (n & 3) poaq_code_store<T>(M, ((n - 1) >> 2) * GW + gl, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the kept rows (and global code words) are read by other lanes in the traceback */
    int bi, bj = L;
    if (CW_POA_OV) { /* the best over the group's lanes: value, then lowest rank, then lowest column */
        int bs = bs0, br = bi0, bc = j0;
        if (bs1 > bs || (bs1 == bs && bi1 < br)) { bs = bs1; br = bi1; bc = j1; }
        if (bs == (int)0x80000000) br = 0x7FFFFFFF;
        for (int o = GW / 2; o > 0; o >>= 1) {
            const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o), oc = __shfl_xor(bc, o);
            if (os > bs || (os == bs && (orr < br || (orr == br && oc < bc)))) { bs = os; br = orr; bc = oc; }
        }
        bi = br; bj = bc;
    } else {
        bi = g_bcast<T>((L & 1) ? bi1 : bi0, L >> 1);
    }
    return bi | (bj << 16);
}

/* Follows the recorded codes from (bi, bj) towards the virtual start; writes seqrank[j] = rank aligned to sequence position j (diagonal
   moves).  Stops in column 0: what is left of the path there is vertical and aligns nothing.  Returns false when the walk does not
   end (cannot happen; reported as an internal error). */
template <class T>
__device__ __forceinline__ bool poaq_trace_c(const PoaQ<T>& M, const int n, const int bi, const int bj, const int gl) {
    typedef __attribute__((address_space(1))) const int* gcint;
    constexpr int GW = T::GW;
    const int G4 = 4 * CW_POA_GAP, MS4 = 4 * (CW_POA_MATCH - CW_POA_GAP), XS4 = 4 * (CW_POA_MISMATCH - CW_POA_GAP);
    gcint keep = (gcint)M.keep;
    int i = bi, j = bj, trips = 0;
    while (i > 0 && j > 0) {
        if (++trips > n + 2 * GW + 8) return false;
        /* lane t looks at cell (i - t, j - t): a run of diagonal moves through first in-edges that lead to the rank before */
        const int ri = i - gl, cj = j - gl;
        bool ok = false;
        int nib = 15;
        uint32_t meta = 0u;
        if (ri >= 1 && cj >= 1) {
            const uint32_t cw = poaq_code_load<T>(M, ((ri - 1) >> 2) * GW + (cj >> 1));
            nib = (int)((cw >> (((cj & 1) << 4) + ((ri - 1) & 3) * 4)) & 15u);
            meta = M.rmeta[ri - 1];
            ok = nib == 0 && (int)M.rpred0[ri - 1] == ri - 1;
        }
        const unsigned okb = g_ballot<T>(ok);
        const int run = __ffsll((long long)~(unsigned long long)okb) - 1; /* leading lanes that continue the run (0 .. GW) */
        if (run > 0) {
            if (gl < run) M.seqrank[cj - 1] = (uint8_t)(ri - 1);
            i -= run; j -= run;
            continue;
        }
        /* one step, decided by the group's lane 0 */
        const int nib0 = g_bcast<T>(nib, 0);
        const uint32_t meta0 = (uint32_t)g_bcast<T>((int)meta, 0);
        if (nib0 == 8) { j--; continue; }
        const int mv = nib0 >> 2;
        int q = nib0 & 3;
        const int np = CW_RM_NP(meta0), off = CW_RM_X(meta0);
        if (q == 3) { /* fourth in-edge or later: the first of them whose candidate equals the cell (the values of these rows are kept) */
            const int base = (int)(meta0 & 3u);
            const int cjv = mv == 0 ? j - 1 : j;
            const int hw = keep[i * GW + (j >> 1)];
            const int h = (j & 1) ? (hw >> 16) : (int)(short)(hw & 0xFFFF);
            const int add = mv == 0 ? (((int)M.sq[j - 1] == base) ? MS4 : XS4) : G4;
            q = -1;
            for (int t = 3; t < np && q < 0; ++t) {
                const int pr = (int)M.plist[off + t];
                const int pw = pr == 0 ? 0x00030003 : keep[pr * GW + (cjv >> 1)];
                const int pv = (cjv & 1) ? (pw >> 16) : (int)(short)(pw & 0xFFFF);
                if (h == pv + add) q = t; /* both values carry the low bits 3 */
            }
            if (q < 0) return false;
        }
        const int pr = np == 1 ? off : (int)M.plist[off + q];
        if (mv == 0) { if (gl == 0) M.seqrank[j - 1] = (uint8_t)(i - 1); j--; }
        i = pr;
    }
    return true;
}

/* What a group carries from member to member of its task (registers; every lane of the group holds the same values). */
struct PoaQSt {
    int n, ne, nseq, tpl_nodes, prev_L;
    bool meta_ok, prev_clean; /* prev_clean: the member before this one was aligned and added neither a node nor an edge */
    unsigned long long pt;    /* phase clock (POAQ_PROF) */
    __device__ __forceinline__ void reset() { n = 0; ne = 0; nseq = 0; tpl_nodes = 0; prev_L = -1; meta_ok = false; prev_clean = false; }
};
#define POAQ_PROF(slot) do { const unsigned long long _n = __builtin_readcyclecounter(); acc[slot] += _n - S.pt; S.pt = _n; } while (0)

/* Round 6, the replay of a repeated member (CW_POAQ_REPLAY).  A member that is, base for base, the member aligned just before it -- and that one changed nothing
   in the graph but coverage counts -- meets the same graph with the same bases: the same fill, the same walk back, the same path.  Its merge is the coverage
   counts of that path once more (M.pcur still holds it, M.sq the bases), and the alignment is not run.  Measured on the checker (a scratch build of the
   restatement over the bench piles, every replayed path compared with the one the alignment gives: 0 of 14 000 differ): 38 % of this tier's members at depth
   150 -- the error-free copy of a short piece comes again and again.  The four tasks of a wave share one instruction stream, so a group does not skip a round
   the others run: it takes its repeated members in a short loop of its own HERE and joins the round with the first member that needs an alignment.
   (Not under the heaviest-bundle policy: a replay would have to raise the path's edge weights too.) */
/* Takes the members from mi on that repeat the member aligned before them; returns with pm = member mi, the first one that needs an alignment (or mi == n_members). */
template <class T>
__device__ __forceinline__ void poaq_replay(const PoaQ<T>& M, PoaQSt& S, const PoaTask& t, const DevBatch& b, const DevScratch& sc, uint32_t& mi, PoaMember& pm, const int gl) {
#if CW_POAQ_REPLAY
    constexpr int GW = T::GW;
    if (!CW_CONS_HEAVIEST_BUNDLE) {
        while (S.prev_clean && (int)pm.len == S.prev_L) {
            const uint32_t* words_ = b.bases + b.seq_word_off[pm.seq];
            bool same = true;
            for (int j = gl; j < S.prev_L; j += GW) same = same && M.sq[j] == (uint8_t)cw_base_at(words_, pm.start + j);
            if (g_ballot<T>(!same) != 0u) break;
            for (int j = gl; j < S.prev_L; j += GW) { const int cur = M.pcur[j]; M.ncov[cur] = (uint8_t)(M.ncov[cur] + 1); } /* (a path visits a node once; lane j owns position j, as in the merge) */
            S.nseq++;
            if (++mi >= t.n_members) break;
            pm = sc.members[t.member_off + mi];
        }
    }
#endif
}

/* the member's bases into M.sq; the first member of a task becomes a chain (returns 1: nothing to align), a later one is aligned by poaq_member (0); 2 = beyond this tier */
template <class T>
__device__ __forceinline__ int poaq_take(const PoaQ<T>& M, PoaQSt& S, const PoaMember& pm, const DevBatch& b, const int gl) {
    constexpr int GW = T::GW;
    const int L = (int)pm.len;
    if ((uint32_t)L > (uint32_t)T::LC) return 2;
    {
        const uint32_t* words = b.bases + b.seq_word_off[pm.seq];
        for (int j = gl; j < L; j += GW) M.sq[j] = (uint8_t)cw_base_at(words, pm.start + j);
    }
    cw_wave_sync();
    S.nseq++;
    S.prev_clean = false; S.prev_L = L;
    if (S.n != 0) return 0;
    /* first member: a chain */
    if ((uint32_t)L > (uint32_t)T::NC || (uint32_t)L > (uint32_t)T::EC) return 2;
    for (int j = gl; j < L; j += GW) {
        M.nbase[j] = M.sq[j]; M.ncov[j] = 1; M.nalc[j] = 0;
        M.in_head[j] = j ? (uint8_t)(j - 1) : CW_NONE8; M.in_tail[j] = M.in_head[j];
        M.indeg[j] = j ? 1 : 0; M.has_out[j] = (j < L - 1) ? 1 : 0;
        M.r2n[j] = (uint8_t)j; M.n2r[j] = (uint8_t)j;
        if (j) { M.efrom[j - 1] = (uint8_t)(j - 1); M.enext[j - 1] = CW_NONE8; if (CW_CONS_HEAVIEST_BUNDLE) M.ew[j - 1] = 1; }
    }
    S.n = L; S.ne = L - 1; S.tpl_nodes = L; S.meta_ok = false;
    cw_wave_sync();
    return 1;
}

/* one member (its bases are in M.sq) against the graph: rank metadata, fill, walk back, merge.  0 = done, 2 = beyond this tier, 3 = failed */
template <class T>
__device__ __forceinline__ int poaq_member(const PoaQ<T>& M, PoaQSt& S, const int L, const int gl, unsigned long long (&acc)[5]) {
    constexpr int GW = T::GW;
    int& n = S.n; int& ne = S.ne; bool& meta_ok = S.meta_ok; bool& prev_clean = S.prev_clean;
    const unsigned lt_mask = (1u << gl) - 1u;
    const int cols = L + 1;

    /* ---- per-rank metadata: the row words (CW_RM_WORD), predecessor lists, which rows the fill keeps in the slab ---- */
    if (!meta_ok) {
        int run = 0;
        for (int w = gl; w < T::GFLAG_WORDS; w += GW) M.gflag[w] = 0u;
        cw_wave_sync();
        for (int r0 = 0; r0 < n; r0 += GW) {
            const int r = r0 + gl;
            const int node = r < n ? M.r2n[r] : 0;
            const int d = r < n ? M.indeg[node] : 0;
            const int inc = g_scan_add<T>(d);
            const int off = run + inc - d;
            if (r < n) {
                int q = off, first = 0;
                for (uint32_t e = M.in_head[node]; e != CW_NONE8; e = M.enext[e]) {
                    const int pr = M.n2r[M.efrom[e]] + 1;
                    if (q == off) first = pr;
                    M.plist[q++] = (uint8_t)pr;
                    if (r + 1 - pr > T::RING || q - off > 3) /* read back from further than the ring reaches / compared by the traceback (fourth in-edge and later) */
                        __hip_atomic_fetch_or((cwc_l32)M.gflag + ((pr - 1) >> 5), 1u << ((pr - 1) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                if (d > 3) __hip_atomic_fetch_or((cwc_l32)M.gflag + (r >> 5), 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                M.rpred0[r] = (uint8_t)first;
                const uint32_t np_ = (uint32_t)(d ? d : 1);
                M.rmeta[r] = CW_RM_WORD(M.nbase[node], np_, np_ == 1u && first == r, !M.has_out[node], 0u, np_ == 1u ? first : off);
            }
            run += g_bcast<T>(inc, GW - 1);
        }
        meta_ok = true;
        cw_wave_sync();
        for (int r = gl; r < n; r += GW) if ((M.gflag[r >> 5] >> (r & 31)) & 1u) M.rmeta[r] |= 16u;
        cw_wave_sync();
    }
    POAQ_PROF(0);

    /* ---- DP fill (records the decisions), end cell ---- */
    for (int j = gl; j < L; j += GW) M.seqrank[j] = CW_NONE8;
    cw_wave_sync();
    const int be = poaq_fill_c<T>(M, n, cols, gl);
    cw_wave_sync();
    POAQ_PROF(1);

    /* ---- traceback over the code words ---- */
    if (!poaq_trace_c<T>(M, n, be & 0xFFFF, be >> 16, gl)) return 3;
    cw_wave_sync();
    POAQ_PROF(2);

    /* ---- merge the path into the graph: one lane per sequence position, 16 at a time (cf. poa_run) ---- */
    {
        const int n_old = n;
        const int chunks = (L + GW - 1) / GW;
        bool edges_added = false;
        int next_rank = -1;
        for (int c = chunks - 1; c >= 0; --c) {
            const int j = c * GW + gl;
            const bool act = j < L;
            const uint32_t rk = act ? M.seqrank[j] : CW_NONE8;
            const unsigned has = g_ballot<T>(act && rk != CW_NONE8);
            const unsigned later = has & ~(lt_mask | (1u << gl));
            const int later_rank = (int)(uint32_t)g_bcast<T>((int)rk, later ? (__ffs((int)later) - 1) : 0);
            const int qr = later ? later_rank : next_rank;
            const int first_rank = (int)(uint32_t)g_bcast<T>((int)rk, has ? (__ffs((int)has) - 1) : 0);
            uint32_t cur = CW_NONE8, at = CW_NONE8;
            if (act) {
                const int bcode = M.sq[j];
                if (rk != CW_NONE8) {
                    const int pn = M.r2n[rk];
                    if (M.nbase[pn] == bcode) cur = (uint32_t)pn;
                    else {
                        const int ac = M.nalc[pn];
                        int last = (int)rk;
                        for (int a = 0; a < ac; ++a) {
                            const int v = M.nal[pn * 3 + a];
                            if (M.nbase[v] == bcode) cur = (uint32_t)v;
                            last = max(last, (int)M.n2r[v]);
                        }
                        if (cur == CW_NONE8) at = (uint32_t)(last + 1);
                    }
                } else if (qr < 0) {
                    at = (uint32_t)n_old;
                } else {
                    const int q = M.r2n[qr];
                    int first = qr;
                    for (int a = 0; a < M.nalc[q]; ++a) first = min(first, (int)M.n2r[M.nal[q * 3 + a]]);
                    at = (uint32_t)first;
                }
                M.pcur[j] = (uint8_t)cur;
                M.pat[j] = (uint8_t)at;
            }
            if (has) next_rank = first_rank;
        }
        cw_wave_sync();
        int fresh_total = 0;
        for (int c = 0; c < chunks; ++c) {
            const int j = c * GW + gl;
            const bool act = j < L;
            const bool fresh = act && M.pcur[j] == CW_NONE8;
            const unsigned fb = g_ballot<T>(fresh);
            if (fresh) {
                const int cur = n_old + fresh_total + __popc(fb & lt_mask);
                if ((uint32_t)cur < (uint32_t)T::NC) {
                    M.pcur[j] = (uint8_t)cur;
                    M.nbase[cur] = M.sq[j]; M.ncov[cur] = 1; M.nalc[cur] = 0;
                    M.in_head[cur] = CW_NONE8; M.in_tail[cur] = CW_NONE8; M.indeg[cur] = 0; M.has_out[cur] = 0;
                    const uint32_t rk = M.seqrank[j];
                    if (rk != CW_NONE8) {
                        const int pn = M.r2n[rk];
                        const int ac = M.nalc[pn];
                        for (int a = 0; a < ac; ++a) {
                            const int v = M.nal[pn * 3 + a];
                            M.nal[cur * 3 + a] = (uint8_t)v;
                            M.nal[v * 3 + M.nalc[v]] = (uint8_t)cur; M.nalc[v] = (uint8_t)(M.nalc[v] + 1);
                        }
                        M.nal[cur * 3 + ac] = (uint8_t)pn; M.nalc[cur] = (uint8_t)(ac + 1);
                        M.nal[pn * 3 + ac] = (uint8_t)cur; M.nalc[pn] = (uint8_t)(ac + 1);
                    }
                }
            } else if (act) {
                const int cur = M.pcur[j];
                M.ncov[cur] = (uint8_t)(M.ncov[cur] + 1);
            }
            fresh_total += __popc(fb);
        }
        if ((uint32_t)(n_old + fresh_total) > (uint32_t)T::NC) return 2;
        cw_wave_sync();
        if (fresh_total > 0) {
            uint32_t* hist = M.hist; /* n_old + 1 counters: the code words (or the row ring) are dead between the traceback and the next fill */
            for (int r = gl; r <= n_old; r += GW) hist[r] = 0;
            cw_wave_sync();
            for (int c = 0; c < chunks; ++c) {
                const int j = c * GW + gl;
                if (j < L && M.pat[j] != CW_NONE8) atomicAdd(&hist[M.pat[j]], 1u);
            }
            cw_wave_sync();
            int run = 0;
            for (int r0 = 0; r0 < n_old; r0 += GW) {
                const int r = r0 + gl;
                const int hcount = r < n_old ? (int)hist[r] : 0;
                const int inc = g_scan_add<T>(hcount);
                if (r < n_old) {
                    const int nr = r + run + inc;
                    const int v = M.r2n[r];
                    M.rtmp[nr] = (uint8_t)v;
                    M.n2r[v] = (uint8_t)nr;
                }
                run += g_bcast<T>(inc, GW - 1);
            }
            for (int c = 0; c < chunks; ++c) {
                const int j = c * GW + gl;
                if (j < L && M.pat[j] != CW_NONE8) {
                    const int cur = M.pcur[j];
                    const int nr = (int)M.pat[j] + (cur - n_old);
                    M.rtmp[nr] = (uint8_t)cur;
                    M.n2r[cur] = (uint8_t)nr;
                }
            }
            cw_wave_sync();
            n = n_old + fresh_total;
            for (int r = gl; r < n; r += GW) M.r2n[r] = M.rtmp[r];
            meta_ok = false;
            cw_wave_sync();
        }
        for (int c = 0; c < chunks; ++c) {
            const int j = c * GW + gl;
            const bool act = j < L && j > 0;
            int head = 0, cur = 0;
            bool add = false;
            if (act) {
                head = M.pcur[j - 1]; cur = M.pcur[j];
                add = true;
                for (uint32_t e = M.in_head[cur]; e != CW_NONE8; e = M.enext[e])
                    if (M.efrom[e] == (uint8_t)head) { add = false; if (CW_CONS_HEAVIEST_BUNDLE) M.ew[e] = (uint8_t)(M.ew[e] + 1); break; }
            }
            const unsigned ab = g_ballot<T>(add);
            const int total = __popc(ab);
            if ((uint32_t)(ne + total) > (uint32_t)T::EC) return 2;
            if (add) {
                const int e = ne + __popc(ab & lt_mask);
                M.efrom[e] = (uint8_t)head; M.enext[e] = CW_NONE8;
                if (CW_CONS_HEAVIEST_BUNDLE) M.ew[e] = 1;
                const uint32_t tl = M.in_tail[cur];
                if (tl == CW_NONE8) M.in_head[cur] = (uint8_t)e; else M.enext[tl] = (uint8_t)e;
                M.in_tail[cur] = (uint8_t)e;
                M.indeg[cur] = (uint8_t)(M.indeg[cur] + 1);
                M.has_out[head] = 1;
            }
            if (total) { ne += total; meta_ok = false; edges_added = true; }
        }
        cw_wave_sync();
        prev_clean = n == n_old && !edges_added; /* nothing but coverage counts changed: the next member may be a replay of this one */
    }
    POAQ_PROF(3);
    return 0;
}

/* the task's consensus into its arena slot; 1 = done, 3 = the slot is too small */
template <class T>
__device__ __forceinline__ int poaq_finish(const PoaQ<T>& M, PoaQSt& S, const PoaTask& t, const DevScratch& sc, const int gl, unsigned long long (&acc)[5]) {
    constexpr int GW = T::GW;
    const int n = S.n, nseq = S.nseq, tpl_nodes = S.tpl_nodes;
    const unsigned lt_mask = (1u << gl) - 1u;
    (void)nseq; (void)tpl_nodes; (void)lt_mask;
    /* ---- consensus: column-majority vote, or the heaviest bundle (cw_policy.h CW_POA_CONSENSUS) ---- */
    uint32_t out_len = 0;
#if CW_CONS_HEAVIEST_BUNDLE
    cw_wave_sync();
    if (gl == 0) out_len = poaq_consensus_hb<T>(M, n, t, sc);
    out_len = (uint32_t)g_bcast<T>((int)out_len, 0);
#else
    for (int r0 = 0; r0 < n; r0 += GW) {
        const int r = r0 + gl;
        int emit = -1;
        if (r < n) {
            const int v = M.r2n[r];
            const int ac = M.nalc[v];
            bool first = true;
            for (int a = 0; a < ac; ++a) if (M.n2r[M.nal[v * 3 + a]] < r) first = false;
            if (first) {
                int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                int tpl_code = -1;
                for (int c = 0; c <= ac; ++c) {
                    const int u = M.r2n[r + c];
                    const int code = M.nbase[u];
                    const int cv = M.ncov[u];
                    c0 += code == 0 ? cv : 0; c1 += code == 1 ? cv : 0; c2 += code == 2 ? cv : 0; c3 += code == 3 ? cv : 0;
                    if (u < tpl_nodes) tpl_code = code;
                }
                const int gaps = nseq - (c0 + c1 + c2 + c3);
                int top = 0, tc = c0;
                if (c1 > tc) { top = 1; tc = c1; }
                if (c2 > tc) { top = 2; tc = c2; }
                if (c3 > tc) { top = 3; tc = c3; }
                if (!CW_CONS_DROPS(gaps, tc)) { /* cw_policy.h "switches" */
                    const int tplc = tpl_code == 0 ? c0 : tpl_code == 1 ? c1 : tpl_code == 2 ? c2 : tpl_code == 3 ? c3 : -1;
                    if (CW_CONS_TEMPLATE_WINS_TIES && tplc == tc) top = tpl_code;
                    emit = top;
                }
            }
        }
        const unsigned bal = g_ballot<T>(emit >= 0);
        const uint32_t idx = out_len + (uint32_t)__popc(bal & lt_mask);
        if (emit >= 0 && idx < t.out_cap) sc.arena[t.out_off + idx] = CW_ACGT(emit);
        out_len += (uint32_t)__popc(bal);
    }
#endif
    if (out_len > t.out_cap) return 3;
    if (gl == 0) sc.seg_len[t.seg_slot] = out_len;
    POAQ_PROF(4);
    return 1;
}

/* a whole task, member after member */
template <class T>
__device__ int poaq_run(const PoaQ<T>& M, const PoaTask& t, const DevBatch& b, const DevScratch& sc, const int gl, unsigned long long (&acc)[5]) {
    PoaQSt S;
    S.reset(); S.pt = __builtin_readcyclecounter();
    if (t.n_members > 255u) return 2; /* coverage counts and edge weights are bytes here */
    for (uint32_t mi = 0; mi < t.n_members; ++mi) {
        PoaMember pm = sc.members[t.member_off + mi];
        poaq_replay<T>(M, S, t, b, sc, mi, pm, gl);
        if (mi >= t.n_members) break;
        const int tk = poaq_take<T>(M, S, pm, b, gl);
        if (tk == 2) return 2;
        if (tk == 1) continue;
        const int rc = poaq_member<T>(M, S, (int)pm.len, gl, acc);
        if (rc) return rc;
    }
    return poaq_finish<T>(M, S, t, sc, gl, acc);
}

/* ---- tier Q (four tasks per wave) and tier H (two): one kernel body ------------------------------------------------------------------ */
#ifndef CW_POAQ_FLAT
#define CW_POAQ_FLAT 0 /* 1: the flat loop below (built, bit-identical, measured slower: off; the variant `qflat` of tests/test_gpu_variants.py) */
#endif
/* Round 6, the flat loop -- tried, measured, off.  The groups of a wave share one instruction stream, so the loop "per group: take a task, run all its members"
   makes every group wait, task after task, for the slowest of the four: the consensus phase of the profile, which the first group's clock charges with that wait,
   is 10.2 of this tier's 62.2 G wave-cycles per depth-150 batch although the tier list is sorted by size.  The flat loop is over ALIGNMENTS instead: each round
   every group brings one member of its own task to the alignment (a new task, its first member's chain, its repeated members and the consensus of the task it has
   just finished on the way, in a loop of its own), and a group whose task ends takes the next one without waiting.  Measured (same box, four alternating runs):
   the kernel 8.84 -> 9.24 ms.  The wait is gone (consensus 10.2 -> 1.3 G) and comes back twice: the work at a task's boundary (a chain of five dependent loads,
   the chain, the consensus) now runs once per GROUP with the other three idle, where the per-task loop runs it once per wave for all four (+6.0 G), and groups
   that are at different members of their tasks meet graphs of different sizes -- the fill runs as many rows as the largest (+4.9 G). */
template <class T, int TIER_LIST, int NEXT_TIER, int PROF_BASE, bool PRODUCER>
__device__ __forceinline__ void poaq_kernel_body(const DevBatch& b, const DevScratch& sc, uint8_t* lds, uint8_t* slab_base) {
    const int gl = threadIdx.x & (T::GW - 1);
    const uint32_t grp = threadIdx.x / T::GW; /* 0 .. (64 / GW) * waves - 1 */
    typedef __attribute__((address_space(1))) uint8_t* cw_gptr;
    const PoaQ<T> M = poaq_carve<T>(lds + (size_t)grp * T::TASK_BYTES, (uint8_t*)(cw_gptr)(slab_base + (size_t)(blockIdx.x * (blockDim.x / T::GW) + grp) * T::SLAB_BYTES));
    const uint32_t* list = sc.tier_list[TIER_LIST];
    const uint32_t n_work = min(sc.ctr->n_tier[TIER_LIST], sc.list_cap);
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
#if CW_POAQ_FLAT
    PoaQSt S;
    S.reset(); S.pt = __builtin_readcyclecounter();
    PoaTask t = {};
    uint32_t ti = 0, mi = 0;
    bool have = false, done = false;
    for (;;) {
        bool aligning = false;
        int L = 0;
        while (!done) { /* (per group) until this group has a member to align or the list is empty */
            if (have && mi >= t.n_members) { /* the task's last member is in: its consensus */
                const int rc = poaq_finish<T>(M, S, t, sc, gl, acc);
                if (gl == 0) poa_hand_over(sc, t, ti, rc, NEXT_TIER);
                cw_wave_sync();
                have = false;
            }
            if (!have) {
                uint32_t m = 0;
                if (gl == 0) m = atomicAdd(&sc.ctr->next_tier[TIER_LIST], 1u);
                m = (uint32_t)g_bcast<T>((int)m, 0);
                if (m >= n_work) { done = true; break; }
                ti = list[m];
                t = sc.tasks[ti];
                if (t.n_members == 0) continue; /* a neutral entry (cw_chain.h "cap_ok") */
                if (t.n_members > 255u) { if (gl == 0) poa_hand_over(sc, t, ti, 2, NEXT_TIER); continue; } /* coverage counts and edge weights are bytes here */
                S.reset();
                mi = 0; have = true;
            }
            PoaMember pm = sc.members[t.member_off + mi];
            poaq_replay<T>(M, S, t, b, sc, mi, pm, gl);
            if (mi >= t.n_members) continue;
            const int tk = poaq_take<T>(M, S, pm, b, gl);
            if (tk == 2) { if (gl == 0) poa_hand_over(sc, t, ti, 2, NEXT_TIER); cw_wave_sync(); have = false; continue; }
            if (tk == 1) { ++mi; continue; } /* the first member: a chain */
            L = (int)pm.len;
            aligning = true;
            break;
        }
        if (__ballot(aligning) == 0ull) break; /* every group has seen the end of the list */
        if (aligning) {
            const int rc = poaq_member<T>(M, S, L, gl, acc);
            if (rc) { if (gl == 0) poa_hand_over(sc, t, ti, rc, NEXT_TIER); cw_wave_sync(); have = false; }
            else ++mi;
        }
    }
#else
    for (;;) {
        uint32_t mi = 0;
        if (gl == 0) mi = atomicAdd(&sc.ctr->next_tier[TIER_LIST], 1u);
        mi = (uint32_t)g_bcast<T>((int)mi, 0);
        if (mi >= n_work) break;
        const uint32_t ti = list[mi];
        const PoaTask t = sc.tasks[ti];
        if (t.n_members == 0) continue; /* a neutral entry (cw_chain.h "cap_ok") */
        const int rc = poaq_run<T>(M, t, b, sc, gl, acc);
        if (gl == 0) poa_hand_over(sc, t, ti, rc, NEXT_TIER);
        cw_wave_sync();
    }
#endif
    /* per-phase cycles as the first group of every wave saw them (the groups of a wave share one instruction stream) */
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 5; ++q) atomicAdd(&sc.ctr->prof[PROF_BASE + q], acc[q]);
    if (PRODUCER) poa_producer_done(sc);
}
#undef POAQ_PROF

/* tier Q: a task that outgrows it (rc 2) is redone in tier S, whose kernel follows on the stream (hand-over list 0) */
__global__ void __launch_bounds__(64 * CW_POAQ_WAVES) cw_poa_q_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    poaq_kernel_body<PoaQ16, 0, 0, 28, false>(b, sc, lds, sc.q_slab);
}
#ifdef CW_TEST_AIDS /* (opt-in, measured no faster: only the test-aid build carries the kernel) */
/* tier H: members of up to 63 bases, graphs of up to 128 nodes; a task that outgrows it goes to tier L's live queue, like tier S's */
__global__ void __launch_bounds__(64 * 6, 5) cw_poa_h_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    poaq_kernel_body<PoaQ32, 5, 3, 64, true>(b, sc, lds, sc.h_slab);
}
#endif

#endif /* CW_Q_CODES */
#endif
