/*
 * cw_poa_c.h -- round 4: a DP fill that records the traceback's decisions, and the traceback that follows them (A4d).
 *
 * For members of at most 63 bases (one DP column per lane; at depth 150 that is 97 % of the rows the one-task-per-wave tiers fill).
 * Measured on a depth-150 batch with a counting build (-DCW_DIAG, tools/diag_rows.sh): in tier M1 every second row read a predecessor
 * row back from the L2-resident matrix (51 M such reads for 99 M rows: the other arm of a bubble is 3-16 ranks back), and 70 % of the
 * traceback's tile trips ended in a node with several in-edges, which cost two more dependent round trips to the L2.  Both go away
 * when the fill itself says, per cell, which move the traceback will take and through WHICH in-edge:
 *
 *   values    rows are kept as W[i][j] = H[i][j] - j * gap, scaled by 4.  In that form a horizontal move costs nothing, so the horizontal
 *             recurrence is a plain inclusive prefix max of the row's candidates (no subtraction and addition of j * gap round the scan),
 *             and the two free low bits carry the ordinal of the in-edge a candidate came through (3 - q for the first three, 0 beyond).
 *   decision  a cell's value is at least every candidate, so a candidate explains the cell iff it is the largest of its kind and equal
 *             to it: the fill keeps the largest diagonal and the largest vertical candidate (on equal values the earlier in-edge has the
 *             larger key and wins, as cw_policy.h demands: diagonal through the in-edges in order, then vertical through them, then
 *             horizontal) and compares them with the finished cell.  Four bits per cell: in-edge | move << 2, eight rows of a lane's
 *             column to a 32-bit word.
 *   row store a predecessor one or two ranks back is in registers; up to CW_RING ranks back it comes from a ring of rows in LDS (tiers
 *             M1 / M2; tier S keeps its whole matrix in LDS); only rows that a later row needs from further back, and the rows round a node
 *             with more than three in-edges, go to the wave's slab in global memory (flag bits from the rank metadata pass).  No matrix
 *             is written otherwise: the traceback never looks at a value.
 *   end cell  the best sink of the last column is tracked while the rows go by (sink bit in the row word).
 *   traceback the 8x8 tile walk of cw_poa.h over the code words instead of over matrix values: a tile row is 0..7 steps up the
 *             first-predecessor chain, a cell continues inside the tile when its move goes through in-edge 0, and a cell whose move goes
 *             through another in-edge names it -- one LDS read of the predecessor list, no search.  Ordinal 0 (fourth in-edge or later)
 *             is the one case decided from stored values, among the in-edges 3.. only.
 * Results are bit-identical to the matrix path (a -DCW_POA_VERIFY build runs both per member and counts differences).
 */
#ifndef CW_POA_C_H
#define CW_POA_C_H

#define CW_POA_GFLAG_BYTES(NC) ((((NC) / 32 + 2) * 4 + 15) / 16 * 16)
#define CW_RM_SINK(m) (((m) & 8u) != 0u)

typedef __attribute__((address_space(3))) int16_t* cwc_l16;
typedef __attribute__((address_space(3))) uint32_t* cwc_l32;
typedef __attribute__((address_space(1))) int16_t* cwc_g16;
typedef __attribute__((address_space(1))) uint32_t* cwc_g32;

/* Global-memory accesses of the fill as inline instructions: the compiler's wait-count pass does not see them, so the row loop does not
   wait (s_waitcnt vmcnt(0)) for a code word or a flagged row to reach the L2 before it goes on -- only the rare read of a row from more
   than CW_RING ranks back waits, inside its own branch.  (As plain C++ every row, also the linear ones, carried such a wait.) */
__device__ __forceinline__ void cwc_gstore_b32(cwc_g32 p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void cwc_gstore_b16(cwc_g16 p, int v) { asm volatile("global_store_short %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
/* (the rare read of a row from the slab stays a plain load: the compiler then waits for it -- and with it for every store before it --
   where the value is first used, which the empty asm pins inside the branch; as an asm load its result register counted as pending at
   the point where the row kinds meet again, and every row waited there) */
__device__ __forceinline__ int cwc_gload_i16_wait(cwc_g16 p) {
    int v = (int)*p;
    asm volatile("" : "+v"(v));
    return v;
}

/* Tiers S / M1 / M2 (CM 2): ring of the last CW_RING rows in LDS, flagged rows and the code words in the wave's global slab, stride 64.
   Returns the DP row of the end cell (best sink in the last column, lowest rank on ties).
   A stored value is 4 * W | 3: the low bits of a row as it is read back are already the ordinal of in-edge 0, so a row with one in-edge
   never touches them and in-edge q of a longer list subtracts min(q, 3).
   A code is q | move << 2 with move 0 diagonal, 1 vertical, 2 horizontal and q the in-edge the move goes through (3: the fourth or a later one).
   The loop is written for the scalar unit as much as for the vector unit: a SIMD issues one scalar and one vector instruction per
   four cycles, from different waves, so with four waves resident a row costs max(scalar, vector) instructions x 16 cycles -- and
   the first version of this loop, like the matrix fill before it, had 50 scalar instructions per row (branches on where a
   predecessor row lives, flag tests, loop conditions) against 30 vector ones.  Hence the row kinds of the rank metadata pass
   (CW_RM_WORD: straight-line code per kind, predecessor offsets computed on the vector unit for the whole list at once, no
   prefetch logic) and the branch-free tail. */
template <int CM>
__device__ __forceinline__ int poa_fill_c(const PoaMem<int16_t>& M, const int n, const int cols, const int lane, const bool lc) {
    const int G4 = 4 * CW_POA_GAP, MS4 = 4 * (CW_POA_MATCH - CW_POA_GAP), XS4 = 4 * (CW_POA_MISMATCH - CW_POA_GAP);
    const int NEGD = -(1 << 24);
    const int L = cols - 1;
    const int sq = (lane > 0 && lane < cols) ? (int)M.sq[lane - 1] : -1;
    const int xs_l = lane == 0 ? NEGD : XS4; /* column 0 has no diagonal */
    const cwc_l16 rowst = (cwc_l16)M.ring + lane;    /* this lane's column of the row ring */
    const cwc_g16 Hg = (cwc_g16)M.H + lane;          /* flagged rows in the slab, stride 64 */
    const cwc_g32 Cg = (cwc_g32)M.codes + lane;
    /* small alignments keep their code words in LDS, row stride = columns: a lane beyond the columns writes into the NEXT group's first
       cells, which that group's own flush overwrites later (one wave's LDS writes are ordered; 64 words of slack behind the last group) */
    const cwc_l32 Cl = (cwc_l32)M.lcodes + lane;
    int rc0 = 3;                                     /* row i-1 (row 0, the virtual start, is all zero in this form) */
    int bs = (int)0x80000000, bi = 0, bj = L; /* (bj: the overlap mode's end column, cw_policy.h CW_POA_MODE_OV) */
#ifdef CW_DIAG /* rows by kind (0, 1, 2-3, generic), in-edges and slab loads of the generic rows, rows with a flag, all rows */
    uint32_t dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    /* the substitution score of a row without a compare and a select (and the two wait states between them): bit b + 4k of a lane's word says
       "my base is b", so bit (row word & 31) is the answer whatever the flag bits above the base are */
    const uint32_t onehot8 = sq >= 0 ? 0x11111111u << sq : 0u;
#define CWC_SCORE(m) ((int)(__builtin_amdgcn_ubfe(onehot8, (m), 1u) * (uint32_t)(MS4 - XS4)) + xs_l)
#define CWC_FLUSH(last_row) do { if (lc) Cl[((last_row) >> 3) * cols] = acc; else cwc_gstore_b32(Cg + ((last_row) >> 3) * 64, acc); acc = 0u; sh = 0; } while (0)
    uint32_t acc = 0u; /* the code word of the current group of eight rows */
    int sh = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
        /* the row words of the next 64 rows, one per lane: a row costs a v_readlane, not an LDS round trip */
        const uint32_t meta_v = (r0 + lane < n) ? M.rmeta[r0 + lane] : 0u;
        const int cnt = __builtin_amdgcn_readfirstlane(min(64, n - r0));
        /* plain rows: one in-edge from the row before, no flag -- nearly half of all rows; runs of them go through a loop of their own */
        unsigned long long plain = __ballot((meta_v & 0x1Cu) == 0x04u);
        uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)meta_v, 0);
        int s_l = CWC_SCORE(meta);
        int rl = 0;
        while (rl < cnt) {
            if (plain & 1ull) {
                do {
                    const int i = r0 + rl + 1;
                    const int kD = cw_wave_shr1(rc0, 0) + s_l, kV = rc0 + G4;
                    const int v = CW_POA_OV && lane == 0 ? 3 : max(kD, kV); /* (overlap mode: column 0 is free) */
                    ++rl; plain >>= 1;
                    meta = (uint32_t)__builtin_amdgcn_readlane((int)meta_v, rl);
                    s_l = CWC_SCORE(meta);
                    const int nv = cw_wave_scan_max(v) | 3;
                    /* in-edge 0 throughout, the candidates carry the cell's low bits: 0 if the diagonal candidate is the cell, else 4 if the vertical one is, else 8
                       (differences, not compares: no condition register between them, no wait states) */
                    acc |= min(min((uint32_t)(nv - kD) << 1, (uint32_t)(nv - kV) + 4u), 8u) << sh;
                    rowst[(i & (CW_RING - 1)) * 64] = (int16_t)nv;
                    rc0 = nv;
                    sh += 4;
                    if (sh == 32) CWC_FLUSH(i - 1);
#ifdef CW_DIAG
                    dg[0]++; dg[7]++;
#endif
                } while (plain & 1ull); /* (rows beyond the graph have no row word: their bits are clear) */
                continue;
            }
            const int i = r0 + rl + 1;
            int kD, kV;
#ifdef CW_DIAG
            {
                const uint32_t kd_ = CW_RM_LIN(meta) ? 0u : (meta >> 5) & 7u;
                dg[kd_ == 0u ? 0 : kd_ == 1u ? 1 : kd_ <= 3u ? 2 : 3]++; dg[7]++;
                if (meta & 24u) dg[6]++;
                if (kd_ > 3u) { const int np_ = CW_RM_NP(meta); dg[4] += (uint32_t)np_; for (int q = 0; q < np_; ++q) { const int pr_ = np_ == 1 ? CW_RM_X(meta) : (int)M.plist[CW_RM_X(meta) + q]; if (pr_ != 0 && i - pr_ > CW_RING) dg[5]++; } }
            }
#endif
            if (CW_RM_LIN(meta)) {
                kD = cw_wave_shr1(rc0, 0) + s_l; kV = rc0 + G4;
            } else {
                const uint32_t kind = (meta >> 5) & 7u;
                const int x = CW_RM_X(meta);
                if (kind == 1u) {
                    const int up = (int)rowst[(x & (CW_RING - 1)) * 64];
                    kD = cw_wave_shr1(up, 0) + s_l; kV = up + G4;
                } else if (kind <= 3u) {
                    /* two or three in-edges, every predecessor in the ring: entry q of the list in lane q, its ring offset computed there too */
                    const int po = ((int)M.plist[x + lane] & (CW_RING - 1)) * 64;
                    const int up0 = (int)rowst[__builtin_amdgcn_readlane(po, 0)];
                    const int up1 = (int)rowst[__builtin_amdgcn_readlane(po, 1)] - 1;
                    int dm = max(cw_wave_shr1(up0, 0), cw_wave_shr1(up1, 0)), vm = max(up0, up1);
                    if (kind == 3u) {
                        const int up2 = (int)rowst[__builtin_amdgcn_readlane(po, 2)] - 2;
                        dm = max(dm, cw_wave_shr1(up2, 0)); vm = max(vm, up2);
                    }
                    kD = dm + s_l; kV = vm + G4;
                } else { /* the general row: any number of in-edges, rows from anywhere (the virtual start, the slab) */
                    const int np = CW_RM_NP(meta);
                    int dm = NEGD, vm = NEGD;
                    for (int q = 0; q < np; ++q) {
                        const int prow = np == 1 ? x : __builtin_amdgcn_readfirstlane((int)M.plist[x + q]);
                        int up;
                        if (prow == 0) up = 3;
                        else if (i - prow <= CW_RING) up = (int)rowst[(prow & (CW_RING - 1)) * 64];
                        else up = cwc_gload_i16_wait(Hg + prow * 64);
                        up -= q < 3 ? q : 3;
                        dm = max(dm, cw_wave_shr1(up, 0)); vm = max(vm, up);
                    }
                    kD = dm + s_l; kV = vm + G4;
                }
            }
            const int v = CW_POA_OV && lane == 0 ? 3 : max(kD, kV);
            const uint32_t meta_c = meta;
            /* the next row's word and match scores: independent of the scan, they fill the wait states between its DPP steps */
            ++rl; plain >>= 1;
            meta = (uint32_t)__builtin_amdgcn_readlane((int)meta_v, rl); /* (lane 64 = lane 0: not used) */
            s_l = CWC_SCORE(meta);
#if defined(CW_EXP_SALU) || defined(CW_EXP_VALU) /* experiment (tools/exp_issue.sh): what one more scalar / vector instruction per row costs */
            {
                int xs_ = rl, xv_ = lane;
#ifdef CW_EXP_SALU
#pragma unroll
                for (int e_ = 0; e_ < CW_EXP_SALU; ++e_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(xs_));
#endif
#ifdef CW_EXP_VALU
#pragma unroll
                for (int e_ = 0; e_ < CW_EXP_VALU; ++e_) asm volatile("v_add_u32 %0, %0, 1" : "+v"(xv_));
#endif
                asm volatile("" :: "s"(xs_), "v"(xv_));
            }
#endif
            const int nv = cw_wave_scan_max(v) | 3;
            /* the code: in-edge q (low bits of candidate ^ cell, the cell's being 3) for a diagonal, 4 + q for a vertical, 8 for a horizontal move */
            const uint32_t t = (uint32_t)(kD ^ nv), u4 = __builtin_elementwise_add_sat((uint32_t)(kV ^ nv), 4u); /* saturating: candidate and cell may differ in sign */
            acc |= min(min(t < 4u ? t : 8u, u4), 8u) << sh;
            rowst[(i & (CW_RING - 1)) * 64] = (int16_t)nv;
            if (meta_c & 24u) { /* a sink (the end cell is the best of them), a row some later row or the traceback reads from the slab */
                if (meta_c & 16u) cwc_gstore_b16(Hg + i * 64, nv);
                if (CW_RM_SINK(meta_c)) {
                    if (CW_POA_OV) { /* the best cell of the row, columns 1..L, in H form; lowest column on ties */
                        const int hv = (lane >= 1 && lane <= L) ? (nv >> 2) + lane * CW_POA_GAP : (int)0x80000000;
                        const int hm = cw_wave_max(hv);
                        if (hm > bs) { bs = hm; bi = i; bj = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)__ballot(hv == hm)) - 1); }
                    } else {
                        const int h = __builtin_amdgcn_readlane(nv, L);
                        if (h > bs) { bs = h; bi = i; } /* ranks ascend: the lowest rank keeps a tie */
                    }
                }
            }
            rc0 = nv;
            sh += 4;
            if (sh == 32) CWC_FLUSH(i - 1);
        }
    }
    if (sh) CWC_FLUSH(n - 1);
#undef CWC_SCORE
#undef CWC_FLUSH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cw_wave_sync();
#ifdef CW_DIAG
    if (lane == 0) { for (int k = 0; k < 7; ++k) atomicAdd(&M.diag[k], (unsigned long long)dg[k]); atomicAdd(&M.diag[10], (unsigned long long)dg[7]); if (cols <= 32) atomicAdd(&M.diag[11], (unsigned long long)dg[7]); }
#endif
    return bi | (bj << 16);
}

/* Follows the recorded codes from (bi, L) to the virtual start; writes seqrank[j] = rank aligned to sequence position j (diagonal moves).
   Returns false when the walk does not end (cannot happen; reported as an internal error). */
template <int CM>
__device__ __forceinline__ bool poa_trace_c(const PoaMem<int16_t>& M, const int n, const int bi, const int bj, const int cols, const int lane, const bool lc) {
    const int G4 = 4 * CW_POA_GAP, MS4 = 4 * (CW_POA_MATCH - CW_POA_GAP), XS4 = 4 * (CW_POA_MISMATCH - CW_POA_GAP);
    const int L = cols - 1;
    const int tr = lane >> 3, tc = lane & 7;
    const int cs = 64;
    const cwc_g32 Cg = (cwc_g32)M.codes;
    int i = bi, j = bj, trips = 0;
    while (i > 0) {
        if (++trips > n + cols + 4) return false;
        i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);
        int row = i; /* tr steps up the first-predecessor chain: 4 + 2 + 1 -- three unconditional reads and selects (as `if (tr & 4) ...` each
                        level was an execution-mask region of its own: ~45 instructions for what is 15) */
        { uint32_t v = M.p4[row - 1]; asm volatile("" : "+v"(v)); row = (tr & 4) ? (v == CW_NONE16 ? -1 : (int)v) : row; }
        { uint32_t v = M.p2[max(row, 1) - 1]; asm volatile("" : "+v"(v)); row = (tr & 2) ? ((row > 0 && v != CW_NONE16) ? (int)v : -1) : row; }
        { uint32_t v = M.rpred0[max(row, 1) - 1]; asm volatile("" : "+v"(v)); row = (tr & 1) ? (row > 0 ? (int)v : -1) : row; }
        const int col = j - tc;
        uint32_t nib = 0u;
        if (row >= 1 && col >= 0) {
            const uint32_t cw = lc ? ((cwc_l32)M.lcodes)[((row - 1) >> 3) * cols + col] : Cg[((row - 1) >> 3) * cs + col];
            nib = (cw >> (((row - 1) & 7) * 4)) & 15u;
        }
        const int mv = (int)(nib >> 2), qn = (int)(nib & 3u);
        /* 0 diagonal, 1 vertical (both through in-edge 0: the next tile row), 2 horizontal; 3 through another in-edge, 5 neighbours outside
           the tile, 6 the virtual start row */
        const bool c_start = row <= 0 || (CW_POA_OV && col <= 0), c_edge = tr == 7 || (tc == 7 && col > 0) || col < 0; /* (overlap mode: the walk stops in column 0) */
        const int code = c_start ? 6 : c_edge ? 5 : mv == 2 ? 2 : qn != 0 ? 3 : mv;
        const unsigned long long m_d = __ballot(code == 0), m_v = __ballot(code == 1), m_h = __ballot(code == 2);
        int pos = 0;
        unsigned long long on_diag = 0ull;
        for (;;) { /* a whole diagonal run per trip, see cw_poa.h */
            const unsigned long long dm = 0x8040201008040201ull << pos;
            const unsigned long long stop = ~m_d & dm;
            const int first = stop ? __ffsll((long long)stop) - 1 : 63;
            on_diag |= dm & ((1ull << first) - 1ull);
            pos = first;
            if ((m_v >> pos) & 1ull) pos += 8;
            else if ((m_h >> pos) & 1ull) pos += 1;
            else break;
        }
        if ((on_diag >> lane) & 1ull) M.seqrank[col - 1] = (uint16_t)(row - 1);
        const int end_code = __builtin_amdgcn_readlane(code, pos);
        i = __builtin_amdgcn_readlane(row, pos);
        j -= pos & 7;
        if (end_code == 6) i = 0;
        else if (end_code == 3) { /* the move goes through in-edge q > 0 */
            const int mv_e = __builtin_amdgcn_readlane(mv, pos), q_e = __builtin_amdgcn_readlane(qn, pos);
            const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.rmeta[i - 1]);
            const int np = CW_RM_NP(meta), off = CW_RM_X(meta);
            int q = q_e;
            if (q_e == 3) { /* fourth in-edge or later: the first of them whose candidate equals the cell (values of these rows are kept) */
                const int base = (int)(meta & 3u);
                const int h = (int)((cwc_g16)M.H)[i * cs + j];
                const int sx = (j > 0 && (int)M.sq[j - 1] == base) ? MS4 : XS4;
                q = -1;
                for (int t = 3; t < np && q < 0; ++t) {
                    const int pr = (int)M.plist[off + t];
                    const int cj = mv_e == 0 ? j - 1 : j;
                    const int pv = (int)((cwc_g16)M.H)[pr * cs + cj];
                    if (h == pv + (mv_e == 0 ? sx : G4)) q = t; /* both values carry the low bits 3 */
                }
                q = __builtin_amdgcn_readfirstlane(q);
                if (q < 0) return false;
            }
            const int pr = np == 1 ? off : __builtin_amdgcn_readfirstlane((int)M.plist[off + q]);
            if (mv_e == 0) { if (lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1); j--; }
            i = pr;
        }
    }
    return true;
}

#endif
