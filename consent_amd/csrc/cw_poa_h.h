/*
 * cw_poa_h.h -- tier H of the partial-order alignment (A4d): TWO tasks per wavefront, one per 32-lane half.
 *
 * What tier Q (cw_poa_q.h) does for the smallest tasks, for the middle of the range: members of 32..63 bases against graphs of up
 * to 128 nodes -- at depth 150 four in five of the tasks that used to go to tier M1 (median longest member 45) -- fill a 64-lane
 * wave's rows to two thirds and spend a whole instruction stream on one alignment.  The POA stage is bound by instruction issue
 * (measured, DESIGN.md: a narrower tier M1 with 1.6x the resident waves ran not a millisecond faster), so the way to more
 * alignments per second is fewer instructions per alignment: here a task owns a 32-lane half -- two DP columns per lane in packed
 * int16 (64 columns), a five-step prefix max (four row_shr + one row_bcast:15, which never leaves the half), the two tasks of a wave
 * in lock step under their own predicates.
 *
 * Memory: the DP matrix lives in the task's global slab (stride 64; the fill reads it back only for a predecessor more than three
 * rows up) together with the merge-only graph arrays (as tiers M1/M2/L); LDS holds the hot graph arrays and, instead of the
 * matrix, DIRECTION WORDS: per row and column the move the traceback will take out of that cell -- diagonal / vertical / horizontal
 * and, for a node with up to four predecessors, WHICH predecessor (decided during the fill, while every predecessor row is in
 * registers, in the order of preference of cw_policy.h: diagonal through the in-edges in order, then vertical through them, then
 * horizontal).  32 bytes per row; the traceback is then a walk over LDS words, one or two round trips per step and no matrix read
 * at all (tier M1 reads 8x8 tiles of the matrix back from L2: a third of its cycles).  A node with more than four predecessors is
 * decided from the cell values as everywhere else.
 *
 * Same policy, same arithmetic, same results as poa_run (cw_poa.h): the tests compare every tier with the oracle.  A task that
 * outgrows a capacity is handed to tier L.
 */
#ifndef CW_POA_H_H
#define CW_POA_H_H

#include "cw_poa.h"

#define CW_POAH_NC 128
#define CW_POAH_EC 384
#define CW_POAH_LC 63
#define CW_POAH_HS 64 /* row stride of the DP matrix: columns 0..63 */
#define CW_POAH_DIR_BYTES (CW_POAH_NC * 32) /* one direction byte per row and lane (two columns) */
#define CW_POAH_TASK_LDS ((CW_POAH_DIR_BYTES + CW_POA_HOT2_BYTES(CW_POAH_NC, CW_POAH_EC, CW_POAH_LC) + 15) / 16 * 16)
#define CW_POAH_SLAB_BYTES (CW_POA_HSLAB_BYTES(CW_POAH_NC, CW_POAH_LC) + CW_POA_COLD2_BYTES(CW_POAH_NC, CW_POAH_EC, CW_POAH_LC))
#define CW_POAH_WAVES 2        /* four tasks per work-group */
#define CW_POAH_ROUTE_NODES 104 /* tasks whose graph is expected to stay below this many nodes come here (1.7 x longest member) */
#define CW_POAH_MIN_LEN 32     /* shorter members: tiers Q and S */

#if defined(CW_TEST_AIDS) && CW_CONS_HEAVIEST_BUNDLE
#error "tier H (test-aid build) has no heaviest-bundle consensus: build it with the column vote"
#endif
#ifdef CW_TEST_AIDS /* the kernel itself exists in the test-aid build only (measured slower in the mix: DESIGN.md); the constants above are the chain kernel's */

/* ---- 32-lane half primitives ----------------------------------------------------------------------------------------------- */
__device__ __forceinline__ unsigned h_ballot(bool p) { return (unsigned)(__ballot(p) >> (threadIdx.x & 32u)); }
__device__ __forceinline__ int h_bcast(int v, int src) { return __shfl(v, (int)(threadIdx.x & 32u) + src); }
__device__ __forceinline__ int h_scan_add(int v) {
    v += CW_DPP(0, v, 0x111, 0xF);
    v += CW_DPP(0, v, 0x112, 0xF);
    v += CW_DPP(0, v, 0x114, 0xF);
    v += CW_DPP(0, v, 0x118, 0xF);
    v += CW_DPP(0, v, 0x142, 0xA); /* lane 15 of the half's first row into its second row */
    return v;
}
__device__ __forceinline__ unsigned h_scan_max_u32(unsigned v) {
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x111, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x112, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x114, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x118, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x142, 0xA));
    return v;
}
/* lane gl receives v of lane gl - 1 of its half; lane 0 of the half receives `fill` (wave_shr:1 would hand it the other half's last lane) */
__device__ __forceinline__ int h_shr1(int v, int fill, int gl) {
    const int s = CW_DPP(fill, v, 0x138, 0xF);
    return gl == 0 ? fill : s;
}

typedef __attribute__((address_space(1))) int* cw_gint;
typedef __attribute__((address_space(1))) const int16_t* cw_gs16;

/* Row metadata of tier H (its own encoding, built in poah_run): base | in-edges << 2 (13 bits) | linear << 15 | x << 16, where
   linear = the node's only predecessor is the rank before it, and x = the predecessor's DP row when there is one in-edge, else the
   offset of the node's predecessor list -- the fill reads ONE word per row and branches on one bit. */
#define CW_HM_NP(m) (int)(((m) >> 2) & 0x1FFFu)
#define CW_HM_LIN(m) (((m) >> 15) & 1u)
#define CW_HM_X(m) (int)((m) >> 16)
__device__ __forceinline__ int pk_shl(int a, int n) { return __builtin_bit_cast(int, (cw_s2)(__builtin_bit_cast(cw_s2, a) << (cw_s2)(short)n)); }
__device__ __forceinline__ int pk_sar(int a, int n) { return __builtin_bit_cast(int, (cw_s2)(__builtin_bit_cast(cw_s2, a) >> (cw_s2)(short)n)); }

/* Packed DP fill of one member against the graph for BOTH halves of the wave (cf. poaq_fill, poa_fill_pk<1>): lane gl of a half owns its
   task's columns 2gl and 2gl + 1.  Control is the wave's, not the lanes': one scalar loop over the rows of the longer graph, one scalar
   branch per row ("is every row of this step linear?"), scalar loop bounds for the predecessors; what differs between the two tasks is
   data (selects and predicates).  Written per lane with divergent loops the same fill cost 250 instructions per row, most of them the
   bookkeeping of execution masks.  Writes the matrix rows to the slab and, per row and lane, one byte of directions to LDS:
   move | ordinal << 2 for the even column in the low nibble, for the odd column in the high nibble. */
__device__ __forceinline__ void poah_fill(const PoaMem<int16_t>& M, uint8_t* dirs, const int n, const int cols, const int gl) {
    const int G = CW_POA_GAP;
    const int GPK = pk_make(G, G);
    cw_gint Hw = (cw_gint)(int*)M.H;
    const int j0 = 2 * gl, j1 = j0 + 1;
    const int jg = pk_make(j0 * G, j1 * G);
    int rc0 = jg, rc1 = jg, rc2 = jg; /* the last three rows, rc0 = the previous one */
    const bool on = j0 < cols;
    const int amask = (j0 < cols ? 0xFFFF : 0) | (j1 < cols ? (int)0xFFFF0000 : 0);
    const int q0 = (j0 >= 1 && j0 < cols) ? (int)M.sq[j0 - 1] : -1, q1 = (j1 < cols) ? (int)M.sq[j1 - 1] : -1;
    const int qpk = (q0 >= 0 ? 1 << q0 : 0) | (q1 >= 0 ? 1 << (16 + q1) : 0);
    if (on) Hw[j0 >> 1] = jg; /* row 0, the virtual start: read back by a source node more than three ranks down */
    /* (scalar loop bounds come from ballots, never from readlane: the other half may be masked off -- its task ended earlier -- and a masked
       lane's register holds whatever was there) */
    uint32_t meta_n = n > 0 ? M.rmeta[0] : 0u;
    for (int r = 0; __ballot(r < n) != 0ull; ++r) {
        const int i = r + 1;
        const uint32_t meta = meta_n;
        const bool act = r < n;
        meta_n = (r + 1 < n) ? M.rmeta[r + 1] : 0u;
        const int base = (int)(meta & 3u), np = CW_HM_NP(meta), hx = CW_HM_X(meta);
        const int srow = pk_score(qpk, base);
        int v, ce, co, qe = 0, qo = 0;
        if (__ballot(act && !CW_HM_LIN(meta)) == 0ull) {
            /* every row of this step hangs off the row before it: no choice of a source row, no list, no loop */
            const int up = rc0;
            const int sh = h_shr1(up, CW_NEGPK, gl);                    /* lane gl-1's pair; column 0 has no left neighbour */
            const int dgv = pk_add(__builtin_amdgcn_alignbit(up, sh, 16), srow), upv = pk_add(up, GPK);
            v = pk_max(dgv, upv);
            int w = pk_sub(v, jg);
            w = (w & amask) | (CW_NEGPK & ~amask);
            w = pk_max(w, (w << 16) | 0x8AD0);                          /* odd column sees the even one of its lane */
            const unsigned inc = h_scan_max_u32(((unsigned)w >> 16) ^ 0x8000u);
            const unsigned ex = (unsigned)h_shr1((int)inc, 0, gl);
            w = pk_max(w, pk_splat_lo((int)(ex ^ 0x8000u)));
            const int nv = pk_add(w, jg);
            rc2 = rc1; rc1 = rc0; rc0 = nv;
            if (act && on) Hw[(i * CW_POAH_HS + j0) >> 1] = nv;
            const unsigned xd = (unsigned)(nv ^ dgv), xv = (unsigned)(nv ^ upv);
            ce = ((xd & 0xFFFFu) == 0u && j0 > 0) ? 0 : (xv & 0xFFFFu) == 0u ? 1 : 2;
            co = (xd >> 16) == 0u ? 0 : (xv >> 16) == 0u ? 1 : 2;
        } else {
            /* the general step.  The first in-edge whose diagonal (then: vertical) explains the cell is found without a second pass: the
               cell's value is at least every candidate, so a candidate explains it iff it is the largest and equal to it -- keep the
               largest diagonal and the largest vertical candidate with the ordinal in the two low bits (value * 4 + 3 - q: on equal
               values the earlier in-edge wins).  More than four in-edges: the traceback decides from the cell values (code 3). */
            v = CW_NEGPK;
            int kd = CW_NEGPK, kv = CW_NEGPK;
            for (int q = 0; __ballot(act && q < np) != 0ull; ++q) {
                const bool qa = act && q < np;
                const int prow = np == 1 ? hx : (qa ? (int)M.plist[hx + q] : i);
                const int dist = i - prow;
                int up = dist == 1 ? rc0 : dist == 2 ? rc1 : rc2;
                const bool far = qa && dist > 3;
                if (__ballot(far) != 0ull) {
                    if (far && on) up = Hw[(prow * CW_POAH_HS + j0) >> 1];
                    /* the loaded row is "used" here, inside the branch: otherwise the wait for it (s_waitcnt vmcnt(0), which also counts the
                       stores of the row before) lands where the branches meet, i.e. in every step (cw_poa.h, poa_fill_pk) */
                    asm volatile("" : "+v"(up));
                }
                up = qa ? up : CW_NEGPK;
                const int sh = h_shr1(up, CW_NEGPK, gl);
                const int dgv = pk_add(__builtin_amdgcn_alignbit(up, sh, 16), srow), upv = pk_add(up, GPK);
                v = pk_max(v, pk_max(dgv, upv));
                if (q < 4) {
                    const int tag = pk_make(3 - q, 3 - q);
                    kd = pk_max(kd, qa ? (pk_shl(dgv, 2) | tag) : CW_NEGPK);
                    kv = pk_max(kv, qa ? (pk_shl(upv, 2) | tag) : CW_NEGPK);
                }
            }
            int w = pk_sub(v, jg);
            w = (w & amask) | (CW_NEGPK & ~amask);
            w = pk_max(w, (w << 16) | 0x8AD0);
            const unsigned inc = h_scan_max_u32(((unsigned)w >> 16) ^ 0x8000u);
            const unsigned ex = (unsigned)h_shr1((int)inc, 0, gl);
            w = pk_max(w, pk_splat_lo((int)(ex ^ 0x8000u)));
            const int nv = pk_add(w, jg);
            rc2 = rc1; rc1 = rc0; rc0 = nv;
            if (act && on) Hw[(i * CW_POAH_HS + j0) >> 1] = nv;
            const unsigned xd = (unsigned)(nv ^ pk_sar(kd, 2)), xv = (unsigned)(nv ^ pk_sar(kv, 2));
            const bool de = (xd & 0xFFFFu) == 0u && j0 > 0, dd = (xd >> 16) == 0u, ve = (xv & 0xFFFFu) == 0u, vo = (xv >> 16) == 0u;
            ce = de ? 0 : ve ? 1 : 2;
            co = dd ? 0 : vo ? 1 : 2;
            qe = de ? 3 - (kd & 3) : ve ? 3 - (kv & 3) : 0;
            qo = dd ? 3 - ((kd >> 16) & 3) : vo ? 3 - ((kv >> 16) & 3) : 0;
            if (np > 4) { ce = co = 3; qe = qo = 0; }
        }
        if (act) dirs[r * 32 + gl] = (uint8_t)(ce | (qe << 2) | (co << 4) | (qo << 6));
    }
}

/* Returns 1 = done, 2 = a capacity of this tier was exceeded, 3 = output capacity exceeded / internal.  Every value below is per lane
   and equal inside the 32-lane half; `gl` = lane inside the half. */
__device__ int poah_run(const PoaMem<int16_t>& M, uint8_t* dirs, const PoaTask& t, const DevBatch& b, const DevScratch& sc, const int gl,
                        unsigned long long (&acc)[5]) {
    unsigned long long _pt = __builtin_readcyclecounter();
#define POAH_PROF(slot) do { const unsigned long long _n = __builtin_readcyclecounter(); acc[slot] += _n - _pt; _pt = _n; } while (0)
    const int G = CW_POA_GAP, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH, HS = CW_POAH_HS;
    cw_gs16 Hg = (cw_gs16)M.H;
    int n = 0, ne = 0, nseq = 0, tpl_nodes = 0;
    bool meta_ok = false;
    const unsigned lt_mask = (1u << gl) - 1u;

    for (uint32_t mi = 0; mi < t.n_members; ++mi) {
        const PoaMember pm = sc.members[t.member_off + mi];
        const int L = (int)pm.len;
        if ((uint32_t)L > M.l_cap) return 2;
        {
            const uint32_t* words = b.bases + b.seq_word_off[pm.seq];
            for (int j = gl; j < L; j += 32) M.sq[j] = (uint8_t)cw_base_at(words, pm.start + j);
        }
        cw_wave_sync();
        nseq++;
        if (n == 0) { /* first member: a chain */
            if ((uint32_t)L > M.n_cap || (uint32_t)L > M.e_cap) return 2;
            for (int j = gl; j < L; j += 32) {
                M.nbase[j] = M.sq[j]; M.ncov[j] = 1; M.nalc[j] = 0;
                M.in_head[j] = j ? (uint16_t)(j - 1) : CW_NONE16; M.in_tail[j] = M.in_head[j];
                M.indeg[j] = j ? 1 : 0; M.has_out[j] = (j < L - 1) ? 1 : 0;
                M.r2n[j] = (uint16_t)j; M.n2r[j] = (uint16_t)j;
                if (j) { M.efrom[j - 1] = (uint16_t)(j - 1); M.enext[j - 1] = CW_NONE16; }
            }
            n = L; ne = L - 1; tpl_nodes = L; meta_ok = false;
            cw_wave_sync();
            continue;
        }
        const int cols = L + 1;

        /* ---- per-rank metadata ---- */
        if (!meta_ok) {
            int run = 0;
            for (int r0 = 0; r0 < n; r0 += 32) {
                const int r = r0 + gl;
                const int node = r < n ? M.r2n[r] : 0;
                const int d = r < n ? M.indeg[node] : 0;
                const int inc = h_scan_add(d);
                const int off = run + inc - d;
                if (r < n) {
                    int q = off, first = 0;
                    for (uint32_t e = M.in_head[node]; e != CW_NONE16; e = M.enext[e]) {
                        const int pr = M.n2r[M.efrom[e]] + 1;
                        if (q == off) first = pr;
                        M.plist[q++] = (uint16_t)pr;
                    }
                    const uint32_t np_ = (uint32_t)(d ? d : 1);
                    M.rmeta[r] = (uint32_t)M.nbase[node] | (np_ << 2) | ((np_ == 1u && first == r) ? 0x8000u : 0u) | ((uint32_t)(np_ == 1u ? first : off) << 16);
                }
                run += h_bcast(inc, 31);
            }
            meta_ok = true;
            cw_wave_sync();
        }
        POAH_PROF(0);

        /* ---- DP fill ---- */
        for (int j = gl; j < L; j += 32) M.seqrank[j] = CW_NONE16;
        cw_wave_sync();
        poah_fill(M, dirs, n, cols, gl);
        cw_wave_sync();
        POAH_PROF(1);

        /* ---- end cell: best sink in the last column, lowest rank on ties ---- */
        int bi;
        {
            int bs = CW_NEG * 2, br = 0x7FFFFFFF;
            for (int r = gl; r < n; r += 32) {
                if (M.has_out[M.r2n[r]]) continue;
                const int h = Hg[(r + 1) * HS + L];
                if (h > bs) { bs = h; br = r; } /* ranks ascend within a lane */
            }
            for (int o = 16; o > 0; o >>= 1) {
                const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o);
                if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; }
            }
            bi = br + 1;
        }

        /* ---- traceback over the direction words: every lane of the half walks the same path ---- */
        {
            int i = bi, j = L;
            int steps_left = n + L + 2; /* a path has at most nodes + bases moves: anything longer is an internal error, never a hang */
            while (i > 0) {
                if (--steps_left < 0 || j < 0) return 3;
                const uint32_t db = dirs[(i - 1) * 32 + (j >> 1)];
                const uint32_t meta = M.rmeta[i - 1];
                const int nib = (int)((j & 1) ? db >> 4 : db & 15u);
                const int code = nib & 3, qv = nib >> 2;
                const int np = CW_HM_NP(meta), off = CW_HM_X(meta);
                int pi = i, pj = j;
                if (code == 2) {
                    pj = j - 1;
                } else if (code != 3) {
                    pi = np == 1 ? off : (int)M.plist[off + qv];
                    if (code == 0) pj = j - 1;
                } else { /* more in-edges than the ordinal holds: decided from the cell values, same order of preference */
                    const int base = (int)(meta & 3u);
                    const int h = Hg[i * HS + j];
                    const int sx = (j != 0 && (int)M.sq[j - 1] == base) ? MS : XS;
                    bool found = false;
                    if (j != 0) {
                        for (int q = 0; q < np && !found; ++q) {
                            const int pr = (int)M.plist[off + q];
                            if (h == (int)Hg[pr * HS + j - 1] + sx) { pi = pr; pj = j - 1; found = true; }
                        }
                    }
                    for (int q = 0; q < np && !found; ++q) {
                        const int pr = (int)M.plist[off + q];
                        if (h == (int)Hg[pr * HS + j] + G) { pi = pr; found = true; }
                    }
                    if (!found) {
                        if (j != 0 && h == (int)Hg[i * HS + j - 1] + G) pj = j - 1;
                        else return 3;
                    }
                }
                if (pj != j && pi != i && gl == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                if (pi == i && pj == j) return 3; /* cannot happen: every code moves */
                i = pi; j = pj;
            }
        }
        cw_wave_sync();
        POAH_PROF(2);

        /* ---- merge the path into the graph: one lane per sequence position, 32 at a time (cf. poa_run) ---- */
        {
            const int n_old = n;
            const int chunks = (L + 31) >> 5;
            int next_rank = -1;
            for (int c = chunks - 1; c >= 0; --c) {
                const int j = c * 32 + gl;
                const bool act = j < L;
                const uint32_t rk = act ? M.seqrank[j] : CW_NONE16;
                const unsigned has = h_ballot(act && rk != CW_NONE16);
                const unsigned later = has & ~(lt_mask | (1u << gl));
                const int later_rank = (int)(uint32_t)h_bcast((int)rk, later ? (__ffs((int)later) - 1) : 0);
                const int qr = later ? later_rank : next_rank;
                const int first_rank = (int)(uint32_t)h_bcast((int)rk, has ? (__ffs((int)has) - 1) : 0);
                uint32_t cur = CW_NONE16, at = CW_NONE16;
                if (act) {
                    const int bcode = M.sq[j];
                    if (rk != CW_NONE16) {
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
                            if (cur == CW_NONE16) at = (uint32_t)(last + 1);
                        }
                    } else if (qr < 0) {
                        at = (uint32_t)n_old;
                    } else {
                        const int q = M.r2n[qr];
                        int first = qr;
                        for (int a = 0; a < M.nalc[q]; ++a) first = min(first, (int)M.n2r[M.nal[q * 3 + a]]);
                        at = (uint32_t)first;
                    }
                    M.pcur[j] = (uint16_t)cur;
                    M.pat[j] = (uint16_t)at;
                }
                if (has) next_rank = first_rank;
            }
            cw_wave_sync();
            int fresh_total = 0;
            for (int c = 0; c < chunks; ++c) {
                const int j = c * 32 + gl;
                const bool act = j < L;
                const bool fresh = act && M.pcur[j] == CW_NONE16;
                const unsigned fb = h_ballot(fresh);
                if (fresh) {
                    const int cur = n_old + fresh_total + __popc(fb & lt_mask);
                    if ((uint32_t)cur < M.n_cap) {
                        M.pcur[j] = (uint16_t)cur;
                        M.nbase[cur] = M.sq[j]; M.ncov[cur] = 1; M.nalc[cur] = 0;
                        M.in_head[cur] = CW_NONE16; M.in_tail[cur] = CW_NONE16; M.indeg[cur] = 0; M.has_out[cur] = 0;
                        const uint32_t rk = M.seqrank[j];
                        if (rk != CW_NONE16) {
                            const int pn = M.r2n[rk];
                            const int ac = M.nalc[pn];
                            for (int a = 0; a < ac; ++a) {
                                const int v = M.nal[pn * 3 + a];
                                M.nal[cur * 3 + a] = (uint16_t)v;
                                M.nal[v * 3 + M.nalc[v]] = (uint16_t)cur; M.nalc[v] = (uint8_t)(M.nalc[v] + 1);
                            }
                            M.nal[cur * 3 + ac] = (uint16_t)pn; M.nalc[cur] = (uint8_t)(ac + 1);
                            M.nal[pn * 3 + ac] = (uint16_t)cur; M.nalc[pn] = (uint8_t)(ac + 1);
                        }
                    }
                } else if (act) {
                    const int cur = M.pcur[j];
                    M.ncov[cur] = (uint16_t)(M.ncov[cur] + 1);
                }
                fresh_total += __popc(fb);
            }
            if ((uint32_t)(n_old + fresh_total) > M.n_cap) return 2;
            cw_wave_sync();
            if (fresh_total > 0) {
                uint32_t* hist = (uint32_t*)M.plist; /* n_old + 1 counters (EC * 2 bytes >= 4 * (NC + 1)) */
                for (int r = gl; r <= n_old; r += 32) hist[r] = 0;
                cw_wave_sync();
                for (int c = 0; c < chunks; ++c) {
                    const int j = c * 32 + gl;
                    if (j < L && M.pat[j] != CW_NONE16) atomicAdd(&hist[M.pat[j]], 1u);
                }
                cw_wave_sync();
                int run = 0;
                for (int r0 = 0; r0 < n_old; r0 += 32) {
                    const int r = r0 + gl;
                    const int hcount = r < n_old ? (int)hist[r] : 0;
                    const int inc = h_scan_add(hcount);
                    if (r < n_old) {
                        const int nr = r + run + inc;
                        const int v = M.r2n[r];
                        M.rtmp[nr] = (uint16_t)v;
                        M.n2r[v] = (uint16_t)nr;
                    }
                    run += h_bcast(inc, 31);
                }
                for (int c = 0; c < chunks; ++c) {
                    const int j = c * 32 + gl;
                    if (j < L && M.pat[j] != CW_NONE16) {
                        const int cur = M.pcur[j];
                        const int nr = (int)M.pat[j] + (cur - n_old);
                        M.rtmp[nr] = (uint16_t)cur;
                        M.n2r[cur] = (uint16_t)nr;
                    }
                }
                cw_wave_sync();
                n = n_old + fresh_total;
                for (int r = gl; r < n; r += 32) M.r2n[r] = M.rtmp[r];
                meta_ok = false;
                cw_wave_sync();
            }
            for (int c = 0; c < chunks; ++c) {
                const int j = c * 32 + gl;
                const bool act = j < L && j > 0;
                int head = 0, cur = 0;
                bool add = false;
                if (act) {
                    head = M.pcur[j - 1]; cur = M.pcur[j];
                    add = true;
                    for (uint32_t e = M.in_head[cur]; e != CW_NONE16; e = M.enext[e])
                        if (M.efrom[e] == (uint16_t)head) { add = false; break; }
                }
                const unsigned ab = h_ballot(add);
                const int total = __popc(ab);
                if ((uint32_t)(ne + total) > M.e_cap) return 2;
                if (add) {
                    const int e = ne + __popc(ab & lt_mask);
                    M.efrom[e] = (uint16_t)head; M.enext[e] = CW_NONE16;
                    const uint32_t tl = M.in_tail[cur];
                    if (tl == CW_NONE16) M.in_head[cur] = (uint16_t)e; else M.enext[tl] = (uint16_t)e;
                    M.in_tail[cur] = (uint16_t)e;
                    M.indeg[cur] = (uint16_t)(M.indeg[cur] + 1);
                    M.has_out[head] = 1;
                }
                if (total) { ne += total; meta_ok = false; }
            }
            cw_wave_sync();
        }
        POAH_PROF(3);
    }

    /* ---- column-majority consensus ---- */
    uint32_t out_len = 0;
    for (int r0 = 0; r0 < n; r0 += 32) {
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
        const unsigned bal = h_ballot(emit >= 0);
        const uint32_t idx = out_len + (uint32_t)__popc(bal & lt_mask);
        if (emit >= 0 && idx < t.out_cap) sc.arena[t.out_off + idx] = "ACGT"[emit];
        out_len += (uint32_t)__popc(bal);
    }
    if (out_len > t.out_cap) return 3;
    if (gl == 0) sc.seg_len[t.seg_slot] = out_len;
    POAH_PROF(4);
#undef POAH_PROF
    return 1;
}

/* ---- tier H: two tasks per wave ------------------------------------------------------------------------------------------- */
#define CW_TIER_H 5
__global__ void __launch_bounds__(64 * CW_POAH_WAVES) cw_poa_h_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int gl = threadIdx.x & 31;
    const uint32_t grp = threadIdx.x >> 5; /* 0 .. 2 * waves - 1 */
    /* every half claims a slab (matrix + merge-only graph arrays); there are more slabs than halves the hardware can hold at once */
    uint32_t gw = 0;
    if (gl == 0) {
        const uint32_t n_slots = sc.slots[CW_TIER_H];
        uint32_t s = (uint32_t)(((unsigned long long)(blockIdx.x * 2u * CW_POAH_WAVES + grp) * 2654435761ull) % n_slots);
        for (;;) {
            if (__hip_atomic_load(&sc.slot_busy[CW_TIER_H][s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
                atomicCAS(&sc.slot_busy[CW_TIER_H][s], 0u, 1u) == 0u) break;
            s = s + 1u == n_slots ? 0u : s + 1u;
        }
        gw = s;
    }
    gw = (uint32_t)h_bcast((int)gw, 0);
    typedef __attribute__((address_space(1))) uint8_t* cw_gptr;
    uint8_t* my_slab = (uint8_t*)(cw_gptr)(sc.slab[CW_TIER_H] + (size_t)gw * sc.slab_bytes[CW_TIER_H]);
    uint8_t* my_lds = lds + (size_t)grp * CW_POAH_TASK_LDS;
    uint8_t* dirs = my_lds;
    PoaMem<int16_t> M = poa_carve<int16_t>(my_lds + CW_POAH_DIR_BYTES, CW_POAH_NC, CW_POAH_EC, CW_POAH_LC, (CW_POAH_NC + 1) * CW_POAH_HS, 0, (int16_t*)my_slab,
                                           (unsigned long long*)dirs, my_slab + CW_POA_HSLAB_BYTES(CW_POAH_NC, CW_POAH_LC), true, false);
    const uint32_t* list = sc.tier_list[CW_TIER_H];
    const uint32_t n_work = min(sc.ctr->n_tier[CW_TIER_H], sc.list_cap);
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
    for (;;) {
        uint32_t mi = 0;
        if (gl == 0) mi = atomicAdd(&sc.ctr->next_tier[CW_TIER_H], 1u);
        mi = (uint32_t)h_bcast((int)mi, 0);
        if (mi >= n_work) break;
        const uint32_t ti = list[mi];
        const PoaTask t = sc.tasks[ti];
        if (t.n_members == 0) continue; /* a neutral entry (cw_chain.h "cap_ok") */
        const int rc = poah_run(M, dirs, t, b, sc, gl, acc);
        if (gl == 0) poa_hand_over(sc, t, ti, rc, 3); /* rc 2: redone in tier L, which takes it from its live queue */
        cw_wave_sync();
    }
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 5; ++q) atomicAdd(&sc.ctr->prof[64 + q], acc[q]); /* as the first half of every wave saw them */
    if (gl == 0) __hip_atomic_store(&sc.slot_busy[CW_TIER_H][gw], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); /* the slab goes back */
    poa_producer_done(sc);
}

#endif /* CW_TEST_AIDS */

#endif
