/*
 * cw_poa.h -- partial-order alignment of one segment pile per wavefront (A4d).
 *
 * One 64-lane wave owns one task from the list the index kernel emitted.  Same code, three memory tiers:
 *   S (cw_poa_kernel)      graph + DP matrix in LDS            small segments (most tasks)
 *   M (cw_poa_mid_kernel)  graph in LDS, matrix in an L2-resident per-wave slab   long segments
 *   G (cw_poa_big_kernel)  everything in a per-wave global slab                   the rare huge graph
 * A task that outgrows its tier is handed to the next one and redone there from scratch.
 *
 *   DP fill     lanes = sequence positions (columns); rows = graph nodes in rank order; the horizontal
 *               gap recurrence H[i][j] = max(H[i][j], H[i][j-1]+g) is a wave-level inclusive prefix-max
 *               of H[i][j] - j*g (linear gaps), so a row costs one pass regardless of its length.
 *   traceback   wave-uniform walk, preference order of cw_policy.h (diagonal, vertical, horizontal).
 *   graph edit  wave-uniform; rank order maintained by insertion (cw_policy.h "rank order"), the shift of
 *               the rank arrays is done by all lanes.
 *   consensus   column-majority vote, one lane per rank, ordered compaction by ballot.
 * All arithmetic is integer; int16 cells in LDS (|score| <= 8*(nodes+len) < 2^15 under the LDS caps), int32
 * cells in the global slab.
 */
#ifndef CW_POA_H
#define CW_POA_H

#include "cw_device.h"

/* tier S: graph and DP matrix in LDS (per wave) */
#define CW_POA_NC 160   /* nodes            */
#define CW_POA_EC 448   /* edges            */
#define CW_POA_LC 255   /* member length    */
#define CW_POA_HC 4096  /* DP cells (int16) */
/* tier M: graph in LDS, DP matrix (int16) in a per-wave global slab that stays L2-resident */
#define CW_POAM_NC 512
#define CW_POAM_EC 1280
#define CW_POAM_LC 511
#define CW_POAM_HC ((CW_POAM_NC + 1) * (CW_POAM_LC + 1))
#define CW_POAM_WAVES 2
/* tier G: everything in a per-wave global slab (int32 cells) */
#define CW_POAB_NC 2048
#define CW_POAB_EC 8192
#define CW_POAB_LC 1023
#define CW_POAB_HC ((CW_POAB_NC + 1) * (CW_POAB_LC + 1))

#define CW_POA_WAVES 4

template <typename HT>
struct PoaMem {
    HT* H;
    uint8_t* nbase;     /* node -> base code                           */
    uint16_t* ncov;     /* node -> sequences through it                */
    uint8_t* nalc;      /* node -> number of aligned nodes (0..3)      */
    uint16_t* nal;      /* node -> 3 aligned node ids                  */
    uint16_t* in_head;  /* node -> first in-edge, CW_NONE16 if none    */
    uint16_t* in_tail;
    uint16_t* indeg;
    uint8_t* has_out;
    uint16_t* efrom;    /* edge -> source node                         */
    uint16_t* enext;    /* edge -> next in-edge of the same target     */
    uint16_t* r2n;      /* rank -> node                                */
    uint16_t* n2r;
    uint16_t* poff;     /* rank -> first entry of its predecessor rows */
    uint16_t* plist;    /* predecessor DP rows (rank+1), in-edge order */
    uint16_t* pnode;    /* traceback path (reversed)                   */
    uint16_t* pseq;
    uint8_t* sq;        /* current member, base codes                  */
    uint32_t n_cap, e_cap, l_cap, h_cap;
};

template <typename HT>
__device__ __forceinline__ size_t poa_mem_bytes(uint32_t nc, uint32_t ec, uint32_t lc, uint32_t hc) {
    size_t b = (size_t)hc * sizeof(HT);
    b += nc * (1 + 2 + 1 + 6 + 2 + 2 + 2 + 1 + 2 + 2) + ec * 6 + 2 * (nc + 1) + 4 * (nc + lc + 2) + (lc + 1);
    return (b + 15) & ~(size_t)15;
}

template <typename HT>
__device__ __forceinline__ PoaMem<HT> poa_carve(uint8_t* base, uint32_t nc, uint32_t ec, uint32_t lc, uint32_t hc, HT* h_ext = nullptr) {
    PoaMem<HT> M;
    uint8_t* p = base;
    if (h_ext) M.H = h_ext;
    else { M.H = (HT*)p; p += (size_t)hc * sizeof(HT); }
    M.ncov = (uint16_t*)p; p += 2 * nc;
    M.nal = (uint16_t*)p; p += 6 * nc;
    M.in_head = (uint16_t*)p; p += 2 * nc;
    M.in_tail = (uint16_t*)p; p += 2 * nc;
    M.indeg = (uint16_t*)p; p += 2 * nc;
    M.r2n = (uint16_t*)p; p += 2 * nc;
    M.n2r = (uint16_t*)p; p += 2 * nc;
    M.efrom = (uint16_t*)p; p += 2 * ec;
    M.enext = (uint16_t*)p; p += 2 * ec;
    M.plist = (uint16_t*)p; p += 2 * ec;
    M.poff = (uint16_t*)p; p += 2 * (nc + 1);
    M.pnode = (uint16_t*)p; p += 2 * (nc + lc + 2);
    M.pseq = (uint16_t*)p; p += 2 * (nc + lc + 2);
    M.nbase = p; p += nc;
    M.nalc = p; p += nc;
    M.has_out = p; p += nc;
    M.sq = p; p += lc + 1;
    M.n_cap = nc; M.e_cap = ec; M.l_cap = lc; M.h_cap = hc;
    return M;
}

/* Returns 1 = done, 2 = a capacity of this memory class was exceeded, 3 = output capacity exceeded. */
template <typename HT>
__device__ int poa_run(const PoaMem<HT>& M, const PoaTask& t, const DevBatch& b, const DevScratch& sc, const int lane,
                       unsigned long long (&acc)[6]) {
    unsigned long long _pt = __builtin_readcyclecounter();
#define POA_PROF(slot) do { const unsigned long long _n = __builtin_readcyclecounter(); acc[slot] += _n - _pt; _pt = _n; } while (0)
    const int G = CW_POA_GAP, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH;
    int n = 0, ne = 0, nseq = 0, tpl_nodes = 0;
    bool csr_ok = false;

    for (uint32_t mi = 0; mi < t.n_members; ++mi) {
        const PoaMember pm = sc.members[t.member_off + mi];
        const int L = pm.len;
        if ((uint32_t)L > M.l_cap) return 2;
        {
            const uint32_t* words = b.bases + b.seq_word_off[pm.seq];
            for (int j = lane; j < L; j += 64) M.sq[j] = (uint8_t)cw_base_at(words, pm.start + j);
        }
        cw_wave_sync();
        nseq++;
        if (n == 0) { /* first member: a chain */
            if ((uint32_t)L > M.n_cap || (uint32_t)L > M.e_cap) return 2;
            for (int j = lane; j < L; j += 64) {
                M.nbase[j] = M.sq[j]; M.ncov[j] = 1; M.nalc[j] = 0;
                M.in_head[j] = j ? (uint16_t)(j - 1) : CW_NONE16; M.in_tail[j] = M.in_head[j];
                M.indeg[j] = j ? 1 : 0; M.has_out[j] = (j < L - 1) ? 1 : 0;
                M.r2n[j] = (uint16_t)j; M.n2r[j] = (uint16_t)j;
                if (j) { M.efrom[j - 1] = (uint16_t)(j - 1); M.enext[j - 1] = CW_NONE16; }
            }
            n = L; ne = L - 1; tpl_nodes = L; csr_ok = false;
            cw_wave_sync();
            continue;
        }
        const int cols = L + 1;
        if ((uint32_t)((n + 1) * cols) > M.h_cap) return 2;

        /* ---- predecessor rows in CSR form (parallel over ranks) ---- */
        if (!csr_ok) {
            int run = 0;
            for (int r0 = 0; r0 < n; r0 += 64) {
                const int r = r0 + lane;
                const int node = r < n ? M.r2n[r] : 0;
                const int d = r < n ? M.indeg[node] : 0;
                int inc = d;
                for (int o = 1; o < 64; o <<= 1) { int x = __shfl_up(inc, o); if (lane >= o) inc += x; }
                const int off = run + inc - d;
                if (r < n) {
                    M.poff[r] = (uint16_t)off;
                    int q = off;
                    for (uint32_t e = M.in_head[node]; e != CW_NONE16; e = M.enext[e]) M.plist[q++] = (uint16_t)(M.n2r[M.efrom[e]] + 1);
                }
                run += __shfl(inc, 63);
            }
            if (lane == 0) M.poff[n] = (uint16_t)run;
            csr_ok = true;
            cw_wave_sync();
        }

        POA_PROF(0);
        /* ---- DP fill ---- */
        for (int j = lane; j < cols; j += 64) M.H[j] = (HT)(j * G);
        cw_wave_sync();
        if (cols <= 128) {
            /* rows of <= 2 chunks: the previous row stays in registers, so a chain of nodes never waits on memory */
            const bool two = cols > 64;
            const int j0 = lane, j1 = 64 + lane;
            const bool act0 = j0 < cols, act1 = two && j1 < cols;
            const int s0q = (j0 > 0 && act0) ? (int)M.sq[j0 - 1] : -1, s1q = act1 ? (int)M.sq[j1 - 1] : -1;
            int prev0 = j0 * G, prev1 = j1 * G; /* row 0 */
            for (int r = 0; r < n; ++r) {
                const int i = r + 1;
                const int base = M.nbase[M.r2n[r]];
                const int p0 = M.poff[r], p1 = M.poff[r + 1];
                const int sc0 = (s0q == base) ? MS : XS, sc1 = (s1q == base) ? MS : XS;
                int v0 = CW_NEG, v1 = CW_NEG;
                const int np = (p0 == p1) ? 1 : p1 - p0;
                for (int q = 0; q < np; ++q) {
                    const int prow = (p0 == p1) ? 0 : (int)M.plist[p0 + q];
                    int up0, up1, dg0, dg1;
                    if (prow == i - 1) {
                        up0 = prev0; up1 = prev1;
                        dg0 = cw_wave_shr1(prev0, CW_NEG);
                        dg1 = cw_wave_shr1(prev1, cw_lane_value(prev0, 63));
                    } else {
                        cw_wave_sync(); /* rows written by other lanes of this wave must have landed */
                        const int pr = prow * cols;
                        up0 = act0 ? (int)M.H[pr + j0] : CW_NEG;
                        dg0 = (act0 && j0 > 0) ? (int)M.H[pr + j0 - 1] : CW_NEG;
                        up1 = act1 ? (int)M.H[pr + j1] : CW_NEG;
                        dg1 = act1 ? (int)M.H[pr + j1 - 1] : CW_NEG;
                    }
                    v0 = max(v0, max(dg0 + sc0, up0 + G));
                    v1 = max(v1, max(dg1 + sc1, up1 + G));
                }
                int w0 = act0 ? v0 - j0 * G : CW_NEG;
                w0 = cw_wave_scan_max(w0, lane);
                prev0 = w0 + j0 * G;
                if (act0) M.H[i * cols + j0] = (HT)prev0;
                if (two) {
                    int w1 = act1 ? v1 - j1 * G : CW_NEG;
                    w1 = cw_wave_scan_max(w1, lane);
                    w1 = max(w1, cw_lane_value(w0, 63));
                    prev1 = w1 + j1 * G;
                    if (act1) M.H[i * cols + j1] = (HT)prev1;
                }
            }
            cw_wave_sync();
        } else
        for (int r = 0; r < n; ++r) {
            const int i = r + 1;
            const int base = M.nbase[M.r2n[r]];
            const int p0 = M.poff[r], p1 = M.poff[r + 1];
            int carry = CW_NEG;
            for (int c0 = 0; c0 < cols; c0 += 64) {
                const int j = c0 + lane;
                const bool act = j < cols;
                int v = CW_NEG;
                if (act) {
                    const int s = (j > 0 && M.sq[j - 1] == base) ? MS : XS;
                    if (p0 == p1) {
                        const int up = M.H[j];
                        const int dg = j > 0 ? (int)M.H[j - 1] : CW_NEG;
                        v = max(dg + s, up + G);
                    } else {
                        for (int q = p0; q < p1; ++q) {
                            const int pr = M.plist[q] * cols;
                            const int up = M.H[pr + j];
                            const int dg = j > 0 ? (int)M.H[pr + j - 1] : CW_NEG;
                            v = max(v, max(dg + s, up + G));
                        }
                    }
                }
                int wv = act ? v - j * G : CW_NEG;
                wv = cw_wave_scan_max(wv, lane);
                wv = max(wv, carry);
                carry = cw_lane_value(wv, 63);
                if (act) M.H[i * cols + j] = (HT)(wv + j * G);
            }
            cw_wave_sync();
        }

        POA_PROF(1);
        /* ---- end cell: best sink in the last column, lowest rank on ties ---- */
        int bi;
        {
            int bs = CW_NEG * 2, br = 0x7FFFFFFF;
            for (int r = lane; r < n; r += 64) {
                if (M.has_out[M.r2n[r]]) continue;
                const int h = M.H[(r + 1) * cols + L];
                if (h > bs) { bs = h; br = r; } /* ranks ascend within a lane */
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o);
                if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; }
            }
            bi = br + 1;
        }

        /* ---- traceback (wave-uniform) ---- */
        int plen = 0;
        {
            int i = bi, j = L;
            while (!(i == 0 && j == 0)) {
                const int h = M.H[i * cols + j];
                int pi = i, pj = j;
                bool found = false;
                int node = 0, p0 = 0, p1 = 0;
                if (i != 0) { node = M.r2n[i - 1]; p0 = M.poff[i - 1]; p1 = M.poff[i]; }
                if (i != 0 && j != 0) {
                    const int s = (M.sq[j - 1] == M.nbase[node]) ? MS : XS;
                    if (p0 == p1) { if (h == (int)M.H[j - 1] + s) { pi = 0; pj = j - 1; found = true; } }
                    else for (int q = p0; q < p1 && !found; ++q) {
                        const int pr = M.plist[q];
                        if (h == (int)M.H[pr * cols + j - 1] + s) { pi = pr; pj = j - 1; found = true; }
                    }
                }
                if (!found && i != 0) {
                    if (p0 == p1) { if (h == (int)M.H[j] + G) { pi = 0; pj = j; found = true; } }
                    else for (int q = p0; q < p1 && !found; ++q) {
                        const int pr = M.plist[q];
                        if (h == (int)M.H[pr * cols + j] + G) { pi = pr; pj = j; found = true; }
                    }
                }
                if (!found && j != 0) {
                    if (h == (int)M.H[i * cols + j - 1] + G) { pi = i; pj = j - 1; found = true; }
                }
                if (!found) return 3; /* cannot happen: the matrix is self-consistent */
                if (lane == 0) {
                    M.pnode[plen] = (i == pi) ? CW_NONE16 : (uint16_t)node;
                    M.pseq[plen] = (j == pj) ? CW_NONE16 : (uint16_t)(j - 1);
                }
                plen++;
                i = pi; j = pj;
            }
        }
        cw_wave_sync();
        POA_PROF(2);

        /* ---- merge the path into the graph (wave-uniform, rank shifts by all lanes) ---- */
        {
            int head = -1;
            int q_node = -1, q_idx = 0x7FFFFFFF; /* cached look-ahead of the next ranked path node */
            for (int tix = plen - 1; tix >= 0; --tix) {
                const uint32_t ps = M.pseq[tix];
                if (ps == CW_NONE16) continue;
                const uint32_t pn = M.pnode[tix];
                const int bcode = M.sq[ps];
                int cur = -1;
                bool fresh = false;
                int at = 0;
                if (pn == CW_NONE16) {
                    if (tix <= q_idx || q_idx == 0x7FFFFFFF) {
                        q_node = -1; q_idx = -1;
                        for (int u = tix - 1; u >= 0; --u)
                            if (M.pseq[u] != CW_NONE16 && M.pnode[u] != CW_NONE16) { q_node = M.pnode[u]; q_idx = u; break; }
                    }
                    if (q_node < 0) at = n;
                    else {
                        at = M.n2r[q_node];
                        for (int a = 0; a < M.nalc[q_node]; ++a) at = min(at, (int)M.n2r[M.nal[q_node * 3 + a]]);
                    }
                    fresh = true;
                } else if (M.nbase[pn] == bcode) {
                    cur = (int)pn;
                } else {
                    const int ac = M.nalc[pn];
                    for (int a = 0; a < ac; ++a) {
                        const int v = M.nal[pn * 3 + a];
                        if (M.nbase[v] == bcode) { cur = v; break; }
                    }
                    if (cur < 0) {
                        at = M.n2r[pn];
                        for (int a = 0; a < ac; ++a) at = max(at, (int)M.n2r[M.nal[pn * 3 + a]]);
                        at += 1;
                        fresh = true;
                    }
                }
                if (!fresh) {
                    if (lane == 0) M.ncov[cur] = (uint16_t)(M.ncov[cur] + 1);
                } else {
                    if ((uint32_t)n >= M.n_cap) return 2;
                    cur = n;
                    /* shift ranks [at, n) up by one, highest chunk first */
                    for (int hi = n; hi > at; hi -= 64) {
                        const int r = hi - 1 - lane;
                        uint16_t v = 0;
                        if (r >= at) v = M.r2n[r];
                        cw_wave_sync();
                        if (r >= at) { M.r2n[r + 1] = v; M.n2r[v] = (uint16_t)(r + 1); }
                        cw_wave_sync();
                    }
                    if (lane == 0) {
                        M.r2n[at] = (uint16_t)cur; M.n2r[cur] = (uint16_t)at;
                        M.nbase[cur] = (uint8_t)bcode; M.ncov[cur] = 1; M.nalc[cur] = 0;
                        M.in_head[cur] = CW_NONE16; M.in_tail[cur] = CW_NONE16; M.indeg[cur] = 0; M.has_out[cur] = 0;
                        if (pn != CW_NONE16) { /* joins pn's column */
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
                    n++;
                    csr_ok = false;
                    cw_wave_sync();
                }
                if (head >= 0) {
                    bool exists = false;
                    for (uint32_t e = M.in_head[cur]; e != CW_NONE16; e = M.enext[e])
                        if (M.efrom[e] == (uint16_t)head) { exists = true; break; }
                    if (!exists) {
                        if ((uint32_t)ne >= M.e_cap) return 2;
                        if (lane == 0) {
                            M.efrom[ne] = (uint16_t)head; M.enext[ne] = CW_NONE16;
                            const uint32_t tl = M.in_tail[cur];
                            if (tl == CW_NONE16) M.in_head[cur] = (uint16_t)ne; else M.enext[tl] = (uint16_t)ne;
                            M.in_tail[cur] = (uint16_t)ne;
                            M.indeg[cur] = (uint16_t)(M.indeg[cur] + 1);
                            M.has_out[head] = 1;
                        }
                        ne++;
                        csr_ok = false;
                        cw_wave_sync();
                    }
                }
                head = cur;
            }
        }
        cw_wave_sync();
        POA_PROF(3);
    }

    /* ---- column-majority consensus ---- */
    uint32_t out_len = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
        const int r = r0 + lane;
        int emit = -1;
        if (r < n) {
            const int v = M.r2n[r];
            const int ac = M.nalc[v];
            bool first = true;
            for (int a = 0; a < ac; ++a) if (M.n2r[M.nal[v * 3 + a]] < r) first = false;
            if (first) {
                int cnt[4] = {0, 0, 0, 0};
                int tpl_code = -1;
                for (int c = 0; c <= ac; ++c) {
                    const int u = M.r2n[r + c];
                    const int code = M.nbase[u];
                    cnt[code] += M.ncov[u];
                    if (u < tpl_nodes) tpl_code = code;
                }
                const int gaps = nseq - (cnt[0] + cnt[1] + cnt[2] + cnt[3]);
                int top = 0;
                for (int c = 1; c < 4; ++c) if (cnt[c] > cnt[top]) top = c;
                if (!(gaps > cnt[top])) {
                    if (tpl_code != -1 && cnt[tpl_code] == cnt[top]) top = tpl_code;
                    emit = top;
                }
            }
        }
        const unsigned long long bal = __ballot(emit >= 0);
        const uint32_t idx = out_len + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (emit >= 0) {
            if (idx < t.out_cap) sc.arena[t.out_off + idx] = "ACGT"[emit];
        }
        out_len += (uint32_t)__popcll(bal);
    }
    if (out_len > t.out_cap) return 3;
    if (lane == 0) sc.seg_len[t.seg_slot] = out_len;
    POA_PROF(4);
#undef POA_PROF
    return 1;
}

/* ---- tier S: one task per wave, graph + DP matrix in LDS, work-stealing over the task list -------- */
#define CW_POA_GRAPH_BYTES(NC, EC, LC) ((NC) * 21 + (EC) * 6 + 2 * ((NC) + 1) + 4 * ((NC) + (LC) + 2) + ((LC) + 1))
#define CW_POA_SLAB_BYTES ((CW_POA_HC * 2 + CW_POA_GRAPH_BYTES(CW_POA_NC, CW_POA_EC, CW_POA_LC) + 15) / 16 * 16)
#define CW_POAM_SLAB_BYTES ((CW_POA_GRAPH_BYTES(CW_POAM_NC, CW_POAM_EC, CW_POAM_LC) + 15) / 16 * 16)

__device__ __forceinline__ void poa_flush_prof(const DevScratch& sc, int base, const unsigned long long (&acc)[6], int lane) {
    if (lane == 0) for (int q = 0; q < 5; ++q) atomicAdd(&sc.ctr->prof[base + q], acc[q]);
}

__global__ void __launch_bounds__(64 * CW_POA_WAVES) cw_poa_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PoaMem<int16_t> M = poa_carve<int16_t>(lds + (size_t)wave * CW_POA_SLAB_BYTES, CW_POA_NC, CW_POA_EC, CW_POA_LC, CW_POA_HC);
    const uint32_t n_tasks = min(sc.ctr->n_tasks, sc.task_cap);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (;;) {
        uint32_t ti = 0;
        if (lane == 0) ti = atomicAdd(&sc.ctr->next_task, 1u);
        ti = (uint32_t)__shfl((int)ti, 0);
        if (ti >= n_tasks) break;
        const PoaTask t = sc.tasks[ti];
        if (t.state != 0) continue; /* routed to a larger tier by the index kernel */
        const int rc = poa_run<int16_t>(M, t, b, sc, lane, acc);
        if (lane == 0) {
            if (rc == 2) {
                const uint32_t bi = atomicAdd(&sc.ctr->n_mid, 1u);
                if (bi < sc.big_cap) sc.mid_list[bi] = ti;
                else { sc.win[t.window].status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            } else if (rc == 3) {
                sc.win[t.window].status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1;
            }
            sc.tasks[ti].state = (uint32_t)rc;
        }
        cw_wave_sync();
    }
    poa_flush_prof(sc, 8, acc, lane);
}

/* ---- tier M: graph in LDS, DP matrix in this wave's global slab --------------------------------- */
__global__ void __launch_bounds__(64 * CW_POAM_WAVES) cw_poa_mid_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * CW_POAM_WAVES + wave;
    if (gw >= sc.mid_slots) return;
    int16_t* hslab = (int16_t*)(sc.mid_scratch + (size_t)gw * sc.mid_slab_bytes);
    const PoaMem<int16_t> M = poa_carve<int16_t>(lds + (size_t)wave * CW_POAM_SLAB_BYTES, CW_POAM_NC, CW_POAM_EC, CW_POAM_LC, CW_POAM_HC, hslab);
    const uint32_t n_mid = min(sc.ctr->n_mid, sc.big_cap);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (;;) {
        uint32_t mi = 0;
        if (lane == 0) mi = atomicAdd(&sc.ctr->next_mid, 1u);
        mi = (uint32_t)__shfl((int)mi, 0);
        if (mi >= n_mid) break;
        const uint32_t ti = sc.mid_list[mi];
        const PoaTask t = sc.tasks[ti];
        const int rc = poa_run<int16_t>(M, t, b, sc, lane, acc);
        if (lane == 0) {
            if (rc == 2) {
                const uint32_t bi = atomicAdd(&sc.ctr->n_big, 1u);
                if (bi < sc.big_cap) sc.big_list[bi] = ti;
                else { sc.win[t.window].status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            } else if (rc == 3) {
                sc.win[t.window].status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1;
            }
            sc.tasks[ti].state = (uint32_t)rc;
        }
        cw_wave_sync();
    }
    poa_flush_prof(sc, 14, acc, lane);
}

/* ---- tier G: everything in this wave's global slab (int32 cells) -------------------------------- */
__global__ void __launch_bounds__(64 * CW_POA_WAVES) cw_poa_big_kernel(DevBatch b, DevScratch sc) {
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * CW_POA_WAVES + (threadIdx.x >> 6);
    if (gw >= sc.big_slots) return;
    const PoaMem<int32_t> M = poa_carve<int32_t>(sc.big_scratch + (size_t)gw * sc.big_slab_bytes, CW_POAB_NC, CW_POAB_EC, CW_POAB_LC, CW_POAB_HC);
    const uint32_t n_big = min(sc.ctr->n_big, sc.big_cap);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (;;) {
        uint32_t bi = 0;
        if (lane == 0) bi = atomicAdd(&sc.ctr->next_big, 1u);
        bi = (uint32_t)__shfl((int)bi, 0);
        if (bi >= n_big) break;
        const uint32_t ti = sc.big_list[bi];
        const PoaTask t = sc.tasks[ti];
        const int rc = poa_run<int32_t>(M, t, b, sc, lane, acc);
        if (lane == 0) {
            if (rc != 1) { sc.win[t.window].status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            sc.tasks[ti].state = (uint32_t)(rc == 1 ? 1 : 3);
        }
        cw_wave_sync();
    }
    poa_flush_prof(sc, 19, acc, lane);
}

#endif
