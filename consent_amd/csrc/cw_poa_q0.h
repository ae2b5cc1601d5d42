/*
 * cw_poa_q0.h -- tier Q as rounds 3-4 ran it (-DCW_Q_CODES=0): the DP matrix of every task in LDS (40 nodes / 120 edges / 31 bases), a step-wise
 * traceback over its values.  Kept as a build variant of cw_poa_q.h (same results; tests/test_gpu_variants.py), included from there.
 */
#ifndef CW_POA_Q0_H
#define CW_POA_Q0_H

#define CW_POAQ_NC 40
#define CW_POAQ_EC 120
#define CW_POAQ_LC 31
#define CW_POAQ_HS 32 /* row stride of the DP matrix: columns 0..31 */
#define CW_POAQ_HC ((CW_POAQ_NC + 1) * CW_POAQ_HS)
#define CW_POAQ_TASK_BYTES ((CW_POAQ_HC * 2 + CW_POA_GRAPH_BYTES(CW_POAQ_NC, CW_POAQ_EC, CW_POAQ_LC) + 15) / 16 * 16)
#define CW_POAQ_WAVES 8 /* 32 tasks per work-group: 150 KB of LDS, one work-group per CU (measured: 5 waves of 64-node slabs 26 ms, 7 x 48 13 ms, 8 x 40 8.7 ms, 10 x 32 5.2 ms for the tasks they take; the step is best here) */
#define CW_POAQ_ROUTE_NODES 35 /* tasks expected to stay below this many nodes come here */
#define CW_POAQ_SLAB_BYTES 64 /* (no slab in this variant) */
/* (no tier H in this variant: cw_poa_q.h) */
#define CW_POAH_LC 63
#define CW_POAH_ROUTE_NODES 112
#define CW_POAH_MIN_LEN 1
#define CW_POAH_TASK_BYTES 64
#define CW_POAH_SLAB_BYTES 64
#define CW_POAH_WAVES 1

/* packed DP fill of one member against the graph: lane gl owns columns 2gl and 2gl + 1 (cf. poa_fill_pk<1>) */
__device__ __forceinline__ void poaq_fill(const PoaMem<int16_t>& M, const int n, const int cols, const int gl) {
    const int G = CW_POA_GAP;
    const int GPK = pk_make(G, G);
    int* Hw = (int*)M.H;
    const int j0 = 2 * gl, j1 = j0 + 1;
    const int jg = pk_make(j0 * G, j1 * G);
    int rc0 = CW_POA_SW ? 0 : jg, rc1 = rc0, rc2 = rc0; /* the last three rows, rc0 = the previous one (row 0: j * gap, or 0 in the local mode) */
    const int amask = (j0 < cols ? 0xFFFF : 0) | (j1 < cols ? (int)0xFFFF0000 : 0);
    const int q0 = (j0 >= 1 && j0 < cols) ? (int)M.sq[j0 - 1] : -1, q1 = (j1 < cols) ? (int)M.sq[j1 - 1] : -1;
    const int qpk = (q0 >= 0 ? 1 << q0 : 0) | (q1 >= 0 ? 1 << (16 + q1) : 0);
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        const uint32_t meta = M.rmeta[r];
        const int pr0 = (int)M.rpred0[r];
        const int base = (int)(meta & 3u), np = (int)((meta >> 2) & 0x3FFFu), off = (int)(meta >> 16);
        const int srow = pk_score(qpk, base);
        int v = CW_NEGPK;
        for (int q = 0; q < np; ++q) {
            const int prow = (np == 1) ? pr0 : (int)M.plist[off + q];
            const int dist = i - prow;
            int up;
            if (dist <= 3) up = dist == 1 ? rc0 : dist == 2 ? rc1 : rc2;
            else up = (j0 < cols) ? Hw[(prow * CW_POAQ_HS + j0) >> 1] : CW_NEGPK;
            const int sh = CW_DPP(CW_NEGPK, up, 0x111, 0xF);            /* lane l-1's pair inside the row; column 0 has no left neighbour */
            const int dg = __builtin_amdgcn_alignbit(up, sh, 16);       /* (col 2l-1, col 2l) of the row above */
            v = pk_max(v, pk_max(pk_add(dg, srow), pk_add(up, GPK)));
        }
        if (CW_POA_OV && gl == 0) v = (int)((unsigned)v & 0xFFFF0000u); /* overlap mode (cw_policy.h): column 0 -- this lane's even half -- is free */
        if (CW_POA_SW) v = pk_max(v, 0); /* local mode: no cell below 0 */
        int w = pk_sub(v, jg);
        w = (w & amask) | (CW_NEGPK & ~amask);
        w = pk_max(w, (w << 16) | 0x8AD0);                              /* odd column sees the even one of its lane */
        const unsigned inc = q_scan_max_u32(((unsigned)w >> 16) ^ 0x8000u);
        const unsigned ex = (unsigned)CW_DPP(0, (int)inc, 0x111, 0xF);
        w = pk_max(w, pk_splat_lo((int)(ex ^ 0x8000u)));
        const int nv = pk_add(w, jg);
        rc2 = rc1; rc1 = rc0; rc0 = nv;
        if (j0 < cols) Hw[(i * CW_POAQ_HS + j0) >> 1] = nv;
    }
}

/* Returns 1 = done, 2 = a capacity of this tier was exceeded, 3 = output capacity exceeded / internal.  Every value below is
   per lane and equal inside the 16-lane row; `gl` = lane inside the row. */
__device__ int poaq_run(const PoaMem<int16_t>& M, const PoaTask& t, const DevBatch& b, const DevScratch& sc, const int gl, unsigned long long (&acc)[5]) {
    unsigned long long _pt = __builtin_readcyclecounter();
#define POAQ_PROF(slot) do { const unsigned long long _n = __builtin_readcyclecounter(); acc[slot] += _n - _pt; _pt = _n; } while (0)
    const int G = CW_POA_GAP, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH, HS = CW_POAQ_HS;
    int n = 0, ne = 0, nseq = 0, tpl_nodes = 0;
    bool meta_ok = false;
    const unsigned lt_mask = (1u << gl) - 1u;

    for (uint32_t mi = 0; mi < t.n_members; ++mi) {
        const PoaMember pm = sc.members[t.member_off + mi];
        const int L = (int)pm.len;
        if ((uint32_t)L > M.l_cap) return 2;
        {
            const uint32_t* words = b.bases + b.seq_word_off[pm.seq];
            for (int j = gl; j < L; j += 16) M.sq[j] = (uint8_t)cw_base_at(words, pm.start + j);
        }
        cw_wave_sync();
        nseq++;
        if (n == 0) { /* first member: a chain */
            if ((uint32_t)L > M.n_cap || (uint32_t)L > M.e_cap) return 2;
            for (int j = gl; j < L; j += 16) {
                M.nbase[j] = M.sq[j]; M.ncov[j] = 1; M.nalc[j] = 0;
                M.in_head[j] = j ? (uint16_t)(j - 1) : CW_NONE16; M.in_tail[j] = M.in_head[j];
                M.indeg[j] = j ? 1 : 0; M.has_out[j] = (j < L - 1) ? 1 : 0;
                M.r2n[j] = (uint16_t)j; M.n2r[j] = (uint16_t)j;
                if (j) { M.efrom[j - 1] = (uint16_t)(j - 1); M.enext[j - 1] = CW_NONE16; if (CW_CONS_HEAVIEST_BUNDLE) M.ew[j - 1] = 1; }
            }
            n = L; ne = L - 1; tpl_nodes = L; meta_ok = false;
            cw_wave_sync();
            continue;
        }
        const int cols = L + 1;
        if ((uint32_t)((n + 1) * HS) > M.h_cap) return 2;

        /* ---- per-rank metadata ---- */
        if (!meta_ok) {
            int run = 0;
            for (int r0 = 0; r0 < n; r0 += 16) {
                const int r = r0 + gl;
                const int node = r < n ? M.r2n[r] : 0;
                const int d = r < n ? M.indeg[node] : 0;
                const int inc = q_scan_add(d);
                const int off = run + inc - d;
                if (r < n) {
                    int q = off, first = 0;
                    for (uint32_t e = M.in_head[node]; e != CW_NONE16; e = M.enext[e]) {
                        const int pr = M.n2r[M.efrom[e]] + 1;
                        if (q == off) first = pr;
                        M.plist[q++] = (uint16_t)pr;
                    }
                    M.rpred0[r] = (uint16_t)first;
                    M.rmeta[r] = (uint32_t)M.nbase[node] | ((uint32_t)(d ? d : 1) << 2) | ((uint32_t)off << 16);
                }
                run += q_bcast(inc, 15);
            }
            meta_ok = true;
            cw_wave_sync();
        }
        POAQ_PROF(0);

        /* ---- DP fill ---- */
        for (int j = gl; j < cols; j += 16) M.H[j] = (int16_t)(CW_POA_SW ? 0 : j * G);
        for (int j = gl; j < L; j += 16) M.seqrank[j] = CW_NONE16;
        cw_wave_sync();
        poaq_fill(M, n, cols, gl);
        cw_wave_sync();
        POAQ_PROF(1);

        /* ---- end cell: best sink in the last column, lowest rank on ties ---- */
        int bi, bj = L;
        if (CW_POA_OV) { /* overlap mode: the best cell of a sink's row, columns 1..L; lowest rank, then lowest column on ties */
            int bs = CW_NEG * 2, br = 0x7FFFFFFF, bc = L;
            for (int r = gl; r < n; r += 16) {
                if (!CW_POA_SW && M.has_out[M.r2n[r]]) continue; /* (local mode: any row) */
                for (int j = 1; j <= L; ++j) {
                    const int h = M.H[(r + 1) * HS + j];
                    if (h > bs) { bs = h; br = r; bc = j; }
                }
            }
            for (int o = 8; o > 0; o >>= 1) {
                const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o), oc = __shfl_xor(bc, o);
                if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; bc = oc; }
            }
            bi = br + 1; bj = bc;
        } else {
            int bs = CW_NEG * 2, br = 0x7FFFFFFF;
            for (int r = gl; r < n; r += 16) {
                if (M.has_out[M.r2n[r]]) continue;
                const int h = M.H[(r + 1) * HS + L];
                if (h > bs) { bs = h; br = r; } /* ranks ascend within a lane */
            }
            for (int o = 8; o > 0; o >>= 1) {
                const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o);
                if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; }
            }
            bi = br + 1;
        }

        /* ---- traceback: every lane of the row walks the same path (diagonal through the in-edges in order, then vertical through them,
           then horizontal); the three candidate cells of a single-predecessor node are requested together ---- */
        {
            int i = bi, j = bj;
            while (i > 0 && (!CW_POA_OV || j > 0)) { /* (overlap mode: the walk stops in column 0) */
                const uint32_t meta = M.rmeta[i - 1];
                const int pr0 = (int)M.rpred0[i - 1];
                const int base = (int)(meta & 3u), np = (int)((meta >> 2) & 0x3FFFu), off = (int)(meta >> 16);
                const int h = M.H[i * HS + j];
                if (CW_POA_SW && h == 0) break; /* local mode: the alignment starts where the score does */
                const int sx = (j != 0 && (int)M.sq[j - 1] == base) ? MS : XS;
                const int hh = j != 0 ? (int)M.H[i * HS + j - 1] : CW_NEG * 2;
                int pi = i, pj = j;
                bool found = false;
                if (np == 1) {
                    const int hv = (int)M.H[pr0 * HS + j];
                    const int hd = j != 0 ? (int)M.H[pr0 * HS + j - 1] : CW_NEG * 2;
                    if (j != 0 && h == hd + sx) { pi = pr0; pj = j - 1; found = true; }
                    else if (h == hv + G) { pi = pr0; found = true; }
                } else {
                    if (j != 0) {
                        for (int q = 0; q < np && !found; ++q) {
                            const int pr = (int)M.plist[off + q];
                            if (h == (int)M.H[pr * HS + j - 1] + sx) { pi = pr; pj = j - 1; found = true; }
                        }
                    }
                    for (int q = 0; q < np && !found; ++q) {
                        const int pr = (int)M.plist[off + q];
                        if (h == (int)M.H[pr * HS + j] + G) { pi = pr; found = true; }
                    }
                }
                if (!found) {
                    if (j != 0 && h == hh + G) { pj = j - 1; found = true; }
                    else return 3;
                }
                if (pj != j && pi != i && gl == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                i = pi; j = pj;
            }
        }
        cw_wave_sync();
        POAQ_PROF(2);

        /* ---- merge the path into the graph: one lane per sequence position, 16 at a time (cf. poa_run) ---- */
        {
            const int n_old = n;
            const int chunks = (L + 15) >> 4;
            int next_rank = -1;
            for (int c = chunks - 1; c >= 0; --c) {
                const int j = c * 16 + gl;
                const bool act = j < L;
                const uint32_t rk = act ? M.seqrank[j] : CW_NONE16;
                const unsigned has = q_ballot(act && rk != CW_NONE16);
                const unsigned later = has & ~(lt_mask | (1u << gl));
                const int later_rank = (int)(uint32_t)q_bcast((int)rk, later ? (__ffs((int)later) - 1) : 0);
                const int qr = later ? later_rank : next_rank;
                const int first_rank = (int)(uint32_t)q_bcast((int)rk, has ? (__ffs((int)has) - 1) : 0);
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
                const int j = c * 16 + gl;
                const bool act = j < L;
                const bool fresh = act && M.pcur[j] == CW_NONE16;
                const unsigned fb = q_ballot(fresh);
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
                for (int r = gl; r <= n_old; r += 16) hist[r] = 0;
                cw_wave_sync();
                for (int c = 0; c < chunks; ++c) {
                    const int j = c * 16 + gl;
                    if (j < L && M.pat[j] != CW_NONE16) atomicAdd(&hist[M.pat[j]], 1u);
                }
                cw_wave_sync();
                int run = 0;
                for (int r0 = 0; r0 < n_old; r0 += 16) {
                    const int r = r0 + gl;
                    const int hcount = r < n_old ? (int)hist[r] : 0;
                    const int inc = q_scan_add(hcount);
                    if (r < n_old) {
                        const int nr = r + run + inc;
                        const int v = M.r2n[r];
                        M.rtmp[nr] = (uint16_t)v;
                        M.n2r[v] = (uint16_t)nr;
                    }
                    run += q_bcast(inc, 15);
                }
                for (int c = 0; c < chunks; ++c) {
                    const int j = c * 16 + gl;
                    if (j < L && M.pat[j] != CW_NONE16) {
                        const int cur = M.pcur[j];
                        const int nr = (int)M.pat[j] + (cur - n_old);
                        M.rtmp[nr] = (uint16_t)cur;
                        M.n2r[cur] = (uint16_t)nr;
                    }
                }
                cw_wave_sync();
                n = n_old + fresh_total;
                for (int r = gl; r < n; r += 16) M.r2n[r] = M.rtmp[r];
                meta_ok = false;
                cw_wave_sync();
            }
            for (int c = 0; c < chunks; ++c) {
                const int j = c * 16 + gl;
                const bool act = j < L && j > 0;
                int head = 0, cur = 0;
                bool add = false;
                if (act) {
                    head = M.pcur[j - 1]; cur = M.pcur[j];
                    add = true;
                    for (uint32_t e = M.in_head[cur]; e != CW_NONE16; e = M.enext[e])
                        if (M.efrom[e] == (uint16_t)head) { add = false; if (CW_CONS_HEAVIEST_BUNDLE) M.ew[e] = (uint16_t)(M.ew[e] + 1); break; }
                }
                const unsigned ab = q_ballot(add);
                const int total = __popc(ab);
                if ((uint32_t)(ne + total) > M.e_cap) return 2;
                if (add) {
                    const int e = ne + __popc(ab & lt_mask);
                    M.efrom[e] = (uint16_t)head; M.enext[e] = CW_NONE16;
                    if (CW_CONS_HEAVIEST_BUNDLE) M.ew[e] = 1;
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
        POAQ_PROF(3);
    }

    /* ---- consensus: column-majority vote, or the heaviest bundle (cw_policy.h CW_POA_CONSENSUS) ---- */
    uint32_t out_len = 0;
#if CW_CONS_HEAVIEST_BUNDLE
    cw_wave_sync();
    if (gl == 0) out_len = poa_consensus_hb(M, n, t, sc);
    out_len = (uint32_t)__shfl((int)out_len, (int)(threadIdx.x & 48u));
#else
    for (int r0 = 0; r0 < n; r0 += 16) {
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
        const unsigned bal = q_ballot(emit >= 0);
        const uint32_t idx = out_len + (uint32_t)__popc(bal & lt_mask);
        if (emit >= 0 && idx < t.out_cap) sc.arena[t.out_off + idx] = CW_ACGT(emit);
        out_len += (uint32_t)__popc(bal);
    }
#endif
    if (out_len > t.out_cap) return 3;
    if (gl == 0) sc.seg_len[t.seg_slot] = out_len;
    POAQ_PROF(4);
#undef POAQ_PROF
    return 1;
}

/* ---- tier Q: four tasks per wave ------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64 * CW_POAQ_WAVES) cw_poa_q_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int gl = threadIdx.x & 15;
    const uint32_t grp = threadIdx.x >> 4; /* 0 .. 4 * waves - 1 */
    const PoaMem<int16_t> M = poa_carve<int16_t>(lds + (size_t)grp * CW_POAQ_TASK_BYTES, CW_POAQ_NC, CW_POAQ_EC, CW_POAQ_LC, CW_POAQ_HC, 0);
    const uint32_t* list = sc.tier_list[0];
    const uint32_t n_work = min(sc.ctr->n_tier[0], sc.list_cap);
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
    for (;;) {
        uint32_t mi = 0;
        if (gl == 0) mi = atomicAdd(&sc.ctr->next_tier[0], 1u);
        mi = (uint32_t)q_bcast((int)mi, 0);
        if (mi >= n_work) break;
        const uint32_t ti = list[mi];
        const PoaTask t = sc.tasks[ti];
        if (t.n_members == 0) continue; /* a neutral entry (cw_chain.h "cap_ok") */
        const int rc = poaq_run(M, t, b, sc, gl, acc);
        if (gl == 0) poa_hand_over(sc, t, ti, rc, 0); /* rc 2: redone in tier S, whose kernel follows this one on the stream */
        cw_wave_sync();
    }
    /* per-phase cycles as the first row of every wave saw them (the four rows of a wave share one instruction stream): slots of tier G,
       which never runs beside tier Q in practice, offset by 28 */
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 5; ++q) atomicAdd(&sc.ctr->prof[28 + q], acc[q]);
}

#endif
