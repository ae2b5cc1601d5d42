/*
 * cw_finish.h -- per-window finish: concatenate the segment consensuses, weightConsensus
 * (correctionMSA.cpp:6-27) and the local de-Bruijn polish (correctionDBG.cpp:93-205 with DBG.cpp:18-169),
 * then write the caller's outputs.  One 64-lane wave per window; the string lives in LDS.
 *
 * The reference's merCounts map is replaced by the window's ascending solid table (keys + exact counts,
 * produced by the index kernel): every test the polish makes is "count >= solidThresh" or a comparison
 * of counts of solid k-mers, except the anchor-pair ordering (correctionDBG.cpp:77-83) which may touch a
 * non-solid k-mer; that rare case recounts the k-mer over the packed pile, so results stay exact.
 * The reference's `visited` string set (correctionDBG.cpp:94, never cleared) holds only solid k-mers, so
 * it is a bitmap over the solid table.
 */
#ifndef CW_FINISH_H
#define CW_FINISH_H

#include "cw_device.h"

#ifndef CW_FIN_WAVES
#define CW_FIN_WAVES 4
#endif
#ifndef CW_FIN_WGS_PER_CU
#define CW_FIN_WGS_PER_CU 2
#endif
#ifndef CW_FIN_CB
#define CW_FIN_CB 3072
#endif
/* CW_FIN_CB: string capacity per buffer, first pass                                    */
#define CW_FIN_CB_BIG 32768   /* ... of the second pass (round 5): windows whose consensus or polish outgrew the first are done again, one wave per
                                 work-group with 107 KB of LDS -- k < 8 (chance anchors: consensuses of several templates), heaviest-bundle policy */
#ifndef CW_FIN_VIS_WORDS
#define CW_FIN_VIS_WORDS 1024 /* visited bitmap in LDS: up to 32768 solid k-mers */
#endif
#define CW_FIN_VIS_GLB_WORDS 8192 /* per-wave bitmap in global memory beyond that: 4^9 keys, the most a direct count table can export */
#define CW_FIN_FRAMES 56
#define CW_FIN_SKEYS 1024     /* solid keys staged in LDS (every lookup of the polish is a binary search in them) */
/* 20.4 KB per wave, 81.5 KB per work-group: exactly two work-groups per CU (162.9 of 163.8 KB) -- and the kernel is latency-bound: staging 3072
   solid keys instead of 1024 (a depth-150 pile has ~2500, so its lookups are binary searches in global memory; round 4 tried) makes the
   work-group 124 KB, one per CU, and the kernel 1.75x as long at depth 30 and 1.15x at depth 150; a third work-group per CU (smaller buffers) changed nothing */
#define CW_FIN_K16_MAX ((CW_FIN_VIS_WORDS + CW_FIN_SKEYS) * 32 / 17 / 64 * 64) /* compact table: n / 32 bitmap words + n / 2 words of 16-bit keys in the bitmap's and the key table's space (3840) */
#define CW_FIN_SLAB_OF(CB) (3 * (CB) + 4 * CW_FIN_VIS_WORDS + CW_FIN_FRAMES * 48 + 256 + 4 * CW_FIN_SKEYS + 16)
#define CW_FIN_SLAB CW_FIN_SLAB_OF(CW_FIN_CB)

struct FinOut {
    char* cons;
    const uint64_t* cons_off;
    uint32_t* cons_len;
    uint8_t* win_status;
    uint32_t* solid;
    const uint64_t* solid_off;
    uint32_t* solid_len;
};

struct FinCtx {
    const uint32_t* skey; /* window's solid keys (ascending) */
    const uint32_t* scnt;
    const uint16_t* scnt16; /* counts staged in LDS (all <= 65535), or NULL */
    bool staged;            /* skey points into LDS and holds at most CW_FIN_SKEYS keys */
    const uint16_t* k16;    /* or (round 4): more keys than that, k <= 9: their low 16 bits staged in LDS (fin_find), or NULL ... */
    uint32_t b16_1, b16_2, b16_3; /* ... and where the keys with bits 17:16 = 1, 2, 3 begin (three scalars: as an array the lookups' select
                                     chains became indexed scratch loads) */
    uint32_t n_solid;
    uint32_t k, solid, kmask;
    /* pile, for exact recounts (the three arrays themselves: a pointer to the kernel's DevBatch argument made the compiler keep a
       copy of it -- and this struct with it -- in scratch memory, re-read at every table lookup: 136 B per lane in round 5's code object) */
    const uint32_t* seq_len;
    const uint64_t* seq_word_off;
    const uint32_t* bases;
    uint32_t s0, N;
};

/* The key table is in LDS (staged) or in the scratch arena, the visited bitmap in LDS or in the wave's global slot: said with the address
   space at every access, or the compiler merges the two arms into flat_ loads (LDS data at global-memory latency). */
typedef __attribute__((address_space(3))) const uint32_t* fin_l32;
typedef __attribute__((address_space(1))) const uint32_t* fin_g32;
typedef __attribute__((address_space(3))) const uint16_t* fin_l16;
typedef __attribute__((address_space(3))) uint32_t* fin_l32w;
typedef __attribute__((address_space(1))) uint32_t* fin_g32w;
__device__ __forceinline__ uint32_t fin_skey(const FinCtx& c, uint32_t i) { return c.staged ? ((fin_l32)c.skey)[i] : ((fin_g32)c.skey)[i]; }
__device__ __forceinline__ uint32_t fin_scnt(const FinCtx& c, uint32_t i) { return c.scnt16 ? (uint32_t)((fin_l16)c.scnt16)[i] : ((fin_g32)c.scnt)[i]; }

__device__ __forceinline__ bool fin_upper(uint8_t c) { return c >= 'A' && c <= 'Z'; }
__device__ __forceinline__ uint32_t fin_code(uint8_t c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; default: return 3; }
}
__device__ __forceinline__ uint32_t fin_key_at(const uint8_t* s, uint32_t p, uint32_t k) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < k; ++i) v = (v << 2) | fin_code(s[p + i]);
    return v;
}

/* index of key in the solid table or -1 (per-lane, global memory binary search) */
__device__ __forceinline__ int fin_find(const FinCtx& c, uint32_t key) {
    if (c.k16) { /* the compact table: the two high bits pick the range, the search compares 16-bit halves at LDS latency */
        const uint32_t h = key >> 16, kl = key & 0xFFFFu;
        const int e1 = (int)c.b16_1, e2 = (int)c.b16_2, e3 = (int)c.b16_3, e4 = (int)c.n_solid; /* (read first, selected as values) */
        int lo = h == 0u ? 0 : h == 1u ? e1 : h == 2u ? e2 : e3;
        int hi = (h == 0u ? e1 : h == 1u ? e2 : h == 2u ? e3 : e4) - 1;
        if (h > 3u) return -1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const uint32_t v = (uint32_t)((fin_l16)c.k16)[mid];
            if (v == kl) return mid;
            if (v < kl) lo = mid + 1; else hi = mid - 1;
        }
        return -1;
    }
    int lo = 0, hi = (int)c.n_solid - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t v = fin_skey(c, (uint32_t)mid);
        if (v == key) return mid;
        if (v < key) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* Four keys at once when the table is staged in LDS (n_solid <= CW_FIN_SKEYS, 16-byte aligned): lanes 16g..16g+15 look up key_g.
 * Two dependent LDS reads instead of the ten of a binary search: 16 pivots (every 64th key) pick the 64-key bucket, then every lane
 * of the group compares four consecutive keys of it.  Returns the index for the lane's group, or -1. */
__device__ __forceinline__ int fin_find4(const FinCtx& c, uint32_t key_g, int lane) {
    const int g = lane >> 4, l = lane & 15;
    const uint32_t n = c.n_solid;
    const uint32_t pv = (uint32_t)l * 64u;
    const fin_l32 sk = (fin_l32)c.skey; /* only called with the table staged */
    const bool ge = pv < n && key_g >= sk[pv];
    const uint32_t bits = (uint32_t)(__ballot(ge) >> (16 * g)) & 0xFFFFu;
    int mine = -1;
    if (bits) {
        const uint32_t base = ((uint32_t)__popc(bits) - 1u) * 64u + (uint32_t)l * 4u;
        if (base < n) {
            uint4 kk; kk.x = sk[base]; kk.y = sk[base + 1u]; kk.z = sk[base + 2u]; kk.w = sk[base + 3u]; /* one ds_read_b128: 16-byte aligned */ /* the staging area is padded to a multiple of four words */
            mine = kk.x == key_g ? (int)base : (base + 1 < n && kk.y == key_g) ? (int)base + 1 : (base + 2 < n && kk.z == key_g) ? (int)base + 2
                   : (base + 3 < n && kk.w == key_g) ? (int)base + 3 : -1;
        }
    }
    const uint32_t hit = (uint32_t)(__ballot(mine >= 0) >> (16 * g)) & 0xFFFFu;
    const int src = hit ? 16 * g + (__ffs((int)hit) - 1) : lane;
    const int got = __shfl(mine, src);
    return hit ? got : -1;
}

/* exact pile-wide count of any k-mer (wave-wide; uniform result) */
__device__ uint32_t fin_count_scan(const FinCtx& c, uint32_t key, int lane);
__device__ uint32_t fin_count_exact(const FinCtx& c, uint32_t key, int lane) {
    int idx = fin_find(c, key);
    if (idx >= 0) return fin_scnt(c, (uint32_t)idx);
    return fin_count_scan(c, key, lane);
}
/* a k-mer below the solidity threshold is not in the table: count it in the pile */
__device__ uint32_t fin_count_scan(const FinCtx& c, uint32_t key, int lane) {
    uint32_t n = 0;
    for (uint32_t s = 0; s < c.N; ++s) {
        const uint32_t len = c.seq_len[c.s0 + s];
        const uint32_t* words = c.bases + c.seq_word_off[c.s0 + s];
        const uint32_t nk = len >= c.k ? len - c.k + 1 : 0;
        for (uint32_t p = lane; p < nk; p += 64) n += (cw_kmer_at(words, p, c.k) == key) ? 1u : 0u;
    }
    return (uint32_t)cw_wave_sum((int)n);
}

/*
 * getNeighbours (DBG.cpp:18-54): solid successors (left==0) or predecessors (left==1) of `key`, best count
 * first, ties in generation order (A,C,G,T to the right; T,G,C,A to the left -- the order in which the
 * reference pushes them; its std::sort on <=4 elements is an insertion sort, hence stable).
 * Results in nbk/nbi (keys / solid-table indices), returns how many.  Wave-uniform.
 */
__device__ int fin_neighbours(const FinCtx& c, uint32_t key, int left, uint32_t* nbk, uint32_t* nbi, int lane) {
    uint32_t cand = 0;
    int idx = -1;
    uint32_t cnt = 0;
    if (c.staged) {
        /* candidate g is looked up by lanes 16g..16g+15; lanes 0..3 then pick up result g */
        const uint32_t gq = (uint32_t)(lane >> 4);
        const uint32_t cg = !left ? (((key << 2) & c.kmask) | gq) : (((3u - gq) << (2 * (c.k - 1))) | (key >> 2));
        const int ig = fin_find4(c, cg, lane);
        idx = __shfl(ig, (lane & 3) * 16);
        cand = (uint32_t)__shfl((int)cg, (lane & 3) * 16);
        if (lane < 4 && idx >= 0) cnt = fin_scnt(c, (uint32_t)idx);
        if (lane >= 4) idx = -1;
    } else if (lane < 4) {
        if (!left) cand = ((key << 2) & c.kmask) | (uint32_t)lane;
        else cand = ((uint32_t)(3 - lane) << (2 * (c.k - 1))) | (key >> 2);
        idx = fin_find(c, cand);
        if (idx >= 0) cnt = ((fin_g32)c.scnt)[idx];
    }
    const bool ok = lane < 4 && idx >= 0; /* table holds exactly the k-mers with count >= solid */
    const unsigned long long bal = __ballot(ok);
    int rank = 0;
    for (int o = 0; o < 4; ++o) {
        const uint32_t oc = (uint32_t)__shfl((int)cnt, o);
        const bool ook = (bal >> o) & 1ull;
        if (ook && o != lane && (oc > cnt || (oc == cnt && o < lane))) rank++;
    }
    if (ok) { nbk[rank] = cand; nbi[rank] = (uint32_t)idx; }
    cw_wave_sync();
    return __popcll(bal);
}

/* getNextSrc / getNextDst (correctionDBG.cpp:13-43), wave-uniform scans over the LDS string */
__device__ int fin_next_src(const uint8_t* s, uint32_t len, uint32_t beg, uint32_t m) {
    uint32_t run = 0, i = beg;
    while (i < len && (fin_upper(s[i]) || run < m)) { run = fin_upper(s[i]) ? run + 1 : 0; i++; }
    return run >= m ? (int)i - 1 : -1;
}
__device__ int fin_next_dst(const uint8_t* s, uint32_t len, uint32_t beg, uint32_t m) {
    uint32_t run = 0, i = beg;
    while (i < len && run < m) { run = fin_upper(s[i]) ? run + 1 : 0; i++; }
    return run >= m ? (int)i - 1 : -1;
}

struct FinLds {
    uint8_t* s;      /* current string      */
    uint8_t* alt;    /* edit target         */
    uint8_t* path;   /* link() path / extension scratch */
    uint32_t* vis;
    bool vis_glb;    /* vis is the wave's slot in global memory (deep polishing piles), else LDS */
    uint32_t* f_nbk; /* frames: 4 keys      */
    uint32_t* f_nbi; /*         4 indices   */
    uint32_t* f_meta;/*         n | it<<8 | plen<<16 */
    uint32_t* f_dist;
    uint32_t* f_key;
    uint32_t* tmp;   /* 64 words            */
    uint32_t cb;     /* capacity of s / alt / path (CW_FIN_CB, or CW_FIN_CB_BIG in the second pass) */
};

__device__ __forceinline__ uint32_t fin_vis_get(const FinLds& M, uint32_t w) { return M.vis_glb ? ((fin_g32)M.vis)[w] : ((fin_l32)M.vis)[w]; }
__device__ __forceinline__ void fin_vis_or(const FinLds& M, uint32_t w, uint32_t bits) { if (M.vis_glb) ((fin_g32w)M.vis)[w] |= bits; else ((fin_l32w)M.vis)[w] |= bits; }
__device__ __forceinline__ void fin_vis_clear(const FinLds& M, uint32_t w) { if (M.vis_glb) ((fin_g32w)M.vis)[w] = 0u; else ((fin_l32w)M.vis)[w] = 0u; }

/*
 * link (DBG.cpp:99-169) without recursion.  path[] starts as the k characters of src; on success returns
 * the path length (the reference's missingPart), else 0.  Returns -1 if the path buffer would overflow.
 */
__device__ int fin_link(const FinCtx& c, const FinLds& M, uint32_t src, uint32_t dst, uint32_t max_len, int lane) {
    uint32_t branches = 0;
    int depth = 0;
    uint32_t plen = c.k, dist = 0, cur = src;
    uint32_t* nbk = M.tmp;      /* current neighbour list (4 + 4 words) */
    uint32_t* nbi = M.tmp + 4;
    int n = 0, it = 0;
    bool found = false;
    int child_ret = -1; /* -1: entering a frame, 0/1: a child just returned */
    for (;;) {
        if (child_ret < 0) {
            /* ---- frame entry (DBG.cpp:100-116) ---- */
            if (branches > CW_DBG_MAX_BRANCHES || dist > max_len) { child_ret = 0; goto frame_return; }
            found = (cur == dst);
            n = fin_neighbours(c, cur, 0, nbk, nbi, lane);
            it = 0;
            /* ---- linear stretch (DBG.cpp:119-138) ---- */
            while (!found && n == 1 && it < n && dist <= max_len) {
                const uint32_t ck = nbk[0], ci = nbi[0];
                const bool seen = (fin_vis_get(M, ci >> 5) >> (ci & 31)) & 1u;
                found = (ck == dst);
                if (!found && !seen) {
                    if (plen + 2 > M.cb) return -1;
                    if (lane == 0) { fin_vis_or(M, ci >> 5, 1u << (ci & 31)); M.path[plen] = CW_ACGT(ck & 3u); }
                    plen++; dist++;
                    cw_wave_sync();
                    n = fin_neighbours(c, ck, 0, nbk, nbi, lane);
                    it = 0;
                } else if (found) {
                    if (plen + 2 > M.cb) return -1;
                    if (lane == 0) M.path[plen] = CW_ACGT(ck & 3u);
                    plen++;
                } else {
                    it++;
                }
            }
        } else {
            /* ---- back in a parent after a child returned (DBG.cpp:150-154) ---- */
            if (child_ret == 1) { if (depth == 0) return (int)plen; child_ret = 1; goto frame_return; }
            it++;
        }
        /* ---- branching stretch (DBG.cpp:141-160) ---- */
        {
            bool descended = false;
            while (!found && n > 1 && it < n && dist <= max_len) {
                const uint32_t ck = nbk[it], ci = nbi[it];
                const bool seen = (fin_vis_get(M, ci >> 5) >> (ci & 31)) & 1u;
                found = (ck == dst);
                if (!found && !seen) {
                    if (depth + 1 >= CW_FIN_FRAMES || plen + 2 > M.cb) return -1;
                    branches++;
                    if (lane == 0) {
                        fin_vis_or(M, ci >> 5, 1u << (ci & 31));
                        for (int q = 0; q < 4; ++q) { M.f_nbk[depth * 4 + q] = nbk[q]; M.f_nbi[depth * 4 + q] = nbi[q]; }
                        M.f_meta[depth] = (uint32_t)n | ((uint32_t)it << 8) | (plen << 16);
                        M.f_dist[depth] = dist; M.f_key[depth] = cur;
                        M.path[plen] = CW_ACGT(ck & 3u);
                    }
                    cw_wave_sync();
                    depth++; plen++; dist++; cur = ck;
                    child_ret = -1;
                    descended = true;
                    break;
                } else if (found) {
                    if (plen + 2 > M.cb) return -1;
                    if (lane == 0) M.path[plen] = CW_ACGT(ck & 3u);
                    plen++;
                } else {
                    it++;
                }
            }
            if (descended) continue;
        }
        child_ret = found ? 1 : 0;
    frame_return:
        if (depth == 0) return child_ret == 1 ? (int)plen : 0;
        /* pop: restore the parent frame; on success the path is kept as is (DBG.cpp:153 returns at once) */
        depth--;
        {
            const uint32_t meta = M.f_meta[depth];
            n = (int)(meta & 0xFF); it = (int)((meta >> 8) & 0xFF);
            if (child_ret != 1) plen = meta >> 16;
            dist = M.f_dist[depth]; cur = M.f_key[depth];
            const uint32_t k0 = M.f_nbk[depth * 4 + 0], k1 = M.f_nbk[depth * 4 + 1], k2 = M.f_nbk[depth * 4 + 2], k3 = M.f_nbk[depth * 4 + 3];
            const uint32_t i0 = M.f_nbi[depth * 4 + 0], i1 = M.f_nbi[depth * 4 + 1], i2 = M.f_nbi[depth * 4 + 2], i3 = M.f_nbi[depth * 4 + 3];
            cw_wave_sync();
            if (lane == 0) { nbk[0] = k0; nbk[1] = k1; nbk[2] = k2; nbk[3] = k3; nbi[0] = i0; nbi[1] = i1; nbi[2] = i2; nbi[3] = i3; }
            cw_wave_sync();
            found = false;
        }
    }
}

/* polishCorrection (correctionDBG.cpp:93-205).  Returns the new length, or -1 on a capacity overflow. */
__device__ int fin_polish(const FinCtx& c, FinLds& M, uint32_t len, int lane) {
    const uint32_t k = c.k, zone = CW_DBG_ZONE, m = k + zone;
    uint32_t tmp_src_beg = 0, tmp_src_end = 0, tmp_dst_beg = 0, tmp_dst_end = 0; /* :104 */
    uint32_t* nbk = M.tmp;
    uint32_t* nbi = M.tmp + 4;

    /* head (:116-130) */
    uint32_t i = 0;
    while (i < len && !fin_upper(M.s[i])) i++;
    if (i > 0 && i < len && len - i >= k) {
        const uint32_t ext_len = i;
        uint32_t key = fin_key_at(M.s, i, k), dist = 0;
        int n = fin_neighbours(c, key, 1, nbk, nbi, lane);
        while (n == 1 && dist < ext_len) { /* DBG.cpp:66 */
            key = nbk[0];
            if (lane == 0) M.s[i - 1 - dist] = CW_ACGT(key >> (2 * (k - 1)));
            dist++;
            cw_wave_sync();
            n = fin_neighbours(c, key, 1, nbk, nbi, lane);
        }
        i = dist; /* :128 (and :122 when fully extended: ext_len == dist) */
        cw_wave_sync();
    }

    /* bridge weak regions (:133-187) */
    while (i < len) {
        const int src_end = fin_next_src(M.s, len, i, m);
        const int dst_end = fin_next_dst(M.s, len, (uint32_t)(src_end + 1), m);
        const int src_beg = src_end - (int)m + 1, dst_beg = dst_end - (int)m + 1;
        if (src_end == -1 || dst_end == -1) break; /* :185 */
        /* anchors (:47-91): 4 k-mers per zone, unique inside their zone, pairs in src-major order */
        uint32_t my_src = 0, my_dst = 0;
        bool pair_ok = false;
        uint32_t sum = 0;
        {
            const int a = lane >> 2, d = lane & 3; /* lanes 0..15 = pairs */
            uint32_t ks[4], kd[4];
            for (int q = 0; q < 4; ++q) { ks[q] = fin_key_at(M.s, (uint32_t)src_beg + q, k); kd[q] = fin_key_at(M.s, (uint32_t)dst_beg + q, k); }
            uint32_t cs[4], cd[4];
            if (c.staged) { /* the eight table lookups as two grouped ones (lanes 16g..16g+15 take k-mer g of each zone) */
                const int g = lane >> 4;
                const uint32_t ksg = g == 0 ? ks[0] : g == 1 ? ks[1] : g == 2 ? ks[2] : ks[3];
                const uint32_t kdg = g == 0 ? kd[0] : g == 1 ? kd[1] : g == 2 ? kd[2] : kd[3];
                const int is = fin_find4(c, ksg, lane), id = fin_find4(c, kdg, lane);
                for (int q = 0; q < 4; ++q) {
                    const int iq = __builtin_amdgcn_readlane(is, q * 16), jq = __builtin_amdgcn_readlane(id, q * 16);
                    cs[q] = iq >= 0 ? fin_scnt(c, (uint32_t)iq) : fin_count_scan(c, ks[q], lane);
                    cd[q] = jq >= 0 ? fin_scnt(c, (uint32_t)jq) : fin_count_scan(c, kd[q], lane);
                }
            } else if (c.k16) { /* the compact table: lanes 0..7 search one k-mer each, their counts come back in one round trip */
                const uint32_t kq = lane == 0 ? ks[0] : lane == 1 ? ks[1] : lane == 2 ? ks[2] : lane == 3 ? ks[3] : lane == 4 ? kd[0] : lane == 5 ? kd[1] : lane == 6 ? kd[2] : kd[3];
                const int iq = lane < 8 ? fin_find(c, kq) : -1;
                const uint32_t cq = iq >= 0 ? fin_scnt(c, (uint32_t)iq) : 0u;
                for (int q = 0; q < 4; ++q) {
                    const int is = __builtin_amdgcn_readlane(iq, q), id = __builtin_amdgcn_readlane(iq, 4 + q);
                    cs[q] = is >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)cq, q) : fin_count_scan(c, ks[q], lane);
                    cd[q] = id >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)cq, 4 + q) : fin_count_scan(c, kd[q], lane);
                }
            } else {
                for (int q = 0; q < 4; ++q) { cs[q] = fin_count_exact(c, ks[q], lane); cd[q] = fin_count_exact(c, kd[q], lane); }
            }
            if (lane < 16) {
                uint32_t sk = 0, dk = 0, sc_ = 0, dc_ = 0;
                int so = 0, dof = 0;
                for (int q = 0; q < 4; ++q) {
                    if (q == a) { sk = ks[q]; sc_ = cs[q]; }
                    if (q == d) { dk = kd[q]; dc_ = cd[q]; }
                }
                for (int q = 0; q < 4; ++q) { so += (ks[q] == sk) ? 1 : 0; dof += (kd[q] == dk) ? 1 : 0; }
                pair_ok = (so == 1 && dof == 1);
                my_src = sk; my_dst = dk; sum = sc_ + dc_;
            }
        }
        unsigned long long avail = __ballot(pair_ok);
        int region_len = 0;
        for (int tries = 0; tries < CW_DBG_MAX_ANCHORS && avail != 0ull && region_len == 0; ++tries) {
            /* stable descending order == repeatedly take the best remaining, lowest pair index on ties */
            int best = -1;
            {
                const bool in = (avail >> lane) & 1ull;
                int bs = in ? (int)sum : -1, bl = in ? lane : 64;
                for (int o = 32; o > 0; o >>= 1) {
                    const int os = __shfl_xor(bs, o), ol = __shfl_xor(bl, o);
                    if (os > bs || (os == bs && ol < bl)) { bs = os; bl = ol; }
                }
                best = bl;
            }
            avail &= ~(1ull << best);
            const uint32_t src = (uint32_t)__shfl((int)my_src, best), dst = (uint32_t)__shfl((int)my_dst, best);
            tmp_src_beg = (uint32_t)src_beg + (uint32_t)(best >> 2); /* :153-156 */
            tmp_src_end = tmp_src_beg + k - 1;
            tmp_dst_beg = (uint32_t)dst_beg + (uint32_t)(best & 3);
            tmp_dst_end = tmp_dst_beg + k - 1;
            if (src != dst) { /* :158 */
                const uint32_t gap = tmp_dst_beg - tmp_src_end - 1;
                /* :163 in IEEE double, no contraction: ((0.15*2.0)*gap + gap) + k, truncated */
                const double t0 = 15.0 / 100.0 * 2.0;
                const double t1 = __dmul_rn(t0, (double)gap);
                const double t2 = __dadd_rn(t1, (double)gap);
                const double t3 = __dadd_rn(t2, (double)k);
                const uint32_t max_size = (uint32_t)t3;
                for (uint32_t q = lane; q < k; q += 64) M.path[q] = CW_ACGT((src >> (2 * (k - 1 - q))) & 3u);
                cw_wave_sync();
                const int r = fin_link(c, M, src, dst, max_size, lane);
                if (r < 0) return -1;
                region_len = r;
            }
        }
        if (region_len > 0) { /* :169-177 */
            const uint32_t rl = tmp_dst_end - tmp_src_beg + 1;
            /* first occurrence of s[tmp_src_beg, +rl) in s (:173) -- it exists, at tmp_src_beg at the latest */
            uint32_t bpos = tmp_src_beg;
            for (uint32_t p0 = 0; p0 < tmp_src_beg; p0 += 64) {
                const uint32_t p = p0 + lane;
                bool eq = p < tmp_src_beg;
                if (eq) for (uint32_t q = 0; q < rl; ++q) if (M.s[p + q] != M.s[tmp_src_beg + q]) { eq = false; break; }
                const unsigned long long bal = __ballot(eq);
                if (bal) { bpos = p0 + (uint32_t)(__ffsll((long long)bal) - 1); break; }
            }
            const uint32_t new_len = len - rl + (uint32_t)region_len;
            if (new_len > M.cb) return -1;
            for (uint32_t q = lane; q < new_len; q += 64) {
                uint8_t ch;
                if (q < bpos) ch = M.s[q];
                else if (q < bpos + (uint32_t)region_len) ch = M.path[q - bpos];
                else ch = M.s[q - (uint32_t)region_len + rl];
                M.alt[q] = ch;
            }
            cw_wave_sync();
            uint8_t* sw = M.s; M.s = M.alt; M.alt = sw;
            len = new_len;
            i = bpos;
        } else {
            i = tmp_dst_beg > i ? tmp_dst_beg : (uint32_t)dst_beg; /* :179,:182 */
        }
    }

    /* tail (:189-202) */
    i = len - 1;
    while (i > 0 && !fin_upper(M.s[i])) i--;
    if (i > 0 && i < len - 1 && i + 1 >= k) {
        const uint32_t ext_len = len - 1 - i;
        uint32_t key = fin_key_at(M.s, i + 1 - k, k), dist = 0;
        int n = fin_neighbours(c, key, 0, nbk, nbi, lane);
        while (n > 0 && dist < ext_len) { /* DBG.cpp:87 */
            key = nbk[0];
            if (lane == 0) M.s[i + 1 + dist] = CW_ACGT(key & 3u);
            dist++;
            cw_wave_sync();
            n = fin_neighbours(c, key, 0, nbk, nbi, lane);
        }
        cw_wave_sync();
    }
    return (int)len;
}

/* CB: string capacity; WAVES: waves per work-group; RETRY: the second pass over the windows the first pass could not hold (sc.fin_retry) */
template <int CB, int WAVES, bool RETRY>
__global__ void __launch_bounds__(64 * WAVES, RETRY ? 4 : 1) /* (the second pass at 128 registers: it has to find room beside running kernels, normally to read one counter) */ cw_finish_kernel(DevBatch b, DevScratch sc, cw_params prm, FinOut out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    /* The second pass keeps its three strings (3 x 32 KB) in global memory: as 107 KB of LDS its work-groups waited for a CU with that much
       free -- among the persistent POA kernels of the other batches in flight 1 ms at depth 150, 5 ms in the driver -- to read one counter and end. */
    constexpr int CBL = RETRY ? 0 : CB; /* string bytes in LDS */
    uint8_t* slab = lds + (size_t)wave * CW_FIN_SLAB_OF(CBL);
    FinLds M;
    M.cb = CB;
    uint8_t* const strings = RETRY ? sc.fin_big + (size_t)blockIdx.x * 3 * CB : slab;
    uint8_t* buf0 = strings; uint8_t* buf1 = strings + CB;
    M.path = strings + 2 * CB;
    uint32_t* const vis_lds = (uint32_t*)(slab + 3 * CBL);
    M.vis = vis_lds; M.vis_glb = false;
    uint32_t* skey_lds = vis_lds + CW_FIN_VIS_WORDS; /* CW_FIN_SKEYS + 4 words, right behind the bitmap: the compact table runs through both */
    M.f_nbk = skey_lds + CW_FIN_SKEYS + 4;
    M.f_nbi = M.f_nbk + CW_FIN_FRAMES * 4;
    M.f_meta = M.f_nbi + CW_FIN_FRAMES * 4;
    M.f_dist = M.f_meta + CW_FIN_FRAMES;
    M.f_key = M.f_dist + CW_FIN_FRAMES;
    M.tmp = M.f_key + CW_FIN_FRAMES; /* 64 words */

    const uint32_t n_retry = RETRY ? min(sc.ctr->n_fin_retry, b.n_windows) : 0u;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(RETRY ? &sc.ctr->next_fin_retry : &sc.ctr->next_finish, 1u);
        w = (uint32_t)__shfl((int)w, 0);
        if (RETRY) { if (w >= n_retry) break; w = sc.fin_retry[w]; }
        else if (w >= b.n_windows) break;
        const WinInfo wi = sc.win[w];
        const uint32_t s0 = b.win_first_seq[w];
        const uint64_t o_beg = out.cons_off[w], o_cap = out.cons_off[w + 1] - o_beg;
        uint32_t status = wi.status;
        uint32_t why = 0;
        int len = 0;
        M.s = buf0; M.alt = buf1;

        if (status == CW_WIN_TEMPLATE && wi.n_seqs == 0) {
            len = 0; /* a window beyond its template has an empty pile (alignmentWindows.cpp:95-97) */
        } else if (status == CW_WIN_TEMPLATE) { /* correctionMSA.cpp:34-36: the raw template */
            const uint32_t* words = b.bases + b.seq_word_off[s0];
            if (wi.tpl_len > o_cap) { status = CW_WIN_OVERFLOW; why = CW_WHY_OUT_CONS; }
            else {
                for (uint32_t q = lane; q < wi.tpl_len; q += 64) out.cons[o_beg + q] = CW_ACGT(cw_base_at(words, q));
                len = (int)wi.tpl_len;
            }
        } else if (status == CW_WIN_CONSENSUS) {
            /* concatenate the segments in chain order */
            uint32_t total = 0;
            bool bad = false;
            for (uint32_t g0 = 0; g0 < wi.n_segs; g0 += 64) {
                const uint32_t g = g0 + lane;
                const uint32_t sl = g < wi.n_segs ? sc.seg_len[wi.seg_base + g] : 0;
                uint32_t inc = sl;
                for (int o = 1; o < 64; o <<= 1) { uint32_t x = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += x; }
                const uint32_t off = total + inc - sl;
                const uint32_t tot = total + (uint32_t)__shfl((int)inc, 63);
                if (tot > (uint32_t)CB) { bad = true; break; }
                if (sl) {
                    const uint8_t* src = sc.arena + sc.seg_off[wi.seg_base + g];
                    for (uint32_t q = 0; q < sl; ++q) M.s[off + q] = src[q];
                }
                total = tot;
            }
            cw_wave_sync();
            /* the visited bitmap of link(): in LDS for up to 32768 solid k-mers (every correction pile); the piles of assembly polishing are
               as deep as the coverage and can hold more: then this wave's slot in global memory */
            const bool vis_glb = wi.n_solid > 32u * CW_FIN_VIS_WORDS;
            M.vis = vis_glb ? sc.fin_vis + (size_t)(blockIdx.x * WAVES + wave) * sc.fin_vis_words : vis_lds; M.vis_glb = vis_glb;
            if (bad || wi.n_solid > 32u * sc.fin_vis_words) { status = CW_WIN_OVERFLOW; why = bad ? CW_WHY_FIN_LEN : CW_WHY_FIN_SOLID; }
            else {
                len = (int)total;
                if ((uint32_t)len >= prm.k) { /* correctionMSA.cpp:43-46 */
                    FinCtx c;
                    c.skey = sc.solid_key + wi.solid_base; c.scnt = sc.solid_cnt + wi.solid_base; c.n_solid = wi.n_solid;
                    c.scnt16 = nullptr; c.staged = false; c.k16 = nullptr; c.b16_1 = c.b16_2 = c.b16_3 = 0u;
                    if (wi.n_solid > CW_FIN_SKEYS && wi.n_solid <= CW_FIN_K16_MAX && prm.k <= 9u && !vis_glb) {
                        /* a deep pile (depth 150: ~2500 solid k-mers): the keys do not fit as words -- and every lookup of the polish was a
                           twelve-step binary search in global memory.  Their low halves fit behind the bitmap words this many k-mers need. */
                        uint16_t* k16 = (uint16_t*)(vis_lds + CW_FIN_K16_MAX / 32);
                        uint32_t n1 = 0, n2 = 0, n3 = 0;
#pragma unroll 4
                        for (uint32_t q = lane; q < wi.n_solid; q += 64) {
                            const uint32_t key = ((fin_g32)c.skey)[q];
                            k16[q] = (uint16_t)key;
                            const uint32_t h = key >> 16;
                            n1 += h < 1u ? 1u : 0u; n2 += h < 2u ? 1u : 0u; n3 += h < 3u ? 1u : 0u;
                        }
                        c.b16_1 = (uint32_t)cw_wave_sum((int)n1); c.b16_2 = (uint32_t)cw_wave_sum((int)n2); c.b16_3 = (uint32_t)cw_wave_sum((int)n3);
                        c.k16 = k16;
                        cw_wave_sync();
                    }
                    if (wi.n_solid <= CW_FIN_SKEYS) { /* the usual case: searches run at LDS latency */
                        /* the visited bitmap needs 32 words for that many k-mers: the counts go behind it, as u16 when they all fit */
                        uint16_t* cnt_lds = (uint16_t*)(vis_lds + 64);
                        bool big = false;
                        for (uint32_t q = lane; q < wi.n_solid; q += 64) {
                            skey_lds[q] = c.skey[q];
                            const uint32_t cv = c.scnt[q];
                            big = big || cv > 0xFFFFu;
                            cnt_lds[q] = (uint16_t)cv;
                        }
                        if (lane < 4) skey_lds[wi.n_solid + lane] = 0xFFFFFFFFu; /* padding read by the 4-key compares (never matched: index >= n) */
                        c.skey = skey_lds; c.staged = true;
                        if (__ballot(big) == 0ull) c.scnt16 = cnt_lds;
                        cw_wave_sync();
                    }
                    c.k = prm.k; c.solid = prm.solid; c.kmask = (prm.k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * prm.k)) - 1u);
                    c.seq_len = b.seq_len; c.seq_word_off = b.seq_word_off; c.bases = b.bases; c.s0 = s0; c.N = wi.n_seqs;
                    /* weightConsensus: case[p] = solid(k-mer at min(p, len-k)) */
                    for (uint32_t p0 = 0; p0 < (uint32_t)len; p0 += 64) {
                        const uint32_t p = p0 + lane;
                        if (p < (uint32_t)len) {
                            const uint32_t q = min(p, (uint32_t)len - prm.k);
                            const bool strong = fin_find(c, fin_key_at(M.s, q, prm.k)) >= 0;
                            M.alt[p] = strong ? M.s[p] : (uint8_t)(M.s[p] + 32);
                        }
                    }
                    cw_wave_sync();
                    { uint8_t* sw = M.s; M.s = M.alt; M.alt = sw; }
                    for (uint32_t q = lane; q < (wi.n_solid + 31) / 32; q += 64) fin_vis_clear(M, q);
                    cw_wave_sync();
                    len = fin_polish(c, M, (uint32_t)len, lane);
                    if (len < 0) { status = CW_WIN_OVERFLOW; why = CW_WHY_FIN_POLISH; len = 0; }
                }
                if (status == CW_WIN_CONSENSUS) {
                    if ((uint64_t)len > o_cap) { status = CW_WIN_OVERFLOW; why = CW_WHY_OUT_CONS; len = 0; }
                    else for (uint32_t q = lane; q < (uint32_t)len; q += 64) out.cons[o_beg + q] = (char)M.s[q];
                }
            }
        }
        if (!RETRY && status == CW_WIN_OVERFLOW && (why == CW_WHY_FIN_LEN || why == CW_WHY_FIN_POLISH)) {
            /* the strings outgrew this pass's buffers: the window goes on the list of the second pass (CW_FIN_CB_BIG), which writes its result */
            if (lane == 0) { const uint32_t at = atomicAdd(&sc.ctr->n_fin_retry, 1u); if (at < b.n_windows) sc.fin_retry[at] = w; }
            cw_wave_sync();
            continue;
        }
        /* solid set for the caller */
        uint32_t n_sol = 0;
        if (out.solid && status != CW_WIN_OVERFLOW) {
            const uint64_t so = out.solid_off[w], scap = out.solid_off[w + 1] - so;
            if (wi.n_solid > scap) { status = CW_WIN_OVERFLOW; why = CW_WHY_OUT_SOLID; len = 0; }
            else {
                for (uint32_t q = lane; q < wi.n_solid; q += 64) out.solid[so + q] = sc.solid_key[wi.solid_base + q];
                n_sol = wi.n_solid;
            }
        }
        if (lane == 0) {
            if (status == CW_WIN_OVERFLOW) { len = 0; sc.ctr->any_overflow = 1; if (why) sc.win[w].pad_ = why; }
            out.cons_len[w] = (uint32_t)len;
            out.win_status[w] = (uint8_t)status;
            if (out.solid) out.solid_len[w] = (status == CW_WIN_OVERFLOW) ? 0u : n_sol;
        }
        cw_wave_sync();
    }
    if (lane == 0) atomicMax(sc.step_clock, (unsigned long long)wall_clock64()); /* when this batch ended (see cw_setup_need_kernel) */
}

#endif
