/*
 * cw_index.h -- per-window setup + k-mer index kernel.
 *
 * cw_setup_kernel : sizes every window's slice of the scratch arrays (single work-group scan).
 * cw_index_kernel : one 1024-thread work-group per window, everything staged in LDS:
 *    A  pile-wide k-mer counts (A4a; consumers correctionMSA.cpp:18, DBG.cpp:38): a direct-addressed table
 *       of 4-bit saturating counters over all 4^k keys (k<=9 -> 128 KiB) + a small exact table for the
 *       keys that saturate; exported as the ascending solid set with exact counts.
 *    B  anchor candidates: template k-mers that are repeated in no sequence and occur in >= support
 *       sequences; position matrix P[candidate][sequence] in LDS.
 *    C  longest ordered chain (cw_policy.h "chaining"), evaluated level by level so that only the pairs
 *       that can win are scored.
 *    D  segmentation: identical-by-construction segments are written straight to the arena, the others
 *       become POA tasks.
 * Roofline: HBM-bound by construction -- the pile is read from HBM (L2) and only the solid set, the task
 * list and the trivial segments are written; see DESIGN.md for the bytes.
 */
#ifndef CW_INDEX_H
#define CW_INDEX_H

#include "cw_device.h"

#define CW_IDX_THREADS 1024
#define CW_IDX_WAVES 16
#define CW_IDX_LDS_BYTES 163840
#define CW_TMAX 1024 /* template k-mer slots */
#define CW_EX_SLOTS 2048
#define CW_TH_SLOTS 2048

/* ------------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(1024) cw_setup_kernel(DevBatch b, DevScratch sc, cw_params prm, uint64_t solid_total_cap,
                                                         uint64_t seg_total_cap, uint64_t arena_total_cap) {
    __shared__ uint32_t part[3][1024];
    __shared__ uint64_t run[3];
    const int tid = threadIdx.x;
    if (tid < 3) run[tid] = 0;
    __syncthreads();
    for (uint32_t w0 = 0; w0 < b.n_windows; w0 += 1024) {
        const uint32_t w = w0 + tid;
        uint32_t need_solid = 0, need_seg = 0, need_arena = 0;
        uint32_t nk = 0, tl = 0, ns = 0;
        if (w < b.n_windows) {
            const uint32_t s0 = b.win_first_seq[w], s1 = b.win_first_seq[w + 1];
            ns = s1 - s0;
            for (uint32_t s = s0; s < s1; ++s) {
                uint32_t l = b.seq_len[s];
                if (l >= prm.k) nk += l - prm.k + 1;
            }
            tl = ns ? b.seq_len[s0] : 0;
            need_solid = nk / prm.solid + 1;
            need_seg = (tl >= prm.k) ? tl - prm.k + 3 : 1;
            need_arena = 16 * tl + 4096;
        }
        part[0][tid] = need_solid; part[1][tid] = need_seg; part[2][tid] = need_arena;
        __syncthreads();
        /* simple in-LDS inclusive scan, 10 steps */
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t v0 = 0, v1 = 0, v2 = 0;
            if (tid >= o) { v0 = part[0][tid - o]; v1 = part[1][tid - o]; v2 = part[2][tid - o]; }
            __syncthreads();
            part[0][tid] += v0; part[1][tid] += v1; part[2][tid] += v2;
            __syncthreads();
        }
        if (w < b.n_windows) {
            WinInfo wi;
            wi.status = CW_WIN_CONSENSUS;
            wi.n_seqs = ns; wi.tpl_len = tl; wi.n_kmers = nk;
            uint64_t sb = run[0] + part[0][tid] - need_solid, gb = run[1] + part[1][tid] - need_seg,
                     ab = run[2] + part[2][tid] - need_arena;
            bool over = sb + need_solid > solid_total_cap || gb + need_seg > seg_total_cap || ab + need_arena > arena_total_cap;
            wi.solid_base = (uint32_t)sb; wi.solid_cap = need_solid; wi.n_solid = 0;
            wi.seg_base = (uint32_t)gb; wi.seg_cap = need_seg; wi.n_segs = 0;
            wi.arena_base = (uint32_t)ab; wi.arena_cap = need_arena; wi.arena_used = 0;
            wi.pad[0] = wi.pad[1] = wi.pad[2] = 0;
            if (over) { wi.status = CW_WIN_OVERFLOW; wi.solid_cap = wi.seg_cap = wi.arena_cap = 0; wi.solid_base = wi.seg_base = wi.arena_base = 0; }
            sc.win[w] = wi;
        }
        __syncthreads();
        if (tid < 3) run[tid] += part[tid][1023];
        __syncthreads();
    }
}

/* ---- block-wide helpers (1024 threads) --------------------------------------------------------- */
/* exclusive prefix sum of v over the block; total returned through *total.  scratch: 17 uint32 in LDS. */
__device__ __forceinline__ uint32_t cw_block_exscan(uint32_t v, uint32_t* scratch, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < CW_IDX_WAVES; ++i) { uint32_t t = scratch[i]; scratch[i] = run; run += t; }
        scratch[CW_IDX_WAVES] = run;
    }
    __syncthreads();
    *total = scratch[CW_IDX_WAVES];
    return scratch[wave] + inc - v;
}

__device__ __forceinline__ uint32_t cw_hash32(uint32_t x) { return x * 2654435761u; }

/* lookup of a template k-mer: returns its representative template position or -1 */
__device__ __forceinline__ int cw_tpl_lookup(const uint32_t* th, const uint32_t* tkey, uint32_t key) {
    uint32_t slot = cw_hash32(key) >> (32 - 11);
    for (;;) {
        uint32_t e = th[slot];
        if (e == 0) return -1;
        if (tkey[e - 1] == key) return (int)e - 1;
        slot = (slot + 1) & (CW_TH_SLOTS - 1);
    }
}

/* ------------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(CW_IDX_THREADS) cw_index_kernel(DevBatch b, DevScratch sc, cw_params prm) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k = prm.k;
    const bool direct = k <= 9;                             /* 4^k nibbles fit the LDS table */
    const uint32_t n_keys = direct ? 1u << (2 * k) : 0u;
    const uint32_t nib_words = direct ? (n_keys >= 8 ? n_keys / 8 : 1) : 0u;

    /* phase A carve */
    uint32_t* tab = (uint32_t*)lds;                                           /* nib_words                */
    unsigned long long* ex = (unsigned long long*)(lds + 131072);             /* CW_EX_SLOTS               */
    uint32_t* scan_tmp = (uint32_t*)(lds + 131072 + CW_EX_SLOTS * 8);         /* 32 words                  */
    uint32_t* flags = scan_tmp + 32;                                          /* [0] overflow [1..] misc   */
    /* phase B..D carve (reuses the same bytes once phase A has been exported) */
    uint32_t* th = (uint32_t*)lds;                                            /* 2048 x u32      @0      */
    uint32_t* tkey = (uint32_t*)(lds + 8192);                                 /* 1024 x u32      @8192   */
    uint32_t* tsup = (uint32_t*)(lds + 12288);                                /* 1024 x u32      @12288  */
    uint8_t* trep = lds + 16384;                                              /* 1024 x u8       @16384  */
    int16_t* tcand = (int16_t*)(lds + 17408);                                 /* 1024 x i16      @17408  */
    uint16_t* cand_tp = (uint16_t*)(lds + 19456);                             /* 1024 x u16      @19456  */
    uint32_t* seen = (uint32_t*)(lds + 21504);                                /* 16 x 32 x u32   @21504  */
    int16_t* clen = (int16_t*)(lds + 23552);
    int16_t* cnxt = (int16_t*)(lds + 25600);
    int16_t* bnext = (int16_t*)(lds + 27648);
    int16_t* lvl_head = (int16_t*)(lds + 29696);                              /* 1025 x i16 (2064 B)     */
    int32_t* csc = (int32_t*)(lds + 31760);
    uint16_t* chain = (uint16_t*)(lds + 35856);
    uint32_t* misc = (uint32_t*)(lds + 37904);                                /* 64 words                */
    uint16_t* const P_lds = (uint16_t*)(lds + 38400);
    const uint32_t p_cap = (CW_IDX_LDS_BYTES - 38400) / 2;
    /* position matrix: in LDS when it fits next to the presence bitsets, else in this work-group's global slot
       (high-identity deep piles: every template k-mer is an anchor).  Accessors pick the address space with a
       block-uniform branch so that the common case keeps ds_ instructions. */
    uint16_t* const P_glb = sc.p_fallback + (size_t)blockIdx.x * sc.p_fallback_elems;
#define PRD(i) (pg ? (uint32_t)P_glb[i] : (uint32_t)P_lds[i])
#define PWR(i, v) do { if (pg) P_glb[i] = (uint16_t)(v); else P_lds[i] = (uint16_t)(v); } while (0)

    for (;;) {
        __syncthreads();
        if (tid == 0) misc[63] = atomicAdd(&sc.ctr->next_window, 1u);
        __syncthreads();
        const uint32_t w = misc[63];
        if (w >= b.n_windows) break;
        WinInfo* wi = &sc.win[w];
        if (wi->status == CW_WIN_OVERFLOW) continue;
        const uint32_t s0 = b.win_first_seq[w];
        const uint32_t N = wi->n_seqs;
        const uint32_t L0 = wi->tpl_len;

        CW_PROF_T0();
        /* ================= phase A: counts ================= */
        if (!direct) {
            /* ---- k > 9: the key space no longer fits a direct table.  Exact counts in an LDS hash table (key<<32 | count),
               the pile scanned P times, pass p owning the keys whose hash falls in partition p (P chosen so that even an
               all-distinct pile stays under half load); after each pass the solid entries are appended to the window's
               slice; at the end the slice is bitonic-sorted by key in LDS. ---- */
            unsigned long long* hs_tab = (unsigned long long*)lds; /* 16384 slots = 128 KiB */
            const uint32_t HS = 16384u;
            const uint32_t P_ = (wi->n_kmers + HS / 2 - 1) / (HS / 2) ? (wi->n_kmers + HS / 2 - 1) / (HS / 2) : 1u;
            uint32_t written = 0;
            bool fits = true;
            if (tid < 8) flags[tid] = 0;
            for (uint32_t pass = 0; pass < P_; ++pass) {
                for (uint32_t i = tid; i < HS; i += CW_IDX_THREADS) hs_tab[i] = 0ull;
                __syncthreads();
                for (uint32_t sp = 0; sp < N; sp += 2) {
                    const uint32_t s = sp + (tid >> 9);
                    if (s < N) {
                        const uint32_t len = b.seq_len[s0 + s];
                        const uint32_t* words = b.bases + b.seq_word_off[s0 + s];
                        const uint32_t nk = len >= k ? len - k + 1 : 0;
                        for (uint32_t p = tid & 511; p < nk; p += 512) {
                            const uint32_t key = cw_kmer_at(words, p, k);
                            const uint32_t h = cw_hash32(key ^ 0x9E3779B9u);
                            if ((uint32_t)(((unsigned long long)h * P_) >> 32) != pass) continue;
                            uint32_t slot = cw_hash32(key) >> (32 - 14);
                            const unsigned long long fresh = ((unsigned long long)key << 32) | 1ull;
                            for (uint32_t probe = 0;; ++probe) {
                                if (probe >= HS) { flags[0] = 1; break; }
                                const unsigned long long cur = atomicCAS(&hs_tab[slot], 0ull, fresh);
                                if (cur == 0ull) break;
                                if ((uint32_t)(cur >> 32) == key) { atomicAdd(&hs_tab[slot], 1ull); break; }
                                slot = (slot + 1) & (HS - 1);
                            }
                        }
                    }
                }
                __syncthreads();
                uint32_t mine = 0;
                for (uint32_t i = tid; i < HS; i += CW_IDX_THREADS) mine += ((uint32_t)hs_tab[i] >= prm.solid) ? 1u : 0u;
                uint32_t total;
                const uint32_t off = cw_block_exscan(mine, scan_tmp, &total);
                if (written + total > wi->solid_cap || flags[0]) fits = false;
                if (fits && mine) {
                    uint32_t o = wi->solid_base + written + off;
                    for (uint32_t i = tid; i < HS; i += CW_IDX_THREADS) {
                        const unsigned long long e = hs_tab[i];
                        if ((uint32_t)e >= prm.solid) { sc.solid_key[o] = (uint32_t)(e >> 32); sc.solid_cnt[o] = (uint32_t)e; o++; }
                    }
                }
                written += total;
                __syncthreads();
                if (!fits) break;
            }
            uint32_t np2 = 2;
            while (np2 < written) np2 <<= 1;
            if (fits && np2 > HS) fits = false;
            if (tid == 0) {
                wi->n_solid = fits ? written : 0;
                if (!fits) { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            }
            __threadfence_block();
            __syncthreads();
            if (!fits) continue;
            if (written > 1) {
                for (uint32_t x = tid; x < np2; x += CW_IDX_THREADS)
                    hs_tab[x] = x < written ? (((unsigned long long)sc.solid_key[wi->solid_base + x] << 32) | sc.solid_cnt[wi->solid_base + x]) : ~0ull;
                __syncthreads();
                for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
                    for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                        for (uint32_t x = tid; x < np2; x += CW_IDX_THREADS) {
                            const uint32_t y = x ^ j2;
                            if (y > x) {
                                const unsigned long long ax = hs_tab[x], ay = hs_tab[y];
                                const bool up = (x & k2) == 0;
                                if ((ax > ay) == up) { hs_tab[x] = ay; hs_tab[y] = ax; }
                            }
                        }
                        __syncthreads();
                    }
                }
                for (uint32_t x = tid; x < written; x += CW_IDX_THREADS) {
                    sc.solid_key[wi->solid_base + x] = (uint32_t)(hs_tab[x] >> 32);
                    sc.solid_cnt[wi->solid_base + x] = (uint32_t)hs_tab[x];
                }
                __syncthreads();
            }
        } else {
        for (uint32_t i = tid; i < nib_words; i += CW_IDX_THREADS) tab[i] = 0;
        for (uint32_t i = tid; i < CW_EX_SLOTS; i += CW_IDX_THREADS) ex[i] = 0ull;
        if (tid < 8) flags[tid] = 0;
        __syncthreads();
        for (uint32_t sp = 0; sp < N; sp += 2) {
            const uint32_t s = sp + (tid >> 9);
            if (s < N) {
                const uint32_t len = b.seq_len[s0 + s];
                const uint32_t* words = b.bases + b.seq_word_off[s0 + s];
                const uint32_t nk = len >= k ? len - k + 1 : 0;
                for (uint32_t p = tid & 511; p < nk; p += 512) {
                    const uint32_t key = cw_kmer_at(words, p, k);
                    const uint32_t wd = key >> 3, sh = (key & 7) * 4;
                    uint32_t old = tab[wd];
                    for (;;) {
                        if (((old >> sh) & 15u) == 15u) break;
                        uint32_t prev = atomicCAS(&tab[wd], old, old + (1u << sh));
                        if (prev == old) break;
                        old = prev;
                    }
                }
            }
        }
        __syncthreads();
        CW_PROF(sc.ctr, 0, tid == 0);
        /* exact counts for the keys whose 4-bit counter saturated */
        for (uint32_t sp = 0; sp < N; sp += 2) {
            const uint32_t s = sp + (tid >> 9);
            if (s < N) {
                const uint32_t len = b.seq_len[s0 + s];
                const uint32_t* words = b.bases + b.seq_word_off[s0 + s];
                const uint32_t nk = len >= k ? len - k + 1 : 0;
                for (uint32_t p = tid & 511; p < nk; p += 512) {
                    const uint32_t key = cw_kmer_at(words, p, k);
                    if (((tab[key >> 3] >> ((key & 7) * 4)) & 15u) != 15u) continue;
                    uint32_t slot = cw_hash32(key) >> (32 - 11);
                    const unsigned long long fresh = ((unsigned long long)(key + 1) << 32) | 1ull;
                    for (uint32_t probe = 0;; ++probe) {
                        if (probe >= CW_EX_SLOTS) { flags[0] = 1; break; }
                        unsigned long long cur = atomicCAS(&ex[slot], 0ull, fresh);
                        if (cur == 0ull) break;
                        if ((uint32_t)(cur >> 32) == key + 1) { atomicAdd(&ex[slot], 1ull); break; }
                        slot = (slot + 1) & (CW_EX_SLOTS - 1);
                    }
                }
            }
        }
        __syncthreads();
        CW_PROF(sc.ctr, 1, tid == 0);
        if (flags[0]) { /* more saturated keys than the exact table holds */
            if (tid == 0) { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            continue;
        }
        /* export the solid set in ascending key order.  Thread t owns the contiguous words [t*wpt, (t+1)*wpt) (so thread
           order = key order) but visits them rotated by t, which spreads a wave's reads over all LDS banks; the few
           solid keys it finds are kept in registers, ranked locally, and written after one block-wide prefix sum. */
        {
            const uint32_t keys_per_word = n_keys >= 8 ? 8 : n_keys;
            const uint32_t wpt = (nib_words + CW_IDX_THREADS - 1) / CW_IDX_THREADS;
            const uint32_t w_beg = min(nib_words, (uint32_t)tid * wpt), w_cnt = min(nib_words, w_beg + wpt) - w_beg;
            uint32_t lk[8], lc[8];
            uint32_t mine = 0;
            for (uint32_t i = 0; i < w_cnt; ++i) {
                uint32_t r = i + (uint32_t)tid;
                r = r >= w_cnt ? r % w_cnt : r;
                const uint32_t wd = w_beg + r;
                const uint32_t v = tab[wd];
                if (v == 0) continue;
                for (uint32_t q = 0; q < keys_per_word; ++q) {
                    const uint32_t nib = (v >> (4 * q)) & 15u;
                    if (!nib) continue;
                    const uint32_t key = wd * 8 + q;
                    uint32_t c = nib;
                    if (nib == 15u) {
                        uint32_t slot = cw_hash32(key) >> (32 - 11);
                        while ((uint32_t)(ex[slot] >> 32) != key + 1) slot = (slot + 1) & (CW_EX_SLOTS - 1);
                        c = (uint32_t)ex[slot];
                    }
                    if (c < prm.solid) continue;
#pragma unroll
                    for (int z = 0; z < 8; ++z) if ((uint32_t)z == mine) { lk[z] = key; lc[z] = c; }
                    mine++;
                }
            }
            if (mine > 8) flags[1] = 1; /* more than the register slots hold: the whole block re-walks in key order */
            uint32_t total;
            const uint32_t off = cw_block_exscan(mine, scan_tmp, &total);
            const bool fits = total <= wi->solid_cap;
            if (fits && flags[1]) {
                uint32_t o = wi->solid_base + off;
                for (uint32_t i = 0; i < w_cnt && mine; ++i) {
                    const uint32_t wd = w_beg + i;
                    const uint32_t v = tab[wd];
                    if (v == 0) continue;
                    for (uint32_t q = 0; q < keys_per_word; ++q) {
                        const uint32_t nib = (v >> (4 * q)) & 15u;
                        if (!nib) continue;
                        const uint32_t key = wd * 8 + q;
                        uint32_t c = nib;
                        if (nib == 15u) {
                            uint32_t slot = cw_hash32(key) >> (32 - 11);
                            while ((uint32_t)(ex[slot] >> 32) != key + 1) slot = (slot + 1) & (CW_EX_SLOTS - 1);
                            c = (uint32_t)ex[slot];
                        }
                        if (c >= prm.solid) { sc.solid_key[o] = key; sc.solid_cnt[o] = c; o++; }
                    }
                }
            } else if (fits && mine) {
#pragma unroll
                for (int z = 0; z < 8; ++z) {
                    if ((uint32_t)z < mine) {
                        uint32_t rank = 0;
#pragma unroll
                        for (int y = 0; y < 8; ++y) rank += ((uint32_t)y < mine && lk[y] < lk[z]) ? 1u : 0u;
                        sc.solid_key[wi->solid_base + off + rank] = lk[z];
                        sc.solid_cnt[wi->solid_base + off + rank] = lc[z];
                    }
                }
            }
            if (tid == 0) {
                wi->n_solid = fits ? total : 0;
                if (!fits) { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            }
            __syncthreads();
            if (!fits) continue;
        }

        } /* direct table */

        CW_PROF(sc.ctr, 2, tid == 0);
        /* ================= phase B: anchor candidates ================= */
        const uint32_t nk0 = L0 >= k ? L0 - k + 1 : 0;
        const int sup_min = min((int)prm.common_kmers, (int)N / 2); /* correctionMSA.cpp:31 */
        if (nk0 == 0 || nk0 > CW_TMAX) {
            if (tid == 0) {
                if (nk0 == 0) wi->status = CW_WIN_TEMPLATE;
                else { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            }
            continue;
        }
        for (uint32_t i = tid; i < CW_TH_SLOTS; i += CW_IDX_THREADS) th[i] = 0;
        for (uint32_t i = tid; i < CW_TMAX; i += CW_IDX_THREADS) { tsup[i] = 0; trep[i] = 0; tcand[i] = -1; lvl_head[i] = -1; }
        if (tid == 0) lvl_head[CW_TMAX] = -1;
        const uint32_t* tpl_words = b.bases + b.seq_word_off[s0];
        if ((uint32_t)tid < nk0) tkey[tid] = cw_kmer_at(tpl_words, tid, k);
        __syncthreads();
        if ((uint32_t)tid < nk0) {
            const uint32_t key = tkey[tid];
            uint32_t slot = cw_hash32(key) >> (32 - 11);
            for (;;) {
                uint32_t prev = atomicCAS(&th[slot], 0u, (uint32_t)tid + 1);
                if (prev == 0) break;
                if (tkey[prev - 1] == key) { trep[prev - 1] = 1; break; } /* repeated inside the template */
                slot = (slot + 1) & (CW_TH_SLOTS - 1);
            }
        }
        __syncthreads();
        /* support + repeat detection: one wave per sequence */
        for (uint32_t s = wave; s < N; s += CW_IDX_WAVES) {
            uint32_t* my_seen = seen + wave * 32;
            if (lane < 32) my_seen[lane] = 0;
            cw_wave_sync();
            const uint32_t len = b.seq_len[s0 + s];
            const uint32_t* words = b.bases + b.seq_word_off[s0 + s];
            const uint32_t nk = len >= k ? len - k + 1 : 0;
            for (uint32_t p = lane; p < nk; p += 64) {
                const int e = cw_tpl_lookup(th, tkey, cw_kmer_at(words, p, k));
                if (e < 0) continue;
                const uint32_t bit = 1u << (e & 31);
                const uint32_t old = atomicOr(&my_seen[e >> 5], bit);
                if (old & bit) trep[e] = 1;
                else atomicAdd(&tsup[e], 1u);
            }
            cw_wave_sync();
        }
        __syncthreads();
        CW_PROF(sc.ctr, 3, tid == 0);
        /* candidates in template order */
        uint32_t A;
        {
            uint32_t ok = 0;
            if ((uint32_t)tid < nk0) {
                const int rep = cw_tpl_lookup(th, tkey, tkey[tid]);
                ok = (rep == tid && trep[tid] == 0 && (int)tsup[tid] >= sup_min) ? 1u : 0u;
            }
            const uint32_t off = cw_block_exscan(ok, scan_tmp, &A);
            if (ok) { tcand[tid] = (int16_t)off; cand_tp[off] = (uint16_t)tid; }
        }
        __syncthreads();
        /* P is anchor-major, P[a * Np + s]; Np is even with Np/2 odd so that rows read as u32 pairs by consecutive
           lanes fall on distinct banks */
        uint32_t Np = (N + 1u) & ~1u;
        if (((Np >> 1) & 1u) == 0u) Np += 2u;
        const uint32_t Nw = (N + 63u) >> 6;
        /* LDS needs: the matrix (A*Np u16) + presence bitsets (A*Nw u64) + dirty list (N u16) */
        const bool pg = (uint64_t)A * Np * 2 + (uint64_t)A * Nw * 8 + (uint64_t)N * 2 + 16 > (uint64_t)p_cap * 2;
        if (pg && (uint64_t)A * Np > sc.p_fallback_elems) {
            if (tid == 0) { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            continue;
        }
        for (uint32_t i = tid; i < A * Np; i += CW_IDX_THREADS) PWR(i, CW_NONE16);
        __syncthreads();
        for (uint32_t s = wave; s < N; s += CW_IDX_WAVES) {
            const uint32_t len = b.seq_len[s0 + s];
            const uint32_t* words = b.bases + b.seq_word_off[s0 + s];
            const uint32_t nk = len >= k ? len - k + 1 : 0;
            for (uint32_t p = lane; p < nk; p += 64) {
                const int e = cw_tpl_lookup(th, tkey, cw_kmer_at(words, p, k));
                if (e < 0) continue;
                const int a = tcand[e];
                if (a >= 0) PWR((uint32_t)a * Np + s, p);
            }
        }
        __syncthreads();

        /* A sequence whose anchor positions increase with the anchor index ("clean") satisfies pos(a) < pos(b) for every
           pair a < b it holds, so its contribution to score(a,b) is one bit of presence(a) & presence(b); only the few
           sequences with an out-of-order (spurious) anchor hit need their positions compared.  Exact, and ~20x cheaper
           than comparing positions for all N sequences. */
        uint8_t* clean = (uint8_t*)seen;                               /* N flags (2 KiB available) */
        unsigned long long* pres = (unsigned long long*)(P_lds + (pg ? 0 : (((size_t)A * Np + 3u) & ~(size_t)3u))); /* A x Nw, 8-byte aligned */
        uint16_t* dirty = (uint16_t*)(pres + (size_t)A * Nw);          /* up to N ids */
        const bool use_bits = N <= 2048u && ((uint8_t*)(dirty + N) <= lds + CW_IDX_LDS_BYTES);
        if (use_bits) {
            for (uint32_t s = wave; s < N; s += CW_IDX_WAVES) {
                int run = -1;
                bool bad = false;
                for (uint32_t a0 = 0; a0 < A; a0 += 64) {
                    const uint32_t a = a0 + lane;
                    const uint32_t pv = a < A ? PRD(a * Np + s) : (uint32_t)CW_NONE16;
                    const int v = pv != CW_NONE16 ? (int)pv : -1;
                    const int inc = cw_wave_scan_max(v);
                    int before = cw_wave_shr1(inc, -1);
                    before = max(before, run);
                    bad = bad || (__ballot(v >= 0 && v <= before) != 0ull);
                    run = max(run, cw_lane_value(inc, 63));
                }
                if (lane == 0) clean[s] = bad ? 0 : 1;
            }
            if (tid == 0) misc[3] = 0;
            __syncthreads();
            for (uint32_t a = wave; a < A; a += CW_IDX_WAVES) {
                for (uint32_t w = 0; w < Nw; ++w) {
                    const uint32_t s = w * 64 + lane;
                    const bool on = s < N && PRD(a * Np + s) != CW_NONE16 && clean[s];
                    const unsigned long long bal = __ballot(on);
                    if (lane == 0) pres[(size_t)a * Nw + w] = bal;
                }
            }
            for (uint32_t s = tid; s < N; s += CW_IDX_THREADS)
                if (!clean[s]) dirty[atomicAdd(&misc[3], 1u)] = (uint16_t)s;
            __syncthreads();
            /* the dirty list must not depend on thread timing: sort the few ids (insertion sort by one thread) */
            if (tid == 0) {
                const uint32_t nd = misc[3];
                for (uint32_t x = 1; x < nd; ++x) {
                    const uint16_t v = dirty[x];
                    uint32_t y = x;
                    while (y > 0 && dirty[y - 1] > v) { dirty[y] = dirty[y - 1]; --y; }
                    dirty[y] = v;
                }
            }
            __syncthreads();
        }
        const uint32_t n_dirty = use_bits ? misc[3] : 0u;

        CW_PROF(sc.ctr, 4, tid == 0);
        /* ================= phase C: chain ================= */
        /* best(a) for a = A-1 .. 0 by one wave: lanes score 64 successors b > a at a time against all N sequences,
           nearest successors first; smax[b] = max length from b onwards lets the scan stop as soon as no later
           successor can tie or beat the best link found (exact; cw_policy.h "chaining"). */
        if (wave == 0) {
            int16_t* smax = bnext; /* A + 1 entries */
            if (lane == 0) smax[A] = -1;
            cw_wave_sync();
            for (int a = (int)A - 1; a >= 0; --a) {
                unsigned long long best = 0ull;
                const uint32_t* pa_row = (const uint32_t*)((pg ? P_glb : P_lds) + (uint32_t)a * Np);
                const uint32_t half = Np >> 1; /* pairs of sequences; padding entries are CW_NONE16 and never count */
                for (uint32_t b0 = (uint32_t)a + 1u; b0 < A; b0 += 64) {
                    const uint32_t bb = b0 + (uint32_t)lane;
                    unsigned long long key = 0ull;
                    if (bb < A) {
                        uint32_t cnt = 0;
                        if (use_bits) {
                            for (uint32_t w = 0; w < Nw; ++w) cnt += (uint32_t)__popcll(pres[(size_t)a * Nw + w] & pres[(size_t)bb * Nw + w]);
                            for (uint32_t d = 0; d < n_dirty; ++d) {
                                const uint32_t sd = dirty[d];
                                const uint32_t pa = PRD((uint32_t)a * Np + sd), pb = PRD(bb * Np + sd);
                                cnt += (pa < pb && pb != CW_NONE16) ? 1u : 0u;
                            }
                        } else {
                            const uint32_t* pb_row = (const uint32_t*)((pg ? P_glb : P_lds) + bb * Np);
#pragma unroll 8
                            for (uint32_t s = 0; s < half; ++s) {
                                const uint32_t va = pa_row[s], vb = pb_row[s];
                                const uint32_t a0 = va & 0xFFFFu, a1 = va >> 16, b0_ = vb & 0xFFFFu, b1_ = vb >> 16;
                                cnt += (a0 < b0_ && b0_ != CW_NONE16) ? 1u : 0u;
                                cnt += (a1 < b1_ && b1_ != CW_NONE16) ? 1u : 0u;
                            }
                        }
                        if ((int)cnt >= sup_min)
                            key = ((unsigned long long)((uint32_t)clen[bb] + 1u) << 48) | ((unsigned long long)((uint32_t)csc[bb] + cnt) << 16) |
                                  (unsigned long long)(0xFFFFu - bb);
                    }
                    key = cw_wave_max_u64(key);
                    best = key > best ? key : best;
                    if (best != 0ull && b0 + 64 < A) {
                        const int blen = (int)(best >> 48) - 1;
                        if ((int)smax[b0 + 64] < blen) break;
                    }
                }
                if (lane == 0) {
                    int la = 0;
                    if (best == 0ull) { clen[a] = 0; csc[a] = 0; cnxt[a] = -1; }
                    else {
                        la = (int)(best >> 48); /* stored length+1 of b == length of a */
                        clen[a] = (int16_t)la;
                        csc[a] = (int32_t)((best >> 16) & 0xFFFFFFFFull);
                        cnxt[a] = (int16_t)(0xFFFFu - (uint32_t)(best & 0xFFFFull));
                    }
                    const int sm = smax[a + 1];
                    smax[a] = (int16_t)(la > sm ? la : sm);
                }
                cw_wave_sync();
            }
        }
        if (wave == 0) {
            /* chain start: longest, then best score, then largest index; a chain needs at least one edge */
            int b_len = 0, b_sc = 0, b_a = -1;
            for (int a = (int)A - 1 - lane; a >= 0; a -= 64) {
                const int l = clen[a], s = csc[a];
                if (l > b_len || (l == b_len && l > 0 && (s > b_sc || b_a < 0))) { b_len = l; b_sc = s; b_a = a; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int ol = __shfl_xor(b_len, o), os = __shfl_xor(b_sc, o), oa = __shfl_xor(b_a, o);
                const bool take = (oa >= 0) && (b_a < 0 || ol > b_len || (ol == b_len && (os > b_sc || (os == b_sc && oa > b_a))));
                if (take) { b_len = ol; b_sc = os; b_a = oa; }
            }
            int m = 0;
            if (b_a >= 0 && b_len > 0) {
                for (int a = b_a; a != -1; a = cnxt[a]) {
                    if (lane == 0) chain[m] = (uint16_t)a;
                    m++;
                }
            }
            if (lane == 0) misc[0] = (uint32_t)m;
        }
        __syncthreads();
        CW_PROF(sc.ctr, 5, tid == 0);
        const uint32_t m = misc[0];
        if (m == 0 || m < prm.min_anchors) {
            if (tid == 0) wi->status = CW_WIN_TEMPLATE;
            continue;
        }
        if (m + 1 > wi->seg_cap) {
            if (tid == 0) { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            continue;
        }
        if (tid == 0) { misc[1] = 0; /* arena used */ misc[2] = 0; /* overflow */ }
        __syncthreads();

        /* ================= phase D: segments ================= */
        for (uint32_t seg = wave; seg <= m; seg += CW_IDX_WAVES) {
            const int ca = seg > 0 ? (int)chain[seg - 1] : -1;
            const int cb = seg < m ? (int)chain[seg] : -1;
            /* pass 1: count members, min/max piece length */
            uint32_t n_mem = 0, mn = 0xFFFFFFFFu, mx = 0, first_seq = 0, first_start = 0;
            for (uint32_t sb = 0; sb < N && n_mem < prm.max_msa; sb += 64) {
                const uint32_t s = sb + lane;
                bool is = false;
                uint32_t st = 0, ln = 0;
                if (s < N) {
                    const uint32_t pa = ca >= 0 ? PRD((uint32_t)ca * Np + s) : 0u;
                    const uint32_t pb = cb >= 0 ? PRD((uint32_t)cb * Np + s) : 0u;
                    if (seg == 0) { is = pb != CW_NONE16 && pb > 0; st = 0; ln = pb; }
                    else if (seg == m) { is = pa != CW_NONE16; st = pa; ln = b.seq_len[s0 + s] - pa; }
                    else { is = pa != CW_NONE16 && pb != CW_NONE16 && pa < pb; st = pa; ln = pb - pa; }
                }
                const unsigned long long bal = __ballot(is);
                const uint32_t before = n_mem + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                const bool keep = is && before < prm.max_msa;
                uint32_t lmn = keep ? ln : 0xFFFFFFFFu, lmx = keep ? ln : 0u;
                for (int o = 32; o > 0; o >>= 1) { lmn = min(lmn, (uint32_t)__shfl_xor((int)lmn, o)); lmx = max(lmx, (uint32_t)__shfl_xor((int)lmx, o)); }
                mn = min(mn, lmn); mx = max(mx, lmx);
                if (n_mem == 0 && bal) {
                    const int fl = __ffsll((long long)bal) - 1;
                    first_seq = sb + fl; first_start = (uint32_t)__shfl((int)st, fl);
                }
                n_mem = min(prm.max_msa, n_mem + (uint32_t)__popcll(bal));
            }
            const uint32_t slot = wi->seg_base + seg;
            if (n_mem == 0) {
                if (lane == 0) { sc.seg_off[slot] = wi->arena_base; sc.seg_len[slot] = 0; }
                continue;
            }
            const bool by_anchor = (seg > 0 && seg < m && mn == mx && mx <= k); /* all pieces = first mx bases of anchor a */
            const bool single = n_mem == 1;
            const uint32_t need = (by_anchor || single) ? mx : 2 * mx + 2;
            uint32_t aoff = 0;
            if (lane == 0) aoff = atomicAdd(&misc[1], need);
            aoff = (uint32_t)__shfl((int)aoff, 0);
            if (aoff + need > wi->arena_cap) { if (lane == 0) misc[2] = 1; continue; }
            const uint32_t abs_off = wi->arena_base + aoff;
            if (by_anchor) {
                const uint32_t key = tkey[cand_tp[ca]];
                if ((uint32_t)lane < mx) sc.arena[abs_off + lane] = "ACGT"[(key >> (2 * (k - 1 - lane))) & 3u];
                if (lane == 0) { sc.seg_off[slot] = abs_off; sc.seg_len[slot] = mx; }
                continue;
            }
            if (single) {
                const uint32_t* words = b.bases + b.seq_word_off[s0 + first_seq];
                for (uint32_t i = lane; i < mx; i += 64) sc.arena[abs_off + i] = "ACGT"[cw_base_at(words, first_start + i)];
                if (lane == 0) { sc.seg_off[slot] = abs_off; sc.seg_len[slot] = mx; }
                continue;
            }
            /* POA task */
            uint32_t t_idx = 0, m_off = 0;
            if (lane == 0) { t_idx = atomicAdd(&sc.ctr->n_tasks, 1u); m_off = atomicAdd(&sc.ctr->n_members, n_mem); }
            t_idx = (uint32_t)__shfl((int)t_idx, 0); m_off = (uint32_t)__shfl((int)m_off, 0);
            if (t_idx >= sc.task_cap || m_off + n_mem > sc.member_cap) { if (lane == 0) misc[2] = 1; continue; }
            uint32_t done = 0;
            for (uint32_t sb = 0; sb < N && done < n_mem; sb += 64) {
                const uint32_t s = sb + lane;
                bool is = false;
                uint32_t st = 0, ln = 0;
                if (s < N) {
                    const uint32_t pa = ca >= 0 ? PRD((uint32_t)ca * Np + s) : 0u;
                    const uint32_t pb = cb >= 0 ? PRD((uint32_t)cb * Np + s) : 0u;
                    if (seg == 0) { is = pb != CW_NONE16 && pb > 0; st = 0; ln = pb; }
                    else if (seg == m) { is = pa != CW_NONE16; st = pa; ln = b.seq_len[s0 + s] - pa; }
                    else { is = pa != CW_NONE16 && pb != CW_NONE16 && pa < pb; st = pa; ln = pb - pa; }
                }
                const unsigned long long bal = __ballot(is);
                const uint32_t idx = done + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (is && idx < n_mem) {
                    PoaMember pm;
                    pm.seq = s0 + s; pm.start = (uint16_t)st; pm.len = (uint16_t)ln;
                    sc.members[m_off + idx] = pm;
                }
                done += (uint32_t)__popcll(bal);
            }
            if (lane == 0) {
                PoaTask t;
                t.window = w; t.seg_slot = slot; t.member_off = m_off; t.n_members = n_mem; t.max_len = mx;
                t.out_off = abs_off; t.out_cap = need;
                /* route by the expected graph size: the graph has at least max_len nodes once its longest member is in
                   and typically ends at 1.4-1.6x that; a task that still outgrows its tier is redone in the next one */
                const uint32_t est = (mx * 17u + 9u) / 10u;
                const uint32_t tier = ((est + 1) * (mx + 1) <= 4096u && est <= 160u) ? 0u
                                      : (est <= 256u && mx <= 255u)                 ? 1u
                                      : (est <= 512u && mx <= 511u)                 ? 2u
                                                                                    : 3u;
                t.state = tier ? 2u : 0u;
                sc.tasks[t_idx] = t;
                if (tier) {
                    const uint32_t bi = atomicAdd(&sc.ctr->n_tier[tier], 1u);
                    if (bi < sc.list_cap) sc.tier_list[tier][bi] = t_idx; else misc[2] = 1;
                }
                sc.seg_off[slot] = abs_off; sc.seg_len[slot] = 0;
            }
        }
        __syncthreads();
        CW_PROF(sc.ctr, 6, tid == 0);
        if (tid == 0) {
            if (misc[2]) { wi->status = CW_WIN_OVERFLOW; sc.ctr->any_overflow = 1; }
            else { wi->n_segs = m + 1; wi->arena_used = misc[1]; }
        }
    }
}

#undef PRD
#undef PWR

#endif
