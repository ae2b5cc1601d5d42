/*
 * cw_chain.h -- anchor chaining + segmentation (A4b, A4c), one wave per window.
 *
 * cw_index_kernel leaves every window an anchor block in HBM/L2 (candidate keys, presence bitsets of the "clean"
 * sequences, the list of "dirty" ones, the anchor-major position matrix).  Chaining is a serial recurrence over the
 * anchors, so a window is one wave here and the CU holds many windows at once (8 waves against the single busy wave the
 * 160 KiB index work-group could offer).
 *    C  longest ordered chain (cw_policy.h "chaining"): best(a) for a = A-1 .. 0, lanes score 64 successors at a time,
 *       nearest first; smax[b] = longest chain from b onwards stops the scan as soon as no later successor can tie.
 *    D  segmentation: pieces between consecutive chain anchors; identical-by-construction segments go straight to the
 *       arena, the others become POA tasks routed to a memory tier by their expected graph size.
 * Per-wave LDS: the DP arrays (16 B per candidate) + presence bitsets + the dirty sequences' columns of the position
 * matrix; when the last two do not fit they are read from the block in place.  The matrix rows phase D needs are read
 * from the block (consecutive lanes = consecutive sequences of one anchor row: coalesced).
 */
#ifndef CW_CHAIN_H
#define CW_CHAIN_H

#include <type_traits>
#include "cw_device.h"
#include "cw_index.h"
#include "cw_poa.h" /* tier capacities for the routing rule */
#include "cw_poa_q.h"

#define CW_CH_WAVES 4
#ifndef CW_CH_SLAB
#define CW_CH_SLAB 20480 /* bytes of LDS per wave: 12 B per anchor + phase D's tile + the pending list + what fits of the bitsets: up to 1230 anchors */
#endif
#define CW_CH_SLAB_LONG 32768 /* ... of the instance an engine configured for long templates launches (cw_configure; round 6): up to CW_TMAX anchors, one work-group per CU */
#define CW_CH_LIST_BYTES 1792
#ifndef CW_POALW_MIN_MEAN
#define CW_POALW_MIN_MEAN 160 /* tier LW takes the tier-L tasks whose mean member length is at least this (two or more chunks per row for most members) ... */
#endif
#ifndef CW_POALW_MIN_MEMBERS
#define CW_POALW_MIN_MEMBERS 6 /* ... and that have at least this many members */
#endif
#define CW_CH_TILE_STRIDE 66u /* u16 per row of phase D's tile: 64 sequences + 2 (33 words: a column read by 64 lanes hits every bank twice) */

/* the successor's index inside a chain key (length + 1 << 48 | score << 16 | this): the largest key wins, so the field is 0xFFFF - b when equal
   length and score go to the SMALLEST successor (the default) and b itself when they go to the largest (cw_policy.h CW_CHAIN_TIE) */
#define CW_CH_BFIELD(b) (CW_CHAIN_TIE == CW_CHAIN_TIE_LARGEST_SUCCESSOR ? (uint32_t)(b) : 0xFFFFu - (uint32_t)(b))
__device__ __forceinline__ int ch_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t ch_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int SLAB>
__global__ void __launch_bounds__(64 * CW_CH_WAVES) cw_chain_kernel(DevBatch b, DevScratch sc, cw_params prm) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* slab = lds + (size_t)wave * SLAB;
    const uint32_t k = prm.k;

    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(&sc.ctr->next_chain, 1u);
        w = (uint32_t)cw_lane_value((int)w, 0);
        if (w >= b.n_windows) break;
        WinInfo* wi = &sc.win[w];
        /* one way through the body: every window ends at the status store + wave barrier at the bottom */
        const bool ready = ch_uni(wi->status) == CW_WIN_CONSENSUS; /* the index kernel got to the end of this window */
        uint32_t new_status = 0xFFFFFFFFu; /* unchanged */
        uint32_t why = 0;
        uint32_t n_segs_out = 0, arena_used = 0;
        CW_PROF_T0();
        /* more anchors than this instance's slab holds (a template beyond ~1240 bases in an engine that was not configured for long templates, cw_configure):
           12 bytes per anchor, and phase D's tile behind the first 8 of them */
        bool fits_slab = true;
        if (ready) {
            const uint32_t A0 = ch_uni(((const uint32_t*)(sc.ablock + ((size_t)ch_uni(wi->ab_base) << 4)))[0]);
            const uint32_t oc = (8u * A0 + 2u + 3u) & ~3u, ov = (oc + 4u * A0 + 7u) & ~7u;
            fits_slab = ov <= (uint32_t)(SLAB - CW_CH_LIST_BYTES) && oc + 65u * CW_CH_TILE_STRIDE * 2u + 4u + 256u <= (uint32_t)(SLAB - CW_CH_LIST_BYTES);
            if (!fits_slab) { new_status = CW_WIN_OVERFLOW; why = CW_WHY_ANCHORS; }
        }
        if (ready && fits_slab) {
            const uint32_t s0 = ch_uni(b.win_first_seq[w]);
            const uint8_t* blk = sc.ablock + ((size_t)ch_uni(wi->ab_base) << 4);
            const uint32_t* hdr = (const uint32_t*)blk;
            const uint32_t A = ch_uni(hdr[0]), N = ch_uni(hdr[1]), n_dirty = ch_uni(hdr[2]);
            const uint32_t ab_flags = ch_uni(hdr[3]);
            const bool use_bits = (ab_flags & 1u) != 0u;
            const bool has_bm = (ab_flags & 2u) != 0u; /* bad-anchor masks: the dirty sequences are in the presence bits except at their out-of-order anchors */
            const uint32_t n_rows = (ab_flags & 4u) ? ch_uni(hdr[4]) : 0u; /* correction rows: what those anchors add to a pair's score, ready-made */
            const uint32_t Ap = cw_ab_ap(A);
            const uint32_t Np = cw_ab_np(N), Nw = (N + 63u) >> 6;
            const uint32_t* ckey = (const uint32_t*)(blk + CW_AB_HDR);
            const unsigned long long* gpres = (const unsigned long long*)((const uint8_t*)ckey + cw_ab_align((uint64_t)A * 4));
            const uint16_t* gdirty = (const uint16_t*)((const uint8_t*)gpres + cw_ab_align((uint64_t)A * Nw * 8));
            const unsigned long long* gbadm = (const unsigned long long*)((const uint8_t*)gdirty + cw_ab_align((uint64_t)n_dirty * 2));
            const uint8_t* growid = (const uint8_t*)gbadm + cw_ab_align((uint64_t)A * 8);
            const uint8_t* gdelta = growid + cw_ab_align((uint64_t)A);
            const uint16_t* P = (const uint16_t*)(n_rows ? gdelta + (size_t)n_rows * Ap : growid);
            const int sup_min = min((int)prm.common_kmers, (int)N / 2); /* correctionMSA.cpp:31 */
            if (lane == 0) { atomicAdd(&sc.ctr->prof[50], (ab_flags & 4u) ? 1ull : 0ull); atomicAdd(&sc.ctr->prof[51], (unsigned long long)n_rows); atomicAdd(&sc.ctr->prof[52], has_bm ? 1ull : 0ull); }
            if (lane == 0) { atomicAdd(&sc.ctr->prof[42], (unsigned long long)A); atomicAdd(&sc.ctr->prof[43], (unsigned long long)n_dirty); atomicAdd(&sc.ctr->prof[44], 1ull); }
            const uint32_t seg_base = ch_uni(wi->seg_base), seg_cap = ch_uni(wi->seg_cap);
            const uint32_t arena_base = ch_uni(wi->arena_base), arena_cap = ch_uni(wi->arena_cap);

            /* LDS carve */
            int16_t* clen = (int16_t*)slab;                    /* A     */
            int16_t* cnxt = clen + A;                          /* A     */
            int16_t* smax = cnxt + A;                          /* A + 1 */
            uint16_t* chain = (uint16_t*)(smax + A + 1);       /* A     */
            /* offsets, not pointer arithmetic through integers: the compiler must keep seeing LDS pointers (ds_ instead of flat_ instructions) */
            const uint32_t off_csc = (8u * A + 2u + 3u) & ~3u, off_var = (off_csc + 4u * A + 7u) & ~7u;
            int32_t* csc = (int32_t*)(slab + off_csc);         /* A     */
            uint8_t* var = slab + off_var;
            /* the last CW_CH_LIST_BYTES of the slab: segments waiting for the whole wave (phase D), 64 entries */
            uint32_t* q_off = (uint32_t*)(slab + SLAB - CW_CH_LIST_BYTES);   /* arena offset            */
            uint32_t* q_need = q_off + 64;                                           /* arena bytes reserved    */
            uint16_t* q_seg = (uint16_t*)(q_need + 64);
            int16_t* q_ca = (int16_t*)(q_seg + 64);
            int16_t* q_cb = q_ca + 64;
            uint16_t* q_n = (uint16_t*)(q_cb + 64);
            uint16_t* q_mx = q_n + 64;
            uint16_t* q_fs = q_mx + 64;
            uint16_t* q_fst = q_fs + 64;
            uint32_t* q_sl = (uint32_t*)(q_fst + 64);                               /* sum of the members' lengths */
            const size_t var_room = (size_t)(SLAB - CW_CH_LIST_BYTES) - off_var;
            /* behind the presence bitsets: the correction rows (row ids + rows) when the block has them and they fit */
            const size_t pres_only = use_bits ? (size_t)A * Nw * 8 : 0;
            const size_t delta_bytes = (size_t)Ap + (size_t)n_rows * Ap;
            const bool pres_lds = use_bits && pres_only <= var_room;
            const bool use_delta = pres_lds && n_rows > 0u && pres_only + delta_bytes <= var_room;
            const bool far_delta = pres_lds && n_rows > 0u && !use_delta && pres_only + Ap <= var_room; /* row ids in LDS, the rows stay in the block */
            unsigned long long* lpres = (unsigned long long*)var;
            const uint8_t* lrowid = (const uint8_t*)(lpres + (size_t)A * Nw); /* anchor -> correction row or 0xFF */
            const uint8_t* ldelta = lrowid + Ap;                              /* row r: ldelta[r * Ap + other anchor] */
            if (pres_lds) for (uint32_t i = lane; i < A * Nw; i += 64) lpres[i] = gpres[i];
            if (use_delta || far_delta) {
                unsigned long long* dst = lpres + (size_t)A * Nw;
                const unsigned long long* src = (const unsigned long long*)growid; /* row ids and rows are contiguous, 16-byte granules */
                for (uint32_t i = lane; i < (use_delta ? delta_bytes : (size_t)Ap) / 8; i += 64) dst[i] = src[i];
            }
            if (lane == 0) smax[A] = -1;
            cw_wave_sync();
            CW_PROF(sc.ctr, 48, lane == 0);

            /* ================= phase C: chain ================= */
            /* The recurrence is serial over the anchors, so its cost is the latency of one step.  In the usual case (presence bitsets in
               LDS, at most 256 sequences, no dirty sequence or correction rows for them) the 64 nearest successors of the current anchor
               -- all that is looked at unless the early stop fails -- are a register window: lane l holds length, score, presence and
               correction row of anchor a+1+l, shifted by one lane per step (DPP), so that a step neither waits for lane 0's LDS writes of
               the step before nor re-reads what it already had; what a step needs from LDS besides is requested one step ahead. */
            const bool narrow = (uint64_t)A * N < (1ull << 21) && A < 2047u; /* (length + 1, score) fit 11 + 21 bits: a chain's score is at most its length x N */
            auto wave_best = [&](unsigned long long key, uint32_t b0) -> unsigned long long {
                if (!narrow) return cw_wave_max_u64(key);
                const uint32_t hi = key ? (((uint32_t)(key >> 48) << 21) | (uint32_t)((key >> 16) & 0x1FFFFFull)) : 0u;
                const uint32_t mx = (uint32_t)cw_lane_value((int)cw_wave_scan_max_u32(hi), 63); /* one fused 32-bit prefix max */
                if (!mx) return 0ull;
                const unsigned long long who = __ballot(hi == mx); /* the first lane holding it = the smallest b (the last lane: the largest, CW_CHAIN_TIE) */
                const uint32_t fb = b0 + (CW_CHAIN_TIE == CW_CHAIN_TIE_LARGEST_SUCCESSOR ? 63u - (uint32_t)__clzll((long long)who) : (uint32_t)(__ffsll((long long)who) - 1));
                return ((unsigned long long)(mx >> 21) << 48) | ((unsigned long long)(mx & 0x1FFFFFu) << 16) | (unsigned long long)CW_CH_BFIELD(fb);
            };
            /* the window path, compiled once per (words of presence, correction rows or not): everything it tests is then a constant */
            auto chain_window = [&](auto nw_tag, auto delta_tag) {
                constexpr uint32_t NW = decltype(nw_tag)::value;
                constexpr uint32_t DELTA = decltype(delta_tag)::value; /* 0 no dirty sequence, 1 correction rows in LDS, 2 in the block (too many for LDS) */
                auto row_at = [&](const uint32_t row, const uint32_t col) -> uint32_t {
                    if constexpr (DELTA == 2u) return (uint32_t)gdelta[(size_t)row * Ap + col];
                    else return (uint32_t)ldelta[row * Ap + col];
                };
                int r_len = 0, r_sc = 0, sm_run = -1;
                unsigned long long r_p[NW];
                unsigned long long pa_n[NW];
#pragma unroll
                for (uint32_t x = 0; x < NW; ++x) { r_p[x] = 0ull; pa_n[x] = A > 0 ? lpres[(size_t)(A - 1u) * NW + x] : 0ull; }
                int r_row = 0xFF;        /* correction row of anchor a + 1 + lane */
                uint32_t d_b = 0;        /* row(bb)[a] of this step, requested during the step before */
                uint32_t row_n = DELTA && A > 0 ? (uint32_t)lrowid[A - 1u] : 0xFFu;
                int sm_n = 0;
                for (int a = (int)A - 1; a >= 0; --a) {
                    unsigned long long pa[NW];
#pragma unroll
                    for (uint32_t x = 0; x < NW; ++x) pa[x] = pa_n[x];
                    const uint32_t row_a = row_n;
                    const uint32_t d_b_cur = d_b;
                    const int sm_far = sm_n; /* smax[a + 65] when that anchor exists */
                    const int r_row_next = DELTA ? cw_wave_shr1(r_row, (int)row_a) : 0xFF; /* the window as the next step sees it */
                    if (a > 0) {
#pragma unroll
                        for (uint32_t x = 0; x < NW; ++x) pa_n[x] = lpres[(size_t)(a - 1) * NW + x];
                        if (DELTA) {
                            row_n = (uint32_t)lrowid[a - 1];
                            d_b = r_row_next != 0xFF ? row_at((uint32_t)r_row_next, (uint32_t)(a - 1)) : 0u;
                        }
                        sm_n = (uint32_t)a + 64u < A ? (int)smax[a + 64] : 0;
                    }
                    unsigned long long best = 0ull;
                    bool stop = false;
                    {   /* successors a+1 .. a+64: from the window */
                        const uint32_t b0 = (uint32_t)a + 1u, bb = b0 + (uint32_t)lane;
                        unsigned long long key = 0ull;
                        if (bb < A) {
                            uint32_t cnt = 0;
#pragma unroll
                            for (uint32_t x = 0; x < NW; ++x) cnt += (uint32_t)__popcll(pa[x] & r_p[x]);
                            if (DELTA) { cnt += d_b_cur; if (row_a != 0xFFu) cnt += row_at(row_a, bb); }
                            if ((int)cnt >= sup_min)
                                key = ((unsigned long long)((uint32_t)r_len + 1u) << 48) | ((unsigned long long)((uint32_t)r_sc + cnt) << 16) |
                                      (unsigned long long)CW_CH_BFIELD(bb);
                        }
                        best = wave_best(key, b0);
                        if (best != 0ull && b0 + 64 < A) stop = sm_far < (int)(best >> 48) - 1;
                        stop = ch_uni(stop ? 1 : 0) != 0u;
                    }
                    if (!stop) {
                        for (uint32_t b0 = (uint32_t)a + 65u; b0 < A; b0 += 64) { /* farther successors: from LDS (written at least 64 steps ago) */
                            const uint32_t bb = b0 + (uint32_t)lane;
                            unsigned long long key = 0ull;
                            if (bb < A) {
                                uint32_t cnt = 0;
#pragma unroll
                                for (uint32_t x = 0; x < NW; ++x) cnt += (uint32_t)__popcll(pa[x] & lpres[(size_t)bb * NW + x]);
                                if (DELTA) {
                                    const uint32_t row_b = lrowid[bb];
                                    if (row_a != 0xFFu) cnt += row_at(row_a, bb);
                                    if (row_b != 0xFFu) cnt += row_at(row_b, (uint32_t)a);
                                }
                                if ((int)cnt >= sup_min)
                                    key = ((unsigned long long)((uint32_t)clen[bb] + 1u) << 48) | ((unsigned long long)((uint32_t)csc[bb] + cnt) << 16) |
                                          (unsigned long long)CW_CH_BFIELD(bb);
                            }
                            key = wave_best(key, b0);
                            best = key > best ? key : best;
                            bool st2 = false;
                            if (best != 0ull && b0 + 64 < A) st2 = (int)smax[b0 + 64] < (int)(best >> 48) - 1;
                            if (ch_uni(st2 ? 1 : 0)) break;
                        }
                    }
                    const int la = best == 0ull ? 0 : (int)(best >> 48); /* stored length + 1 of b == length of a */
                    const int sca = best == 0ull ? 0 : (int)((best >> 16) & 0xFFFFFFFFull);
                    sm_run = la > sm_run ? la : sm_run;
                    if (lane == 0) {
                        clen[a] = (int16_t)la; csc[a] = sca;
                        cnxt[a] = best == 0ull ? (int16_t)-1 : (int16_t)CW_CH_BFIELD((uint32_t)(best & 0xFFFFull));
                        smax[a] = (int16_t)sm_run;
                    }
                    /* slide the window: lane l takes lane l-1, lane 0 takes anchor a */
                    r_len = cw_wave_shr1(r_len, la);
                    r_sc = cw_wave_shr1(r_sc, sca);
#pragma unroll
                    for (uint32_t x = 0; x < NW; ++x) {
                        const int lo_ = cw_wave_shr1((int)(uint32_t)r_p[x], (int)(uint32_t)pa[x]);
                        const int hi_ = cw_wave_shr1((int)(uint32_t)(r_p[x] >> 32), (int)(uint32_t)(pa[x] >> 32));
                        r_p[x] = ((unsigned long long)(uint32_t)hi_ << 32) | (uint32_t)lo_;
                    }
                    r_row = r_row_next;
                }
                cw_wave_sync();
            };
            const bool fast_path = pres_lds && Nw <= 4u && (n_dirty == 0u || use_delta || far_delta);
            const unsigned long long _c0 = __builtin_readcyclecounter();
            if (fast_path) {
                const uint32_t dm = use_delta ? 1u : far_delta ? 2u : 0u;
#define CW_CH_CASE(NWV)                                                                                                 \
                    if (dm == 0u) chain_window(std::integral_constant<uint32_t, NWV>{}, std::integral_constant<uint32_t, 0>{});        \
                    else if (dm == 1u) chain_window(std::integral_constant<uint32_t, NWV>{}, std::integral_constant<uint32_t, 1>{});   \
                    else chain_window(std::integral_constant<uint32_t, NWV>{}, std::integral_constant<uint32_t, 2>{});
                if (Nw == 1u) { CW_CH_CASE(1) } else if (Nw == 2u) { CW_CH_CASE(2) } else if (Nw == 3u) { CW_CH_CASE(3) } else { CW_CH_CASE(4) }
#undef CW_CH_CASE
            } else
            /* everything else (deep polishing piles, more dirty sequences than a mask holds, rows that do not fit): from the block in place */
            for (int a = (int)A - 1; a >= 0; --a) {
                unsigned long long best = 0ull;
                const uint32_t* pa_row = (const uint32_t*)(P + (uint32_t)a * Np);
                const uint32_t half = Np >> 1; /* pairs of sequences; padding entries are CW_NONE16 and never count */
                for (uint32_t b0 = (uint32_t)a + 1u; b0 < A; b0 += 64) {
                    const uint32_t bb = b0 + (uint32_t)lane;
                    unsigned long long key = 0ull;
                    if (bb < A) {
                        uint32_t cnt = 0;
                        if (use_bits) {
                            if (pres_lds) { for (uint32_t x = 0; x < Nw; ++x) cnt += (uint32_t)__popcll(lpres[(size_t)a * Nw + x] & lpres[(size_t)bb * Nw + x]); }
                            else { for (uint32_t x = 0; x < Nw; ++x) cnt += (uint32_t)__popcll(gpres[(size_t)a * Nw + x] & gpres[(size_t)bb * Nw + x]); }
                            if (n_rows) { /* correction rows, in the block */
                                const uint32_t row_a = growid[a], row_b = growid[bb];
                                if (row_a != 0xFFu) cnt += gdelta[(size_t)row_a * Ap + bb];
                                if (row_b != 0xFFu) cnt += gdelta[(size_t)row_b * Ap + (uint32_t)a];
                            } else if (has_bm) { /* bad masks: the dirty sequences that are out of order at a or at bb */
                                unsigned long long mm = gbadm[a] | gbadm[bb];
                                while (mm) {
                                    const uint32_t d = (uint32_t)__ffsll((long long)mm) - 1u;
                                    mm &= mm - 1ull;
                                    const uint32_t sd = gdirty[d];
                                    const uint32_t pa_ = P[(uint32_t)a * Np + sd], pb = P[bb * Np + sd];
                                    cnt += (pa_ < pb && pb != CW_NONE16) ? 1u : 0u;
                                }
                            } else { /* every dirty sequence */
                                for (uint32_t d = 0; d < n_dirty; ++d) {
                                    const uint32_t sd = gdirty[d];
                                    const uint32_t pa_ = P[(uint32_t)a * Np + sd], pb = P[bb * Np + sd];
                                    cnt += (pa_ < pb && pb != CW_NONE16) ? 1u : 0u;
                                }
                            }
                        } else {
                            const uint32_t* pb_row = (const uint32_t*)(P + bb * Np);
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
                                  (unsigned long long)CW_CH_BFIELD(bb);
                    }
                    key = cw_wave_max_u64(key);
                    best = key > best ? key : best;
                    bool stop = false;
                    if (best != 0ull && b0 + 64 < A) {
                        const int blen = (int)(best >> 48) - 1;
                        stop = (int)smax[b0 + 64] < blen;
                    }
                    if (ch_uni(stop ? 1 : 0)) break;
                }
                if (lane == 0) {
                    int la = 0;
                    if (best == 0ull) { clen[a] = 0; csc[a] = 0; cnxt[a] = -1; }
                    else {
                        la = (int)(best >> 48); /* stored length+1 of b == length of a */
                        clen[a] = (int16_t)la;
                        csc[a] = (int32_t)((best >> 16) & 0xFFFFFFFFull);
                        cnxt[a] = (int16_t)CW_CH_BFIELD((uint32_t)(best & 0xFFFFull));
                    }
                    const int sm = smax[a + 1];
                    smax[a] = (int16_t)(la > sm ? la : sm);
                }
                cw_wave_sync();
            }
            if (!fast_path && lane == 0) { atomicAdd(&sc.ctr->prof[53], 1ull); atomicAdd(&sc.ctr->prof[54], __builtin_readcyclecounter() - _c0); }
            /* chain start: longest, then best score, then largest index; a chain needs at least one edge */
            uint32_t m = 0;
            {
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
                b_len = ch_uni(b_len); b_a = ch_uni(b_a);
                if (b_a >= 0 && b_len > 0) {
                    int a = b_a;
                    while (a != -1) {
                        if (lane == 0) chain[m] = (uint16_t)a;
                        m++;
                        a = ch_uni((int)cnxt[a]);
                    }
                }
                cw_wave_sync();
            }
            CW_PROF(sc.ctr, 5, lane == 0);

            if (m == 0 || m < prm.min_anchors) {
                new_status = CW_WIN_TEMPLATE;
            } else if (m + 1 > seg_cap) {
                new_status = CW_WIN_OVERFLOW; why = CW_WHY_SEGMENTS;
            } else {
                /* ================= phase D: segments =================
                   Lanes = segments, 64 at a time.  Each lane walks the pile once (two matrix reads per sequence, no stores
                   in between, so the loads pipeline), then writes its own segment if it is trivial: empty, a prefix of its
                   left anchor (all pieces equal and no longer than k), or a single short piece.  The few segments that need
                   the whole wave -- POA tasks and long single pieces -- follow one by one. */
#if CW_SEG_MISSING_ANCHOR == CW_SEG_MISSING_ANCHOR_EXTRAPOLATE
                /* cw_policy.h CW_SEG_MISSING_ANCHOR_EXTRAPOLATE (a policy build, round 6): the chain anchors a sequence lacks get the position the template's
                   spacing gives them, counted from the nearest chain anchor the sequence holds (the one before, else the one after), written into the
                   block's position matrix in place -- everything below then cuts the segments as ever.  Lanes = sequences; the template (sequence 0)
                   holds every anchor at its template position. */
                {
                    uint16_t* Pw = const_cast<uint16_t*>(P);
                    for (uint32_t sb = 0; sb < N; sb += 64) {
                        const uint32_t s = sb + (uint32_t)lane;
                        const bool in = s < N;
                        const int top = in ? (int)min(b.seq_len[s0 + s], 65534u) : 0;
                        int last_p = -1, last_t = 0, first_i = -1, first_p = 0, first_t = 0;
                        for (uint32_t i = 0; i < m; ++i) {
                            const uint32_t a = chain[i];
                            const int t_i = (int)P[a * Np];
                            const uint32_t pv = in ? (uint32_t)P[a * Np + s] : (uint32_t)CW_NONE16;
                            if (pv != CW_NONE16) { last_p = (int)pv; last_t = t_i; if (first_i < 0) { first_i = (int)i; first_p = (int)pv; first_t = t_i; } }
                            else if (in && last_p >= 0) Pw[a * Np + s] = (uint16_t)min(top, last_p + (t_i - last_t));
                        }
                        const int lead = cw_wave_max(in ? first_i : -1); /* the longest run of leading anchors any lane has to fill */
                        for (int i = 0; i < lead; ++i) {
                            const uint32_t a = chain[i];
                            const int t_i = (int)P[a * Np];
                            if (in && i < first_i) Pw[a * Np + s] = (uint16_t)min(top, max(0, first_p - (first_t - t_i)));
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); /* the rows are read back below by other lanes of this wave, through the vector cache */
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    cw_wave_sync();
                }
#endif
                bool over = false, over_arena = false; /* (over_arena is wave-uniform: decided from wave-uniform totals) */
                uint32_t q_cnt = 0;
                unsigned long long t_flush = 0;
                uint16_t* d_tile = (uint16_t*)(slab + off_csc);                         /* 65 rows x 66 u16 */
                uint32_t* d_len = (uint32_t*)(slab + off_csc + 65u * CW_CH_TILE_STRIDE * 2u + 4u); /* 64 sequence lengths */
                auto flush = [&]() {
                    const unsigned long long _f0 = __builtin_readcyclecounter();
                    const bool e = (uint32_t)lane < q_cnt;
                    const uint32_t e_n = e ? q_n[lane] : 0u, e_mx = e ? q_mx[lane] : 0u;
                    const bool poa = e && e_n > 1u;
                    /* route by the expected graph size: the graph has at least max_len nodes once its longest member is in
                       and typically ends at 1.4-1.6x that; a task that still outgrows its tier is redone in the next one */
                    const uint32_t est = (e_mx * 17u + 9u) / 10u;
                    /* deep piles grow wider graphs: the smallest tier is only worth trying when the graph will very likely stay in it
                       (a task that outgrows tier S is redone in tier L, the scarcest one) */
                    const uint32_t est_s = (e_mx * (15u + e_n / 5u) + 9u) / 10u;
                    /* tier Q (four tasks per wave, cw_poa_q.h): members of at most 31 bases and a graph that should stay small */
                    /* tier H (two tasks per wave, cw_poa_q.h): members of up to 63 bases, graph expected (depth-aware) to stay inside its 128 nodes */
                    const bool fits_h = sc.use_h != 0u && e_mx <= (uint32_t)CW_POAH_LC && e_mx >= sc.h_min_len && est_s <= (uint32_t)CW_POAH_ROUTE_NODES;
                    const bool fits_s = est_s <= sc.s_route_cells && e_mx <= (uint32_t)CW_POA_LC; /* s_route_cells: a node count since round 4 */
                    uint32_t tier = !poa ? 0xFFu
                                          : (sc.use_q && e_mx <= (uint32_t)CW_POAQ_LC && est_s <= (uint32_t)CW_POAQ_ROUTE_NODES) ? 4u
                                          : (fits_h && (sc.use_h > 1u || !fits_s)) ? 5u
                                          : fits_s ? 0u
                                          : ((sc.m1_route_depth && est_s > est ? est_s : est) <= (uint32_t)CW_POAM1_ROUTE && e_mx <= (uint32_t)CW_POAM1_LC) ? 1u
                                          : (est <= (uint32_t)CW_POAM2_ROUTE && e_mx <= (uint32_t)CW_POAM2_LC) ? 2u
                                                                                            : 3u;
                    /* tier LW (round 6, cw_poa_w.h): a tier-L task whose members are wide ON AVERAGE (several 128-column chunks per DP row) and many runs
                       on the four waves of a work-group -- list 5, which the product build has free (tier H is a test aid; with it on, no tier LW) */
                    if (tier == 3u && sc.use_lw && e_n >= (uint32_t)CW_POALW_MIN_MEMBERS && q_sl[lane] / e_n >= (uint32_t)CW_POALW_MIN_MEAN) tier = 5u;
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const unsigned long long pm = __ballot(poa);
                    const unsigned long long tm1 = __ballot(tier == 1u), tm2 = __ballot(tier == 2u), tm3 = __ballot(tier == 3u), tmq = __ballot(tier == 4u), tmh = __ballot(tier == 5u);
                    const int minc = cw_wave_scan_add(poa ? (int)e_n : 0);
                    const uint32_t m_total = (uint32_t)cw_lane_value(minc, 63);
                    uint32_t tb = 0, mb = 0, b1 = 0, b2 = 0, b3 = 0, bq = 0, bh = 0;
                    if (lane == 0 && pm) {
                        tb = atomicAdd(&sc.ctr->n_tasks, (uint32_t)__popcll(pm));
                        mb = atomicAdd(&sc.ctr->n_members, m_total);
                        if (tm1) b1 = atomicAdd(&sc.ctr->n_tier[1], (uint32_t)__popcll(tm1));
                        if (tm2) b2 = atomicAdd(&sc.ctr->n_tier[2], (uint32_t)__popcll(tm2));
                        if (tm3) b3 = atomicAdd(&sc.ctr->n_tier[3], (uint32_t)__popcll(tm3));
                        if (tmq) bq = atomicAdd(&sc.ctr->n_tier[0], (uint32_t)__popcll(tmq));
                        if (tmh) bh = atomicAdd(&sc.ctr->n_tier[5], (uint32_t)__popcll(tmh));
                    }
                    tb = (uint32_t)cw_lane_value((int)tb, 0); mb = (uint32_t)cw_lane_value((int)mb, 0);
                    b1 = (uint32_t)cw_lane_value((int)b1, 0); b2 = (uint32_t)cw_lane_value((int)b2, 0); b3 = (uint32_t)cw_lane_value((int)b3, 0);
                    bq = (uint32_t)cw_lane_value((int)bq, 0); bh = (uint32_t)cw_lane_value((int)bh, 0);
                    const bool cap_ok = (uint64_t)tb + (uint32_t)__popcll(pm) <= sc.task_cap && (uint64_t)mb + m_total <= sc.member_cap &&
                                        b1 + (uint32_t)__popcll(tm1) <= sc.list_cap && b2 + (uint32_t)__popcll(tm2) <= sc.list_cap &&
                                        b3 + (uint32_t)__popcll(tm3) <= sc.list_cap && bq + (uint32_t)__popcll(tmq) <= sc.list_cap &&
                                        bh + (uint32_t)__popcll(tmh) <= sc.list_cap;
                    if (!cap_ok) {
                        /* The counters have advanced and are never rolled back; consumers clamp them to the capacities and walk every slot
                           below.  Whatever this flush reserved inside a capacity is therefore given a neutral content -- a finished task
                           without members, list entries naming the batch's permanent neutral task (tasks[task_cap], cw_setup_kernel) -- or the
                           tier kernels would run the records a previous batch left there. */
                        over = true;
                        const uint32_t n_t = (uint32_t)__popcll(pm);
                        for (uint32_t x = lane; x < n_t; x += 64)
                            if ((uint64_t)tb + x < sc.task_cap) { PoaTask t; t.window = w; t.seg_slot = seg_base; t.member_off = 0; t.n_members = 0; t.max_len = 0; t.out_off = 0; t.out_cap = 0; t.state = 1u; sc.tasks[tb + x] = t; }
                        const uint32_t lb[5] = {bq, b1, b2, b3, bh};
                        const unsigned long long lm[5] = {tmq, tm1, tm2, tm3, tmh};
                        const int lt[5] = {0, 1, 2, 3, 5};
                        for (int li = 0; li < 5; ++li) {
                            const uint32_t n_l = (uint32_t)__popcll(lm[li]);
                            for (uint32_t x = lane; x < n_l; x += 64)
                                if ((uint64_t)lb[li] + x < sc.list_cap) sc.tier_list[lt[li]][lb[li] + x] = sc.task_cap;
                        }
                    }
                    else {
                        const uint32_t t_idx = tb + (uint32_t)__popcll(pm & below);
                        const uint32_t m_off = mb + (uint32_t)minc - (poa ? e_n : 0u);
                        if (poa) { /* task records: one lane each */
                            PoaTask t;
                            t.window = w; t.seg_slot = seg_base + q_seg[lane]; t.member_off = m_off; t.n_members = e_n;
                            t.max_len = e_mx | ((e_n ? q_sl[lane] / e_n : 0u) << 16); /* longest member | mean member length: what the tier sort goes by */
                            t.out_off = q_off[lane]; t.out_cap = q_need[lane];
                            t.state = tier ? 2u : 0u; /* 0 = tier S takes it from the task array; anything else is on a list */
                            sc.tasks[t_idx] = t;
                            if (tier == 1u) sc.tier_list[1][b1 + (uint32_t)__popcll(tm1 & below)] = t_idx;
                            else if (tier == 2u) sc.tier_list[2][b2 + (uint32_t)__popcll(tm2 & below)] = t_idx;
                            else if (tier == 3u) sc.tier_list[3][b3 + (uint32_t)__popcll(tm3 & below)] = t_idx;
                            else if (tier == 4u) sc.tier_list[0][bq + (uint32_t)__popcll(tmq & below)] = t_idx;
                            else if (tier == 5u) sc.tier_list[5][bh + (uint32_t)__popcll(tmh & below)] = t_idx;
                            sc.seg_off[t.seg_slot] = t.out_off; sc.seg_len[t.seg_slot] = 0;
                        }
                        /* the wave-wide part, entry by entry: member lists (coalesced matrix rows) and long single pieces */
                        for (uint32_t q = 0; q < q_cnt; ++q) {
                            const uint32_t g_seg = q_seg[q], g_n = q_n[q], g_mx = q_mx[q], g_off = q_off[q];
                            const int g_ca = q_ca[q], g_cb = q_cb[q];
                            if (g_n == 1u) {
                                const uint32_t* words = b.bases + b.seq_word_off[s0 + q_fs[q]];
                                const uint32_t fst = q_fst[q];
                                for (uint32_t i = lane; i < g_mx; i += 64) sc.arena[g_off + i] = CW_ACGT(cw_base_at(words, fst + i));
                                if (lane == 0) { sc.seg_off[seg_base + g_seg] = g_off; sc.seg_len[seg_base + g_seg] = g_mx; }
                            } else {
                                const uint32_t g_moff = (uint32_t)cw_lane_value((int)m_off, (int)q);
                                uint32_t done = 0;
                                for (uint32_t sb = 0; sb < N && done < g_n; sb += 64) {
                                    const uint32_t s = sb + lane;
                                    bool is = false;
                                    uint32_t st = 0, ln = 0;
                                    if (s < N) {
                                        const uint32_t pa = g_ca >= 0 ? (uint32_t)P[(uint32_t)g_ca * Np + s] : 0u;
                                        const uint32_t pb = g_cb >= 0 ? (uint32_t)P[(uint32_t)g_cb * Np + s] : 0u;
                                        if (g_seg == 0) { is = pb != CW_NONE16 && pb > 0; st = 0; ln = pb; }
                                        else if (g_seg == m) { const uint32_t sl_ = b.seq_len[s0 + s]; is = pa != CW_NONE16 && pa < sl_; st = pa; ln = sl_ - pa; } /* (pa < length: always, unless the position is an extrapolated one) */
                                        else { is = pa != CW_NONE16 && pb != CW_NONE16 && pa < pb; st = pa; ln = pb - pa; }
                                    }
                                    const unsigned long long bal = __ballot(is);
                                    const uint32_t idx = done + (uint32_t)__popcll(bal & below);
                                    if (is && idx < g_n) {
                                        PoaMember pmb;
                                        pmb.seq = s0 + s; pmb.start = (uint16_t)st; pmb.len = (uint16_t)ln;
                                        sc.members[g_moff + idx] = pmb;
                                    }
                                    done += (uint32_t)__popcll(bal);
                                }
                            }
                            cw_wave_sync();
                        }
                    }
                    q_cnt = 0;
                    cw_wave_sync();
                    t_flush += __builtin_readcyclecounter() - _f0;
                };
                for (uint32_t seg0 = 0; seg0 <= m && !over; seg0 += 64) {
                    const uint32_t seg = seg0 + (uint32_t)lane;
                    const bool valid = seg <= m;
                    const int ca = (valid && seg > 0) ? (int)chain[seg - 1] : -1;
                    const int cb = (valid && seg < m) ? (int)chain[seg] : -1;
                    uint32_t n_mem = 0, mn = 0xFFFFFFFFu, mx = 0, first_seq = 0, first_start = 0, sl = 0;
                    /* A lane needs the rows of its two chain anchors, all N sequences of them: read per lane that is two scattered 2-byte loads per
                       sequence (64 cache lines per instruction).  Instead the 65 rows of this round's chain anchors go through an LDS tile 64
                       sequences at a time -- a row is one coalesced 128-byte load, lanes = sequences -- and the lanes read their two rows back
                       column-wise (row stride 33 words: conflict-free).  The tile lies over the chain DP arrays, which are dead by now. */
                    const int cb63 = cw_lane_value(cb, 63);
                    for (uint32_t sb = 0; sb < N; sb += 64) {
                        const uint32_t ns = min(64u, N - sb);
                        const bool s_in = (uint32_t)lane < ns;
#pragma unroll 1
                        for (uint32_t r0 = 0; r0 < 66u; r0 += 33u) { /* 33 loads in flight: the wave has the registers (two waves per SIMD) */
                            uint32_t v[33];
#pragma unroll
                            for (uint32_t q = 0; q < 33u; ++q) {
                                const uint32_t r = r0 + q;
                                const int an = r < 64u ? __builtin_amdgcn_readlane(ca, (int)r) : r == 64u ? cb63 : -1; /* row r = left anchor of lane r's segment */
                                v[q] = (an >= 0 && s_in) ? (uint32_t)P[(uint32_t)an * Np + sb + (uint32_t)lane] : (uint32_t)CW_NONE16;
                            }
#pragma unroll
                            for (uint32_t q = 0; q < 33u; ++q) if (r0 + q < 65u) d_tile[(r0 + q) * CW_CH_TILE_STRIDE + (uint32_t)lane] = (uint16_t)v[q];
                        }
                        d_len[lane] = s_in ? b.seq_len[s0 + sb + (uint32_t)lane] : 0u;
                        cw_wave_sync();
                        const uint16_t* row_a = d_tile + (uint32_t)lane * CW_CH_TILE_STRIDE;
                        const uint16_t* row_b = row_a + CW_CH_TILE_STRIDE;
                        /* One rule for the three kinds of segment, without branches (a branch per kind made every LDS read of the loop wait on its
                           own): the piece of sequence s runs from pa (0 in front of the first anchor) to pb (the sequence's end behind the last
                           one; a k-mer starts before the end, so pa < pb there whenever pa is a hit) and exists iff both are hits and pa < pb. */
                        const bool seg_first = seg == 0u, seg_last = seg == m;
                        for (uint32_t j0 = 0; j0 < ns; j0 += 8u) {
                            uint32_t va[8], vb[8], sln[8];
#pragma unroll
                            for (uint32_t q = 0; q < 8u; ++q) { va[q] = row_a[j0 + q]; vb[q] = row_b[j0 + q]; sln[q] = d_len[(j0 + q) & 63u]; } /* rows are padded: reading past ns is harmless */
#pragma unroll
                            for (uint32_t q = 0; q < 8u; ++q) {
                                const uint32_t pa = seg_first ? 0u : va[q];
                                const uint32_t pb = seg_last ? sln[q] : vb[q];
                                const bool hit_b = seg_last || vb[q] != CW_NONE16;
                                const bool take = valid && j0 + q < ns && n_mem < prm.max_msa && pa != CW_NONE16 && hit_b && pa < pb;
                                const uint32_t ln = pb - pa;
                                const bool fst = take && n_mem == 0u;
                                first_seq = fst ? sb + j0 + q : first_seq;
                                first_start = fst ? pa : first_start;
                                n_mem += take ? 1u : 0u;
                                mn = take ? min(mn, ln) : mn;
                                mx = take ? max(mx, ln) : mx;
                                sl += take ? ln : 0u;
                            }
                        }
                        cw_wave_sync();
                    }
                    /* all pieces = first mx bases of anchor a (not under CW_SEG_MISSING_ANCHOR_EXTRAPOLATE: a piece may start where the anchor is only assumed to be) */
                    const bool by_anchor = CW_SEG_MISSING_ANCHOR == CW_SEG_MISSING_ANCHOR_DROP && n_mem > 0 && seg > 0 && seg < m && mn == mx && mx <= k;
                    const bool single = n_mem == 1;
                    const uint32_t need = n_mem == 0 ? 0u : (by_anchor || single) ? mx : 2 * mx + 2;
                    const int inc = cw_wave_scan_add((int)need);
                    const uint32_t total = (uint32_t)cw_lane_value(inc, 63);
                    if (arena_used + total > arena_cap) { over = true; over_arena = true; }
                    else {
                        const uint32_t abs_off = arena_base + arena_used + (uint32_t)inc - need;
                        const uint32_t slot = seg_base + seg;
                        bool serial = false;
                        if (valid) {
                            if (n_mem == 0) { sc.seg_off[slot] = arena_base; sc.seg_len[slot] = 0; }
                            else if (by_anchor) {
                                const uint32_t key = ckey[ca];
                                for (uint32_t i = 0; i < mx; ++i) sc.arena[abs_off + i] = CW_ACGT((key >> (2 * (k - 1 - i))) & 3u);
                                sc.seg_off[slot] = abs_off; sc.seg_len[slot] = mx;
                            } else if (single && mx <= 16u) {
                                const uint32_t* words = b.bases + b.seq_word_off[s0 + first_seq];
                                for (uint32_t i = 0; i < mx; ++i) sc.arena[abs_off + i] = CW_ACGT(cw_base_at(words, first_start + i));
                                sc.seg_off[slot] = abs_off; sc.seg_len[slot] = mx;
                            } else serial = true;
                        }
                        arena_used += total;
                        /* queue the segments that need the whole wave; the queue is emptied when it could overflow and
                           at the end of the window, so that the batch-wide counters see a handful of atomics per window
                           instead of three per task (same-address atomics from 2048 waves serialise in L2) */
                        const unsigned long long sm = __ballot(serial);
                        const uint32_t n_new = (uint32_t)__popcll(sm);
                        if (q_cnt + n_new > 64u) { flush(); }
                        if (serial) {
                            const uint32_t qi = q_cnt + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
                            q_off[qi] = abs_off; q_need[qi] = need; q_seg[qi] = (uint16_t)seg; q_ca[qi] = (int16_t)ca; q_cb[qi] = (int16_t)cb;
                            q_n[qi] = (uint16_t)n_mem; q_mx[qi] = (uint16_t)mx; q_fs[qi] = (uint16_t)first_seq; q_fst[qi] = (uint16_t)first_start; q_sl[qi] = sl;
                        }
                        q_cnt += n_new;
                        cw_wave_sync();
                    }
                    over = ch_uni(over ? 1 : 0) != 0;
                    cw_wave_sync();
                }
                if (!over && q_cnt) flush();
                if (lane == 0) atomicAdd(&sc.ctr->prof[49], t_flush);
                if (over) { new_status = CW_WIN_OVERFLOW; why = over_arena ? CW_WHY_ARENA : CW_WHY_TASKS; }
                else n_segs_out = m + 1;
            }
            CW_PROF(sc.ctr, 6, lane == 0);
        }
        new_status = ch_uni(new_status); n_segs_out = ch_uni(n_segs_out); arena_used = ch_uni(arena_used);
        if (lane == 0) {
            if (new_status != 0xFFFFFFFFu) { wi->status = new_status; if (new_status == CW_WIN_OVERFLOW) { wi->pad_ = why; sc.ctr->any_overflow = 1; } }
            else if (ready) { wi->n_segs = n_segs_out; wi->arena_used = arena_used; }
        }
        cw_wave_sync();
    }
}

#endif
