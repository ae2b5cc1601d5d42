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

typedef __attribute__((address_space(3))) uint32_t* cw_l32w;
typedef __attribute__((address_space(3))) const uint32_t* cw_l32; /* a pile's words staged in LDS ... */
typedef __attribute__((address_space(1))) const uint32_t* cw_g32; /* ... or where the batch has them */
#define CW_IDX_THREADS 1024
#define CW_IDX_WAVES 16
#define CW_IDX_LDS_BYTES 163840
#define CW_TMAX 2048 /* template k-mer slots (round 6: 2048 -- templates of up to 2048 + k - 1 bases, `-l 1500` runs; 1024 through round 5) */
#define CW_EX_SLOTS 1024 /* in LDS; a pile that saturates more keys than this is counted again with the table in global memory */
#define CW_EX_BITS 10
#define CW_EXP_SLOTS 8 /* solid keys a thread keeps in registers during the export of the count table; a thread that finds more walks its words again */
#define CW_EXG_SLOTS 262144 /* per-work-group exact table in global memory for piles so deep that more than CW_EX_SLOTS / 2 keys can saturate */
#define CW_TH_SLOTS 4096
#ifndef CW_IDX_BYTES
#ifndef CW_IDX_BYTES_RTN
#define CW_IDX_BYTES_RTN 0 /* 1: the byte counters of phase A with returning adds, as round 4 had them */
#endif
#define CW_IDX_BYTES 1 /* phase A counts in byte counters first (two halves of the key space for k = 9); 0 = the nibble table only */
#endif
#ifndef CW_IDX_BYTES_MIN_N
#define CW_IDX_BYTES_MIN_N 64u /* k = 9: from this many sequences on (below, few keys pass fifteen occurrences and the nibble table's single pass wins:
                                  depth 30 measured 3.11 ms against 3.32 ms per batch; depth 150: 10.9 against 9.7) */
#endif
/* the window's pile staged in LDS (behind the phase A tables, in front of nothing: the position matrix stops short of it): sequence
   lengths, word offsets and the 2-bit words themselves, so that the four passes over the pile's k-mers read LDS instead of walking
   seq_len -> seq_word_off -> bases in global memory (three dependent round trips per sequence and pass, with all 16 waves waiting
   at the same time) */
#define CW_IDX_STAGE_OFF 139776 /* behind the count table (128 KiB), the exact table (8 KiB) and 512 B of flags and scan scratch */
#define CW_IDX_STAGE_N 192
#define CW_IDX_STAGE_WORDS ((CW_IDX_LDS_BYTES - CW_IDX_STAGE_OFF - 16 - CW_IDX_STAGE_N * 8) / 4)

/* ------------------------------------------------------------------------------------------------ */
/* anchor block of one window (index kernel -> chain kernel): sizes in bytes, everything 16-byte aligned */
__host__ __device__ __forceinline__ uint32_t cw_ab_np(uint32_t N) { uint32_t Np = (N + 1u) & ~1u; if (((Np >> 1) & 1u) == 0u) Np += 2u; return Np; }
__host__ __device__ __forceinline__ uint64_t cw_ab_align(uint64_t x) { return (x + 15ull) & ~15ull; }
#define CW_AB_HDR 64u
#define CW_AB_ROWS_MAX 254u /* correction rows (anchors that are out of order in some dirty sequence) a block can carry */
__host__ __device__ __forceinline__ uint32_t cw_ab_ap(uint32_t A) { return (A + 15u) & ~15u; } /* bytes per correction row */
__host__ __device__ __forceinline__ uint64_t cw_ab_bytes(uint32_t A, uint32_t N, uint32_t n_dirty, uint32_t n_rows = 0) {
    const uint32_t Np = cw_ab_np(N), Nw = (N + 63u) >> 6;
    return CW_AB_HDR + cw_ab_align((uint64_t)A * 4) + cw_ab_align((uint64_t)A * Nw * 8) + cw_ab_align((uint64_t)n_dirty * 2) + cw_ab_align((uint64_t)A * 8) +
           (n_rows ? cw_ab_align((uint64_t)A) + (uint64_t)n_rows * cw_ab_ap(A) : 0ull) + cw_ab_align((uint64_t)A * Np * 2);
}

/* one wave per window: the pile's k-mer count (the sum over its sequences), read coalesced; the single work-group of cw_setup_kernel
   then only scans per-window numbers (walking the sequences there, one thread per window, took 1.9 ms of a depth-150 batch) */
__global__ void __launch_bounds__(256) cw_setup_need_kernel(DevBatch b, DevScratch sc, cw_params prm) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc.ctr->prof[57] = wall_clock64() - sc.step_clock[0]; /* 10 ns units since the batch before ended */
    if (w >= b.n_windows) return;
    const uint32_t s0 = b.win_first_seq[w], s1 = b.win_first_seq[w + 1];
    uint32_t nk = 0;
    for (uint32_t s = s0 + lane; s < s1; s += 64) { const uint32_t l = b.seq_len[s]; nk += l >= prm.k ? l - prm.k + 1 : 0; }
    nk = (uint32_t)cw_wave_sum((int)nk);
    if (lane == 0) { WinInfo* wi = &sc.win[w]; wi->n_seqs = s1 - s0; wi->tpl_len = s1 > s0 ? b.seq_len[s0] : 0; wi->n_kmers = nk; }
}

__global__ void __launch_bounds__(1024) cw_setup_kernel(DevBatch b, DevScratch sc, cw_params prm, uint64_t solid_total_cap,
                                                         uint64_t seg_total_cap, uint64_t arena_total_cap, uint32_t arena_scale) {
    __shared__ uint32_t part[4][1024];
    __shared__ uint64_t run[4];
    const int tid = threadIdx.x;
    if (tid < 4) run[tid] = 0;
    if (tid == 0) { /* the neutral task: what a list entry names when the chain kernel ran out of task slots (finished, no members) */
        PoaTask t; t.window = 0; t.seg_slot = 0; t.member_off = 0; t.n_members = 0; t.max_len = 0; t.out_off = 0; t.out_cap = 0; t.state = 1u;
        sc.tasks[sc.task_cap] = t;
    }
    __syncthreads();
    for (uint32_t w0 = 0; w0 < b.n_windows; w0 += 1024) {
        const uint32_t w = w0 + tid;
        uint32_t need_solid = 0, need_seg = 0, need_arena = 0, need_ab = 0;
        uint32_t nk = 0, tl = 0, ns = 0;
        if (w < b.n_windows) {
            ns = sc.win[w].n_seqs; tl = sc.win[w].tpl_len; nk = sc.win[w].n_kmers; /* cw_setup_need_kernel */
            need_solid = nk / prm.solid + 1;
            need_seg = (tl >= prm.k) ? tl - prm.k + 3 : 1;
            need_arena = (16 * tl + 4096) * arena_scale;
            const uint32_t nk0 = (tl >= prm.k && tl - prm.k + 1 <= CW_TMAX) ? tl - prm.k + 1 : 0;
            need_ab = (uint32_t)(cw_ab_bytes(nk0, ns, ns) >> 4);
        }
        part[0][tid] = need_solid; part[1][tid] = need_seg; part[2][tid] = need_arena; part[3][tid] = need_ab;
        __syncthreads();
        /* simple in-LDS inclusive scan, 10 steps */
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if (tid >= o) { v0 = part[0][tid - o]; v1 = part[1][tid - o]; v2 = part[2][tid - o]; v3 = part[3][tid - o]; }
            __syncthreads();
            part[0][tid] += v0; part[1][tid] += v1; part[2][tid] += v2; part[3][tid] += v3;
            __syncthreads();
        }
        if (w < b.n_windows) {
            WinInfo wi;
            wi.status = CW_WIN_CONSENSUS;
            wi.n_seqs = ns; wi.tpl_len = tl; wi.n_kmers = nk;
            uint64_t sb = run[0] + part[0][tid] - need_solid, gb = run[1] + part[1][tid] - need_seg,
                     ab = run[2] + part[2][tid] - need_arena, kb = run[3] + part[3][tid] - need_ab;
            /* the bases are stored as 32-bit offsets: a batch whose running totals would wrap is refused window by window (the host
               refuses such batches up front, CW_MAX_BATCH_WINDOWS; this is the second line of defence) */
            bool over = sb + need_solid > solid_total_cap || gb + need_seg > seg_total_cap || ab + need_arena > arena_total_cap ||
                        kb + need_ab > sc.ablock_units || kb + need_ab > 0xFFFFFFFFull || sb + need_solid > 0xFFFFFFFFull ||
                        gb + need_seg > 0xFFFFFFFFull || ab + need_arena > 0xFFFFFFFFull;
            wi.solid_base = (uint32_t)sb; wi.solid_cap = need_solid; wi.n_solid = 0;
            wi.seg_base = (uint32_t)gb; wi.seg_cap = need_seg; wi.n_segs = 0;
            wi.arena_base = (uint32_t)ab; wi.arena_cap = need_arena; wi.arena_used = 0;
            wi.ab_base = (uint32_t)kb; wi.ab_cap = need_ab; wi.pad_ = 0;
            if (over) { wi.status = CW_WIN_OVERFLOW; wi.pad_ = CW_WHY_SETUP; wi.solid_cap = wi.seg_cap = wi.arena_cap = wi.ab_cap = 0; wi.solid_base = wi.seg_base = wi.arena_base = wi.ab_base = 0; }
            sc.win[w] = wi;
        }
        __syncthreads();
        if (tid < 4) run[tid] += part[tid][1023];
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

/* the template's k-mer table: 1024 buckets of four entries, entry = (template position + 1) | the key's low 20 bits << 12 (the whole key
   when k <= 10).  A bucket is one 16-byte LDS read and its four entries are compared in registers; entries fill a bucket front to back
   and overflow into the next bucket, so a bucket with a free last entry ends the search.  (Linear probing over single entries took about
   eight dependent round trips to LDS per wave and lookup round at depth 150 -- half of the support pass.) */
#define CW_TH_POS(e) ((e) & 4095u)
#define CW_TH_FP(key) (((key) & 0xFFFFFu) << 12)
#define CW_TH_BUCKETS (CW_TH_SLOTS / 4)
#define CW_TH_HOME(key) (cw_hash32(key) >> (32 - 10))
static_assert(CW_TH_BUCKETS == 1024, "CW_TH_HOME takes ten bits of the hash");
static_assert(CW_TMAX + 1 <= 4095, "an entry holds the template position + 1 in twelve bits");
/* lookup of a template k-mer: returns its representative template position or -1 */
__device__ __forceinline__ int cw_tpl_lookup(const uint32_t* th, const uint32_t* tkey, uint32_t key) {
    uint32_t bkt = CW_TH_HOME(key);
    const uint32_t fp = CW_TH_FP(key);
    for (;;) {
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t e = th[bkt * 4u + j];
            if (e == 0) return -1;
            if ((e & ~4095u) == fp && tkey[CW_TH_POS(e) - 1] == key) return (int)CW_TH_POS(e) - 1;
        }
        bkt = (bkt + 1) & (CW_TH_BUCKETS - 1);
    }
}

/* ------------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(CW_IDX_THREADS) cw_index_kernel(DevBatch b, DevScratch sc, cw_params prm) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k = prm.k;
    const bool direct = k <= 9;                             /* 4^k nibbles fit the LDS table */
    const uint32_t n_keys = direct ? 1u << (2 * k) : 0u;
    const uint32_t kmask32_ = k >= 16u ? 0xFFFFFFFFu : (1u << (2u * k)) - 1u;
    const uint32_t nib_words = direct ? (n_keys >= 8 ? n_keys / 8 : 1) : 0u;

    /* phase A carve */
    uint32_t* tab = (uint32_t*)lds;                                           /* nib_words                */
    unsigned long long* ex = (unsigned long long*)(lds + 131072);             /* CW_EX_SLOTS               */
    uint32_t* scan_tmp = (uint32_t*)(lds + 131072 + CW_EX_SLOTS * 8);         /* 32 words                  */
    uint32_t* flags = scan_tmp + 32;                                          /* [0] overflow [1..] misc   */
    /* phase B..D carve (reuses the same bytes once phase A has been exported) */
    /* Two layouts, chosen per window by the template's length (round 6): a template of at most 1024 k-mers -- every window of the wrappers' defaults --
       keeps arrays of 1024 entries and the position matrix gets what is left (107 KB: 6 KB more than round 5's layout, which carried 15 KB of arrays
       that moved to the chain kernel in round 3); up to 2048 k-mers the arrays are twice as long.  The template table has 1024 buckets either way. */
    uint32_t* th = (uint32_t*)lds;                                            /* 4096 x u32      @0      */
    uint32_t* tkey = (uint32_t*)(lds + 16384);                                /* 1024 | 2048 x u32       */
    uint32_t* tsup; uint8_t* trep; int16_t* tcand; uint16_t* cand_tp; uint32_t* seen; uint32_t* misc; uint16_t* P_lds; /* set per window, below */
    uint32_t p_cap = 0, seen_words = 32;
    uint32_t* st_hdr = (uint32_t*)(lds + CW_IDX_STAGE_OFF);                    /* [0] words staged          */
    uint32_t* s_len = st_hdr + 4;                                             /* CW_IDX_STAGE_N            */
    uint32_t* s_off = s_len + CW_IDX_STAGE_N;                                 /* CW_IDX_STAGE_N            */
    uint32_t* s_words = s_off + CW_IDX_STAGE_N;                               /* CW_IDX_STAGE_WORDS        */
    /* position matrix: in LDS when it fits next to the presence bitsets, else in this work-group's global slot
       (high-identity deep piles: every template k-mer is an anchor).  Accessors pick the address space with a
       block-uniform branch so that the common case keeps ds_ instructions. */
    uint16_t* const P_glb = sc.p_fallback + (size_t)blockIdx.x * sc.p_fallback_elems;
    /* (address spaces said explicitly: otherwise the two arms are merged into flat_ accesses, LDS data at global-memory latency) */
    typedef __attribute__((address_space(3))) uint16_t* cw_l16w;
    typedef __attribute__((address_space(1))) uint16_t* cw_g16w;
#define PRD(i) (pg ? (uint32_t)((cw_g16w)P_glb)[i] : (uint32_t)((cw_l16w)P_lds)[i])
#define PWR(i, v) do { if (pg) ((cw_g16w)P_glb)[i] = (uint16_t)(v); else ((cw_l16w)P_lds)[i] = (uint16_t)(v); } while (0)

    for (;;) {
        __syncthreads();
        if (tid == 0) { st_hdr[1] = atomicAdd(&sc.ctr->next_window, 1u); st_hdr[0] = 0; } /* (st_hdr[1..3] are free: the staged lengths start at st_hdr + 4) */
        __syncthreads();
        const uint32_t w = st_hdr[1];
        if (w >= b.n_windows) break;
        WinInfo* wi = &sc.win[w];
        if (wi->status == CW_WIN_OVERFLOW) continue;
        const uint32_t s0 = b.win_first_seq[w];
        const uint32_t N = wi->n_seqs;
        const uint32_t L0 = wi->tpl_len;
        {   /* the phase B..D carve for this window's template (wave-uniform values: a few scalar registers) */
            const bool wide = L0 >= k && L0 - k + 1u > 1024u;
            tsup = (uint32_t*)(lds + (wide ? 24576u : 20480u));      /* 1024 | 2048 x u32 */
            trep = lds + (wide ? 32768u : 24576u);                    /* 1024 | 2048 x u8  */
            tcand = (int16_t*)(lds + (wide ? 34816u : 25600u));      /* 1024 | 2048 x i16 */
            cand_tp = (uint16_t*)(lds + (wide ? 38912u : 27648u));   /* 1024 | 2048 x u16 */
            seen = (uint32_t*)(lds + (wide ? 43008u : 29696u));      /* 16 waves x 32 | 64 words: one bit per template k-mer; later 2 | 4 KiB of flags */
            seen_words = wide ? 64u : 32u;
            misc = (uint32_t*)(lds + (wide ? 47104u : 31744u));      /* 64 words */
            const uint32_t p_off = wide ? 47360u : 32000u;
            P_lds = (uint16_t*)(lds + p_off);
            p_cap = (CW_IDX_STAGE_OFF - p_off) / 2u;
        }
        /* read once: a store to the solid table may alias *wi as far as the compiler knows, and every use inside a store loop would be a
           dependent global load (the export's write loop: 24 of them per thread, ~50 k cycles per window) */
        const uint32_t w_solid_base = wi->solid_base, w_solid_cap = wi->solid_cap, w_ab_cap = wi->ab_cap, w_ab_base = wi->ab_base, w_n_kmers = wi->n_kmers;
        /* stage the pile (see CW_IDX_STAGE_OFF): stm = lengths and offsets are in LDS, stw = the words too */
        const bool stm = N <= CW_IDX_STAGE_N;
        bool stw = false;
        if (stm) {
            const uint64_t base_off = b.seq_word_off[s0];
            for (uint32_t s = tid; s < N; s += CW_IDX_THREADS) {
                const uint32_t len = b.seq_len[s0 + s];
                const uint64_t rel = b.seq_word_off[s0 + s] - base_off; /* piles are packed front to back; anything else does not fit */
                s_len[s] = len;
                s_off[s] = (uint32_t)rel;
                atomicMax(&st_hdr[0], rel > 0xFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rel + ((len + 15u) >> 4));
            }
            __syncthreads();
            const uint32_t n_stage = st_hdr[0];
            stw = n_stage <= CW_IDX_STAGE_WORDS;
            if (stw) {
                const uint32_t* src = b.bases + base_off;
                for (uint32_t i = tid; i < n_stage; i += CW_IDX_THREADS) s_words[i] = src[i];
            }
            /* the barrier behind the table clears below orders these writes before the first pass */
        }
/* sequence s for the whole wave, four consecutive k-mers per lane (see CW_IDX_KMERS4): BODY sees p and key */
#define CW_IDX_KMERS4_WAVE(...)                                                                                         \
                const uint32_t nk = len >= k ? len - k + 1 : 0, nwd = (len + 15u) >> 4;                                \
                for (uint32_t p0 = (uint32_t)lane * 4u; p0 < nk; p0 += 256u) {                                          \
                    const uint32_t wi_ = p0 >> 4;                                                                       \
                    uint64_t x_ = ((uint64_t)words[wi_] << 32) | (wi_ + 1u < nwd ? words[wi_ + 1u] : 0u);               \
                    x_ <<= 2u * (p0 & 15u);                                                                             \
                    _Pragma("unroll") for (uint32_t q_ = 0; q_ < 4u; ++q_, x_ <<= 2) {                                  \
                        const uint32_t p = p0 + q_;                                                                     \
                        if (p >= nk) break;                                                                             \
                        const uint32_t key = (uint32_t)(x_ >> (64u - 2u * k));                                          \
                        __VA_ARGS__                                                                                     \
                    }                                                                                                   \
                }
#define CW_IDX_PASS_SEQ(...)                                                                                            \
        if (stw) { const uint32_t len = s_len[s]; const cw_l32 words = (cw_l32)(s_words + s_off[s]); CW_IDX_KMERS4_WAVE(__VA_ARGS__) } \
        else { const uint32_t len = stm ? s_len[s] : b.seq_len[s0 + s];                                                 \
               const cw_g32 words = (cw_g32)(b.bases + b.seq_word_off[s0 + s]); CW_IDX_KMERS4_WAVE(__VA_ARGS__) }
/* one pass over the pile, work-group wide: eight sequences at a time, 128 threads each, four consecutive k-mers per thread out of one
   64-bit window of the packed bases; BODY sees s, p and key (a `continue` in BODY goes to the next k-mer) */
#define CW_IDX_KMERS4(...)                                                                                              \
                const uint32_t nk = len >= k ? len - k + 1 : 0, nwd = (len + 15u) >> 4;                                \
                for (uint32_t p0 = ((uint32_t)tid & 127u) * 4u; p0 < nk; p0 += 512u) {                                  \
                    const uint32_t wi_ = p0 >> 4;                                                                       \
                    uint64_t x_ = ((uint64_t)words[wi_] << 32) | (wi_ + 1u < nwd ? words[wi_ + 1u] : 0u);               \
                    x_ <<= 2u * (p0 & 15u);                                                                             \
                    _Pragma("unroll") for (uint32_t q_ = 0; q_ < 4u; ++q_, x_ <<= 2) {                                  \
                        const uint32_t p = p0 + q_;                                                                     \
                        if (p >= nk) break;                                                                             \
                        const uint32_t key = (uint32_t)(x_ >> (64u - 2u * k));                                          \
                        __VA_ARGS__                                                                                     \
                    }                                                                                                   \
                }
#define CW_IDX_PASS_BLOCK(...)                                                                                          \
        for (uint32_t sp = 0; sp < N; sp += 8) {                                                                        \
            const uint32_t s = sp + ((uint32_t)tid >> 7);                                                               \
            if (s < N) {                                                                                                \
                if (stw) { const uint32_t len = s_len[s]; const cw_l32 words = (cw_l32)(s_words + s_off[s]); CW_IDX_KMERS4(__VA_ARGS__) } \
                else { const uint32_t len = stm ? s_len[s] : b.seq_len[s0 + s];                                         \
                       const cw_g32 words = (cw_g32)(b.bases + b.seq_word_off[s0 + s]); CW_IDX_KMERS4(__VA_ARGS__) }           \
            }                                                                                                           \
        }

/* eight sequences at a time and four consecutive k-mers per thread like CW_IDX_PASS_BLOCK, but the eight thread groups start 64 positions
   apart (rotating inside each block of 512 positions): the sequences of a pile are copies of one stretch of the genome, so groups that
   walk them in step update the same counters at the same moment; 64 positions apart they are in different k-mers */
#define CW_IDX_KMERS4R(...)                                                                                             \
                const uint32_t nk = len >= k ? len - k + 1 : 0, nwd = (len + 15u) >> 4;                                \
                for (uint32_t pb = 0; pb < nk; pb += 512u) {                                                            \
                    const uint32_t p0 = pb + ((((uint32_t)tid & 127u) * 4u + ((uint32_t)tid >> 7) * 64u) & 511u);      \
                    if (p0 >= nk) continue;                                                                             \
                    const uint32_t wi_ = p0 >> 4;                                                                       \
                    const uint64_t x_ = ((uint64_t)words[wi_] << 32) | (wi_ + 1u < nwd ? words[wi_ + 1u] : 0u);         \
                    /* (direct tables: k <= 9.  The four k-mers of a thread start in the first four bases of the window's upper word once base p0 is \
                       at its top: 32-bit field extracts -- round 6; the running 64-bit shift cost two double-width shifts per k-mer) */ \
                    const uint32_t yh_ = (uint32_t)((x_ << (2u * (p0 & 15u))) >> 32), nv_ = nk - p0;                    \
                    _Pragma("unroll") for (uint32_t q_ = 0; q_ < 4u; ++q_) {                                            \
                        if (q_ < nv_) do {                                                                              \
                            const uint32_t p = p0 + q_;                                                                 \
                            const uint32_t keyraw = yh_ >> (32u - 2u * k - 2u * q_), key = keyraw & kmask32_; /* keyraw: bases before the k-mer above it */ \
                            (void)p; (void)keyraw;                                                                      \
                            __VA_ARGS__                                                                                 \
                        } while (0);                                                                                    \
                    }                                                                                                   \
                }
#define CW_IDX_PASS_BLOCKR(...)                                                                                         \
        for (uint32_t sp = 0; sp < N; sp += 8) {                                                                        \
            const uint32_t s = sp + ((uint32_t)tid >> 7);                                                               \
            if (s < N) {                                                                                                \
                if (stw) { const uint32_t len = s_len[s]; const cw_l32 words = (cw_l32)(s_words + s_off[s]); CW_IDX_KMERS4R(__VA_ARGS__) } \
                else { const uint32_t len = stm ? s_len[s] : b.seq_len[s0 + s];                                         \
                       const cw_g32 words = (cw_g32)(b.bases + b.seq_word_off[s0 + s]); CW_IDX_KMERS4R(__VA_ARGS__) }          \
            }                                                                                                           \
        }

/* the same pass, two sequences at a time and one k-mer per thread: the sequences of a pile are copies of one stretch of the genome, so
   threads that work on the same positions of different sequences update the same counters at the same moment (eight-way with
   CW_IDX_PASS_BLOCK: most compare-and-swap rounds of the count pass were retries) */
#define CW_IDX_KMERS1(...)                                                                                              \
                const uint32_t nk = len >= k ? len - k + 1 : 0, nwd = (len + 15u) >> 4;                                \
                for (uint32_t p = (uint32_t)tid & 511u; p < nk; p += 512u) {                                            \
                    const uint32_t wi_ = p >> 4;                                                                        \
                    uint64_t x_ = ((uint64_t)words[wi_] << 32) | (wi_ + 1u < nwd ? words[wi_ + 1u] : 0u);               \
                    x_ <<= 2u * (p & 15u);                                                                              \
                    const uint32_t key = (uint32_t)(x_ >> (64u - 2u * k));                                              \
                    __VA_ARGS__                                                                                         \
                }
#define CW_IDX_PASS_BLOCK2(...)                                                                                         \
        for (uint32_t sp = 0; sp < N; sp += 2) {                                                                        \
            const uint32_t s = sp + ((uint32_t)tid >> 9);                                                               \
            if (s < N) {                                                                                                \
                if (stw) { const uint32_t len = s_len[s]; const cw_l32 words = (cw_l32)(s_words + s_off[s]); CW_IDX_KMERS1(__VA_ARGS__) } \
                else { const uint32_t len = stm ? s_len[s] : b.seq_len[s0 + s];                                         \
                       const cw_g32 words = (cw_g32)(b.bases + b.seq_word_off[s0 + s]); CW_IDX_KMERS1(__VA_ARGS__) }           \
            }                                                                                                           \
        }

        CW_PROF_T0();
        /* ================= phase A: counts ================= */
        if (!direct) {
            /* ---- k > 9: the key space no longer fits a direct table.  Exact counts in an LDS hash table (key<<32 | count),
               the pile scanned P times, pass p owning the keys whose hash falls in partition p (P chosen so that even an
               all-distinct pile stays under half load); after each pass the solid entries are appended to the window's
               slice; at the end the slice is bitonic-sorted by key in LDS. ---- */
            unsigned long long* hs_tab = (unsigned long long*)lds; /* 16384 slots = 128 KiB */
            const uint32_t HS = 16384u;
            const uint32_t P_ = (w_n_kmers + HS / 2 - 1) / (HS / 2) ? (w_n_kmers + HS / 2 - 1) / (HS / 2) : 1u;
            uint32_t written = 0;
            bool fits = true;
            if (tid < 8) flags[tid] = 0;
            for (uint32_t pass = 0; pass < P_; ++pass) {
                for (uint32_t i = tid; i < HS; i += CW_IDX_THREADS) hs_tab[i] = 0ull;
                __syncthreads();
                CW_IDX_PASS_BLOCK({
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
                })
                __syncthreads();
                uint32_t mine = 0;
                for (uint32_t i = tid; i < HS; i += CW_IDX_THREADS) mine += ((uint32_t)hs_tab[i] >= prm.solid) ? 1u : 0u;
                uint32_t total;
                const uint32_t off = cw_block_exscan(mine, scan_tmp, &total);
                if (written + total > w_solid_cap || flags[0]) fits = false;
                if (fits && mine) {
                    uint32_t o = w_solid_base + written + off;
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
            /* the sort runs in LDS when the set fits (16384 keys), else in this work-group's global table (262144: round 4 -- a k > 9 run with a low
               solid threshold, e.g. -k 13 --solid 1 on 900-base windows, has more solid keys than LDS holds and used to stop on a capacity) */
            if (fits && np2 > (uint32_t)CW_EXG_SLOTS) fits = false;
            if (tid == 0) {
                wi->n_solid = fits ? written : 0;
                if (!fits) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_COUNT; sc.ctr->any_overflow = 1; }
            }
            __threadfence_block();
            __syncthreads();
            if (!fits) continue;
            if (written > 1) {
                unsigned long long* const sort_tab = np2 <= HS ? hs_tab : sc.ex_fallback + (size_t)blockIdx.x * CW_EXG_SLOTS;
                for (uint32_t x = tid; x < np2; x += CW_IDX_THREADS)
                    sort_tab[x] = x < written ? (((unsigned long long)sc.solid_key[w_solid_base + x] << 32) | sc.solid_cnt[w_solid_base + x]) : ~0ull;
                __syncthreads();
                for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
                    for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                        for (uint32_t x = tid; x < np2; x += CW_IDX_THREADS) {
                            const uint32_t y = x ^ j2;
                            if (y > x) {
                                const unsigned long long ax = sort_tab[x], ay = sort_tab[y];
                                const bool up = (x & k2) == 0;
                                if ((ax > ay) == up) { sort_tab[x] = ay; sort_tab[y] = ax; }
                            }
                        }
                        __threadfence_block();
                        __syncthreads();
                    }
                }
                for (uint32_t x = tid; x < written; x += CW_IDX_THREADS) {
                    sc.solid_key[w_solid_base + x] = (uint32_t)(sort_tab[x] >> 32);
                    sc.solid_cnt[w_solid_base + x] = (uint32_t)sort_tab[x];
                }
                __syncthreads();
            }
        } else {
        /* keys that occur 16 times or more: at most n_kmers / 16 of them.  The LDS table holds every pile of read correction (<= 151
           sequences) in practice; the piles of assembly polishing are as deep as the coverage (maxSupport = 20000, CONSENT-polish:43):
           when the LDS table overflows the count pass is redone with this work-group's table in global memory */
        /* Round 4: byte counters first.  At read-correction depths no key comes near 255 occurrences (a true k-mer of a depth-150 pile is seen
           ~50 times), so a counter per key that is one byte wide needs neither the compare-and-swap loop of the nibbles (a read, a CAS and
           its retries per k-mer) nor the list of the occurrences beyond the fifteenth and its second phase (a quarter of a deep pile's
           k-mers): one returning add per k-mer.  4^9 bytes are twice the table, so k = 9 counts and exports the lower half of the key space,
           then the upper half (the pile's k-mers are extracted twice: cheap next to the atomics of a deep pile, not of a shallow one:
           CW_IDX_BYTES_MIN_N).  A counter that reaches 200 -- low
           complexity, polishing depths -- sends the window to the nibble path below, which has no such limit. */
        bool done8 = false;
        const uint32_t bkeys = n_keys < 131072u ? n_keys : 131072u, bwords = bkeys >> 2, BNI = bwords >> 12;
        if (CW_IDX_BYTES && N <= 200u && (n_keys <= 131072u || N >= CW_IDX_BYTES_MIN_N) && prm.solid >= 1u && prm.solid <= 127u && (bwords & 4095u) == 0u && BNI >= 1u && BNI <= 8u) {
            const uint32_t n_half = n_keys / bkeys; /* 1 (k <= 8) or 2 (k = 9) */
            uint32_t written = 0;
            bool ok8 = true, fits8 = true;
            if (tid < 8) flags[tid] = 0;
            for (uint32_t h = 0; h < n_half; ++h) {
                for (uint32_t i = tid; i < bwords; i += CW_IDX_THREADS) tab[i] = 0;
                __syncthreads(); /* (flags[2], the byte sum, runs on over the halves: cleared with the other flags above) */
                if (h == 0) CW_PROF(sc.ctr, 55, tid == 0);
#if CW_IDX_BYTES_RTN /* rounds 4: a returning add, the old byte looked at */
                CW_IDX_PASS_BLOCKR({
                    if (n_half > 1u && (key >> 17) != h) continue;
                    const uint32_t kk = key & (bkeys - 1u), sh8 = (kk & 3u) * 8u;
                    const uint32_t old = atomicAdd(&tab[kk >> 2], 1u << sh8);
                    if (((old >> sh8) & 255u) >= 200u) flags[0] = 1; /* (a byte cannot carry into its neighbour unseen: the add that takes it from 255 to 0 returns 255) */
                })
#else
                /* Round 5: fire-and-forget adds.  Nobody waits for an add to come back, so the pass runs at the rate the LDS takes the adds instead of
                   at the latency of a returning atomic per k-mer and thread; whether a byte overflowed is decided AFTER the pass, exactly: a word holds
                   the sum of its four counters x 256^b, so as long as no counter passes 255 the bytes ARE the counts, and every overflow lowers the sum
                   of all bytes of the table by 255 (a carry into the next byte) or 256 (out of the word) against the number of adds made.  The export
                   scan reads every word anyway and adds the bytes up (v_sad_u8); a table whose byte sum is not the number of k-mers counted sends the
                   window to the nibble path below, as a counter at 200 did. */
                /* Round 6: the adds are not counted any more -- every k-mer of the pile is added in exactly one half, so after the last half the byte sums of
                   the halves must add up to the pile's k-mer count (w_n_kmers, cw_setup_need_kernel); an overflow in the first half is then seen a half
                   later, on a window that takes the nibble path anyway.  The word's offset and the half test are one subtraction and one compare. */
                const uint32_t half_off = h * bkeys;
                CW_IDX_PASS_BLOCKR({
                    const uint32_t off = (key & ~3u) - half_off; /* byte offset of the counter's word in this half's table; wraps for a key of the other half */
                    if (off >= bkeys) continue;
                    (void)__hip_atomic_fetch_add((cw_l32w)((__attribute__((address_space(3))) uint8_t*)tab + off), 1u << ((keyraw << 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                })
#endif
                __syncthreads();
                CW_PROF(sc.ctr, 0, tid == 0);
                if (flags[0]) { ok8 = false; break; }
                /* export of this half, as the nibble table's fast export below: wave v owns a sixteenth of the table and reads it lane-interleaved,
                   four words (sixteen keys) per lane and read; key order = (read, lane, bit) */
                const uint32_t qbase = (uint32_t)wave * (BNI * 64u) + (uint32_t)lane;
                const uint32_t addt = (128u - prm.solid) * 0x01010101u;
                auto cmask8 = [&](const uint32_t v) -> uint32_t { /* bit q = byte q of v is >= the threshold */
                    const uint32_t c = (((v & 0x7F7F7F7Fu) + addt) | v) & 0x80808080u;
                    return ((c >> 7) | (c >> 14) | (c >> 21) | (c >> 28)) & 0xFu;
                };
                uint32_t m[8], bsum = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    m[i] = 0u;
                    if ((uint32_t)i < BNI) {
                        const uint4 v4 = *(const uint4*)&tab[(qbase + (uint32_t)i * 64u) * 4u];
                        m[i] = cmask8(v4.x) | (cmask8(v4.y) << 4) | (cmask8(v4.z) << 8) | (cmask8(v4.w) << 12);
                        bsum = __builtin_amdgcn_sad_u8(v4.x, 0u, bsum); bsum = __builtin_amdgcn_sad_u8(v4.y, 0u, bsum);
                        bsum = __builtin_amdgcn_sad_u8(v4.z, 0u, bsum); bsum = __builtin_amdgcn_sad_u8(v4.w, 0u, bsum);
                    }
                }
#if !CW_IDX_BYTES_RTN
                { const uint32_t ws_ = (uint32_t)cw_wave_sum((int)bsum); if (lane == 0) atomicAdd(&flags[2], ws_); } /* complete after the barriers of the scan below */
#endif
                CW_PROF(sc.ctr, 56, tid == 0);
                uint32_t offs[8], wtot = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t c = (uint32_t)__popc(m[i]);
                    const uint32_t inc = (uint32_t)cw_wave_scan_add((int)c);
                    offs[i] = wtot + inc - c;
                    wtot += (uint32_t)cw_lane_value((int)inc, 63);
                }
                uint32_t total;
                uint32_t woff = cw_block_exscan(lane == 0 ? wtot : 0u, scan_tmp, &total);
                woff = (uint32_t)cw_lane_value((int)woff, 0);
#if !CW_IDX_BYTES_RTN
                if (h + 1u == n_half && flags[2] != w_n_kmers) { ok8 = false; break; } /* a counter passed 255 in one of the halves: the bytes are not the counts */
#endif
                fits8 = written + total <= w_solid_cap;
                if (!fits8) { if (h + 1u != n_half) ok8 = false; /* (the counts of this half are not verified yet: the nibble path decides) */ break; }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint32_t mm = m[i], o = w_solid_base + written + woff + offs[i];
                    const uint32_t wd0 = (qbase + (uint32_t)i * 64u) * 4u;
                    while (mm) {
                        const uint32_t bpos = (uint32_t)__ffs((int)mm) - 1u;
                        mm &= mm - 1u;
                        const uint32_t wd = wd0 + (bpos >> 2);
                        sc.solid_key[o] = h * 131072u + wd * 4u + (bpos & 3u);
                        sc.solid_cnt[o] = (tab[wd] >> (8u * (bpos & 3u))) & 255u;
                        ++o;
                    }
                }
                written += total;
                __syncthreads(); /* the table is cleared for the next half */
            }
            if (ok8) {
                if (tid == 0) {
                    wi->n_solid = fits8 ? written : 0;
                    if (!fits8) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_SOLIDCAP; sc.ctr->any_overflow = 1; }
                }
                __syncthreads();
                if (!fits8) continue;
                done8 = true;
            }
            __syncthreads(); /* (fallback: everybody has read the flag before the nibble path clears it) */
        }
        if (!done8) {
        bool big_ex = false; /* the LDS table overflowed and the pass was redone with the table in global memory */
        unsigned long long* const exg = sc.ex_fallback + (size_t)blockIdx.x * CW_EXG_SLOTS;
        /* 4-bit counters in the direct table; occurrences beyond the 15th of a key are counted in a small hash table (key + 1 in the
           high half, the overflow count in the low half), so that a key's exact count is its nibble, plus its overflow when the nibble is 15 */
#define CW_IDX_COUNT_PASS(EXTAB, EXSLOTS, EXBITS)                                                                       \
        CW_IDX_PASS_BLOCK2({                                                                                            \
            const uint32_t wd = key >> 3, sh = (key & 7) * 4;                                                           \
            uint32_t old = tab[wd];                                                                                     \
            bool sat = false;                                                                                           \
            for (;;) {                                                                                                  \
                if (((old >> sh) & 15u) == 15u) { sat = true; break; }                                                  \
                uint32_t prev = atomicCAS(&tab[wd], old, old + (1u << sh));                                             \
                if (prev == old) break;                                                                                 \
                old = prev;                                                                                             \
            }                                                                                                           \
            if (!sat) continue;                                                                                         \
            uint32_t slot = cw_hash32(key) >> (32 - (EXBITS));                                                          \
            const unsigned long long fresh = ((unsigned long long)(key + 1) << 32) | 1ull;                              \
            for (uint32_t probe = 0;; ++probe) {                                                                        \
                if (probe >= (EXSLOTS)) { flags[0] = 1; break; }                                                        \
                unsigned long long cur = atomicCAS(&(EXTAB)[slot], 0ull, fresh);                                        \
                if (cur == 0ull) break;                                                                                 \
                if ((uint32_t)(cur >> 32) == key + 1) { atomicAdd(&(EXTAB)[slot], 1ull); break; }                       \
                slot = (slot + 1) & ((EXSLOTS) - 1);                                                                    \
            }                                                                                                           \
        })
        /* First attempt: the thread that takes a key's counter from 14 to 15 enters the key into the LDS hash table (count 0), and every
           later occurrence only appends the key to a list in this work-group's global scratch; after the pass the list is added up with
           all threads busy, one lookup and one fire-and-forget add per entry.  (Counting the later occurrences in the table as they came
           -- a compare-and-swap and an add per occurrence, a fifth of the lanes active, the probe loop as long as its slowest lane -- was
           70 % of the count pass of a depth-150 pile.)  If the table or the list overflows, the pass is redone the old way with the table
           in global memory. */
        uint32_t* const ovl = (uint32_t*)exg;
        const uint32_t ovl_cap = CW_EXG_SLOTS * 2u;
        for (uint32_t i = tid; i < nib_words; i += CW_IDX_THREADS) tab[i] = 0;
        for (uint32_t i = tid; i < CW_EX_SLOTS; i += CW_IDX_THREADS) ex[i] = 0ull;
        if (tid < 8) flags[tid] = 0;
        __syncthreads();
        CW_PROF(sc.ctr, 55, tid == 0);
        CW_IDX_PASS_BLOCKR({
            const uint32_t wd = key >> 3, sh = (key & 7) * 4;
            uint32_t old = tab[wd];
            bool sat = false;
            for (;;) {
                if (((old >> sh) & 15u) == 15u) { sat = true; break; }
                uint32_t prev = atomicCAS(&tab[wd], old, old + (1u << sh));
                if (prev == old) break;
                old = prev;
            }
            if (sat) { /* flags[2]: the list's cursor */
                const uint32_t oi = atomicAdd(&flags[2], 1u);
                if (oi < ovl_cap) ovl[oi] = key;
                continue;
            }
            if (((old >> sh) & 15u) != 14u) continue;
            uint32_t slot = cw_hash32(key) >> (32 - CW_EX_BITS); /* this increment was the fifteenth: the key's entry */
            const unsigned long long fresh = (unsigned long long)(key + 1) << 32;
            for (uint32_t probe = 0;; ++probe) {
                if (probe >= CW_EX_SLOTS) { flags[0] = 1; break; }
                if (atomicCAS(&ex[slot], 0ull, fresh) == 0ull) break;
                slot = (slot + 1) & (CW_EX_SLOTS - 1);
            }
        })
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* the list: global memory written and read by this work-group only (one CU, one L1) */
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (flags[2] > ovl_cap) flags[0] = 1; /* every thread stores the same value */
        __syncthreads();
        if (!flags[0]) {
            const uint32_t n_ovl = flags[2];
            for (uint32_t i = tid; i < n_ovl; i += CW_IDX_THREADS) {
                const uint32_t key = ovl[i];
                uint32_t slot = cw_hash32(key) >> (32 - CW_EX_BITS), probe = 0;
                while ((uint32_t)(ex[slot] >> 32) != key + 1 && probe < CW_EX_SLOTS) { slot = (slot + 1) & (CW_EX_SLOTS - 1); ++probe; } /* it is there: entered before the list got the key */
                if (probe < CW_EX_SLOTS) atomicAdd(&ex[slot], 1ull);
                else flags[3] = 1; /* cannot happen; if it does, the old way decides */
            }
        }
        __syncthreads();
        if (flags[3]) flags[0] = 1;
        __syncthreads();
        if (flags[0]) { /* rare (deep polishing piles): everything again, saturated keys into this work-group's global table */
            __syncthreads(); /* everybody has read the flag before it is cleared */
            for (uint32_t i = tid; i < nib_words; i += CW_IDX_THREADS) tab[i] = 0;
            for (uint32_t i = tid; i < CW_EXG_SLOTS; i += CW_IDX_THREADS) exg[i] = 0ull;
            if (tid < 8) flags[tid] = 0;
            __syncthreads();
            CW_IDX_COUNT_PASS(exg, CW_EXG_SLOTS, 18)
            __syncthreads();
            big_ex = true;
        }
#undef CW_IDX_COUNT_PASS
        CW_PROF(sc.ctr, 0, tid == 0);
        CW_PROF(sc.ctr, 1, tid == 0);
        if (flags[0]) { /* more saturated keys than even the global exact table holds */
            if (tid == 0) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_COUNT; sc.ctr->any_overflow = 1; }
            __builtin_amdgcn_wave_barrier(); /* the wave meets again before the back edge (see cw_stitch.h) */
            continue;
        }
        /* export the solid set in ascending key order.  Thread t owns the contiguous words [t*wpt, (t+1)*wpt) (so thread
           order = key order) but visits them rotated by t, which spreads a wave's reads over all LDS banks; the few
           solid keys it finds are kept in registers, ranked locally, and written after one block-wide prefix sum. */
        {
            const uint32_t keys_per_word = n_keys >= 8 ? 8 : n_keys;
            const uint32_t wpt = (nib_words + CW_IDX_THREADS - 1) / CW_IDX_THREADS;
            const uint32_t w_beg = min(nib_words, (uint32_t)tid * wpt), w_cnt = min(nib_words, w_beg + wpt) - w_beg;
            uint32_t lk[CW_EXP_SLOTS]; /* key (18 bits: this is the k <= 9 path) | count << 18; a count that does not pack sends the thread to the re-walk below */
            bool wide = false;
            uint32_t mine = 0, n_first = 0xFFFFFFFFu; /* keys found before the rotated walk wrapped to the thread's first word */
            auto ex_lookup = [&](const uint32_t key) -> uint32_t { /* occurrences beyond the 15th */
                if (big_ex) {
                    uint32_t slot = cw_hash32(key) >> (32 - 18);
                    unsigned long long xe = exg[slot];
                    while (xe != 0ull && (uint32_t)(xe >> 32) != key + 1) { slot = (slot + 1) & (CW_EXG_SLOTS - 1); xe = exg[slot]; }
                    return (uint32_t)xe;
                }
                uint32_t slot = cw_hash32(key) >> (32 - CW_EX_BITS);
                unsigned long long xe = ex[slot];
                while (xe != 0ull && (uint32_t)(xe >> 32) != key + 1) { slot = (slot + 1) & (CW_EX_SLOTS - 1); xe = ex[slot]; }
                return (uint32_t)xe;
            };
            const uint32_t add = (16u - (prm.solid < 15u ? prm.solid : 15u)) * 0x01010101u;
            auto scan_word = [&](const uint32_t v, const uint32_t wd) {
                if (v == 0) return;
                /* nibbles that can be solid, all eight at once: (nibble + 16 - t) carries into bit 4 of its byte iff nibble >= t,
                   t = min(solid, 15) (a saturated nibble is decided by its exact count below) */
                uint32_t cand = ((((v & 0x0F0F0F0Fu) + add) & 0x10101010u) >> 4) | (((((v >> 4) & 0x0F0F0F0Fu) + add) & 0x10101010u) >> 3);
                if (keys_per_word < 8) cand &= (1u << (8 * ((keys_per_word + 1) / 2))) - 1u;
                while (cand) {
                    const uint32_t bpos = (uint32_t)__ffs((int)cand) - 1u; /* bit 8*byte + (odd nibble) : ascending = key order */
                    cand &= cand - 1u;
                    const uint32_t q = (bpos >> 3) * 2u + (bpos & 1u);
                    const uint32_t nib = (v >> (4 * q)) & 15u;
                    const uint32_t key = wd * 8 + q;
                    uint32_t c = nib;
                    if (nib == 15u) c = 15u + ex_lookup(key); /* exactly 15 occurrences leave no entry */
                    if (c < prm.solid) continue;
#pragma unroll
                    for (int z = 0; z < CW_EXP_SLOTS; ++z) if ((uint32_t)z == mine) lk[z] = key | (c << 18);
                    wide = wide || c >= (1u << 14);
                    mine++;
                }
            };
            /* The common case -- k = 8 or 9 and a solid threshold the nibble decides (<= 15): no per-key branch while the table is read.
               Wave v owns the 16th part of the table that holds its keys and reads it lane-interleaved, four words per lane and read (no
               bank conflicts without any rotation); a read leaves one 32-bit mask of the solid nibbles of its 32 keys (bit = key within
               the four words), so key order = (read, lane, bit) and a key's place in the output is a wave prefix sum per read plus the
               block prefix over the waves.  Then every lane writes the keys of its masks (a few per mask) with their counts -- instead of
               every wave walking the candidate path (12 register slots, the exact-table probe) for every word in which ANY lane had a
               solid key (measured at depth 150: export scan + write 103 k cycles per window, a fifth of the kernel; now 30 k). */
            const uint32_t NI = nib_words >> 12;
            if (prm.solid <= 15u && (nib_words & 4095u) == 0u && NI >= 1u && NI <= 8u) {
                const uint32_t qbase = (uint32_t)wave * (NI * 64u) + (uint32_t)lane;
                uint32_t m[8];
                auto cmask = [&](const uint32_t v) -> uint32_t { /* bit q = nibble q of v is >= the threshold */
                    uint32_t c = ((((v & 0x0F0F0F0Fu) + add) & 0x10101010u) >> 4) | (((((v >> 4) & 0x0F0F0F0Fu) + add) & 0x10101010u) >> 3);
                    return (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xFFu;
                };
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    m[i] = 0u;
                    if ((uint32_t)i < NI) {
                        const uint4 v4 = *(const uint4*)&tab[(qbase + (uint32_t)i * 64u) * 4u];
                        m[i] = cmask(v4.x) | (cmask(v4.y) << 8) | (cmask(v4.z) << 16) | (cmask(v4.w) << 24);
                    }
                }
                CW_PROF(sc.ctr, 56, tid == 0);
                uint32_t offs[8], wtot = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t c = (uint32_t)__popc(m[i]);
                    const uint32_t inc = (uint32_t)cw_wave_scan_add((int)c);
                    offs[i] = wtot + inc - c;
                    wtot += (uint32_t)cw_lane_value((int)inc, 63);
                }
                uint32_t total;
                uint32_t woff = cw_block_exscan(lane == 0 ? wtot : 0u, scan_tmp, &total);
                woff = (uint32_t)cw_lane_value((int)woff, 0);
                const bool fits = total <= w_solid_cap;
                if (fits) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        uint32_t mm = m[i], o = w_solid_base + woff + offs[i];
                        const uint32_t wd0 = (qbase + (uint32_t)i * 64u) * 4u;
                        while (mm) {
                            const uint32_t bpos = (uint32_t)__ffs((int)mm) - 1u;
                            mm &= mm - 1u;
                            const uint32_t wd = wd0 + (bpos >> 3), key = wd * 8u + (bpos & 7u);
                            uint32_t c = (tab[wd] >> (4u * (bpos & 7u))) & 15u;
                            if (c == 15u) c += ex_lookup(key); /* exactly 15 occurrences leave no entry */
                            sc.solid_key[o] = key;
                            sc.solid_cnt[o] = c;
                            ++o;
                        }
                    }
                }
                if (tid == 0) {
                    wi->n_solid = fits ? total : 0;
                    if (!fits) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_SOLIDCAP; sc.ctr->any_overflow = 1; }
                }
                __syncthreads();
                if (!fits) continue;
            } else {
            if (w_cnt && (w_cnt & 3u) == 0u) {
                /* four words per LDS read (most of the table is empty); the rotation spreads a wave's reads over the banks */
                const uint32_t nq = w_cnt >> 2, q0 = (uint32_t)tid % nq; /* one division per thread */
                for (uint32_t i = 0; i < nq; ++i) {
                    uint32_t r = i + q0;
                    if (r == nq) n_first = mine;
                    r = r >= nq ? r - nq : r;
                    const uint32_t wd = w_beg + 4u * r;
                    const uint4 v4 = *(const uint4*)&tab[wd];
                    if ((v4.x | v4.y | v4.z | v4.w) == 0u) continue;
                    scan_word(v4.x, wd); scan_word(v4.y, wd + 1u); scan_word(v4.z, wd + 2u); scan_word(v4.w, wd + 3u);
                }
            } else {
                const uint32_t r0 = w_cnt ? (uint32_t)tid % w_cnt : 0u;
                for (uint32_t i = 0; i < w_cnt; ++i) {
                    uint32_t r = i + r0;
                    if (r == w_cnt) n_first = mine;
                    r = r >= w_cnt ? r - w_cnt : r;
                    scan_word(tab[w_beg + r], w_beg + r);
                }
            }
            CW_PROF(sc.ctr, 56, tid == 0);
            uint32_t total;
            const uint32_t off = cw_block_exscan(mine, scan_tmp, &total);
            const bool fits = total <= w_solid_cap;
            if (n_first > mine) n_first = mine; /* never wrapped */
            if (fits && (mine > CW_EXP_SLOTS || wide)) { /* more than the register slots hold (deep piles: a few threads per window): this thread walks its words again, in key order */
                uint32_t o = w_solid_base + off;
                for (uint32_t i = 0; i < w_cnt && mine; ++i) {
                    const uint32_t wd = w_beg + i;
                    const uint32_t v = tab[wd];
                    if (v == 0) continue;
                    uint32_t cand = ((((v & 0x0F0F0F0Fu) + add) & 0x10101010u) >> 4) | (((((v >> 4) & 0x0F0F0F0Fu) + add) & 0x10101010u) >> 3);
                    if (keys_per_word < 8) cand &= (1u << (8 * ((keys_per_word + 1) / 2))) - 1u;
                    while (cand) { /* see scan_word */
                        const uint32_t bpos = (uint32_t)__ffs((int)cand) - 1u;
                        cand &= cand - 1u;
                        const uint32_t q = (bpos >> 3) * 2u + (bpos & 1u);
                        const uint32_t nib = (v >> (4 * q)) & 15u;
                        const uint32_t key = wd * 8 + q;
                        uint32_t c = nib;
                        if (nib == 15u) c = 15u + ex_lookup(key);
                        if (c >= prm.solid) { sc.solid_key[o] = key; sc.solid_cnt[o] = c; o++; }
                    }
                }
            } else if (fits && mine) {
                /* found order = the words from the rotation point to the thread's last word, then from its first word: two ascending runs, the
                   second one below the first */
#pragma unroll
                for (int z = 0; z < CW_EXP_SLOTS; ++z) {
                    if ((uint32_t)z < mine) {
                        const uint32_t pos = (uint32_t)z < n_first ? (uint32_t)z + (mine - n_first) : (uint32_t)z - n_first;
                        sc.solid_key[w_solid_base + off + pos] = lk[z] & 0x3FFFFu;
                        sc.solid_cnt[w_solid_base + off + pos] = lk[z] >> 18;
                    }
                }
            }
            if (tid == 0) {
                wi->n_solid = fits ? total : 0;
                if (!fits) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_SOLIDCAP; sc.ctr->any_overflow = 1; }
            }
            __syncthreads();
            if (!fits) continue;
            }
        }

        } /* the nibble path */
        } /* direct table */

        CW_PROF(sc.ctr, 2, tid == 0);
        /* ================= phase B: anchor candidates ================= */
        const uint32_t nk0 = L0 >= k ? L0 - k + 1 : 0;
        const int sup_min = min((int)prm.common_kmers, (int)N / 2); /* correctionMSA.cpp:31 */
        if (nk0 == 0 || nk0 > CW_TMAX) {
            if (tid == 0) {
                if (nk0 == 0) wi->status = CW_WIN_TEMPLATE;
                else { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_TEMPLATE; sc.ctr->any_overflow = 1; }
            }
            __builtin_amdgcn_wave_barrier(); /* the wave meets again before the back edge (see cw_stitch.h) */
            continue;
        }
        /* P is anchor-major, P[a * Np + s]; Np is even with Np/2 odd so that rows read as u32 pairs by consecutive
           lanes fall on distinct banks */
        uint32_t Np = (N + 1u) & ~1u;
        if (((Np >> 1) & 1u) == 0u) Np += 2u;
        const uint32_t Nw = (N + 63u) >> 6;
        /* When a matrix with one row per TEMPLATE k-mer fits (every 500-base window up to depth ~100), the support pass records the hit
           positions as it goes and the anchors' rows are simply picked out of it afterwards: the second pass over the pile (one more
           table lookup per k-mer) is not needed.  Otherwise the matrix has one row per anchor and is filled by that second pass. */
        const bool tfit = (uint64_t)nk0 * Np * 2 + (uint64_t)nk0 * Nw * 8 + (uint64_t)N * 2 + 16 <= (uint64_t)p_cap * 2;
        /* (the matrix in LDS is cleared sixteen bytes per lane and instruction: Np is even, P_lds 16-byte aligned) */
        auto p_clear_lds = [&](const uint32_t n16) { /* n16 u16 entries, even */
            const uint32_t nv = n16 >> 3, pat = (uint32_t)CW_NONE16 * 0x00010001u;
            for (uint32_t i = tid; i < nv; i += CW_IDX_THREADS) ((uint4*)P_lds)[i] = make_uint4(pat, pat, pat, pat);
            for (uint32_t i = (nv << 3) + tid; i < n16; i += CW_IDX_THREADS) P_lds[i] = (uint16_t)CW_NONE16;
        };
        if (tfit) p_clear_lds(nk0 * Np);
        /* When the matrix per template k-mer does not fit (depth > ~100), the support pass also writes every hit (template k-mer, sequence,
           position: 10 + 12 + 10 bits) to a list in this work-group's global scratch, and the anchors' rows are filled from the list: the
           second pass over the pile's k-mers (extraction and a table lookup each, nine in ten for nothing) is only taken when a hit does not
           pack or the list overflows.  (The scratch is the one of phase A's global exact table, which is exported by now.) */
        uint32_t* const hitlist = (uint32_t*)(sc.ex_fallback + (size_t)blockIdx.x * CW_EXG_SLOTS);
        const uint32_t hit_cap = CW_EXG_SLOTS * 2u;
        const bool hl = !tfit && N <= 1024u; /* (a list entry holds the sequence in ten bits) */
        if (tid == 0) { misc[4] = 0; misc[5] = 0; }
        for (uint32_t i = tid; i < CW_TH_SLOTS; i += CW_IDX_THREADS) th[i] = 0;
        for (uint32_t i = tid; i < seen_words * 32u; i += CW_IDX_THREADS) { tsup[i] = 0; trep[i] = 0; tcand[i] = -1; } /* 1024 or 2048 entries */
        /* (one thread per template k-mer, two rounds for a template of more than 1024 k-mers: round 6) */
        for (uint32_t tp = tid; tp < nk0; tp += CW_IDX_THREADS) tkey[tp] = stw ? cw_kmer_at(s_words, tp, k) : cw_kmer_at(b.bases + b.seq_word_off[s0], tp, k);
        __syncthreads();
        for (uint32_t tp = tid; tp < nk0; tp += CW_IDX_THREADS) {
            const uint32_t key = tkey[tp];
            uint32_t bkt = CW_TH_HOME(key);
            for (bool placed = false; !placed; bkt = (bkt + 1) & (CW_TH_BUCKETS - 1)) {
                for (uint32_t j = 0; j < 4u && !placed; ++j) {
                    const uint32_t prev = atomicCAS(&th[bkt * 4u + j], 0u, (tp + 1u) | CW_TH_FP(key));
                    if (prev == 0) placed = true;
                    else if ((prev & ~4095u) == CW_TH_FP(key) && tkey[CW_TH_POS(prev) - 1] == key) { trep[CW_TH_POS(prev) - 1] = 1; placed = true; } /* repeated inside the template */
                }
            }
        }
        __syncthreads();
        CW_PROF(sc.ctr, 7, tid == 0);
        /* support + repeat detection: one wave per sequence, four consecutive k-mers per lane out of one 64-bit window of the packed bases.  The
           four template-table lookups of a lane are requested together, one bucket each (nine in ten end there: not a template k-mer), and so
           are the hits' updates: the four "seen in this sequence" bits go out together, and the wave takes its places in the hit list with
           one add per round. */
        const bool fp_exact = k <= 10u; /* the entry holds the whole key */
        auto support_seq = [&](auto words, const uint32_t len, const uint32_t s, uint32_t* my_seen) {
            const uint32_t nk = len >= k ? len - k + 1 : 0, nwd = (len + 15u) >> 4;
            for (uint32_t p0 = (uint32_t)lane * 4u; p0 < nk; p0 += 256u) {
                const uint32_t wi_ = p0 >> 4;
                uint64_t x_ = ((uint64_t)words[wi_] << 32) | (wi_ + 1u < nwd ? words[wi_ + 1u] : 0u);
                x_ <<= 2u * (p0 & 15u);
                uint32_t key4[4], bkt4[4], e1[4];
                if (k <= 13u) { /* (wave-uniform) four k-mers of up to 13 bases start in the first four bases of the upper word: 32-bit field extracts */
                    const uint32_t yh = (uint32_t)(x_ >> 32);
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) key4[q] = (yh >> (32u - 2u * k - 2u * q)) & kmask32_;
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q, x_ <<= 2) key4[q] = (uint32_t)(x_ >> (64u - 2u * k));
                }
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) bkt4[q] = CW_TH_HOME(key4[q]);
                /* e1[q]: the k-mer's position in the template, anything >= CW_TMAX = not a template k-mer */
                if (fp_exact) {
                    /* An entry of this key is (position + 1) + fp, fp a multiple of 4096: minus (fp + 1) it is the position (< CW_TMAX); an entry of another key
                       comes out as its position plus a non-zero multiple of 4096, an empty slot as 4095 or more.  So the bucket's answer is the minimum of four
                       differences (round 6; four masked compares and a chain of selects before) */
                    auto match = [&](const uint4 v, const uint32_t fp1) -> uint32_t { return min(min(v.x - fp1, v.y - fp1), min(v.z - fp1, v.w - fp1)); };
                    uint4 v4[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) v4[q] = p0 + q < nk ? *(const uint4*)&th[bkt4[q] * 4u] : make_uint4(0u, 0u, 0u, 0u);
                    uint32_t pend = 0;
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) { e1[q] = match(v4[q], CW_TH_FP(key4[q]) + 1u); if (e1[q] >= (uint32_t)CW_TMAX && v4[q].w != 0u) pend |= 1u << q; }
                    while (__ballot(pend != 0u) != 0ull) { /* a full bucket without the key: the next one (rare) */
#pragma unroll
                        for (uint32_t q = 0; q < 4u; ++q) {
                            if ((pend >> q) & 1u) {
                                bkt4[q] = (bkt4[q] + 1u) & (CW_TH_BUCKETS - 1);
                                const uint4 v = *(const uint4*)&th[bkt4[q] * 4u];
                                e1[q] = match(v, CW_TH_FP(key4[q]) + 1u);
                                if (e1[q] < (uint32_t)CW_TMAX || v.w == 0u) pend &= ~(1u << q);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) e1[q] = p0 + q < nk ? (uint32_t)cw_tpl_lookup(th, tkey, key4[q]) : 0xFFFFFFFFu; /* (-1: not there) */
                }
                uint32_t old4[4];
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) {
                    const uint32_t e = e1[q];
                    old4[q] = e < (uint32_t)CW_TMAX ? atomicOr(&my_seen[e >> 5], 1u << (e & 31u)) : 0u;
                }
                uint32_t n_list = 0, my_list = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) {
                    const uint32_t e = e1[q], p = p0 + q;
                    const bool hit = e < (uint32_t)CW_TMAX;
                    if (hit) {
                        if (old4[q] & (1u << (e & 31u))) trep[e] = 1;
                        else atomicAdd(&tsup[e], 1u);
                        if (tfit) P_lds[__umul24(e, Np) + s] = (uint16_t)p; /* a k-mer seen twice never becomes an anchor: any of its positions will do (both factors fit 24 bits: the full-rate multiply) */
                        else if (hl && p >= 2048u) misc[5] = 1;
                    }
                    if (!tfit && hl) {
                        const unsigned long long hm = __ballot(hit && p < 2048u);
                        if (hit && p < 2048u) my_list |= (n_list + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))) << (8u * q); /* at most 256 hits per round */
                        n_list += (uint32_t)__popcll(hm);
                    }
                }
                if (!tfit && hl && n_list) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&misc[4], n_list);
                    base = (uint32_t)cw_lane_value((int)base, 0);
#pragma unroll
                    for (uint32_t q = 0; q < 4u; ++q) {
                        const uint32_t p = p0 + q, hi_ = base + ((my_list >> (8u * q)) & 255u);
                        if (e1[q] < (uint32_t)CW_TMAX && p < 2048u && hi_ < hit_cap) hitlist[hi_] = (e1[q] << 21) | (s << 11) | p; /* template k-mer (11 bits), sequence (10), position (11) */
                    }
                }
            }
        };
        for (uint32_t s = wave; s < N; s += CW_IDX_WAVES) {
            uint32_t* my_seen = seen + wave * seen_words; /* one bit per template k-mer */
            if ((uint32_t)lane < seen_words) my_seen[lane] = 0;
            cw_wave_sync();
            if (stw) support_seq((cw_l32)(s_words + s_off[s]), s_len[s], s, my_seen);
            else support_seq((cw_g32)(b.bases + b.seq_word_off[s0 + s]), stm ? s_len[s] : b.seq_len[s0 + s], s, my_seen);
            cw_wave_sync();
        }
        __syncthreads();
        CW_PROF(sc.ctr, 3, tid == 0);
        /* candidates in template order */
        uint32_t A;
        {
            A = 0;
            for (uint32_t tb = 0; tb < nk0; tb += CW_IDX_THREADS) { /* (the template's k-mers 1024 at a time: template order = round, then thread) */
                const uint32_t tp = tb + (uint32_t)tid;
                uint32_t ok = 0;
                if (tp < nk0) {
                    const int rep = cw_tpl_lookup(th, tkey, tkey[tp]);
                    ok = (rep == (int)tp && trep[tp] == 0 && (int)tsup[tp] >= sup_min) ? 1u : 0u;
                }
                uint32_t a_round;
                const uint32_t off = A + cw_block_exscan(ok, scan_tmp, &a_round);
                if (ok) { tcand[tp] = (int16_t)off; cand_tp[off] = (uint16_t)tp; }
                A += a_round;
                __syncthreads(); /* (scan_tmp is used again by the next round) */
            }
        }
        __syncthreads();
        CW_PROF(sc.ctr, 58, tid == 0);
        /* LDS needs: the matrix (A*Np u16) + presence bitsets (A*Nw u64) + dirty list (N u16) */
        const bool pg = !tfit && (uint64_t)A * Np * 2 + (uint64_t)A * Nw * 8 + (uint64_t)N * 2 + 16 > (uint64_t)p_cap * 2;
        if (pg && (uint64_t)A * Np > sc.p_fallback_elems) {
            if (tid == 0) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_MATRIX; sc.ctr->any_overflow = 1; }
            __builtin_amdgcn_wave_barrier(); /* the wave meets again before the back edge (see cw_stitch.h) */
            continue;
        }
/* matrix row of anchor a */
#define PROW(a) (tfit ? (uint32_t)cand_tp[a] : (uint32_t)(a))
        const uint32_t n_hits = misc[4];
        const bool from_list = hl && n_hits <= hit_cap && misc[5] == 0u;
        if (!tfit) {
            if (!pg) p_clear_lds(A * Np);
            else for (uint32_t i = tid; i < A * Np; i += CW_IDX_THREADS) PWR(i, CW_NONE16);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* the list was written by the other waves of this work-group (same CU, same L1) */
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (from_list) {
                for (uint32_t i0 = tid; i0 < n_hits; i0 += 4u * CW_IDX_THREADS) { /* four list entries per thread in flight (L2 round trips) */
                    uint32_t h4[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) { const uint32_t i = i0 + u * CW_IDX_THREADS; h4[u] = i < n_hits ? hitlist[i] : 0xFFFFFFFFu; }
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        const uint32_t h = h4[u];
                        const int a = i0 + u * CW_IDX_THREADS < n_hits ? tcand[h >> 21] : -1;
                        if (a >= 0) PWR((uint32_t)a * Np + ((h >> 11) & 1023u), h & 2047u);
                    }
                }
            } else
            for (uint32_t s = wave; s < N; s += CW_IDX_WAVES) {
                CW_IDX_PASS_SEQ({
                    const int e = cw_tpl_lookup(th, tkey, key);
                    if (e < 0) continue;
                    const int a = tcand[e];
                    if (a >= 0) PWR((uint32_t)a * Np + s, p);
                })
            }
            __syncthreads();
        }
        CW_PROF(sc.ctr, 59, tid == 0);

        /* A sequence whose anchor positions increase with the anchor index ("clean") satisfies pos(a) < pos(b) for every
           pair a < b it holds, so its contribution to score(a,b) is one bit of presence(a) & presence(b); only the few
           sequences with an out-of-order (spurious) anchor hit ("dirty") need positions compared -- and of those only the pairs
           that involve one of the sequence's out-of-order anchors: take away the anchors whose position is not above every
           earlier one (or, scanning from the end, not below every later one: whichever set is smaller) and what is left of the
           sequence is increasing again, so it goes into the presence bits like a clean one; the anchors taken away are one bit per
           dirty sequence in the anchor's "bad" mask, and the chain kernel compares positions for exactly those.  Exact. */
        uint8_t* clean = (uint8_t*)seen;                               /* N flags (2 KiB available): 1 clean, 0 dirty (forward), 2 dirty (backward); later 0x80 = dirty with masks */
        uint8_t* didx = clean + 1024;                                  /* sequence -> index in the dirty list (piles of at most 1024 sequences) */
        unsigned long long* pres = (unsigned long long*)(P_lds + (pg ? 0 : (((size_t)(tfit ? nk0 : A) * Np + 3u) & ~(size_t)3u))); /* A x Nw, 8-byte aligned */
        uint16_t* dirty = (uint16_t*)(pres + (size_t)A * Nw);          /* up to N ids */
        unsigned long long* badm = (unsigned long long*)(lds + (((size_t)((uint8_t*)(dirty + N) - lds) + 7u) & ~(size_t)7u)); /* A masks over the dirty list */
        const bool use_bits = N <= 2048u && ((uint8_t*)(dirty + N) <= lds + CW_IDX_STAGE_OFF);
        bool has_bm = false, has_delta = false;
        uint32_t n_rows = 0;
        uint32_t W = 1;                                                                 /* 64-bit words per mask: dirty sequences / 64 (at most 4) */
        uint8_t* rowid = (uint8_t*)(badm + (size_t)A * 4);                              /* anchor -> correction row, 0xFF none */
        uint16_t* rowanc = (uint16_t*)(lds + (((size_t)(rowid + A - lds) + 1u) & ~(size_t)1u)); /* correction row -> anchor */
        /* one pass of one wave over sequence s's column of the matrix, anchors in ascending (fwd) or descending order: the anchors that do
           not set a new record.  MARK: set bit d of their masks; else: count them */
        auto scan_seq = [&](const uint32_t s, const bool fwd, const bool mark, const uint32_t d) -> uint32_t {
            int run = -1;
            uint32_t n_bad = 0;
            for (uint32_t a0 = 0; a0 < A; a0 += 64) {
                const uint32_t ai = a0 + lane;
                const uint32_t a = fwd ? ai : A - 1u - ai;
                const uint32_t pv = ai < A ? PRD(PROW(a) * Np + s) : (uint32_t)CW_NONE16;
                const int v = pv != CW_NONE16 ? (fwd ? (int)pv : (int)(0xFFFEu - pv)) : -1;
                const int inc = cw_wave_scan_max(v);
                int before = cw_wave_shr1(inc, -1);
                before = max(before, run);
                const bool bad = v >= 0 && v <= before;
                if (mark) { if (bad) atomicOr(&badm[(size_t)a * W + (d >> 6)], 1ull << (d & 63u)); }
                else n_bad += (uint32_t)__popcll(__ballot(bad));
                run = max(run, cw_lane_value(inc, 63));
            }
            return n_bad;
        };
        if (use_bits) {
            for (uint32_t s = wave; s < N; s += CW_IDX_WAVES) {
                const uint32_t nf = scan_seq(s, true, false, 0);
                uint32_t c = 1;
                if (nf) c = scan_seq(s, false, false, 0) < nf ? 2u : 0u;
                if (lane == 0) clean[s] = (uint8_t)c;
            }
            if (tid == 0) misc[3] = 0;
            __syncthreads();
            CW_PROF(sc.ctr, 60, tid == 0);
            for (uint32_t s = tid; s < N; s += CW_IDX_THREADS)
                if (clean[s] != 1) dirty[atomicAdd(&misc[3], 1u)] = (uint16_t)s;
            __syncthreads();
            /* (the order of the dirty list depends on thread timing and nothing else depends on it: every use is a sum or a bit per entry) */
            const uint32_t nd = misc[3];
            /* One spurious anchor can make most of a deep pile dirty, so the masks may be up to four words (256 dirty sequences; the index of
               a dirty sequence is a byte per sequence: piles of at most 1024).  The chain kernel's fallback without correction rows knows
               one-word masks only: wider ones are used only if the rows can be produced. */
            W = (nd + 63u) >> 6;
            if (W == 0u) W = 1u;
            const bool masks = nd > 0 && nd <= 255u /* a row entry is a byte */ && W <= (N <= 1024u ? 4u : 1u) && (uint8_t*)(rowanc + CW_AB_ROWS_MAX) <= lds + CW_IDX_STAGE_OFF;
            if (masks) {
                for (uint32_t i = tid; i < A * W; i += CW_IDX_THREADS) badm[i] = 0ull;
                __syncthreads();
                for (uint32_t d = wave; d < nd; d += CW_IDX_WAVES) { const uint32_t s = dirty[d]; scan_seq(s, clean[s] == 0, true, d); }
                __syncthreads();
                /* correction rows: the anchors with a non-empty mask, numbered in anchor order */
                /* (anchors 1024 at a time, like the candidates above: two rounds when a long template has more than 1024 anchors) */
                uint32_t ok2[2] = {0u, 0u}, off2[2] = {0u, 0u};
                n_rows = 0;
                for (uint32_t r2 = 0; r2 < 2u; ++r2) {
                    const uint32_t an = r2 * CW_IDX_THREADS + (uint32_t)tid;
                    if (r2 * CW_IDX_THREADS >= A) break;
                    if (an < A) for (uint32_t x = 0; x < W; ++x) ok2[r2] |= badm[(size_t)an * W + x] != 0ull ? 1u : 0u;
                    uint32_t n_round;
                    off2[r2] = n_rows + cw_block_exscan(ok2[r2], misc + 16, &n_round);
                    n_rows += n_round;
                    __syncthreads();
                }
                has_delta = n_rows >= 1u && n_rows <= CW_AB_ROWS_MAX && cw_ab_bytes(A, N, nd, n_rows) <= ((uint64_t)w_ab_cap << 4);
                has_bm = W == 1u || has_delta;
                if (has_delta) {
                    for (uint32_t r2 = 0; r2 < 2u; ++r2) {
                        const uint32_t an = r2 * CW_IDX_THREADS + (uint32_t)tid;
                        if (an < A) rowid[an] = ok2[r2] ? (uint8_t)off2[r2] : (uint8_t)0xFF;
                        if (ok2[r2]) rowanc[off2[r2]] = (uint16_t)an;
                    }
                } else n_rows = 0;
                if (has_bm && (uint32_t)tid < nd) { const uint32_t s = dirty[tid]; clean[s] = (uint8_t)0x80u; if (N <= 1024u) didx[s] = (uint8_t)tid; else clean[s] = (uint8_t)(0x80u | (uint32_t)tid); }
                __syncthreads();
            }
            CW_PROF(sc.ctr, 61, tid == 0);
            for (uint32_t w = 0; w < Nw; ++w) { /* what a lane knows about its sequence is read once, not once per anchor */
                const uint32_t s = w * 64 + lane;
                const uint32_t c = s < N ? (uint32_t)clean[s] : 0u;
                const bool plain = c == 1u, part = has_bm && (c & 0x80u) != 0u; /* part: a dirty sequence counts where it is in order */
                const uint32_t d = part ? (N <= 1024u ? (uint32_t)didx[s] : (c & 63u)) : 0u;
                for (uint32_t a = wave; a < A; a += CW_IDX_WAVES) {
                    bool good = plain;
                    if (part) good = !((badm[(size_t)a * W + (d >> 6)] >> (d & 63u)) & 1ull);
                    const bool on = s < N && PRD(PROW(a) * Np + s) != CW_NONE16 && good;
                    const unsigned long long bal = __ballot(on);
                    if (lane == 0) pres[(size_t)a * Nw + w] = bal;
                }
            }
            __syncthreads();
        }
        const uint32_t n_dirty = use_bits ? misc[3] : 0u;
        CW_PROF(sc.ctr, 62, tid == 0);

        /* ================= hand-over: the window's anchor block =================
           Chaining is a serial recurrence over the anchors: one wave's work.  Doing it here would idle 15 of this
           work-group's 16 waves (and the CU, which the 160 KiB of LDS keeps to itself), so the candidates, the
           presence bitsets, the dirty list and the position matrix go to HBM/L2 and cw_chain_kernel finishes the
           window with one wave per window and many windows per CU. */
        {
            uint8_t* blk = sc.ablock + ((size_t)w_ab_base << 4);
            if (cw_ab_bytes(A, N, n_dirty, n_rows) > ((uint64_t)w_ab_cap << 4)) { /* cannot happen: sized from the template length */
                if (tid == 0) { wi->status = CW_WIN_OVERFLOW; wi->pad_ = CW_WHY_MATRIX; sc.ctr->any_overflow = 1; }
                __builtin_amdgcn_wave_barrier(); /* the wave meets again before the back edge (see cw_stitch.h) */
                continue;
            }
            uint32_t* hdr = (uint32_t*)blk;
            uint32_t* ckey = (uint32_t*)(blk + CW_AB_HDR);
            unsigned long long* gpres = (unsigned long long*)((uint8_t*)ckey + cw_ab_align((uint64_t)A * 4));
            uint16_t* gdirty = (uint16_t*)((uint8_t*)gpres + cw_ab_align((uint64_t)A * Nw * 8));
            unsigned long long* gbadm = (unsigned long long*)((uint8_t*)gdirty + cw_ab_align((uint64_t)n_dirty * 2));
            uint8_t* growid = (uint8_t*)gbadm + cw_ab_align((uint64_t)A * 8);
            uint8_t* gdelta = growid + cw_ab_align((uint64_t)A);
            const uint32_t Ap = cw_ab_ap(A);
            uint16_t* gP = (uint16_t*)(n_rows ? gdelta + (size_t)n_rows * Ap : growid);
            if (tid == 0) { hdr[0] = A; hdr[1] = N; hdr[2] = n_dirty; hdr[3] = (use_bits ? 1u : 0u) | (has_bm && W == 1u ? 2u : 0u) | (has_delta ? 4u : 0u); hdr[4] = n_rows; }
            for (uint32_t a = tid; a < A; a += CW_IDX_THREADS) ckey[a] = tkey[cand_tp[a]];
            if (use_bits) {
                for (uint32_t i = tid; i < A * Nw; i += CW_IDX_THREADS) gpres[i] = pres[i];
                for (uint32_t i = tid; i < n_dirty; i += CW_IDX_THREADS) gdirty[i] = dirty[i];
                if (has_bm && W == 1u) for (uint32_t a = tid; a < A; a += CW_IDX_THREADS) gbadm[a] = badm[a];
                if (has_delta) {
                    /* Correction row of anchor x: for every other anchor y, how many of the dirty sequences in which x is out of order have
                       the pair in template order (the smaller anchor in front).  A pair that is out of order at both ends in one sequence is
                       counted in the row of its smaller anchor only.  The chain kernel adds row(a)[b] + row(b)[a] to the presence count. */
                    for (uint32_t a = tid; a < A; a += CW_IDX_THREADS) growid[a] = rowid[a];
                    for (uint32_t r = wave; r < n_rows; r += CW_IDX_WAVES) {
                        const uint32_t x = rowanc[r];
                        for (uint32_t y0 = 0; y0 < Ap; y0 += 64) {
                            const uint32_t y = y0 + lane;
                            uint32_t cnt = 0;
                            if (y < A && y != x) {
                                for (uint32_t wd = 0; wd < W; ++wd) {
                                    const unsigned long long bmx = badm[(size_t)x * W + wd], bmy = badm[(size_t)y * W + wd];
                                    unsigned long long mm = y < x ? bmx & ~bmy : bmx;
                                    while (mm) {
                                        const uint32_t d = wd * 64u + (uint32_t)__ffsll((long long)mm) - 1u;
                                        mm &= mm - 1ull;
                                        const uint32_t sd = dirty[d];
                                        const uint32_t px = PRD(PROW(x) * Np + sd), py = PRD(PROW(y) * Np + sd);
                                        cnt += y > x ? ((px < py && py != CW_NONE16) ? 1u : 0u) : (py < px ? 1u : 0u); /* px is a hit: x is out of order in sd */
                                    }
                                }
                            }
                            if (y < Ap) gdelta[(size_t)r * Ap + y] = (uint8_t)cnt;
                        }
                    }
                }
            }
            {   /* rows are Np (even) u16: copy as u32 pairs, row by row (the rows of the anchors when the matrix is per template k-mer) */
                const uint32_t* src = (const uint32_t*)(pg ? P_glb : P_lds);
                uint32_t* dst = (uint32_t*)gP;
                const uint32_t half = Np >> 1;
                if (tfit) {
                    for (uint32_t a = wave; a < A; a += CW_IDX_WAVES)
                        for (uint32_t j = lane; j < half; j += 64) dst[a * half + j] = src[(uint32_t)cand_tp[a] * half + j];
                } else {
                    const uint32_t n2 = A * half;
                    if (!pg) { /* out of LDS, both ends 16-byte aligned: four words per lane and instruction */
                        const uint32_t n4 = n2 >> 2;
                        for (uint32_t i = tid; i < n4; i += CW_IDX_THREADS) ((uint4*)dst)[i] = ((const uint4*)src)[i];
                        for (uint32_t i = (n4 << 2) + tid; i < n2; i += CW_IDX_THREADS) dst[i] = src[i];
                    } else
                    for (uint32_t i = tid; i < n2; i += CW_IDX_THREADS) dst[i] = src[i];
                }
            }
            /* no flag, no fence: every early exit above changes wi->status, so "still CW_WIN_CONSENSUS when the kernel has ended"
               means the block is complete, and the kernel boundary makes it visible to cw_chain_kernel */
        }
        CW_PROF(sc.ctr, 4, tid == 0);
    }
}

#undef PRD
#undef PROW
#undef PWR
#undef CW_IDX_PASS_SEQ
#undef CW_IDX_PASS_BLOCK
#undef CW_IDX_KMERS4
#undef CW_IDX_KMERS4_WAVE
#undef CW_IDX_KMERS1
#undef CW_IDX_PASS_BLOCK2

#endif
