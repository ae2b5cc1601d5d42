/*
 * cw_stitch.h -- read re-assembly on the device: alignConsensus (correctionAlignment.cpp:47-140),
 * trimRead(.,1) and dropRead (utils.cpp:96-128, :71-73) for every read of a batch, one wave per read.
 *
 * The local alignment the reference gets from StripedSmithWaterman::Aligner is restated per include/cw_policy.h
 * (library absent: PARITY UNPINNED, bit-identical to the oracle's restatement):
 *   sweep      packed int16, loop over reference columns; H, E and the per-position best in registers; striped since round 5
 *              (st_sweep_st: a slot of NV consecutive positions per lane-half, the in-column gap F down the registers of a lane and ONE
 *              exclusive DPP prefix max per column for its passage between slots, exact because open >= ext; the chunked sweep of
 *              rounds 1-4, st_sweep_pk, is the -DCW_ST_STRIPED=0 variant); forward sweep finds (score, end), reverse sweep on the
 *              reversed prefixes finds begin (stops when the forward score is reached); consensuses beyond 2048 characters: the last
 *              launch with the sweep's state in global memory (st_sweep_mem)
 *   indels     banded traceback when two overlapping windows disagree and the earlier one wins: its rows on the lanes of the wave
 *   the read   lives in its output slot as a gap buffer (left part at the front, untouched tail right-aligned), so a
 *              replace that changes the length moves only the few hundred characters between the gap and the edit.
 */
#ifndef CW_STITCH_H
#define CW_STITCH_H

#include "cw_device.h"
#include "cw_poa.h" /* packed int16 helpers (pk_add, pk_max, pk_splat_lo, ...) */

#ifndef CW_ST_BAND_PAR
#define CW_ST_BAND_PAR 1 /* 0: the banded traceback's rows cell by cell on lane 0, as rounds 2-4 had them (variant test) */
#endif
#ifndef CW_ST_PROF
#define CW_ST_PROF 0 /* 1 (a scratch build): the debug trace holds the shader clocks of a window's phases instead of its alignment (tools/stitch_phases.py) */
#endif
#define CW_ST_WAVES 4
#define CW_ST_QMAX 2048 /* consensus length */
#define CW_ST_RMAX 2048 /* aligned slice of the read: window_size + 2*window_overlap */
#define CW_ST_DIR_BYTES (1u << 20) /* banded traceback scratch (directions, and the rows when they outgrow LDS), per wave, global */
#define CW_ST_MAX_WGS 256
#define CW_ST_ROWS_BYTES 4096u /* banded traceback rows: 3 x (2*band + 3) int32 -> band <= 169 */
/* per wave: slice codes | current consensus | previous consensus | traceback rows | query codes forward, reversed */
#define CW_ST_SLAB (CW_ST_RMAX + 2 * CW_ST_QMAX + CW_ST_ROWS_BYTES + 2 * CW_ST_QMAX)

struct StitchArgs {
    cw_read_set reads;
    const cw_stitch_read* jobs;
    uint32_t n_reads;
    const uint32_t* win_pos; /* [2 * n_windows] */
    DevBatch batch;          /* the piles the consensuses came from (template = first sequence of a window) */
    const char* cons;
    const uint64_t* cons_off;
    const uint32_t* cons_len;
    const uint8_t* win_status;
    const uint32_t* solid;
    const uint64_t* solid_off;
    const uint32_t* solid_len;
    uint32_t window_size, window_overlap, mer_size;
    int do_trim;
    char* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint8_t* read_status; /* 0 ok, 1 dropped (dropRead), 2 capacity */
    uint32_t* cursor;
    uint32_t* order;     /* reads in the order the waves take them: most windows first (cw_stitch_order_kernel) */
    int8_t* dir_scratch; /* dir_bytes per wave of the grid */
    uint32_t dir_bytes;
    uint32_t* trace; /* debug: 8 words per window, NULL in production */
    int prio;        /* wave priorities by read length (CW_STITCH_PRIO) */
    uint8_t* huge;   /* CW_STH_WAVE_BYTES per wave of the last launch (consensuses beyond CW_ST_QMAX): its buffers and the sweep's state, in global memory */
};

__device__ __forceinline__ int st_code(uint8_t c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
__device__ __forceinline__ uint8_t st_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
__device__ __forceinline__ bool st_is_upper(uint8_t c) { return c >= 'A' && c <= 'Z'; }

/* Ordering point for LDS *and* global traffic between the lanes of one wave: the read under construction lives in global memory
   (gap buffer) and is written and read back by different lanes -- of the SAME wave, whose memory instructions go through one vector L1 in
   issue order, so work-group scope (wait for the outstanding accesses; nothing to write back or invalidate) is all it takes.  Until round 5
   these were agent-scope fences: an L2 write-back and invalidate on this XCD at each of the ~30 ordering points of a window, paid by the
   consensus kernels of the other worker running beside this one as well (CW_ST_FENCE_AGENT=1 is that variant). */
#ifndef CW_ST_FENCE_AGENT
#define CW_ST_FENCE_AGENT 0
#endif
__device__ __forceinline__ void st_mem_sync() {
#if CW_ST_FENCE_AGENT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

/* the read under construction: logical string = buf[0, L) + buf[cap-R, cap) */
struct GapBuf {
    uint8_t* buf;
    uint32_t cap, L, R;
    __device__ __forceinline__ uint32_t len() const { return L + R; }
    __device__ __forceinline__ uint8_t at(uint32_t p) const { return p < L ? buf[p] : buf[cap - R + (p - L)]; }
};

/* move the gap so that the left part holds exactly the first `pos` characters (wave-parallel, pos <= len) */
__device__ __forceinline__ void st_gap_to(GapBuf& g, uint32_t pos, int lane) {
    if (pos > g.L) { /* bring characters over from the right part */
        const uint32_t n = pos - g.L;
        for (uint32_t base = 0; base < n; base += 64) { /* lowest first (ranges may overlap) */
            const uint32_t x = base + lane;
            uint8_t v = 0;
            if (x < n) v = g.buf[g.cap - g.R + x];
            st_mem_sync();
            if (x < n) g.buf[g.L + x] = v;
            st_mem_sync();
        }
        g.L += n; g.R -= n;
    } else if (pos < g.L) { /* push the end of the left part to the front of the right part; highest first (ranges may overlap) */
        const uint32_t n = g.L - pos;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t x = base + lane;
            uint8_t v = 0;
            if (x < n) v = g.buf[g.L - 1 - x];
            st_mem_sync();
            if (x < n) g.buf[g.cap - g.R - 1 - x] = v;
            st_mem_sync();
        }
        g.L -= n; g.R += n;
    }
    st_mem_sync();
}

/* Everything that steers control flow is wave-uniform; telling the compiler so (scalar registers, scalar branches) keeps
 * the 64 lanes on one path and the cross-lane operations below well defined. */
__device__ __forceinline__ int st_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t st_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct StSweep { int score, col, row; };

/*
 * One local-alignment sweep.  q[] (m codes, already in sweep order) against r[] walked from r_first in `step` until r_last_excl.
 * Two query positions per lane in packed int16 (scores <= 2 * CW_ST_QMAX fit); the column state (H, E) and the per-position best
 * (value, first column reaching it) stay in registers: no LDS traffic and no per-column reduction.  The in-column gap F is an
 * exclusive prefix max of h'[t] + t*ext in position order (exact because open >= ext).  The reference's "first column reaching
 * the best score, smallest query index in it" falls out at the end: the global maximum, the earliest column among the positions
 * holding it, the smallest position among those.  A reverse sweep stops at the first column holding `terminate` (-1: never).
 * NCH2 = chunks of 128 query positions.
 */
/* x = (query letters of two positions) ^ (the reference letter in both halves): +MATCH where a half is zero, -MISMATCH elsewhere */
__device__ __forceinline__ int st_score(int x) {
    /* min(x, 1) * -(MISMATCH + MATCH) + MATCH in both halves: two packed instructions (written out: the generic vector min was lowered to
       two compares, two selects, a permute and a shift per call -- nine instructions per chunk and column of the sweep) */
    int t, r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(x));
    const int mul = pk_make(-CW_SSW_MISMATCH - CW_SSW_MATCH, -CW_SSW_MISMATCH - CW_SSW_MATCH), add = pk_make(CW_SSW_MATCH, CW_SSW_MATCH);
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(t), "v"(mul), "v"(add));
    return r;
}

/* the column of a position's best: where the best changed (a half of ch is not zero) the half takes the column -- a packed min, a packed
   negate and one bit-field insert instead of two compares and two selects */
__device__ __forceinline__ int st_keep_col(const int ch, const int ipk, const int bcol) {
    int t;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(ch));
    const int mask = pk_sub(0, t); /* 0xFFFF where the half changed */
    return (ipk & mask) | (bcol & ~mask);
}

/* EXACT: the caller picked NCH2 for this query (m > 128 * (NCH2 - 1) or the next smaller variant does not exist): every chunk is walked
   without a branch -- a chunk beyond the query is all masks and feeds nothing below it -- so the column is one basic block and the
   chunks' prefix-max ladders fill each other's wait states. */
template <int NCH2, bool EXACT = false, bool TERM = true> /* TERM: a reverse sweep (terminate > 0); forward sweeps are compiled without the test */
__device__ __forceinline__ StSweep st_sweep_pk(const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step, int terminate,
                                               int lane) {
    const int GO = CW_SSW_GAP_OPEN, GE = CW_SSW_GAP_EXT;
    const int GOPK = pk_make(GO, GO), GEPK = pk_make(GE, GE), FADJ = pk_make(GE - GO, GE - GO);
    m = st_uni(m); r_first = st_uni(r_first); r_last_excl = st_uni(r_last_excl); terminate = st_uni(terminate);
    const int TERMPK = pk_make(terminate, terminate);
    int hprev[NCH2], ee[NCH2], jg[NCH2], amask[NCH2], qpk[NCH2], qok[NCH2], bestv[NCH2], bcol[NCH2]; /* bcol: the first column reaching bestv, two 16-bit halves (-1: none) */
#pragma unroll
    for (int c = 0; c < NCH2; ++c) {
        const int j0 = c * 128 + 2 * lane, j1 = j0 + 1;
        jg[c] = pk_make(j0 * GE, j1 * GE);
        amask[c] = (j0 < m ? 0xFFFF : 0) | (j1 < m ? (int)0xFFFF0000 : 0);
        const int q0 = j0 < m ? (int)q[j0] : 4, q1 = j1 < m ? (int)q[j1] : 4;
        qpk[c] = pk_make(q0, q1);
        qok[c] = (q0 < 4 ? 0xFFFF : 0) | (q1 < 4 ? (int)0xFFFF0000 : 0);
        hprev[c] = 0; ee[c] = 0; bestv[c] = 0; bcol[c] = -1;
    }
    int hit_col = -1;
    for (int i = r_first; i != r_last_excl; i += step) {
        const int rc = st_uni((int)r[i]);
        int carry_pair = 0;      /* H of the previous column at rows (.., 128c - 1) */
        unsigned carry_f = (unsigned)(CW_NEG16 + 32768); /* running max of h'[t] + t*GE over the rows of this column so far (biased by 32768) */
        const int rcpk = rc * 0x00010001, rc_ok = rc <= 3 ? -1 : 0;
        const int ipk = i * 0x00010001; /* the column in both halves (columns stay below 2048) */
        unsigned long long hit = 0ull;
        int zacc = -1;
        if constexpr (EXACT) {
            /* the same column written chunk-interleaved: everything that only needs the previous column for all chunks, then the six
               steps of the prefix-max ladders side by side, then the short chain through the chunks (the running maximum of the gap).
               The compiler finds this order itself for the forward sweeps and not for the reverse ones (whose columns it left chunk
               after chunk, a wait state behind every packed instruction: 346 against 246 instructions for five chunks) */
            int e_[NCH2], hp[NCH2], w[NCH2];
            unsigned key[NCH2];
#pragma unroll
            for (int c = 0; c < NCH2; ++c) {
                const int hp_ = hprev[c];
                e_[c] = pk_max(pk_max(pk_sub(ee[c], GEPK), pk_sub(hp_, GOPK)), 0);
                const int sh = CW_DPP(carry_pair, hp_, 0x138, 0xF);
                carry_pair = cw_lane_value(hp_, 63);
                const int dg = __builtin_amdgcn_alignbit(hp_, sh, 16);
                const int sv = st_score(qpk[c] ^ rcpk) & qok[c] & rc_ok;
                hp[c] = pk_max(pk_max(pk_add(dg, sv), e_[c]), 0);
                w[c] = pk_add(hp[c], jg[c]);
                const int tot = pk_max(w[c], __builtin_amdgcn_perm(w[c], w[c], 0x01000302));
                key[c] = ((unsigned)tot & 0xFFFFu) ^ 0x8000u;
            }
#pragma unroll
            for (int c = 0; c < NCH2; ++c) key[c] = max(key[c], (unsigned)CW_DPP(0, (int)key[c], 0x111, 0xF));
#pragma unroll
            for (int c = 0; c < NCH2; ++c) key[c] = max(key[c], (unsigned)CW_DPP(0, (int)key[c], 0x112, 0xF));
#pragma unroll
            for (int c = 0; c < NCH2; ++c) key[c] = max(key[c], (unsigned)CW_DPP(0, (int)key[c], 0x114, 0xF));
#pragma unroll
            for (int c = 0; c < NCH2; ++c) key[c] = max(key[c], (unsigned)CW_DPP(0, (int)key[c], 0x118, 0xF));
#pragma unroll
            for (int c = 0; c < NCH2; ++c) key[c] = max(key[c], (unsigned)CW_DPP(0, (int)key[c], 0x142, 0xA));
#pragma unroll
            for (int c = 0; c < NCH2; ++c) key[c] = max(key[c], (unsigned)CW_DPP(0, (int)key[c], 0x143, 0xC));
#pragma unroll
            for (int c = 0; c < NCH2; ++c) {
                const unsigned ex = max((unsigned)CW_DPP(0, (int)key[c], 0x138, 0xF), carry_f);
                carry_f = max(carry_f, (unsigned)cw_lane_value((int)key[c], 63));
                const int pre = pk_max(pk_splat_lo((int)(ex ^ 0x8000u)), (w[c] << 16) | (CW_NEGPK & 0xFFFF));
                const int f = pk_max(pk_add(pk_sub(pre, jg[c]), FADJ), 0);
                const int h = pk_max(hp[c], f);
                hprev[c] = h; ee[c] = e_[c];
                const int hm = h & amask[c];
                const int nb = pk_max(bestv[c], hm);
                const int ch = nb ^ bestv[c];
                bestv[c] = nb;
                bcol[c] = st_keep_col(ch, ipk, bcol[c]);
            }
        } else
#pragma unroll
        for (int c = 0; c < NCH2; ++c) {
            if (EXACT || c * 128 < m) {
                const int hp_ = hprev[c];
                int e = pk_max(pk_sub(ee[c], GEPK), pk_sub(hp_, GOPK));
                e = pk_max(e, 0);
                const int sh = CW_DPP(carry_pair, hp_, 0x138, 0xF);
                carry_pair = cw_lane_value(hp_, 63);
                const int dg = __builtin_amdgcn_alignbit(hp_, sh, 16);          /* rows (2l-1, 2l) of the previous column */
                const int sv = st_score(qpk[c] ^ rcpk) & qok[c] & rc_ok; /* positions past the query and letters other than ACGT score 0 */
                const int hp = pk_max(pk_max(pk_add(dg, sv), e), 0);
                /* F[j] = max_{t<j}(h'[t] + t*GE) - GO - (j-1)*GE: exclusive prefix max in position order */
                const int w = pk_add(hp, jg[c]);
                const int tot = pk_max(w, __builtin_amdgcn_perm(w, w, 0x01000302)); /* both halves = the lane's larger key */
                /* prefix max over the lanes on biased unsigned keys: six fused v_max_u32_dpp (see poa_fill_pk) */
                const unsigned inc = cw_wave_scan_max_u32(((unsigned)tot & 0xFFFFu) ^ 0x8000u);
                const unsigned ex = max((unsigned)CW_DPP(0, (int)inc, 0x138, 0xF), carry_f);
                carry_f = max(carry_f, (unsigned)cw_lane_value((int)inc, 63));
                const int pre = pk_max(pk_splat_lo((int)(ex ^ 0x8000u)), (w << 16) | (CW_NEGPK & 0xFFFF)); /* the odd position also sees the even one of its lane */
                const int f = pk_max(pk_add(pk_sub(pre, jg[c]), FADJ), 0);
                const int h = pk_max(hp, f);
                hprev[c] = h; ee[c] = e;
                const int hm = h & amask[c];
                const int nb = pk_max(bestv[c], hm);
                const int ch = nb ^ bestv[c];
                bestv[c] = nb;
                bcol[c] = st_keep_col(ch, ipk, bcol[c]);
            }
        }
        if (TERM) { /* a half of zacc is zero iff some position of this column holds the score: one test per column, from the column as stored */
            typedef unsigned short st_u2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int c = 0; c < NCH2; ++c)
                if (EXACT || c * 128 < m)
                    zacc = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(st_u2, zacc), __builtin_bit_cast(st_u2, (hprev[c] & amask[c]) ^ TERMPK)));
            hit = __ballot((zacc & 0xFFFF) == 0 || ((unsigned)zacc >> 16) == 0u);
        }
        if (hit) { hit_col = i; break; }
    }
    /* the best score, the first column that reached it, the smallest position holding it there */
    int lm = 0;
#pragma unroll
    for (int c = 0; c < NCH2; ++c) lm = max(lm, max((int)(short)(bestv[c] & 0xFFFF), (int)(short)((unsigned)bestv[c] >> 16)));
    const int M = hit_col >= 0 ? terminate : st_uni(cw_wave_max(lm));
    StSweep best{0, -1, 0};
    if (M <= 0) return best;
    /* sweep order = increasing i*step: compare columns through that key */
    int kc = 0x7FFFFFFF;
#pragma unroll
    for (int c = 0; c < NCH2; ++c) {
        if ((int)(short)(bestv[c] & 0xFFFF) == M) kc = min(kc, (int)(short)(bcol[c] & 0xFFFF) * step);
        if ((int)(short)((unsigned)bestv[c] >> 16) == M) kc = min(kc, (int)(short)((unsigned)bcol[c] >> 16) * step);
    }
    kc = st_uni(-cw_wave_max(-kc));
    const int col = kc * step;
    int jr = 0x7FFFFFFF;
#pragma unroll
    for (int c = 0; c < NCH2; ++c) {
        const int j0 = c * 128 + 2 * lane;
        if ((int)(short)(bestv[c] & 0xFFFF) == M && (int)(short)(bcol[c] & 0xFFFF) == col) jr = min(jr, j0);
        if ((int)(short)((unsigned)bestv[c] >> 16) == M && (int)(short)((unsigned)bcol[c] >> 16) == col) jr = min(jr, j0 + 1);
    }
    jr = st_uni(-cw_wave_max(-jr));
    best.score = M; best.col = col; best.row = jr;
    return best;
}

/* ---- the sweep of the product build, STRIPED: the 128 slots of a wave (two per lane, the halves of a packed register) each hold NV CONSECUTIVE
 * query positions (slot s: positions s*NV .. s*NV + NV - 1), one per register, instead of every register holding a chunk of 128 consecutive ones.
 * The in-column gap F then runs down the registers of a lane as plain arithmetic (F' = max(F - ext, H - open)), and only its passage from one slot
 * to the next needs the lanes: ONE exclusive prefix max per column -- over the slots, of (F leaving the slot + slot * NV * ext) -- where the chunked
 * sweep above runs one such ladder per chunk.  What enters a slot is then applied to its positions in a second pass (H = max(H, Fin - v * ext));
 * exact for the same reason as the ladder itself: open >= ext, so a cell raised by F never opens a better gap than the one that raised it.
 * Per register and column: 13 + 7 instructions (9 with the reverse sweep's stop test) against 49 for a chunk, plus ~25 per column; a five-register
 * column (a 500-base window) is ~135 issued instructions for 241 / 269.  Same recurrences, same tie rules, same results (CW_ST_STRIPED=0 is the
 * variant; tests/test_gpu_variants.py). ---- */
#ifndef CW_ST_STRIPED
#define CW_ST_STRIPED 1
#endif
__device__ __forceinline__ int st_subsat(int a, int b) { /* both halves: max(a - b, 0) for unsigned halves */
    int r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* TRACK = false (reverse sweeps): no per-position best and column -- the sweep ends at the first column that holds `terminate`, and the answer is that
   column and the smallest position holding the score in it, read from the column itself.  Returns score -1 when no column holds it (cannot happen:
   the reversed forward alignment is in the rectangle, and no cell of the rectangle exceeds the forward score); the caller then sweeps once more with
   TRACK. */
template <int NV, bool TERM, bool TRACK = true>
__device__ __forceinline__ StSweep st_sweep_st(const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step, int terminate, int lane) {
    const int GO = CW_SSW_GAP_OPEN, GE = CW_SSW_GAP_EXT;
    const int GOPK = pk_make(GO, GO), GEPK = pk_make(GE, GE), D = NV * GE, DPK = pk_make(D, D);
    m = st_uni(m); r_first = st_uni(r_first); r_last_excl = st_uni(r_last_excl); terminate = st_uni(terminate);
    const int TERMPK = pk_make(terminate, terminate);
    const int s0 = 2 * lane, s1 = s0 + 1;
    const int jgs = pk_make(s0 * D, s1 * D); /* at most 127 * 16 * ext */
    int hs[NV], ee[NV], qpk[NV], qok[NV], amask[NV], bestv[NV], bcol[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int j0 = s0 * NV + v, j1 = s1 * NV + v;
        amask[v] = (j0 < m ? 0xFFFF : 0) | (j1 < m ? (int)0xFFFF0000 : 0);
        const int q0 = j0 < m ? (int)q[j0] : 4, q1 = j1 < m ? (int)q[j1] : 4;
        qpk[v] = pk_make(q0, q1);
        qok[v] = (q0 < 4 ? 0xFFFF : 0) | (q1 < 4 ? (int)0xFFFF0000 : 0);
        hs[v] = 0; ee[v] = 0; bestv[v] = 0; bcol[v] = -1;
    }
    int hit_col = -1;
    /* The scores of a column depend on nothing but the slice letter: they are computed a column ahead (svn), between the two passes of the column
       before, where they fill the wait states of its prefix-max ladder; the letter itself is asked for at the top of that column. */
    int svn[NV];
    auto scores_of = [&](int rc_v) {
        const int rc = st_uni(rc_v);
        const int rcpk = rc * 0x00010001;
        /* a letter other than ACGT on the slice scores 0 against everything: the two constants of the score, chosen per column */
        const int mul = rc <= 3 ? pk_make(-CW_SSW_MISMATCH - CW_SSW_MATCH, -CW_SSW_MISMATCH - CW_SSW_MATCH) : 0, add = rc <= 3 ? pk_make(CW_SSW_MATCH, CW_SSW_MATCH) : 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int t, sv;
            asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(qpk[v] ^ rcpk));
            asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(sv) : "v"(t), "v"(mul), "v"(add));
            svn[v] = sv & qok[v];
        }
    };
    scores_of(r_first != r_last_excl ? (int)r[r_first] : 0);
    for (int i = r_first; i != r_last_excl; i += step) {
        const int rc_next = i + step != r_last_excl ? (int)r[i + step] : 0;
        const int ipk = i * 0x00010001;
        /* the diagonal of a slot's first position is the last position of the slot before it */
        const int hl = hs[NV - 1];
        int dg = __builtin_amdgcn_alignbit(hl, CW_DPP(0, hl, 0x138, 0xF), 16);
        int f = 0;
        int hq[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int hold = hs[v];
            const int e = pk_max(pk_sub(ee[v], GEPK), st_subsat(hold, GOPK)); /* >= 0: no cell is negative */
            const int h = pk_max(pk_max(pk_add(dg, svn[v]), e), f);
            ee[v] = e; hq[v] = h;
            f = pk_max(pk_sub(f, GEPK), st_subsat(h, GOPK));
            dg = hold;
        }
        scores_of(rc_next);
        /* what enters slot s: max over the slots t before it of (F leaving t) - (s - 1 - t) * D */
        const int w = pk_add(f, jgs);
        const int tot = pk_max(w, __builtin_amdgcn_perm(w, w, 0x01000302));
        const unsigned inc = cw_wave_scan_max_u32(((unsigned)tot & 0xFFFFu) ^ 0x8000u);
        const unsigned ex = (unsigned)CW_DPP(0, (int)inc, 0x138, 0xF);
        const int pre = pk_max(pk_splat_lo((int)(ex ^ 0x8000u)), (w << 16) | (CW_NEGPK & 0xFFFF)); /* the odd slot also sees the even one of its lane */
        int fi = pk_max(pk_add(pk_sub(pre, jgs), DPK), 0);
        int zacc = -1;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            /* (a forward sweep leaves the positions beyond the query unmasked: nothing flows from them to a position of the query, and their
               best is dropped after the last column) */
            const int h = TERM ? pk_max(hq[v], fi) & amask[v] : pk_max(hq[v], fi);
            hs[v] = h;
            fi = pk_sub(fi, GEPK);
            if (TRACK) {
                int mask; /* all ones where the best improves (written out: the compiler turns a vector shift + select into two compares, two selects and a permute) */
                asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(mask) : "v"(pk_sub(bestv[v], h)));
                bestv[v] = pk_max(bestv[v], h);
                bcol[v] = (ipk & mask) | (bcol[v] & ~mask);
            }
            if (TERM) {
                typedef unsigned short st_u2 __attribute__((ext_vector_type(2)));
                zacc = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(st_u2, zacc), __builtin_bit_cast(st_u2, h ^ TERMPK)));
            }
        }
        if (TERM) {
            if (__ballot((zacc & 0xFFFF) == 0 || ((unsigned)zacc >> 16) == 0u)) { hit_col = i; break; }
        }
    }
    StSweep best{0, -1, 0};
    if constexpr (!TRACK) {
        if (hit_col < 0) { best.score = -1; return best; }
        int jr = 0x7FFFFFFF;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if ((int)(hs[v] & 0xFFFF) == terminate) jr = min(jr, s0 * NV + v);
            if ((int)((unsigned)hs[v] >> 16) == terminate) jr = min(jr, s1 * NV + v);
        }
        best.score = terminate; best.col = hit_col; best.row = st_uni(-cw_wave_max(-jr));
        return best;
    }
    int lm = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (!TERM) bestv[v] &= amask[v];
        lm = max(lm, max((int)(short)(bestv[v] & 0xFFFF), (int)(short)((unsigned)bestv[v] >> 16)));
    }
    const int M = hit_col >= 0 ? terminate : st_uni(cw_wave_max(lm));
    if (M <= 0) return best;
    int kc = 0x7FFFFFFF;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if ((int)(short)(bestv[v] & 0xFFFF) == M) kc = min(kc, (int)(short)(bcol[v] & 0xFFFF) * step);
        if ((int)(short)((unsigned)bestv[v] >> 16) == M) kc = min(kc, (int)(short)((unsigned)bcol[v] >> 16) * step);
    }
    kc = st_uni(-cw_wave_max(-kc));
    const int col = kc * step;
    int jr = 0x7FFFFFFF;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if ((int)(short)(bestv[v] & 0xFFFF) == M && (int)(short)(bcol[v] & 0xFFFF) == col) jr = min(jr, s0 * NV + v);
        if ((int)(short)((unsigned)bestv[v] >> 16) == M && (int)(short)((unsigned)bcol[v] >> 16) == col) jr = min(jr, s1 * NV + v);
    }
    jr = st_uni(-cw_wave_max(-jr));
    best.score = M; best.col = col; best.row = jr;
    return best;
}

/* a reverse sweep: without the per-position bookkeeping first */
template <int NV, bool TERM>
__device__ __forceinline__ StSweep st_sweep_st2(const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step, int terminate, int lane) {
    if constexpr (TERM) {
        const StSweep sw = st_sweep_st<NV, true, false>(q, m, r, r_first, r_last_excl, step, terminate, lane);
        if (sw.score >= 0) return sw;
    }
    return st_sweep_st<NV, TERM, true>(q, m, r, r_first, r_last_excl, step, terminate, lane);
}

/* ---- the same striped sweep for a consensus of ANY length (the slow path of the re-assembly: a window whose consensus is beyond CW_ST_QMAX
 * characters -- chance anchors with k < 8 -- is taken by the last launch, cw_stitch_kernel<CW_STH_QMAX, ...>): NV = ceil(m / 128) registers per
 * slot do not exist, so a slot's positions live in this wave's global scratch, one 16-byte record (H, E, best, first column of the best) and one
 * word of letters per lane and position index, read and written twice per column (the next record is asked for before the present one is worked
 * on).  Same recurrences, same tie rules as st_sweep_st; columns stay below 2048, scores below 2 * 2048. ---- */
#define CW_STH_QMAX 32768
#define CW_STH_NV (CW_STH_QMAX / 128)
#define CW_STH_STATE_BYTES ((size_t)CW_STH_NV * 64 * 20)
#define CW_STH_WAVE_BYTES (((((size_t)CW_ST_RMAX + 4 * (size_t)CW_STH_QMAX + CW_ST_ROWS_BYTES + 255u) & ~(size_t)255u) + CW_STH_STATE_BYTES + 255u) & ~(size_t)255u)
#define CW_STH_MAX_WGS 64
template <bool TERM>
__device__ __forceinline__ StSweep st_sweep_mem(const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step, int terminate, int lane, uint8_t* state) {
    const int GO = CW_SSW_GAP_OPEN, GE = CW_SSW_GAP_EXT;
    const int GOPK = pk_make(GO, GO), GEPK = pk_make(GE, GE);
    m = st_uni(m); r_first = st_uni(r_first); r_last_excl = st_uni(r_last_excl); terminate = st_uni(terminate);
    const int NV = (m + 127) >> 7;
    const int D = NV * GE; /* at most 256 */
    const int TERMPK = pk_make(terminate, terminate);
    const int s0 = 2 * lane, s1 = s0 + 1;
    int4* const rec = (int4*)state + lane;                             /* [v * 64]: x = H, y = E, z = best, w = its first column */
    int* const let = (int*)(state + (size_t)CW_STH_NV * 64 * 16) + lane; /* [v * 64]: the two letters */
    for (int v = 0; v < NV; ++v) {
        const int j0 = s0 * NV + v, j1 = s1 * NV + v;
        const int q0 = j0 < m ? (int)q[j0] : 4, q1 = j1 < m ? (int)q[j1] : 4;
        let[v * 64] = pk_make(q0, q1);
        rec[v * 64] = make_int4(0, 0, 0, -1);
    }
    st_mem_sync();
    int hit_col = -1;
    for (int i = r_first; i != r_last_excl; i += step) {
        const int rc = st_uni((int)r[i]);
        const int rcpk = rc * 0x00010001;
        const int mul = rc <= 3 ? pk_make(-CW_SSW_MISMATCH - CW_SSW_MATCH, -CW_SSW_MISMATCH - CW_SSW_MATCH) : 0, add = rc <= 3 ? pk_make(CW_SSW_MATCH, CW_SSW_MATCH) : 0;
        const int ipk = i * 0x00010001;
        const int hl = rec[(NV - 1) * 64].x;
        int dg = __builtin_amdgcn_alignbit(hl, CW_DPP(0, hl, 0x138, 0xF), 16);
        int f = 0;
        int4 cur = rec[0];
        int ql = let[0];
        for (int v = 0; v < NV; ++v) {
            const int vn = v + 1 < NV ? v + 1 : v;
            const int4 nxt = rec[vn * 64];
            const int qn = let[vn * 64];
            const int hold = cur.x;
            const int qok = ((ql & 0xFFFF) < 4 ? 0xFFFF : 0) | (((unsigned)ql >> 16) < 4u ? (int)0xFFFF0000 : 0);
            int t, sv;
            asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(ql ^ rcpk));
            asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(sv) : "v"(t), "v"(mul), "v"(add));
            sv &= qok;
            const int e = pk_max(pk_sub(cur.y, GEPK), st_subsat(hold, GOPK));
            const int h = pk_max(pk_max(pk_add(dg, sv), e), f);
            f = pk_max(pk_sub(f, GEPK), st_subsat(h, GOPK));
            dg = hold;
            rec[v * 64] = make_int4(h, e, cur.z, cur.w);
            cur = nxt; ql = qn;
        }
        /* what enters a slot, in 32 bits (slot * D outgrows a half here) */
        const int klo = (int)(short)(f & 0xFFFF) + s0 * D, khi = (f >> 16) + s1 * D;
        const unsigned inc = cw_wave_scan_max_u32((unsigned)(max(klo, khi) + 0x40000000));
        const int ex = (int)(unsigned)CW_DPP(0, (int)inc, 0x138, 0xF) - 0x40000000; /* lane 0: far below anything */
        const int flo = max(ex - s0 * D + D, 0), fhi = max(max(ex, klo) - s1 * D + D, 0);
        int fi = pk_make(flo, fhi);
        int zacc = -1;
        st_mem_sync();
        cur = rec[0];
        for (int v = 0; v < NV; ++v) {
            const int vn = v + 1 < NV ? v + 1 : v;
            const int4 nxt = rec[vn * 64];
            const int j0 = s0 * NV + v, j1 = s1 * NV + v;
            const int amask = (j0 < m ? 0xFFFF : 0) | (j1 < m ? (int)0xFFFF0000 : 0);
            const int h = pk_max(cur.x, fi) & amask;
            fi = pk_sub(fi, GEPK);
            int mask;
            asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(mask) : "v"(pk_sub(cur.z, h)));
            rec[v * 64] = make_int4(h, cur.y, pk_max(cur.z, h), (ipk & mask) | (cur.w & ~mask));
            if (TERM) {
                typedef unsigned short st_u2 __attribute__((ext_vector_type(2)));
                zacc = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(st_u2, zacc), __builtin_bit_cast(st_u2, h ^ TERMPK)));
            }
            cur = nxt;
        }
        st_mem_sync();
        if (TERM) {
            if (__ballot((zacc & 0xFFFF) == 0 || ((unsigned)zacc >> 16) == 0u)) { hit_col = i; break; }
        }
    }
    int lm = 0;
    for (int v = 0; v < NV; ++v) { const int b = rec[v * 64].z; lm = max(lm, max((int)(short)(b & 0xFFFF), (int)(short)((unsigned)b >> 16))); }
    const int M = hit_col >= 0 ? terminate : st_uni(cw_wave_max(lm));
    StSweep best{0, -1, 0};
    if (M <= 0) return best;
    int kc = 0x7FFFFFFF;
    for (int v = 0; v < NV; ++v) {
        const int4 c = rec[v * 64];
        if ((int)(short)(c.z & 0xFFFF) == M) kc = min(kc, (int)(short)(c.w & 0xFFFF) * step);
        if ((int)(short)((unsigned)c.z >> 16) == M) kc = min(kc, (int)(short)((unsigned)c.w >> 16) * step);
    }
    kc = st_uni(-cw_wave_max(-kc));
    const int col = kc * step;
    int jr = 0x7FFFFFFF;
    for (int v = 0; v < NV; ++v) {
        const int4 c = rec[v * 64];
        if ((int)(short)(c.z & 0xFFFF) == M && (int)(short)(c.w & 0xFFFF) == col) jr = min(jr, s0 * NV + v);
        if ((int)(short)((unsigned)c.z >> 16) == M && (int)(short)((unsigned)c.w >> 16) == col) jr = min(jr, s1 * NV + v);
    }
    jr = st_uni(-cw_wave_max(-jr));
    best.score = M; best.col = col; best.row = jr;
    return best;
}

/* ---- the same sweep on several waves of one work-group (a read is a serial chain of windows, and a window is two sweeps of ~600
 * columns: with one wave per read a launch lasts as long as its longest read, ~0.4 ms per window).  The query's chunks of 128 positions
 * are dealt to the waves in order -- one chunk per wave up to 128 x waves positions, two beyond -- and a column moves down the waves as
 * a pipeline: within a column, chunk c needs from chunk c - 1 only the running maximum of the in-column gap (16 bits) and, for the
 * next column, the H pair of its last lane (32 bits).  Both travel in ONE 64-bit LDS word per column, tagged with the column (ring of
 * eight per wave, the consumer publishes how far it has read).  Everything of a column that does not depend on the word is issued
 * before the wave polls for it.  A reverse sweep ends at the first column that holds the forward score: the wave that finds it lowers
 * `stop`; waves ahead of it have walked a few columns further, which cannot change the result (no cell of the reverse rectangle
 * exceeds the forward score, and a position's first column is only recorded when its best improves).  Results are reduced over the
 * waves through LDS; three work-group barriers per sweep. ---- */
#define CW_STS_WAVES 5
#define CW_STS_CPW 2 /* chunks per wave at most: consensuses up to 1280 positions */
#define CW_STS_QMAX (CW_STS_WAVES * CW_STS_CPW * 128)
#define CW_STS_RMAX 640
#define CW_STS_RING 8
struct StSys {
    unsigned long long mail[CW_STS_WAVES][CW_STS_RING]; /* [producer][column & 7]: F maximum (biased u16) | H pair << 16 | (column + 1) << 48 */
    uint32_t progress[8];                                /* [consumer]: columns read so far */
    int stop;                                            /* first column index (in sweep order) holding the terminate score; INT_MAX: none */
    int red[3][8];
    uint32_t q_off, r_off;                               /* the request: LDS byte offsets of the query and the slice */
    int m, r_first, r_last_excl, step, terminate, quit;
    int fail;                                            /* a bounded wait ran out (cannot happen): the read is reported as CW_READ_CAPACITY */
};

__device__ __forceinline__ StSweep st_sweep_sys(StSys* sm, const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step, int terminate,
                                                const int lane, const int wv) {
    const int GO = CW_SSW_GAP_OPEN, GE = CW_SSW_GAP_EXT;
    const int GOPK = pk_make(GO, GO), GEPK = pk_make(GE, GE), FADJ = pk_make(GE - GO, GE - GO);
    m = st_uni(m); r_first = st_uni(r_first); r_last_excl = st_uni(r_last_excl); terminate = st_uni(terminate); step = st_uni(step);
    const int TERMPK = pk_make(terminate, terminate);
    const int cpw = m <= CW_STS_WAVES * 128 ? 1 : 2;
    const int chunk0 = wv * cpw;
    const bool active = chunk0 * 128 < m;
    const bool has_prev = wv > 0, has_next = wv + 1 < CW_STS_WAVES && (wv + 1) * cpw * 128 < m;
    int hprev[CW_STS_CPW], ee[CW_STS_CPW], jg[CW_STS_CPW], amask[CW_STS_CPW], qpk[CW_STS_CPW], qok[CW_STS_CPW], bestv[CW_STS_CPW], bce[CW_STS_CPW], bco[CW_STS_CPW];
#pragma unroll
    for (int c = 0; c < CW_STS_CPW; ++c) {
        const int j0 = (chunk0 + c) * 128 + 2 * lane, j1 = j0 + 1;
        jg[c] = pk_make(j0 * GE, j1 * GE);
        amask[c] = (j0 < m ? 0xFFFF : 0) | (j1 < m ? (int)0xFFFF0000 : 0);
        const int q0 = j0 < m ? (int)q[j0] : 4, q1 = j1 < m ? (int)q[j1] : 4;
        qpk[c] = pk_make(q0, q1);
        qok[c] = (q0 < 4 ? 0xFFFF : 0) | (q1 < 4 ? (int)0xFFFF0000 : 0);
        hprev[c] = 0; ee[c] = 0; bestv[c] = 0; bce[c] = -1; bco[c] = -1;
    }
    const int ncols = (r_last_excl - r_first) * step; /* step is +1 or -1 */
    typedef __attribute__((address_space(3))) volatile unsigned long long* st_l64;
    typedef __attribute__((address_space(3))) volatile uint32_t* st_l32;
    typedef __attribute__((address_space(3))) volatile int* st_li32;
    int carry_pair = 0;  /* H of the previous column at the last rows of the wave before */
    uint32_t seen = 0;   /* columns the next wave has read, as last looked up */
    if (active) {
        /* the slice letter and the stop column are requested a column ahead, before the wave polls for its word: a column's only
           dependent round trip to LDS is the poll (a stop seen a column late costs one harmless column, see above) */
        int rc_v = ncols > 0 ? (int)r[r_first] : 0, stop_v = 0x7FFFFFFF;
        for (int k = 0; k < ncols; ++k) {
            if (k > st_uni(stop_v)) break;
            const int i = r_first + k * step;
            const int rc = st_uni(rc_v);
            rc_v = k + 1 < ncols ? (int)r[i + step] : 0;
            stop_v = *(st_li32)&sm->stop;
            const int rcpk = rc * 0x00010001, rc_ok = rc <= 3 ? -1 : 0;
            /* what does not depend on the wave before */
            int e_[CW_STS_CPW], hp[CW_STS_CPW], w[CW_STS_CPW];
            unsigned inc[CW_STS_CPW];
            int cp = carry_pair;
#pragma unroll
            for (int c = 0; c < CW_STS_CPW; ++c) {
                if (c < cpw && (chunk0 + c) * 128 < m) {
                    const int hp_ = hprev[c];
                    int e = pk_max(pk_sub(ee[c], GEPK), pk_sub(hp_, GOPK));
                    e = pk_max(e, 0);
                    const int sh = CW_DPP(cp, hp_, 0x138, 0xF);
                    cp = cw_lane_value(hp_, 63);
                    const int dg = __builtin_amdgcn_alignbit(hp_, sh, 16);
                    const int sv = st_score(qpk[c] ^ rcpk) & qok[c] & rc_ok;
                    hp[c] = pk_max(pk_max(pk_add(dg, sv), e), 0);
                    e_[c] = e;
                    w[c] = pk_add(hp[c], jg[c]);
                    const int tot = pk_max(w[c], __builtin_amdgcn_perm(w[c], w[c], 0x01000302));
                    inc[c] = cw_wave_scan_max_u32(((unsigned)tot & 0xFFFFu) ^ 0x8000u);
                }
            }
            /* the word of the wave before */
            unsigned carry_f = (unsigned)(CW_NEG16 + 32768);
            int pair_next = 0;
            bool gone = false;
            if (has_prev) {
                const st_l64 slot = (st_l64)&sm->mail[wv - 1][k & (CW_STS_RING - 1)];
                for (uint32_t spin = 0;; ++spin) {
                    const unsigned long long msg = *slot;
                    const uint32_t lo = st_uni((uint32_t)msg), hi = st_uni((uint32_t)(msg >> 32));
                    if ((hi >> 16) == (uint32_t)(k + 1)) { carry_f = lo & 0xFFFFu; pair_next = (int)((lo >> 16) | (hi << 16)); break; }
                    if (k > st_uni(*(st_li32)&sm->stop)) { gone = true; break; }
                    if (spin > (1u << 22)) { if (lane == 0) { sm->fail = 1; atomicMin(&sm->stop, -1); } gone = true; break; } /* cannot happen: every wait is bounded all the same */
                }
                if (gone) break;
                if (lane == 0) *(st_l32)&sm->progress[wv] = (uint32_t)(k + 1);
            }
            unsigned long long hit = 0ull;
            int h_last = 0;
#pragma unroll
            for (int c = 0; c < CW_STS_CPW; ++c) {
                if (c < cpw && (chunk0 + c) * 128 < m) {
                    const unsigned ex = max((unsigned)CW_DPP(0, (int)inc[c], 0x138, 0xF), carry_f);
                    carry_f = max(carry_f, (unsigned)cw_lane_value((int)inc[c], 63));
                    const int pre = pk_max(pk_splat_lo((int)(ex ^ 0x8000u)), (w[c] << 16) | (CW_NEGPK & 0xFFFF));
                    const int f = pk_max(pk_add(pk_sub(pre, jg[c]), FADJ), 0);
                    const int h = pk_max(hp[c], f);
                    hprev[c] = h; ee[c] = e_[c];
                    h_last = h;
                    const int hm = h & amask[c];
                    const int nb = pk_max(bestv[c], hm);
                    const int ch = nb ^ bestv[c];
                    bestv[c] = nb;
                    bce[c] = (ch & 0xFFFF) ? i : bce[c];
                    bco[c] = ((unsigned)ch >> 16) ? i : bco[c];
                    if (terminate >= 0) {
                        const int d = hm ^ TERMPK;
                        hit |= __ballot((d & 0xFFFF) == 0 || ((unsigned)d >> 16) == 0u);
                    }
                }
            }
            if (has_next) { /* hand the column on: not before the next wave has read the word this one replaces */
                for (uint32_t spin = 0; (uint32_t)k >= seen + CW_STS_RING; ++spin) {
                    seen = st_uni(*(st_l32)&sm->progress[wv + 1]);
                    if (k > st_uni(*(st_li32)&sm->stop)) { gone = true; break; }
                    if (spin > (1u << 22)) { if (lane == 0) { sm->fail = 1; atomicMin(&sm->stop, -1); } gone = true; break; }
                    if ((uint32_t)k >= seen + CW_STS_RING) __builtin_amdgcn_s_sleep(1);
                }
                if (gone) break;
                const uint32_t pair = (uint32_t)cw_lane_value(h_last, 63);
                const unsigned long long msg = (unsigned long long)(carry_f & 0xFFFFu) | ((unsigned long long)pair << 16) | ((unsigned long long)(k + 1) << 48);
                if (lane == 0) *(st_l64)&sm->mail[wv][k & (CW_STS_RING - 1)] = msg;
            }
            carry_pair = pair_next;
            if (hit) { if (lane == 0) atomicMin(&sm->stop, k); break; }
        }
    }
    /* over the waves: the best score, the first column that reached it, the smallest position holding it there */
    int lm = 0;
#pragma unroll
    for (int c = 0; c < CW_STS_CPW; ++c) lm = max(lm, max((int)(short)(bestv[c] & 0xFFFF), (int)(short)((unsigned)bestv[c] >> 16)));
    lm = st_uni(cw_wave_max(lm));
    if (lane == 0) sm->red[0][wv] = lm;
    __syncthreads();
    const bool was_hit = st_uni(*(st_li32)&sm->stop) != 0x7FFFFFFF;
    int M = 0;
    for (int x = 0; x < CW_STS_WAVES; ++x) M = max(M, sm->red[0][x]);
    M = was_hit ? terminate : st_uni(M);
    /* the rings are free again: every wave is out of its loop */
    if (lane < CW_STS_RING) sm->mail[wv][lane] = 0ull;
    if (lane == 0) sm->progress[wv] = 0u;
    StSweep best{0, -1, 0};
    if (M <= 0) { __syncthreads(); return best; } /* (the master may post its next request only when every wave has read this one's results) */
    int kc = 0x7FFFFFFF;
#pragma unroll
    for (int c = 0; c < CW_STS_CPW; ++c) {
        if ((int)(short)(bestv[c] & 0xFFFF) == M) kc = min(kc, bce[c] * step);
        if ((int)(short)((unsigned)bestv[c] >> 16) == M) kc = min(kc, bco[c] * step);
    }
    kc = st_uni(-cw_wave_max(-kc));
    if (lane == 0) sm->red[1][wv] = kc;
    __syncthreads();
    for (int x = 0; x < CW_STS_WAVES; ++x) kc = min(kc, sm->red[1][x]);
    kc = st_uni(kc);
    const int col = kc * step;
    int jr = 0x7FFFFFFF;
#pragma unroll
    for (int c = 0; c < CW_STS_CPW; ++c) {
        const int j0 = (chunk0 + c) * 128 + 2 * lane;
        if ((int)(short)(bestv[c] & 0xFFFF) == M && bce[c] == col) jr = min(jr, j0);
        if ((int)(short)((unsigned)bestv[c] >> 16) == M && bco[c] == col) jr = min(jr, j0 + 1);
    }
    jr = st_uni(-cw_wave_max(-jr));
    if (lane == 0) sm->red[2][wv] = jr;
    __syncthreads();
    for (int x = 0; x < CW_STS_WAVES; ++x) jr = min(jr, sm->red[2][x]);
    best.score = M; best.col = col; best.row = st_uni(jr);
    return best;
}

/* the master wave (wave 0, which runs the read) posts a sweep and takes part in it; the other waves wait for requests in st_sys_helper */
__device__ __forceinline__ StSweep st_sweep_post(StSys* sm, const uint8_t* lds_base, const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step,
                                                 int terminate, const int lane) {
    if (lane == 0) {
        sm->q_off = (uint32_t)(q - lds_base); sm->r_off = (uint32_t)(r - lds_base);
        sm->m = m; sm->r_first = r_first; sm->r_last_excl = r_last_excl; sm->step = step; sm->terminate = terminate; sm->quit = 0;
        sm->stop = 0x7FFFFFFF;
    }
    __syncthreads();
    const StSweep sw = st_sweep_sys(sm, q, m, r, r_first, r_last_excl, step, terminate, lane, 0);
    if (st_uni(sm->fail)) return StSweep{0, -1, 0}; /* no coordinates from a sweep that was given up */
    return sw;
}
__device__ __forceinline__ void st_sys_helper(StSys* sm, const uint8_t* lds_base, const int lane, const int wv) {
    for (;;) {
        __syncthreads();
        if (st_uni(sm->quit)) break;
        (void)st_sweep_sys(sm, lds_base + st_uni(sm->q_off), st_uni(sm->m), lds_base + st_uni(sm->r_off), st_uni(sm->r_first), st_uni(sm->r_last_excl), st_uni(sm->step),
                           st_uni(sm->terminate), lane, wv);
    }
}

template <int NCHK, bool TERM> /* NCHK: the most registers per slot (chunks) this kernel holds; 0 = the last launch, which has the memory-state sweep for what is longer still */
__device__ __forceinline__ StSweep st_sweep_any(const uint8_t* q, int m, const uint8_t* r, int r_first, int r_last_excl, int step, int terminate, int lane, uint8_t* state = nullptr) {
    if constexpr (NCHK == 0) { /* (registers for the common lengths only: the last launch is compiled for 128 of them, so that its work-groups find room beside running kernels) */
        if (st_uni(m) > 640) return st_sweep_mem<TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane, state);
    }
    /* the narrow kernel (consensuses of at most 640 positions: every 500-base window): five chunks, a third of the registers */
#if CW_ST_STRIPED
    m = st_uni(m);
    if constexpr (NCHK >= 1 && NCHK <= 8) return st_sweep_st2<NCHK, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 128) return st_sweep_st2<1, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 256) return st_sweep_st2<2, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 384) return st_sweep_st2<3, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 512) return st_sweep_st2<4, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 640 || NCHK == 0) return st_sweep_st2<5, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 768) return st_sweep_st2<6, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 1024) return st_sweep_st2<8, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 1536) return st_sweep_st2<12, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    return st_sweep_st2<CW_ST_QMAX / 128, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
#else
    if constexpr (NCHK >= 1 && NCHK <= 8) return st_sweep_pk<NCHK, false, TERM>(q, st_uni(m), r, r_first, r_last_excl, step, terminate, lane);
    /* A variant per chunk count for the common lengths (a 500-base window's consensus is 500-600 positions: five chunks), each a
       branch-free column written chunk-interleaved (st_sweep_pk, EXACT); the rare long ones keep the sixteen-chunk loop that skips the
       chunks beyond the query with a scalar branch.  Round 2 walked everything up to 1024 positions through an eight-chunk loop with those
       skips: a basic block per chunk, nothing to fill the wait states of the prefix-max ladders with, the per-position best copied
       between register sets at every branch -- ~110 issued instructions per chunk and column against 49 now (read in the ISA; a
       column of five chunks: 550 -> 241 instructions forward, 269 reverse).  Round 1 had 4-, 8- and 16-chunk variants side by side and
       the 8-chunk one returned garbage rows on gfx950 (ROCm 7.2 hipcc), each of them alone being correct; the present set is checked by
       tests/test_gpu_stitch.py (lengths on both sides of every boundary), tests/test_gpu_pipeline.py and tools/fuzz_pipeline.py. */
    m = st_uni(m);
    if (m <= 128) return st_sweep_pk<1, true, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 256) return st_sweep_pk<2, true, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 512) return st_sweep_pk<4, true, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 640) return st_sweep_pk<5, true, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 768) return st_sweep_pk<6, true, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    if (m <= 1024) return st_sweep_pk<8, true, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
    return st_sweep_pk<CW_ST_QMAX / 128, false, TERM>(q, m, r, r_first, r_last_excl, step, terminate, lane);
#endif
}

/* banded traceback (ssw banded_sw): totals of inserted / deleted bases between the alignment's ends.  Wave-uniform, serial
 * (lane 0 walks the band; rare: only when two overlapping windows disagree and the earlier one wins).
 * rows: int32 h_b/e_b/h_c (3*width) in LDS; dir: the direction bytes (width_d*readLen*3) in this wave's global scratch. */
__device__ __forceinline__ bool st_banded_indels(const uint8_t* ref, int refLen, const uint8_t* read, int readLen, int score, uint8_t* rows, uint32_t rows_bytes,
                                 int8_t* dir_all, uint32_t dir_bytes, unsigned* ins, unsigned* del, int lane, int8_t* dir_lds = nullptr, uint32_t dir_lds_bytes = 0) {
    const int GO = CW_SSW_GAP_OPEN, GE = CW_SSW_GAP_EXT;
    *ins = 0; *del = 0;
    refLen = st_uni(refLen); readLen = st_uni(readLen); score = st_uni(score);
    if (refLen <= 0 || readLen <= 0) return true;
    int band = abs(refLen - readLen) + 1;
    int8_t* dir_cur = dir_all;
    int width_d = 0, stride_d = 0;
    for (;;) {
        const int width = band * 2 + 3;
        width_d = band * 2 + 1;
        /* a band wider than the matrix only ever touches refLen + 1 cells per row: store that much (the reference allocates the full
           band; which cell holds what is unchanged) */
        const int w_rows = width < refLen + 3 ? width : refLen + 3;
        stride_d = width_d < refLen + 1 ? width_d : refLen + 1;
        const size_t rows_need = (size_t)w_rows * 12, dir_need = (size_t)stride_d * readLen * 3;
        const bool rows_in_lds = rows_need <= rows_bytes;
        if (dir_need + (rows_in_lds ? 0 : ((rows_need + 15) & ~(size_t)15)) > dir_bytes) return false;
        int* h_b = rows_in_lds ? (int*)rows : (int*)dir_all; int* e_b = h_b + w_rows; int* h_c = e_b + w_rows;
        /* the direction bytes of a narrow band over a hundred-odd rows are a kilobyte or two: in LDS when the caller has a free buffer (the walk
           back is a chain of dependent one-byte reads) */
        int8_t* dir = rows_in_lds && dir_need <= dir_lds_bytes ? dir_lds : dir_all + (rows_in_lds ? 0 : ((rows_need + 15) & ~(size_t)15));
        for (int x = lane; x < 3 * w_rows; x += 64) h_b[x] = 0; /* h_b, e_b, h_c are one block */
        for (int x = lane; x < stride_d * readLen * 3; x += 64) dir[x] = 0; /* cells outside the band read as "stop" */
        dir_cur = dir;
        st_mem_sync();
        int mx = 0;
        if (CW_ST_BAND_PAR && rows_in_lds && width_d <= 64) {
            /* a row of the band on the lanes (lane t: cell j = beg + t), the arrays and their index rules -- including the zeroed `edge` slots and whatever
               an earlier row left beyond the cells of the last one -- exactly those of the serial loop below: every lane reads its three cells of the
               row before, then all write; the in-row gap f[t] = max(h[t-1] - open, f[t-1] - ext) is an exclusive prefix max of h'[s] - open + (s+1) ext
               (h' = the cell without its f; exact as open >= ext and no h' is negative), the directions follow from the finished values. */
            for (int i = 0; i < readLen; ++i) {
                int beg = i - band, end = i + band;
                beg = beg > 0 ? beg : 0; end = end < refLen - 1 ? end : refLen - 1;
                const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
                if (lane == 0) { h_b[0] = 0; e_b[0] = 0; h_b[edge] = 0; e_b[edge] = 0; h_c[0] = 0; }
                st_mem_sync();
                const int xo = i - band > 0 ? i - band : 0, xp = i - 1 - band > 0 ? i - 1 - band : 0;
                const int j = beg + lane;
                const bool on = j <= end;
                const int u = j - xo + 1, e_i = j - xp + 1;
                int hbe = 0, ebe = 0, hbd = 0, a = 4;
                if (on) { hbe = h_b[e_i]; ebe = e_b[e_i]; hbd = h_b[e_i - 1]; a = ref[j]; }
                const int bq = st_uni((int)read[i]);
                const int t1e = i == 0 ? -GO : hbe - GO, t2e = i == 0 ? -GE : ebe - GE;
                const int e_new = t1e > t2e ? t1e : t2e;
                const int dir_e = t1e > t2e ? 3 : 2;
                const int e1 = e_new > 0 ? e_new : 0;
                const int diag = hbd + ((a == 4 || bq == 4) ? 0 : (a == bq ? CW_SSW_MATCH : -CW_SSW_MISMATCH));
                const int hq = e1 > diag ? e1 : diag;
                const unsigned key = on ? (unsigned)(hq - GO + (lane + 1) * GE + 0x40000000) : 0u;
                const unsigned inc = cw_wave_scan_max_u32(key);
                const int ex = (int)(unsigned)CW_DPP(0, (int)inc, 0x138, 0xF) - 0x40000000; /* lane 0: far below anything */
                const int f = (ex > -GE ? ex : -GE) - lane * GE;
                const int f1 = f > 0 ? f : 0;
                const int t1h = e1 > f1 ? e1 : f1;
                const int hc = t1h > diag ? t1h : diag;
                const int hc_prev = CW_DPP(0, hc, 0x138, 0xF), f_prev = CW_DPP(0, f, 0x138, 0xF); /* lane 0: h_c[0] = 0 and f = 0 before the first cell */
                const int dir_f = hc_prev - GO > f_prev - GE ? 5 : 4;
                const int dir_h = t1h <= diag ? 1 : (e1 > f1 ? dir_e : dir_f);
                if (on) {
                    e_b[u] = e_new; h_c[u] = hc;
                    int8_t* line = dir + (size_t)stride_d * i * 3 + 3 * lane;
                    line[0] = (int8_t)dir_e; line[1] = (int8_t)dir_f; line[2] = (int8_t)dir_h;
                    mx = hc > mx ? hc : mx;
                }
                st_mem_sync();
                if (on) h_b[u] = hc; /* for (j = 1; j <= u; ++j) h_b[j] = h_c[j] */
            }
            mx = st_uni(cw_wave_max(mx));
        } else if (lane == 0) {
            for (int i = 0; i < readLen; ++i) {
                int beg = 0, end = refLen - 1, u = 0;
                int j = i - band; beg = beg > j ? beg : j;
                j = i + band; end = end < j ? end : j;
                const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
                int f = 0;
                h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
                int8_t* line = dir + (size_t)stride_d * i * 3;
                for (j = beg; j <= end; ++j) {
                    int x;
#define ST_SET_U(res, ii, jj) do { x = (ii) - band; x = x > 0 ? x : 0; (res) = (jj) - x + 1; } while (0)
#define ST_SET_D(res, ii, jj, pp) do { x = (ii) - band; x = x > 0 ? x : 0; x = (jj) - x; (res) = x * 3 + (pp); } while (0)
                    int e, b_, d, de, df, dh;
                    ST_SET_U(u, i, j); ST_SET_U(e, i - 1, j); ST_SET_U(b_, i, j - 1); ST_SET_U(d, i - 1, j - 1);
                    ST_SET_D(de, i, j, 0); ST_SET_D(df, i, j, 1); ST_SET_D(dh, i, j, 2);
                    int t1 = i == 0 ? -GO : h_b[e] - GO;
                    int t2 = i == 0 ? -GE : e_b[e] - GE;
                    e_b[u] = t1 > t2 ? t1 : t2;
                    line[de] = t1 > t2 ? 3 : 2;
                    t1 = h_c[b_] - GO;
                    t2 = f - GE;
                    f = t1 > t2 ? t1 : t2;
                    line[df] = t1 > t2 ? 5 : 4;
                    const int e1 = e_b[u] > 0 ? e_b[u] : 0, f1 = f > 0 ? f : 0;
                    t1 = e1 > f1 ? e1 : f1;
                    const int a = ref[j], bq = read[i];
                    t2 = h_b[d] + ((a == 4 || bq == 4) ? 0 : (a == bq ? CW_SSW_MATCH : -CW_SSW_MISMATCH));
                    h_c[u] = t1 > t2 ? t1 : t2;
                    if (h_c[u] > mx) mx = h_c[u];
                    if (t1 <= t2) line[dh] = 1;
                    else line[dh] = e1 > f1 ? line[de] : line[df];
                }
                for (j = 1; j <= u; ++j) h_b[j] = h_c[j];
            }
        }
        mx = cw_lane_value(mx, 0);
        __threadfence_block();
        st_mem_sync();
        if (mx >= score || band > refLen + readLen) break;
        band *= 2;
    }
    unsigned ni = 0, nd = 0;
    if (lane == 0) {
        int i = readLen - 1, j = refLen - 1, state = 2;
        while (i > 0 && j >= 0) {
            const int8_t* line = dir_cur + (size_t)stride_d * i * 3;
            int x = i - band; x = x > 0 ? x : 0; x = j - x;
            const int idx = x * 3 + state;
            if (idx < 0 || idx >= width_d * 3) break;
            const int dcode = line[idx];
            if (dcode == 1) { --i; --j; state = 2; }
            else if (dcode == 2) { --i; state = 0; ++ni; }
            else if (dcode == 3) { --i; state = 2; ++ni; }
            else if (dcode == 4) { --j; state = 1; ++nd; }
            else if (dcode == 5) { --j; state = 2; ++nd; }
            else break;
        }
    }
#undef ST_SET_U
#undef ST_SET_D
    *ins = (unsigned)cw_lane_value((int)ni, 0);
    *del = (unsigned)cw_lane_value((int)nd, 0);
    return true;
}

struct StAlign { int score, ref_begin, ref_end, query_begin, query_end; };

/* full alignment: forward sweep, reverse sweep.  qfw = query codes; qrv = scratch for the reversed prefix. */
template <int NCHK, bool SYS = false>
__device__ __forceinline__ StAlign st_align(const uint8_t* qfw, int m, uint8_t* qrv, const uint8_t* ref, int n, int lane, StSys* sm = nullptr, const uint8_t* lds_base = nullptr,
                                            uint8_t* state = nullptr) {
    StAlign a{0, 0, -1, 0, -1};
    m = st_uni(m); n = st_uni(n);
    if (m <= 0 || n <= 0) return a;
    StSweep fw;
    if constexpr (SYS) fw = st_sweep_post(sm, lds_base, qfw, m, ref, 0, n, 1, -1, lane);
    else fw = st_sweep_any<NCHK, false>(qfw, m, ref, 0, n, 1, -1, lane, state);
    a.score = fw.score;
    if (fw.score <= 0) return a;
    a.ref_end = fw.col; a.query_end = fw.row;
    const int pm = fw.row + 1;
    for (int x = lane; x < pm; x += 64) qrv[x] = qfw[fw.row - x];
    st_mem_sync();
    StSweep bw;
    if constexpr (SYS) bw = st_sweep_post(sm, lds_base, qrv, pm, ref, fw.col, -1, -1, fw.score, lane);
    else bw = st_sweep_any<NCHK, true>(qrv, pm, ref, fw.col, -1, -1, fw.score, lane, state);
    a.ref_begin = bw.col; a.query_begin = fw.row - bw.row;
    return a;
}

/* number of solid k-mers of a string under the policy "anything but A, C, G is T" (correctionAlignment.cpp:6-15) */
__device__ __forceinline__ int st_nb_solid(const uint8_t* s, int len, const uint32_t* solid, uint32_t n_solid, uint32_t k, int lane) {
    int nb = 0;
    for (int p = lane; p + (int)k <= len; p += 64) {
        uint32_t v = 0;
        for (uint32_t x = 0; x < k; ++x) { const uint8_t c = s[p + x]; v = (v << 2) | (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u); }
        int lo = 0, hi = (int)n_solid - 1;
        bool hit = false;
        while (lo <= hi) { const int mid = (lo + hi) >> 1; const uint32_t t = solid[mid]; if (t == v) { hit = true; break; } if (t < v) lo = mid + 1; else hi = mid - 1; }
        nb += hit ? 1 : 0;
    }
    return st_uni(cw_wave_sum(nb));
}

/* A read is a serial chain of its windows on one wave, so the launch lasts as long as its longest read takes from the moment it is
 * picked up: hand the reads out longest first (counting sort by window count, most windows first; the order inside a class does not
 * matter -- every read writes only its own output slot). */
__global__ void __launch_bounds__(1024) cw_stitch_order_kernel(StitchArgs a) {
    __shared__ uint32_t hist[1024];
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_reads; i += 1024) atomicAdd(&hist[min(a.jobs[i].win_count, 1023u)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { /* exclusive offsets, largest class first */
        uint32_t run = 0;
        for (int c = 1023; c >= 0; --c) { const uint32_t k = hist[c]; hist[c] = run; run += k; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_reads; i += 1024) a.order[atomicAdd(&hist[min(a.jobs[i].win_count, 1023u)], 1u)] = i;
}

/* The instantiations.  The product build launches the WIDE kernel (consensus and slice of at most 2048 positions, sweeps of 1..16 registers per slot)
   over all reads, then the LAST one (NCHK = 0: consensuses up to CW_STH_QMAX, everything in global memory) over the reads the wide one marked
   CW_READ_REDO -- normally none.  REDO = a launch that takes only marked reads.  The test-aid build also has the NARROW kernel (consensus and slice
   of at most 640 positions -- every window of the wrappers' defaults, 500 + 2 x 50 -- five registers per slot, 7.3 KB of LDS per wave: twice the
   reads in flight per CU; what does not fit it is marked and taken by the wide one as a REDO launch) and the several-waves-per-read one (SYS);
   both bit-identical and measured no faster. */
#define CW_READ_REDO 0xFEu
#define CW_ST_SLAB_OF(QMAX, RMAX) ((RMAX) + 2 * (QMAX) + CW_ST_ROWS_BYTES + 2 * (QMAX))
#define CW_STN_QMAX 640
#define CW_STN_RMAX 640
#define CW_STN_WAVES 4
/* SYS: one read per work-group of CW_STS_WAVES waves; wave 0 runs the read exactly as the one-wave kernels do, and every sweep is shared
   with the other waves (st_sweep_sys).  WAVES = 1 then (slabs and scratch are per read). */
template <int QMAX, int RMAX, int NCHK, int WAVES, bool REDO, bool SYS = false>
/* (the wide kernel needs 243 + 16 registers: one wave per SIMD, 1024 reads in flight.  Capped at 256 for two waves per SIMD the launch
   is SLOWER, 60.5 against 55.3 ms per job of 32768 windows: it lasts as long as its longest read, and that read's wave then shares its SIMD) */
/* (the last launch is compiled for 128 registers and holds the register sweeps of the common lengths only: normally it has nothing to do, and as a
   334-register kernel its work-groups waited 4 ms for a SIMD that free beside the other workers' kernels) */
__global__ void __launch_bounds__(SYS ? 64 * CW_STS_WAVES : 64 * WAVES, NCHK == 0 ? 4 : 1) cw_stitch_kernel(StitchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = SYS ? 0 : threadIdx.x >> 6;
    constexpr bool HUGE = NCHK == 0; /* the last launch: one wave per work-group, every buffer in global memory, consensuses up to CW_STH_QMAX */
    const uint32_t too_big = !HUGE && a.huge ? (uint32_t)CW_READ_REDO : (uint32_t)CW_READ_CAPACITY; /* a kernel hands on what the last one reports as a capacity */
    StSys* const sm = (StSys*)(lds + (((size_t)CW_ST_SLAB_OF(QMAX, RMAX) + 15u) & ~(size_t)15u));
    if constexpr (SYS) {
        static_assert(WAVES == 1 && QMAX <= CW_STS_QMAX && RMAX <= CW_STS_RMAX, "one read per work-group");
        const int wv = threadIdx.x >> 6;
        if (lane < CW_STS_RING) sm->mail[wv][lane] = 0ull;
        if (lane == 0) { sm->progress[wv] = 0u; if (wv == 0) sm->fail = 0; }
        if (wv != 0) { st_sys_helper(sm, lds, lane, wv); return; }
    }
    uint8_t* slab = HUGE ? a.huge + (size_t)blockIdx.x * CW_STH_WAVE_BYTES : lds + (size_t)wave * CW_ST_SLAB_OF(QMAX, RMAX);
    uint8_t* const state = HUGE ? slab + (((size_t)CW_ST_SLAB_OF(QMAX, RMAX) + 255u) & ~(size_t)255u) : nullptr;
    uint8_t* refc = slab;                              /* RMAX codes of the aligned slice              */
    uint8_t* cur = refc + RMAX;                        /* current consensus (chars), QMAX              */
    uint8_t* old = cur + QMAX;                         /* previous window's consensus as written, QMAX */
    uint8_t* rows = old + QMAX;                        /* CW_ST_ROWS_BYTES: the three int32 rows of the banded traceback */
    uint8_t* qfw = rows + CW_ST_ROWS_BYTES;            /* QMAX query codes                              */
    uint8_t* qrv = qfw + QMAX;                         /* QMAX reversed prefix / build area             */
    int8_t* dirbuf = a.dir_scratch + ((size_t)blockIdx.x * WAVES + wave) * a.dir_bytes;
    if constexpr (HUGE) { if (st_uni(*(volatile uint32_t*)(a.cursor + 3)) == 0u) return; } /* no read was marked: normally */
    for (;;) {
        uint32_t ri = 0;
        if (lane == 0) ri = atomicAdd(a.cursor + (HUGE ? 2 : REDO ? 1 : 0), 1u);
        ri = (uint32_t)cw_lane_value((int)ri, 0);
        if (ri >= a.n_reads) break;
        /* the launch lasts as long as its longest read: the waves that hold the longest reads (handed out first) issue before the others */
        if (a.prio) { if (ri < a.n_reads / 32u + 1u) __builtin_amdgcn_s_setprio(3); else if (ri < a.n_reads / 8u) __builtin_amdgcn_s_setprio(2); else if (ri < a.n_reads / 2u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        ri = st_uni(a.order[ri]);
        if (REDO && st_uni((uint32_t)a.read_status[ri]) != (uint32_t)CW_READ_REDO) continue;
        cw_stitch_read jb = a.jobs[ri];
        jb.read = st_uni(jb.read); jb.win_first = st_uni(jb.win_first); jb.win_count = st_uni(jb.win_count);
        const uint32_t rlen = st_uni(a.reads.read_len[jb.read]);
        const uint32_t* rwords = a.reads.bases + a.reads.read_word_off[jb.read];
        GapBuf g;
        g.buf = (uint8_t*)a.out + a.out_off[ri];
        g.cap = st_uni((uint32_t)(a.out_off[ri + 1] - a.out_off[ri]));
        /* One way through the loop body: every read ends at the single store + wave barrier at the bottom.  (A lane-0 store
           followed by `continue` lets the compiler send lane 0 round a back edge of its own; the cross-lane broadcast at the
           top of the next iteration then runs without it.) */
        uint32_t status = g.cap < rlen ? (uint32_t)CW_READ_CAPACITY : 0u;
        const uint32_t n_fill = status ? 0u : rlen;
        /* outSequence = lower-case read (:56-57), everything in the right part: the gap starts at 0 */
        for (uint32_t x = lane; x < n_fill; x += 64) g.buf[g.cap - rlen + x] = "acgt"[cw_base_at(rwords, x)];
        g.L = 0; g.R = n_fill;
        st_mem_sync();

        int cur_pos = jb.win_count ? st_uni((int)a.win_pos[2 * jb.win_first]) : 0;   /* startPos = pilesPos[0].first (CONSENT-correction.cpp:47) */
        uint32_t old_end = 0, old_len = 0, old_w = 0;
        bool have_old = false;
        for (uint32_t wi = 0; wi < jb.win_count && status == 0; ++wi) {
            const uint32_t w = jb.win_first + wi;
#if CW_ST_PROF
            long long pc[6] = {0, 0, 0, 0, 0, 0}, pt = clock64();
#define ST_PROF(k) do { const long long n_ = clock64(); pc[k] += n_ - pt; pt = n_; } while (0)
#else
#define ST_PROF(k) do { } while (0)
#endif
            if (st_uni((int)a.win_status[w]) == CW_WIN_OVERFLOW) continue;                 /* no consensus for this window: leave the read as it is */
            uint32_t clen = st_uni(a.cons_len[w]);
            const bool long_enough = clen >= a.mer_size;                      /* :75 / :98 / :125 */
            if (long_enough) {
                if (clen > QMAX) { status = too_big; break; }
                const char* src = a.cons + a.cons_off[w];
                for (uint32_t x = lane; x < clen; x += 64) cur[x] = (uint8_t)src[x];
            } else {                                                          /* :76 the window's template */
                const uint32_t ts = a.batch.win_first_seq[w];
                clen = st_uni((a.batch.win_first_seq[w + 1] > ts) ? a.batch.seq_len[ts] : 0u);
                if (clen > QMAX) { status = too_big; break; }
                const uint32_t* tw = a.batch.bases + a.batch.seq_word_off[ts];
                for (uint32_t x = lane; x < clen; x += 64) cur[x] = CW_ACGT(cw_base_at(tw, x));
            }
            st_mem_sync();
            const int al_pos = max(0, cur_pos - (int)a.window_overlap);                                    /* :83 */
            const uint32_t tot = g.len();
            int size_al;
            if ((uint64_t)al_pos + a.window_size + 2ull * a.window_overlap >= tot) size_al = (int)tot - al_pos;   /* :84-88 */
            else size_al = (int)(a.window_size + 2 * a.window_overlap);
            if (size_al <= 0 || clen == 0) continue;
            if (size_al > RMAX) { status = too_big; break; }
            for (int x = lane; x < size_al; x += 64) refc[x] = (uint8_t)st_code(g.at((uint32_t)al_pos + x));
            for (uint32_t x = lane; x < clen; x += 64) qfw[x] = (uint8_t)st_code(cur[x]);
            st_mem_sync();
            ST_PROF(0);
            const StAlign al = st_align<NCHK, SYS>(qfw, (int)clen, qrv, refc, size_al, lane, sm, lds, state);     /* :90 */
            ST_PROF(1);
            if (!CW_ST_PROF && a.trace && lane == 0) {
                uint32_t* t = a.trace + 8 * (size_t)w;
                t[0] = (uint32_t)al_pos; t[1] = (uint32_t)size_al; t[2] = (uint32_t)al.score; t[3] = (uint32_t)al.ref_begin; t[4] = (uint32_t)al.ref_end;
                t[5] = (uint32_t)al.query_begin; t[6] = (uint32_t)al.query_end; t[7] = clen;
            }
            if (al.score <= 0) continue;
            const uint32_t beg = (uint32_t)(al.ref_begin + al_pos), end = (uint32_t)(al.ref_end + al_pos); /* :91-92 */
            /* curCons = curCons.substr(query_begin, ...) (:93): shift down in place */
            uint32_t cl = (uint32_t)(al.query_end - al.query_begin + 1);
            if (al.query_begin > 0) {
                for (uint32_t base = 0; base < cl; base += 64) {
                    const uint32_t x = base + lane;
                    uint8_t v = 0;
                    if (x < cl) v = cur[al.query_begin + x];
                    st_mem_sync();
                    if (x < cl) cur[x] = v;
                    st_mem_sync();
                }
            }
            bool emptied = false;
            ST_PROF(2);
            if (wi != 0 && have_old && old_end >= beg) {                                                   /* :96 */
                const uint32_t overlap = old_end - beg + 1;
                if (long_enough && old_len >= overlap && cl >= overlap) {                                   /* :98 */
                    const uint8_t* seq1 = old + (old_len - overlap);                                        /* :99 */
                    bool diff = false;
                    for (uint32_t x = lane; x < overlap; x += 64) diff = diff || (st_upper(seq1[x]) != st_upper(cur[x]));
                    if (__ballot(diff) != 0ull) {                                                           /* :101 */
                        int s1, s2;
                        if (overlap >= a.mer_size) {                                                        /* :102-104 */
                            s1 = st_nb_solid(seq1, (int)overlap, a.solid + a.solid_off[old_w], a.solid_len[old_w], a.mer_size, lane);
                            s2 = st_nb_solid(cur, (int)overlap, a.solid + a.solid_off[w], a.solid_len[w], a.mer_size, lane);
                        } else {                                                                            /* :106-107 */
                            int u1 = 0, u2 = 0;
                            for (uint32_t x = lane; x < overlap; x += 64) { u1 += st_is_upper(seq1[x]) ? 1 : 0; u2 += st_is_upper(cur[x]) ? 1 : 0; }
                            s1 = st_uni(cw_wave_sum(u1)); s2 = st_uni(cw_wave_sum(u2));
                        }
                        ST_PROF(3);
                        if (s1 > s2) {                                                                      /* :109-119 */
                            /* Align(seq1, seq2, min(len)) then the cigar's indel totals */
                            if (overlap > RMAX) { status = (uint32_t)CW_READ_CAPACITY; break; } /* (the last launch only: elsewhere a consensus is no longer than a slice may be) */
                            for (uint32_t x = lane; x < overlap; x += 64) { qfw[x] = (uint8_t)st_code(seq1[x]); refc[x] = (uint8_t)st_code(cur[x]); }
                            st_mem_sync();
                            const StAlign sub = st_align<NCHK, SYS>(qfw, (int)overlap, qrv, refc, (int)overlap, lane, sm, lds, state);
                            unsigned ins = 0, del = 0;
                            if (sub.score > 0) {
                                if (!st_banded_indels(refc + sub.ref_begin, sub.ref_end - sub.ref_begin + 1, qfw + sub.query_begin, sub.query_end - sub.query_begin + 1,
                                                      sub.score, rows, CW_ST_ROWS_BYTES, dirbuf, a.dir_bytes, &ins, &del, lane, (int8_t*)qrv, a.dir_bytes >= QMAX ? QMAX : 0)) { status = 2; break; }
                            }
                            ST_PROF(4);
                            const uint32_t cut = overlap - ins + del;
                            if (cut < cl) {                                                                 /* :114-115 curCons = seq1 + curCons.substr(cut) */
                                const uint32_t tail = cl - cut, nl = overlap + tail;
                                if (nl > QMAX) { status = too_big; break; }
                                /* build in qrv (free now), then copy back */
                                for (uint32_t x = lane; x < nl; x += 64) qrv[x] = x < overlap ? seq1[x] : cur[cut + (x - overlap)];
                                st_mem_sync();
                                for (uint32_t x = lane; x < nl; x += 64) cur[x] = qrv[x];
                                cl = nl;
                                st_mem_sync();
                            } else {
                                emptied = true;                                                             /* :117 */
                            }
                        }
                    }
                }
            }
            ST_PROF(3);
            if (!emptied && cl > 0) {                                                                       /* :124 */
                if (long_enough) {                                                                          /* :125-129 replace(beg, end-beg+1, upper(cur)) */
                    const uint32_t rl = end - beg + 1;
                    if (g.len() - rl + cl > g.cap) { status = 2; break; }
                    st_gap_to(g, beg + rl, lane);
                    g.L = beg;
                    for (uint32_t x = lane; x < cl; x += 64) g.buf[g.L + x] = st_upper(cur[x]);
                    g.L += cl;
                    st_mem_sync();
                }
                if (wi + 1 < jb.win_count) {                                                                /* :130-135 */
                    const long long np = (long long)cur_pos + (long long)st_uni(a.win_pos[2 * (w + 1)]) - (long long)st_uni(a.win_pos[2 * w]) - (long long)(end - beg + 1) + (long long)cl;
                    cur_pos = (int)(uint32_t)np;
                    for (uint32_t x = lane; x < cl; x += 64) old[x] = cur[x];
                    old_len = cl; old_w = w; have_old = true;
                    old_end = beg + cl - 1;
                    st_mem_sync();
                }
            }
#if CW_ST_PROF
            ST_PROF(5);
            if (a.trace && lane == 0) { uint32_t* t = a.trace + 8 * (size_t)w; for (int x = 0; x < 6; ++x) t[x] = (uint32_t)pc[x]; t[6] = clen; t[7] = (uint32_t)size_al; }
#endif
        }
        /* make the string contiguous */
        if (status == 0) st_gap_to(g, g.len(), lane);
        uint32_t flen = status == 0 ? g.L : 0u, fbeg = 0;
        if (a.do_trim && status == 0) { /* trimRead(.,1): first and last upper-case character; dropRead: < 10 % upper case */
            uint32_t first = 0xFFFFFFFFu, last = 0, ups = 0;
            for (uint32_t x = lane; x < flen; x += 64) if (st_is_upper(g.buf[x])) { first = min(first, x); last = max(last, x); ups++; }
            for (int o = 32; o > 0; o >>= 1) { first = min(first, (uint32_t)__shfl_xor((int)first, o)); last = max(last, (uint32_t)__shfl_xor((int)last, o)); }
            ups = st_uni((uint32_t)cw_wave_sum((int)ups)); first = st_uni(first); last = st_uni(last);
            if (first == 0xFFFFFFFFu || !(last > first)) { flen = 0; }            /* utils.cpp:123-127: end > beg else "" */
            else {
                fbeg = first; flen = last - first + 1;
                if ((float)ups / (float)flen < 0.1) { flen = 0; status = 1; }     /* dropRead on the trimmed read (CONSENT-correction.cpp:52) */
            }
            if (flen && fbeg) {
                for (uint32_t base = 0; base < flen; base += 64) {
                    const uint32_t x = base + lane;
                    uint8_t v = 0;
                    if (x < flen) v = g.buf[fbeg + x];
                    st_mem_sync();
                    if (x < flen) g.buf[x] = v;
                    st_mem_sync();
                }
            }
        }
        flen = st_uni(flen); status = st_uni(status);
        if constexpr (SYS) { if (st_uni(sm->fail)) { flen = 0; status = (uint32_t)CW_READ_CAPACITY; if (lane == 0) sm->fail = 0; } }
        if (lane == 0) {
            a.out_len[ri] = flen; a.read_status[ri] = (uint8_t)status;
            if (!HUGE && status == (uint32_t)CW_READ_REDO) atomicAdd(a.cursor + 3, 1u); /* the last launch looks at this count first */
        }
        st_mem_sync();
    }
    if constexpr (SYS) { /* the other waves wait for the next request: none */
        if (lane == 0) sm->quit = 1;
        __syncthreads();
    }
}

#endif
