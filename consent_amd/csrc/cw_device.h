/*
 * cw_device.h -- device-side data layout and small helpers shared by the kernels of the engine.
 *
 * Data layout in HBM for one batch (all arrays engine-owned scratch unless marked caller):
 *   caller  bases[]            2-bit, 16 bases / u32, MSB first; every sequence word-aligned
 *   caller  seq_len[], seq_word_off[], win_first_seq[]
 *   scratch WinInfo win[W]     per-window bookkeeping (status, offsets into the arrays below)
 *   scratch solid_key/cnt[]    per window: ascending solid k-mers + their exact pile-wide counts
 *   scratch seg_off/seg_len[]  per window: one slot per segment of the anchor chain (in chain order)
 *   scratch arena[]            per window: segment consensus characters ('A','C','G','T')
 *   scratch tasks[], members[] POA work items emitted by the index kernel (bump-allocated, batch-wide)
 */
#ifndef CW_DEVICE_H
#define CW_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/consent_amd.h"
#include "../../include/cw_policy.h"

#define CW_NONE16 0xFFFFu
#define CW_NEG (-(1 << 28))

struct WinInfo {
    uint32_t status;     /* CW_WIN_* */
    uint32_t n_seqs;     /* pile size N */
    uint32_t tpl_len;    /* length of the template (sequence 0) */
    uint32_t n_kmers;    /* k-mers in the whole pile */
    uint32_t solid_base; /* first slot in solid_key/solid_cnt */
    uint32_t solid_cap;
    uint32_t n_solid;
    uint32_t seg_base;   /* first slot in seg_off/seg_len */
    uint32_t seg_cap;
    uint32_t n_segs;     /* chain nodes + 1, 0 when there is no consensus */
    uint32_t arena_base; /* first byte in arena */
    uint32_t arena_cap;
    uint32_t arena_used;
    uint32_t ab_base;    /* anchor block (index kernel -> chain kernel), in 16-byte units into DevScratch::ablock */
    uint32_t ab_cap;     /* 16-byte units */
    uint32_t pad_;       /* why the window overflowed (CW_WHY_*), 0 otherwise: an inspection aid (cw_debug_win_info word 15) */
};

#define CW_WHY_SETUP 1      /* scratch slices of the batch exhausted (cw_setup_kernel)               */
#define CW_WHY_COUNT 2      /* more saturated / distinct k-mers than the exact count tables hold     */
#define CW_WHY_SOLIDCAP 3   /* solid set larger than the window's slice                              */
#define CW_WHY_TEMPLATE 4   /* template longer than CW_TMAX k-mers                                   */
#define CW_WHY_MATRIX 5     /* position matrix larger than the fallback slot / the anchor block      */
#define CW_WHY_SEGMENTS 6   /* more chain segments than slots                                        */
#define CW_WHY_TASKS 7      /* task / member / list capacity of the batch                            */
#define CW_WHY_POA 8        /* a POA task outgrew every tier, or its output slot                     */
#define CW_WHY_FIN_LEN 9    /* consensus longer than the finish kernel's string buffers              */
#define CW_WHY_FIN_SOLID 10 /* more solid k-mers than the visited bitmap covers                      */
#define CW_WHY_FIN_POLISH 11/* the polish outgrew a buffer                                           */
#define CW_WHY_OUT_CONS 12  /* the caller's consensus slot is too small                              */
#define CW_WHY_OUT_SOLID 13 /* the caller's solid slot is too small                                  */
#define CW_WHY_ANCHORS 15   /* more anchors than the chain kernel's slab holds: an engine that was not configured for long templates (cw_configure) */
#define CW_WHY_ARENA 14     /* the window's slice of the segment arena (round 6: told apart from CW_WHY_TASKS -- a plan whose arena scale is
                               already clamped cannot cure it, and re-running the batch three times for nothing was ADVICE r05's finding)  */


struct PoaTask {
    uint32_t window;
    uint32_t seg_slot;   /* absolute index into seg_off/seg_len */
    uint32_t member_off; /* absolute index into members[] */
    uint32_t n_members;
    uint32_t max_len;    /* longest member */
    uint32_t out_off;    /* absolute byte offset in arena */
    uint32_t out_cap;
    uint32_t state;      /* 0 pending, 1 done, 2 needs the large-graph path, 3 failed (capacity) */
};

struct PoaMember {
    uint32_t seq;   /* absolute sequence index */
    uint16_t start; /* first base of the piece */
    uint16_t len;
};

#define CW_TIERS 6 /* POA memory tiers: 0 = S (LDS), 1 = M1, 2 = M2, 3 = L (graph in LDS, matrix in a slab), 4 = G (all global), 5 = H (two tasks per wave,
                      cw_poa_q.h); list 0 is tier Q's (four tasks per wave, cw_poa_q.h) */
#define CW_PROF_SLOTS 128

/* Batch-wide counters (one struct in scratch, zeroed before every run). */
struct BatchCounters {
    uint32_t n_tasks;
    uint32_t n_members;
    uint32_t next_task;    /* work-stealing cursor of the tier-S kernel */
    uint32_t next_window;  /* work-stealing cursor of the index kernel */
    uint32_t next_finish;  /* work-stealing cursor of the finish kernel */
    uint32_t any_overflow;
    uint32_t n_tier[CW_TIERS];    /* tasks routed to tier t (t >= 1) by the index kernel */
    uint32_t next_tier[CW_TIERS]; /* work-stealing cursors */
    uint32_t n_over[CW_TIERS];    /* tasks that outgrew a tier and were handed to tier t */
    uint32_t next_over[CW_TIERS];
    uint32_t done_wgs;            /* work-groups of the producing tiers (S, M1, M2) that have finished */
    uint32_t next_chain;          /* work-stealing cursor of the chain kernel */
    unsigned long long prof[CW_PROF_SLOTS]; /* cycle totals per phase (0-32; tier H: 64-68), longest single task per POA tier (36-40), see cw_debug_profile; 72 + 12 t ..: row / trip
                                               counts of tier t in a -DCW_DIAG build (cw_poa.h PoaMem::diag) */
    uint32_t n_fin_retry;         /* windows the finish kernel's first pass handed to its second (cw_finish.h) */
    uint32_t next_fin_retry;
};

struct DevBatch {
    uint32_t n_windows;
    const uint32_t* win_first_seq;
    const uint32_t* seq_len;
    const uint64_t* seq_word_off;
    const uint32_t* bases;
};

struct DevScratch {
    WinInfo* win;
    uint32_t* solid_key;
    uint32_t* solid_cnt;
    uint32_t* seg_off;
    uint32_t* seg_len;
    uint8_t* arena;
    PoaTask* tasks;
    uint32_t task_cap;
    PoaMember* members;
    uint32_t member_cap;
    BatchCounters* ctr;
    uint32_t* tier_list[CW_TIERS]; /* task indices routed to tier t by the index kernel */
    uint32_t* over_list[CW_TIERS]; /* task indices that outgrew tier t-1 */
    uint32_t list_cap;
    uint8_t* slab[CW_TIERS];       /* per-wave slabs of tier t (DP matrix; for tier G also the graph) */
    uint64_t slab_bytes[CW_TIERS];
    uint32_t slots[CW_TIERS];      /* slabs of tier t: at least as many as waves of that tier the hardware can hold at once */
    uint32_t* slot_busy[CW_TIERS]; /* one flag per slab: a wave claims a free one when it starts and gives it back when it ends */
    uint32_t persist_wgs[CW_TIERS];/* the LAST so many work-groups of a tier's grid loop until its list is empty; the earlier ones take one
                                      chunk of tasks and end, which returns their LDS to the dispatcher -- tier 0 = S */
    uint16_t* p_fallback;          /* index kernel: per-work-group slot for a position matrix that outgrows LDS */
    uint64_t p_fallback_elems;     /* u16 elements per slot */
    uint8_t* ablock;               /* per-window anchor blocks: candidates, presence bitsets, position matrix */
    uint64_t ablock_units;         /* capacity in 16-byte units */
    unsigned long long* ex_fallback; /* index kernel: per-work-group exact table of saturated keys for very deep piles (CW_EXG_SLOTS each) */
    uint4* task_dbg;               /* NULL unless CW_TASK_TRACE is set: per task (start, duration) in 1024-cycle units, tier|rc|pass, wave */
    uint32_t* fin_retry;           /* finish kernel: windows whose strings outgrew the first pass (n_windows entries) */
    uint8_t* fin_big;              /* finish kernel, second pass: 3 x CW_FIN_CB_BIG bytes per work-group (64 of them) */
    uint32_t* fin_vis;             /* finish kernel: per-wave visited bitmap for windows with more solid k-mers than the LDS bitmap covers */
    uint32_t fin_vis_words;        /* words per wave (4^9 / 32: a window cannot have more distinct k-mers counted) */
    unsigned long long* step_clock; /* [0] wall clock at which the last batch's finish kernel ended (inspection: idle time between batches) */
    uint8_t* q_slab;               /* tier Q (cw_poa_q.h): per resident task CW_POAQ_SLAB_BYTES of kept DP rows */
    uint8_t* h_slab;               /* tier H: per resident task CW_POAH_SLAB_BYTES of kept DP rows and code words */
    uint32_t use_q;                /* route small tasks to tier Q (cw_poa_q.h); 0 = tier S takes them (CW_NO_TIER_Q) */
    uint32_t s_route_cells;        /* tier S takes a task whose graph is expected to stay below this many nodes (its capacity is CW_POA_NC; a task that outgrows
                                      it is redone in tier L, late).  (Until round 3: a bound on the cells of its LDS matrix, hence the name.) */
    uint32_t m1_route_depth;       /* tier M1 is chosen with the depth-aware graph estimate too (deep piles of ~100-base members outgrow its 256 nodes) */
    uint32_t use_h;                /* 0 = no tier H; 1 = tier H takes what would go to tier M1 or further; 2 (default) = also what tier S would take (CW_TIER_H) */
    uint32_t use_lw;               /* round 6: tier L's tasks whose members are wide on average go to list 5 and run on four waves each (tier LW, cw_poa_w.h); off while tier H uses that list */
    uint32_t h_min_len;            /* shortest "longest member" tier H takes (CW_H_MIN_LEN, default CW_POAH_MIN_LEN) */
    uint32_t linger_wgs;           /* tier-L work-groups that stay on the live overflow queue */
    uint32_t producer_wgs;         /* work-groups launched for tiers S + M1 + M2 (tier L's live queue waits for them) */
};

/* ---- packed-sequence access ------------------------------------------------------------------- */
/* the letter of a 2-bit code, by arithmetic (round 6: `"ACGT"[code]` is a load from constant memory -- s_getpc + global_load_ubyte -- and sat in front of a store in
   every consensus loop and, on lane 0, in every step of the finish kernel's serial walks) */
#define CW_ACGT(code) ((char)((0x54474341u >> (8u * ((uint32_t)(code) & 3u))) & 0xFFu))
__device__ __forceinline__ uint32_t cw_base_at(const uint32_t* w, uint32_t j) { return (w[j >> 4] >> (30 - 2 * (j & 15))) & 3u; }

/* k-mer starting at base p (k <= 16), str2num order (first base most significant). */
__device__ __forceinline__ uint32_t cw_kmer_at(const uint32_t* w, uint32_t p, uint32_t k) {
    const uint32_t wi = p >> 4, sh = 2 * (p & 15);
    uint64_t x = (uint64_t)w[wi] << 32;
    if ((p & 15) + k > 16) x |= w[wi + 1];
    x <<= sh;
    return (uint32_t)(x >> (64 - 2 * k));
}

/* phase profiling: lane/thread 0 of a work unit adds elapsed s_memtime ticks into ctr->prof[slot] */
#define CW_PROF_T0() unsigned long long _pt = __builtin_readcyclecounter()
#define CW_PROF(ctr, slot, leader)                                                        \
    do {                                                                                  \
        unsigned long long _now = __builtin_readcyclecounter();                           \
        if (leader) atomicAdd(&(ctr)->prof[slot], _now - _pt);                            \
        _pt = _now;                                                                       \
    } while (0)

__device__ __forceinline__ int cw_lane() { return (int)(threadIdx.x & 63); }

/* Compiler + hardware ordering point for LDS traffic between the lanes of ONE wave. */
__device__ __forceinline__ void cw_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* DPP controls (gfx9 family): row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
 * With bound_ctrl = false a lane without a source keeps `old`, which is the identity of the reduction. */
#define CW_DPP(old, src, ctrl, row_mask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (row_mask), 0xF, false)

/* inclusive prefix max over the 64 lanes: 4 row shifts + 2 row broadcasts, all VALU (no LDS crossbar).  The fill value of the
   shifts is the identity of the operation (INT_MIN / 0), which lets the compiler fold each shift into its max: one v_max_*_dpp per
   step instead of move + shift + max */
__device__ __forceinline__ int cw_wave_scan_max(int v) {
    const int I = (int)0x80000000;
    v = max(v, CW_DPP(I, v, 0x111, 0xF));
    v = max(v, CW_DPP(I, v, 0x112, 0xF));
    v = max(v, CW_DPP(I, v, 0x114, 0xF));
    v = max(v, CW_DPP(I, v, 0x118, 0xF));
    v = max(v, CW_DPP(I, v, 0x142, 0xA));
    v = max(v, CW_DPP(I, v, 0x143, 0xC));
    return v;
}
__device__ __forceinline__ unsigned cw_wave_scan_max_u32(unsigned v) {
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x111, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x112, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x114, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x118, 0xF));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x142, 0xA));
    v = max(v, (unsigned)CW_DPP(0, (int)v, 0x143, 0xC));
    return v;
}
/* inclusive prefix sum over the 64 lanes (same DPP ladder, identity 0) */
__device__ __forceinline__ int cw_wave_scan_add(int v) {
    v += CW_DPP(0, v, 0x111, 0xF);
    v += CW_DPP(0, v, 0x112, 0xF);
    v += CW_DPP(0, v, 0x114, 0xF);
    v += CW_DPP(0, v, 0x118, 0xF);
    v += CW_DPP(0, v, 0x142, 0xA);
    v += CW_DPP(0, v, 0x143, 0xC);
    return v;
}
/* wave-wide max / sum as a wave-uniform value: the DPP ladder of the scans, then the last lane's value through v_readlane -- all VALU / SALU,
   no ds_bpermute round trips through the LDS crossbar (round 5; until then six __shfl_xor steps each).  Callers are in wave-uniform control flow. */
__device__ __forceinline__ int cw_wave_max(int v) { return __builtin_amdgcn_readlane(cw_wave_scan_max(v), 63); }
__device__ __forceinline__ int cw_wave_sum(int v) { return __builtin_amdgcn_readlane(cw_wave_scan_add(v), 63); }
/* wave-wide max of a 64-bit key, returned as a wave-uniform value (two 32-bit DPP ladders with a lexicographic select) */
__device__ __forceinline__ unsigned long long cw_wave_max_u64(unsigned long long v) {
    unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
#define CW_MAX64_STEP(ctrl, mask)                                                        \
    do {                                                                                 \
        const unsigned ohi = (unsigned)CW_DPP(0, (int)hi, ctrl, mask);                   \
        const unsigned olo = (unsigned)CW_DPP(0, (int)lo, ctrl, mask);                   \
        const bool take = ohi > hi || (ohi == hi && olo > lo);                           \
        hi = take ? ohi : hi; lo = take ? olo : lo;                                      \
    } while (0)
    CW_MAX64_STEP(0x111, 0xF); CW_MAX64_STEP(0x112, 0xF); CW_MAX64_STEP(0x114, 0xF); CW_MAX64_STEP(0x118, 0xF);
    CW_MAX64_STEP(0x142, 0xA); CW_MAX64_STEP(0x143, 0xC);
#undef CW_MAX64_STEP
    const unsigned rh = (unsigned)__builtin_amdgcn_readlane((int)hi, 63), rl = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    return ((unsigned long long)rh << 32) | rl;
}
/* lane l receives v of lane l-1; lane 0 receives `fill` */
__device__ __forceinline__ int cw_wave_shr1(int v, int fill) { return CW_DPP(fill, v, 0x138, 0xF); }
/* value of a fixed lane as a wave-uniform scalar */
__device__ __forceinline__ int cw_lane_value(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ int cw_bcast(int v, int src) { return __shfl(v, src); }

#endif
