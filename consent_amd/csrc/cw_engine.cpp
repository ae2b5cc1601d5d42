/*
 * cw_engine.cpp -- host side of libconsent_amd.so: engine life cycle, scratch sizing, kernel launches,
 * host-buffer staging, per-stage timing.  C ABI declared in include/consent_amd.h.
 */
#include "cw_internal.h"
#include "cw_private.h"
#include "cw_device.h"
#include "cw_index.h"
#include "cw_chain.h"
#include "cw_poa.h"
#include "cw_poa_q.h"
#include "cw_finish.h"
#include "cw_extract.h"
#include "cw_stitch.h"
#include "cw_pack.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

/* LDS budget of the POA tiers (160 KiB per CU): the occupancy the tier table of DESIGN.md states depends on these sums */
static_assert(4 * CW_POA_WAVES * CW_POA_SLAB_BYTES <= 163840, "tier S: at least four work-groups per CU (five under the default policy)");
static_assert(4 * CW_POAM1_WAVES * CW_POA_HOT2C_BYTES(CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC) <= 163840, "tier M1: four work-groups per CU");
static_assert(4 * CW_POAM2_WAVES * CW_POA_HOT2T_BYTES(2, CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC) <= 163840 || CW_M2_CHAIN_TABS || CW_M2_CODES, "tier M2: four work-groups per CU");
static_assert(CW_POAL_LDS_BYTES <= 40960, "tier L: a work-group fits the hole an M1/M2 work-group leaves");
static_assert(!CW_POA_LW || CW_POALW_LDS_BYTES <= 61440, "tier LW (cw_poa_w.h): a four-wave work-group takes a third of a CU at most");
static_assert(CW_IDX_STAGE_OFF + 16 + CW_IDX_STAGE_N * 8 + 4 * CW_IDX_STAGE_WORDS <= CW_IDX_LDS_BYTES, "index kernel: stage area inside the LDS allocation");

static_assert(CW_FIN_CB_BIG == CW_CONS_SLOT_MAX, "the largest consensus slot (include/consent_amd.h CW_CONS_SLOT_BYTES, engine.py cons_slot_bytes) is what the finish kernel's second pass holds");

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct TierCfg { uint32_t slots; size_t slab_bytes; };

size_t big_slab_bytes() {
    return align_up((size_t)CW_POAB_HC * 4 + CW_POA_GRAPH_BYTES(CW_POAB_NC, CW_POAB_EC, CW_POAB_LC), 256);
}

/* Work-groups per CU of the four concurrent POA tier kernels (S, M1, M2, L).  The tiers share each CU's 160 KiB of LDS, every kernel
   is persistent, and which kernel's work-groups get onto a CU first is up to the hardware dispatcher: whatever is resident keeps its
   LDS until its list is empty.  Measured with the task timeline (tools/task_trace.py): with the shallow mix on depth-150 piles tier S
   (3 x 45 KB) held every CU for the first 27 ms and tier L, the long pole, started late (POA stage 79 ms); with fewer S and M2
   work-groups and one more of tier L it takes 69 ms; the shallow piles of config 2 prefer the old mix (19.6 vs 21.5 ms).  The choice
   goes by the mean pile depth of the batch.  Slabs are sized for the larger of the two, so the scratch layout does not depend on it.
   CW_WGS_S / _M1 / _M2 / _L override (experiments). */
struct TierMix { uint32_t s, m1, m2, l; };
TierMix tier_mix(bool deep) {
    TierMix m = deep ? TierMix{4, 5, 4, 2u} : TierMix{4, 5, 4, 2u}; /* see DESIGN.md: no static mix was robustly better than this one (round 4: tier S's work-groups take 24 KB, not 45) */
    auto knob = [](const char* name, uint32_t dflt) { const char* v = CW_AID_ENV(name); if (!v) return dflt; const int x = atoi(v); return x >= 1 && x <= 12 ? (uint32_t)x : dflt; };
    m.s = knob("CW_WGS_S", m.s); m.m1 = knob("CW_WGS_M1", m.m1); m.m2 = knob("CW_WGS_M2", m.m2); m.l = knob("CW_WGS_L", m.l);
    if (m.s > 6) m.s = 6;
    if (m.m1 > 12) m.m1 = 12;
    if (m.m2 > 12) m.m2 = 12;
    if (m.l > 8) m.l = 8;
    return m;
}

/* Work-groups of tier t (0 = S, 1 = M1, 2 = M2, 3 = L) worth launching for a batch of n_windows: the full persistent grid for a full batch,
   fewer for a small one (a 16384-window batch at depth 150 has ~5 tier-S, ~4 tier-M1, ~0.7 tier-M2 and ~0.25 tier-L tasks per window) --
   and with the grids the slabs: an engine used to hold ~10 GB of them whatever the batch, a 24-window batch now holds ~0.3 GB. */
uint32_t tier_wgs_cap(int t, uint32_t n_windows, uint32_t full) {
    static const uint32_t div[4] = {2, 4, 8, 8};
    const uint64_t want = (uint64_t)n_windows / div[t] + 4u;
    return want < full ? (uint32_t)want : full;
}

/* work-groups of tier LW (round 6: four waves each, one wide tier-L task at a time): its tasks are few -- a handful per thousand windows of a read
   set's piles, fewer in the synthetic ones -- and long: one work-group per CU at most */
uint32_t lw_wgs_cap(uint32_t n_windows, uint32_t cus) {
    if (!CW_POA_LW) return 0u;
    const uint32_t want = n_windows / 256u + 8u, most = cus / 4u ? cus / 4u : 1u; /* (measured: eight such tasks in a 16 384-window batch of the depth-150 bench piles) */
    return want < most ? want : most;
}

void tier_config(int cus, uint32_t big_slots, uint32_t n_windows, TierCfg (&t)[CW_TIERS]) {
    /* slabs: one per wave the hardware can hold at once plus a margin; waves claim them (slot_busy) */
    const TierMix mix = tier_mix(false);
    const uint32_t pass1 = 64u < (uint32_t)cus * 2u ? 64u : (uint32_t)cus * 2u;
    uint32_t l_wgs = tier_wgs_cap(3, n_windows, (uint32_t)cus * mix.l); /* (3.5 MB a slab: a generous count here was 12 GB, and allocating it stalled an engine's first job by 1-2 s) */
    if (l_wgs < pass1) l_wgs = pass1; /* the overflow pass launches its own tier-L work-groups */
    t[0] = {(tier_wgs_cap(0, n_windows, (uint32_t)cus * mix.s) * 2u + 8u) * CW_POA_WAVES, CW_POA_SLAB2_TOTAL(CW_POA_NC, CW_POA_EC, CW_POA_LC)}; /* tier S (round 4: cold arrays, flagged rows and code words in a slab) */
    /* round 6: a margin of an eighth over the waves the grids hold (a half through round 5): the plan of a 10 240-window job was 16.6 GB, most of it slabs,
       and a fresh process with four workers on a device spent 0.6-1.6 s allocating before its first job ran (tools/plan_sizes.py) */
    t[1] = {(tier_wgs_cap(1, n_windows, (uint32_t)cus * mix.m1) * 9u / 8u + 8u) * CW_POAM1_WAVES, CW_POA_SLAB2_TOTAL(CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC)};
    t[2] = {(tier_wgs_cap(2, n_windows, (uint32_t)cus * mix.m2) * 9u / 8u + 8u) * CW_POAM2_WAVES, CW_POA_SLAB2_TOTAL(CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC)};
    t[3] = {(l_wgs * 9u / 8u + 8u) * CW_POAL_WAVES + lw_wgs_cap(n_windows, (uint32_t)cus), CW_POA_SLAB2_TOTAL(CW_POAL_NC, CW_POAL_EC, CW_POAL_LC)}; /* (+ tier LW's work-groups: they claim tier L's slabs) */
    const uint64_t bs = (uint64_t)n_windows / 16u + 8u;
    t[4] = {bs < big_slots ? (uint32_t)bs / 4u * 4u : big_slots, big_slab_bytes()};
    t[5] = {8u, 64u}; /* (tier H's slabs go by resident task, not by claim: ScratchPlan::hslab) */
}

struct ScratchPlan {
    size_t win, solid_key, solid_cnt, seg_off, seg_len, arena, tasks, members, ctr, list[CW_TIERS], over[CW_TIERS], slab[CW_TIERS], qslab, hslab, finretry, finbig, pfall, ablock, finvis, exg, tdbg, sbusy[CW_TIERS], total;
    uint64_t pfall_elems, ablock_units;
    uint64_t solid_cap, seg_cap, arena_cap;
    uint32_t task_cap, member_cap, arena_scale;
    TierCfg tier[CW_TIERS];
};

/* tmax: template k-mers per window the plan provides for -- 1024 (templates of up to 1032 bases at k = 9: every window of the wrappers' defaults) unless the
   engine was configured for longer ones (cw_configure, up to CW_TMAX): the index kernel takes any template of up to CW_TMAX k-mers, what grows is the plan */
ScratchPlan plan_scratch(const cw_params& prm, uint32_t n_windows, uint32_t n_seqs, uint64_t n_words, int cus, uint32_t big_slots, uint32_t scale = 1, uint32_t tmax = 1024) {
    ScratchPlan p;
    memset(&p, 0, sizeof(p));
    p.solid_cap = (16ull * n_words) / prm.solid + n_windows + 16;
    p.seg_cap = (uint64_t)n_windows * (tmax + 2);
    /* scale: 1, or 4 / 16 / 64 after a run whose windows stopped on the task / member / arena capacities -- heuristics of the batch, which a batch of
       few, heavy windows (900-base windows at depth 100) outgrows: cw_run_device_sync runs such a batch again with the larger plan */
    /* the arena is addressed with 32-bit offsets (WinInfo, PoaTask): its scale is the largest one <= scale that keeps the batch's arena inside them
       (x4 of a 52 000-window batch did not, and the re-run ended in CW_E_INVALID); tasks and members scale on their own */
    const uint64_t arena1 = (uint64_t)n_windows * (16ull * (tmax + 16) + 4096);
    uint64_t as = scale;
    while (as > 1 && arena1 * as > 0xFFFFFFFFull) as /= 2;
    p.arena_scale = (uint32_t)as;
    p.arena_cap = arena1 * as;
    uint64_t tc = (64ull * n_windows + 1024) * scale, mc = (2048ull * n_windows + 4096) * scale;
    if (const char* v = CW_AID_ENV("CW_PLAN_DIV")) { const long x = atol(v); if (x >= 2) { tc = tc / (uint64_t)x + 1; mc = mc / (uint64_t)x + 1; } } /* test aid: a first plan that is too small, so that the growth of cw_run / cw_run_device_sync is exercised */
    p.task_cap = (uint32_t)(tc > 0x7FFFFFFFull ? 0x7FFFFFFFull : tc);
    p.member_cap = (uint32_t)(mc > 0x7FFFFFFFull ? 0x7FFFFFFFull : mc);
    /* test aid: shrink the two heuristic capacities so that the overflow path of the chain kernel's task emission can be exercised */
    if (const char* v = CW_AID_ENV("CW_TASK_CAP")) { const long x = atol(v); if (x >= 1 && (uint64_t)x < p.task_cap) p.task_cap = (uint32_t)x; }
    if (const char* v = CW_AID_ENV("CW_MEMBER_CAP")) { const long x = atol(v); if (x >= 1 && (uint64_t)x < p.member_cap) p.member_cap = (uint32_t)x; }
    tier_config(cus, big_slots, n_windows, p.tier);
    size_t o = 0;
    auto put = [&](size_t& slot, size_t bytes) { slot = o; o = align_up(o + bytes, 256); };
    put(p.win, (size_t)n_windows * sizeof(WinInfo));
    put(p.solid_key, p.solid_cap * 4);
    put(p.solid_cnt, p.solid_cap * 4);
    put(p.seg_off, p.seg_cap * 4);
    put(p.seg_len, p.seg_cap * 4);
    put(p.arena, p.arena_cap);
    put(p.tasks, ((size_t)p.task_cap + 1) * sizeof(PoaTask)); /* + the batch's neutral task at index task_cap (cw_setup_kernel; cw_chain.h "cap_ok") */
    put(p.members, (size_t)p.member_cap * sizeof(PoaMember));
    put(p.ctr, sizeof(BatchCounters));
    for (int t = 0; t < CW_TIERS; ++t) put(p.list[t], (size_t)p.task_cap * 4); /* list 0 = tier Q */
    for (int t = 0; t < CW_TIERS; ++t) put(p.over[t], (size_t)p.task_cap * 4);
    for (int t = 0; t < CW_TIERS; ++t) put(p.slab[t], (size_t)p.tier[t].slots * p.tier[t].slab_bytes);
    put(p.qslab, (size_t)cus * CW_POAQ_WAVES * 4 * CW_POAQ_SLAB_BYTES); /* tier Q: kept rows of every task a CU can hold */
    put(p.hslab, (size_t)cus * CW_POAH_WAVES * 2 * CW_POAH_SLAB_BYTES); /* tier H: kept rows and code words */
    /* anchor blocks (cw_ab_bytes): header + keys + presence bitsets + dirty list + the position matrix, 2 bytes per (template k-mer,
       sequence); at most CW_TMAX template k-mers per window, so the bound is per sequence, whatever the pieces' lengths are
       (a pile of many pieces only k bases long has few packed words but a full-width matrix) */
    p.ablock_units = ((uint64_t)n_seqs * (2ull * tmax + 2 + tmax / 8) + (uint64_t)n_windows * (tmax * (4ull + 8 + 8 + 8) + 256)) / 16 + 64;
    put(p.ablock, (size_t)p.ablock_units * 16);
    /* the position matrix of a window whose matrix does not fit LDS: up to tmax anchors x the sequences of a pile -- sixteen times the batch's mean depth, at
       least 1024 and at most ~4096 of them (round 6; 4100 whatever the batch through round 5: 2.1 GB of a driver job's plan at depth 30).  A pile deeper
       than that with more anchors than LDS holds is a reported capacity (CW_WHY_MATRIX), as it always was beyond 4100 */
    const uint64_t mean_n = n_windows ? (uint64_t)n_seqs / n_windows + 1 : 1;
    const uint64_t pf_n = mean_n * 16 < 1024 ? 1024 : mean_n * 16 > 4100 ? 4100 : mean_n * 16;
    p.pfall_elems = (uint64_t)tmax * pf_n;
    const size_t idx_wgs = n_windows < (uint32_t)cus ? n_windows : (size_t)cus; /* work-groups of the index kernel: one fallback slot each */
    put(p.pfall, idx_wgs * p.pfall_elems * 2);
    for (int t = 0; t < CW_TIERS; ++t) put(p.sbusy[t], (size_t)p.tier[t].slots * 4);
    put(p.exg, idx_wgs * CW_EXG_SLOTS * 8);
    put(p.tdbg, getenv("CW_TASK_TRACE") ? (size_t)p.task_cap * 16 : 0);
    /* one visited bitmap per wave of the first finish pass -- and of the second (one wave per work-group, up to 64 of them): the larger of the two */
    put(p.finvis, std::max<size_t>((size_t)cus * CW_FIN_WGS_PER_CU * CW_FIN_WAVES, 64) * CW_FIN_VIS_GLB_WORDS * 4);
    put(p.finretry, (size_t)n_windows * 4);
    put(p.finbig, (size_t)64 * 3 * CW_FIN_CB_BIG);
    p.total = o;
    return p;
}

int ensure(void** ptr, size_t* have, size_t need) {
    if (*have >= need) return CW_OK;
    if (*ptr) { if (hipFree(*ptr) != hipSuccess) return CW_E_NO_DEVICE; *ptr = nullptr; *have = 0; }
    /* a quarter more than asked for (bounded): hipFree + hipMalloc of a multi-gigabyte arena wait for the device and take up to seconds with a second
       engine at work on it, and the batches of a run differ by a few percent (the native driver's jobs: one regrowth instead of one per
       larger job -- the difference between 1.7 and 3-4 s for the E. coli-scale set) */
    size_t want = need + (need / 4 < ((size_t)1 << 30) ? need / 4 : ((size_t)1 << 30)); /* at most 1 GiB on top: several engines share a device */
    if (hipMalloc(ptr, want) != hipSuccess) {
        (void)hipGetLastError();
        want = need;
        if (hipMalloc(ptr, want) != hipSuccess) { *ptr = nullptr; return CW_E_NOMEM; }
    }
    *have = want;
    return CW_OK;
}

int check_params(const cw_params* p) {
    if (!p) return CW_E_INVALID;
    if (p->k < 2 || p->k > 16) return CW_E_INVALID; /* k-mers are 32-bit keys; k <= 9 counts in a direct LDS table, larger k in a hashed one */
    if (p->solid < 1 || p->max_msa < 1) return CW_E_INVALID;
    return CW_OK;
}

/* LDS request of the wide re-assembly kernel: what it needs, or -- to keep a second work-group off the CU (cw_stitch_device) -- a little more than
   half of what the DEVICE's CUs have (the constant 84 KB encoded gfx950's 160 KB), never more than a work-group may ask for */
size_t stitch_lds_one(const hipDeviceProp_t& prop) {
    size_t need = (size_t)CW_ST_WAVES * CW_ST_SLAB, half = prop.sharedMemPerMultiprocessor / 2 + 4096, cap = prop.sharedMemPerBlock;
    if (prop.sharedMemPerMultiprocessor == 0) half = 84u * 1024u;
    size_t want = half > need ? half : need;
    if (cap >= need && want > cap) want = cap;
    return want;
}

int set_kernel_attributes(const hipDeviceProp_t& prop) {
    const int lds_st = (int)stitch_lds_one(prop);
    if (hipFuncSetAttribute((const void*)cw_index_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CW_IDX_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_chain_kernel<CW_CH_SLAB>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_CH_WAVES * CW_CH_SLAB) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_chain_kernel<CW_CH_SLAB_LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_CH_WAVES * CW_CH_SLAB_LONG) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_sort_tier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CW_SORT_LDS_CLS) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CW_POA_SLAB_BYTES * CW_POA_WAVES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CW_POAQ_TASK_BYTES * 4 * CW_POAQ_WAVES) != hipSuccess ||
#if CW_Q_CODES && defined(CW_TEST_AIDS) /* tier H: an opt-in kernel of the test-aid build (CW_TIER_H), not in the product's code object */
        hipFuncSetAttribute((const void*)cw_poa_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CW_POAH_TASK_BYTES * 2 * 6) != hipSuccess ||
#endif
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC, CW_POAM1_WAVES, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POA_HOT2C_BYTES(CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC) * CW_POAM1_WAVES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC, CW_POAM2_WAVES, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POA_HOT2T_BYTES(2, CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC) * CW_POAM2_WAVES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAL_NC, CW_POAL_EC, CW_POAL_LC, CW_POAL_WAVES, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POAL_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC, CW_POAM1_WAVES, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POA_HOT2C_BYTES(CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC) * CW_POAM1_WAVES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC, CW_POAM2_WAVES, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POA_HOT2T_BYTES(2, CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC) * CW_POAM2_WAVES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAL_NC, CW_POAL_EC, CW_POAL_LC, CW_POAL_WAVES, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POAL_LDS_BYTES) != hipSuccess ||
#if CW_POA_LW
        hipFuncSetAttribute((const void*)cw_poa_slab_kernel<CW_POAL_NC, CW_POAL_EC, CW_POAL_LC, CW_POAL_WAVES, 3, 0, CW_POAL_MW, 5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_POALW_LDS_BYTES) != hipSuccess ||
#endif
        hipFuncSetAttribute((const void*)cw_finish_kernel<CW_FIN_CB, CW_FIN_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_FIN_SLAB * CW_FIN_WAVES) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_finish_kernel<CW_FIN_CB_BIG, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_FIN_SLAB_OF(0)) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_stitch_kernel<CW_ST_QMAX, CW_ST_RMAX, 16, CW_ST_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_st) != hipSuccess ||
#ifdef CW_TEST_AIDS
        hipFuncSetAttribute((const void*)cw_stitch_kernel<CW_ST_QMAX, CW_ST_RMAX, 16, CW_ST_WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_st) != hipSuccess ||
        hipFuncSetAttribute((const void*)cw_stitch_kernel<CW_STN_QMAX, CW_STN_RMAX, 5, CW_STN_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            CW_STN_WAVES * CW_ST_SLAB_OF(CW_STN_QMAX, CW_STN_RMAX)) != hipSuccess ||
#endif
        false)
        return CW_E_NO_DEVICE;
    return CW_OK;
}

int stage_begin(cw_engine* e, hipStream_t st, const char* name) {
    const int i = e->n_stages < CW_MAX_STAGES ? e->n_stages++ : CW_MAX_STAGES - 1;
    e->stage_name[i] = name;
    (void)hipEventRecord(e->ev0[i], st);
    return i;
}
void stage_end(cw_engine* e, hipStream_t st, int i) { (void)hipEventRecord(e->ev1[i], st); }

} // namespace

extern "C" {

const char* cw_version(void) { return "consent_amd 0.1 (gfx950, HIP)"; }

const char* cw_strerror(int s) {
    switch (s) {
    case CW_OK: return "ok";
    case CW_E_INVALID: return "invalid argument or malformed batch";
    case CW_E_NO_DEVICE: return "no HIP device or HIP runtime error";
    case CW_E_NOMEM: return "out of memory";
    case CW_E_CAPACITY: return "at least one window exceeded a capacity (see win_status)";
    case CW_E_INTERNAL: return "internal error";
    default: return "unknown status";
    }
}

int cw_create(const cw_params* params, int device, cw_engine** out) {
    if (!out) return CW_E_INVALID;
    *out = nullptr;
    int rc = check_params(params);
    if (rc) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return CW_E_NO_DEVICE;
    cw_engine* e = new (std::nothrow) cw_engine();
    if (!e) return CW_E_NOMEM;
    e->prm = *params;
    e->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&e->prop, device) != hipSuccess ||
        hipStreamCreate(&e->stream) != hipSuccess) {
        delete e;
        return CW_E_NO_DEVICE;
    }
    bool ok = hipEventCreate(&e->ev_fork) == hipSuccess && hipEventCreate(&e->ev_begin) == hipSuccess && hipEventCreate(&e->ev_end) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_join_s, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&e->host_fb, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) memset(e->host_fb, 0, 64);
    for (int i = 0; i < 4 && ok; ++i) ok = hipStreamCreateWithFlags(&e->side[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < CW_MAX_STAGES && ok; ++i) ok = hipEventCreate(&e->ev0[i]) == hipSuccess && hipEventCreate(&e->ev1[i]) == hipSuccess;
    for (int i = 0; i < CW_SLOTS && ok; ++i)
        ok = hipEventCreateWithFlags(&e->slot[i].ev_in, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e->slot[i].ev_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { cw_destroy(e); return CW_E_NO_DEVICE; }
    /* kernel attributes are per device: set them here, once per engine (not behind a process-wide flag) */
    if (set_kernel_attributes(e->prop) != CW_OK) { cw_destroy(e); return CW_E_NO_DEVICE; }
    *out = e;
    return CW_OK;
}

void cw_destroy(cw_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    if (e->scratch) (void)hipFree(e->scratch);
    for (int i = 0; i < CW_SLOTS; ++i) {
        cw_slot& s = e->slot[i];
        if (s.dev_in) (void)hipFree(s.dev_in);
        if (s.dev_out) (void)hipFree(s.dev_out);
        if (s.pin_out) (void)hipHostFree(s.pin_out);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    }
    if (e->xscratch) (void)hipFree(e->xscratch);
    if (e->step_clock) (void)hipFree(e->step_clock);
    if (e->stitch_scratch) (void)hipFree(e->stitch_scratch);
    if (e->host_fb) (void)hipHostFree(e->host_fb);
    for (int i = 0; i < CW_MAX_STAGES; ++i) { if (e->ev0[i]) (void)hipEventDestroy(e->ev0[i]); if (e->ev1[i]) (void)hipEventDestroy(e->ev1[i]); }
    for (int i = 0; i < 4; ++i) { if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]); if (e->side[i]) (void)hipStreamDestroy(e->side[i]); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join_s) (void)hipEventDestroy(e->ev_join_s);
    if (e->ev_begin) (void)hipEventDestroy(e->ev_begin);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    if (e->copy_in) (void)hipStreamDestroy(e->copy_in);
    if (e->copy_out) (void)hipStreamDestroy(e->copy_out);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

} // extern "C"

namespace {

/* the body of cw_run_device; the caller holds e->mu */
int run_device_locked(cw_engine* e, const cw_batch* batch, const cw_result* res, void* hip_stream) {
    if (!e || !batch || !res || !res->cons || !res->cons_off || !res->cons_len || !res->win_status) return CW_E_INVALID;
    if ((res->solid != nullptr) != (res->solid_off != nullptr) || (res->solid != nullptr) != (res->solid_len != nullptr)) return CW_E_INVALID;
    if (batch->n_windows == 0) return CW_OK;
    if (!batch->win_first_seq || !batch->seq_len || !batch->seq_word_off || !batch->bases) return CW_E_INVALID;
    /* per-window offsets into the arena, the solid table and the segment table are 32-bit (WinInfo, PoaTask): a batch is
       limited so that none of them can wrap -- split larger inputs (the library's own drivers do) */
    if (batch->n_windows > CW_MAX_BATCH_WINDOWS) return CW_E_INVALID;
    CW_HIP(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;

    uint32_t big_slots = 64; /* tier G's kernel is sixteen work-groups of four waves at most (big_wgs below): 64 slabs of 16.8 MB are all it can use (256 through round 5: 4.3 GB) */
    if (const char* env = CW_AID_ENV("CW_BIG_SLOTS")) { int v = atoi(env); if (v >= 4 && v <= 4096) big_slots = (uint32_t)v / 4 * 4; }
    const int cus = e->prop.multiProcessorCount > 0 ? e->prop.multiProcessorCount : 256;
    const ScratchPlan p = plan_scratch(e->prm, batch->n_windows, batch->n_seqs, batch->n_words, cus, big_slots, e->cap_scale, e->tmax_plan);
    if (p.solid_cap > 0xFFFFFFFFull || p.seg_cap > 0xFFFFFFFFull || p.arena_cap > 0xFFFFFFFFull) return CW_E_INVALID; /* see CW_MAX_BATCH_WINDOWS */
    e->last_windows = batch->n_windows; e->last_seqs = batch->n_seqs; e->last_words = batch->n_words; e->last_big_slots = big_slots;
    e->last_ctr_off = p.ctr;
    int rc = ensure(&e->scratch, &e->scratch_bytes, p.total);
    if (rc) return rc;
    if (!e->step_clock) { CW_HIP(hipMalloc((void**)&e->step_clock, 16)); CW_HIP(hipMemset(e->step_clock, 0, 16)); }
    uint8_t* base = (uint8_t*)e->scratch;

    DevBatch db;
    db.n_windows = batch->n_windows;
    db.win_first_seq = batch->win_first_seq; db.seq_len = batch->seq_len; db.seq_word_off = batch->seq_word_off; db.bases = batch->bases;
    DevScratch sc;
    memset(&sc, 0, sizeof(sc));
    sc.win = (WinInfo*)(base + p.win);
    sc.solid_key = (uint32_t*)(base + p.solid_key); sc.solid_cnt = (uint32_t*)(base + p.solid_cnt);
    sc.seg_off = (uint32_t*)(base + p.seg_off); sc.seg_len = (uint32_t*)(base + p.seg_len);
    sc.arena = base + p.arena;
    sc.tasks = (PoaTask*)(base + p.tasks); sc.task_cap = p.task_cap;
    sc.members = (PoaMember*)(base + p.members); sc.member_cap = p.member_cap;
    sc.ctr = (BatchCounters*)(base + p.ctr);
    sc.list_cap = p.task_cap;
    /* how many tier-L work-groups stay on the live overflow queue: about half of what the last finished batch handed over (few when
       nothing overflows, since a lingering work-group holds LDS the other tiers could use).  The count arrives in pinned host memory
       by an asynchronous copy at the end of each run: nothing here waits for the device */
    if (e->timings_valid) { /* a run has been enqueued before: its count, or that of the one before it, is there */
        const uint32_t want = e->host_fb[0] / 2;
        e->linger_wgs = want < 16 ? 16 : want > 256 ? 256 : want;
    }
    sc.p_fallback = (uint16_t*)(base + p.pfall); sc.p_fallback_elems = p.pfall_elems;
    sc.ablock = base + p.ablock; sc.ablock_units = p.ablock_units;
    sc.ex_fallback = (unsigned long long*)(base + p.exg);
    sc.step_clock = e->step_clock;
    sc.task_dbg = getenv("CW_TASK_TRACE") ? (uint4*)(base + p.tdbg) : nullptr;
    e->last_tasks_off = p.tasks; e->last_tdbg_off = p.tdbg; e->last_task_cap = p.task_cap;
    if (sc.task_dbg) CW_HIP(hipMemsetAsync(sc.task_dbg, 0, (size_t)p.task_cap * 16, st));
    sc.fin_vis = (uint32_t*)(base + p.finvis); sc.fin_vis_words = CW_FIN_VIS_GLB_WORDS;
    sc.fin_retry = (uint32_t*)(base + p.finretry);
    sc.fin_big = base + p.finbig;
    sc.linger_wgs = e->linger_wgs ? e->linger_wgs : 64;
    if (const char* env = CW_AID_ENV("CW_LINGER_WGS")) { int v = atoi(env); if (v >= 0 && v <= 1024) sc.linger_wgs = (uint32_t)v; }
    /* grids of the four concurrent tier kernels: `yield` work-groups that run one chunk of tasks and end, then `persist` ones that loop
       until the list is empty (see cw_poa_slab_kernel).  The yielding part is sized for the lists of a full batch and shrinks with the
       batch, so that a small batch does not pay for thousands of empty work-groups. */
    const TierMix mix = tier_mix(batch->n_seqs / batch->n_windows > 64u);
    const uint32_t W_ = batch->n_windows;
    auto ymin = [](uint32_t a, uint32_t b_) { return a < b_ ? a : b_; };
    uint32_t yield_wgs[4] = {ymin(8192u, W_ / 2u + 1u), ymin(16384u, W_), ymin(4096u, W_ / 4u + 1u), ymin(4096u, W_ / 4u + 1u)};
    /* measured: with every tier on the machine all the time the stage is SLOWER (depth 150: 86-92 ms against 69-82; depth 30:
       25.3 against 19.6) -- the CUs are short of issue slots, not of resident waves -- so yielding is off unless CW_YIELD is set */
    if (!CW_AID_ENV("CW_YIELD")) yield_wgs[0] = yield_wgs[1] = yield_wgs[2] = yield_wgs[3] = 0;
    sc.persist_wgs[0] = tier_wgs_cap(0, W_, (uint32_t)cus * mix.s); sc.persist_wgs[1] = tier_wgs_cap(1, W_, (uint32_t)cus * mix.m1);
    sc.persist_wgs[2] = tier_wgs_cap(2, W_, (uint32_t)cus * mix.m2); sc.persist_wgs[3] = tier_wgs_cap(3, W_, (uint32_t)cus * mix.l);
    sc.persist_wgs[4] = 0;
    const uint32_t wgs_s = yield_wgs[0] + sc.persist_wgs[0], wgs_m1 = yield_wgs[1] + sc.persist_wgs[1], wgs_m2 = yield_wgs[2] + sc.persist_wgs[2],
                   wgs_l = yield_wgs[3] + sc.persist_wgs[3];
    /* Tier H (two tasks per wave on 32-lane halves, cw_poa_q.h; round 5: tier Q's code instantiated for members of up to 63 bases and graphs of up
       to 128 nodes -- byte-sized graph arrays, recorded decisions, code words in the task's slab; it replaces round 3's opt-in kernel).  Bit-identical
       (tests/test_gpu_tier_q.py, test_gpu_parity.py) and OFF: measured on one box at depth 150 it takes tier S's 41.7 k tasks in 27.4 ms where tier
       S needs 28.4 -- 64 G wave-cycles against 69 -- and with its 18 waves per CU the other tiers slow down (M2 30.8 -> 36.7, L 41 -> 44 ms): step
       61-63 ms against 55-58; depth 30: 23.6 against 22.7.  Four tasks per wave (tier Q) beat tier S's scalar-controlled row loop six to one in
       machine time; two per wave do not: a per-lane row loop is ~100 instructions, tier S's 45.  CW_TIER_H=1|2 (test-aid build) turns it on. */
    /* Routing margins against late hand-overs.  A task that outgrows tier S or M1 is redone in tier L, whose work-groups only turn to the
       hand-over queue once their own list is empty: measured on a depth-150 batch (tools/task_trace.py), the ~150 tasks of 24-45 members
       x 27-31 bases that the depth-aware estimate put just inside tier S's 2048 cells, and the ~45 tasks of 52-64 members x 90-120 bases
       that outgrew tier M1's 256 nodes, kept a handful of tier-L waves busy for 10-20 ms after every other tier had finished.  With tier S
       taking only what is expected to stay below 1700 cells and tier M1 chosen by the depth-aware estimate as well, one task per batch is
       handed over instead of ~195, and the step (two engines) goes from 73.1 to 68.8 ms on the same box.  CW_S_ROUTE_CELLS /
       CW_M1_ROUTE_DEPTH=0 give the old routing (experiments; results do not depend on the tier). */
    sc.s_route_cells = 112; /* round 4: tier S has no cell limit any more; it takes a task whose graph is expected to stay below this many NODES (its capacity: CW_POA_NC) */
    if (const char* env = CW_AID_ENV("CW_S_ROUTE_NODES")) { const int v = atoi(env); if (v >= 8 && v <= CW_POA_NC) sc.s_route_cells = (uint32_t)v; }
    sc.m1_route_depth = 1u;
    if (const char* env = CW_AID_ENV("CW_M1_ROUTE_DEPTH")) sc.m1_route_depth = (uint32_t)atoi(env);
    sc.use_h = 0u; sc.h_min_len = CW_POAH_MIN_LEN;
    if (const char* env = CW_AID_ENV("CW_TIER_H")) { const int v = atoi(env); if (v >= 0 && v <= 2 && CW_Q_CODES) sc.use_h = (uint32_t)v; }
    if (const char* env = CW_AID_ENV("CW_H_MIN_LEN")) { const int v = atoi(env); if (v >= 1 && v <= CW_POAH_LC) sc.h_min_len = (uint32_t)v; }
    /* tier LW (round 6): on unless tier H has list 5 (test-aid build) or CW_LW=0 (test aid: every tier-L task on one wave, as through round 5) */
    sc.use_lw = CW_POA_LW && !sc.use_h ? 1u : 0u;
    if (const char* env = CW_AID_ENV("CW_LW")) { if (atoi(env) == 0) sc.use_lw = 0u; }
    if (CW_AID_ENV("CW_PHASES")) sc.use_lw = 0u; /* (the staged-launch experiment below has no tier LW) */
    const uint32_t wgs_lw = sc.use_lw ? lw_wgs_cap(W_, (uint32_t)cus) : 0u;
    uint32_t wgs_h = 0;
    const uint32_t h_waves = 3; /* waves per tier-H work-group (two tasks per wave) */
    if (sc.use_h) {
        const uint32_t lds_cu = e->prop.sharedMemPerMultiprocessor ? (uint32_t)e->prop.sharedMemPerMultiprocessor : 160u * 1024u;
        const uint32_t fit = lds_cu / (((uint32_t)(CW_POAH_TASK_BYTES * 2 * h_waves) + 1023u) / 1024u * 1024u);
        uint32_t per_cu = fit < CW_POAH_WAVES / h_waves ? (fit ? fit : 1u) : CW_POAH_WAVES / h_waves;
        if (const char* env = CW_AID_ENV("CW_WGS_H")) { const int v = atoi(env); if (v >= 1 && (uint32_t)v <= per_cu) per_cu = (uint32_t)v; }
        wgs_h = (uint32_t)cus * per_cu;
    }
    sc.persist_wgs[5] = wgs_h;
    sc.producer_wgs = wgs_s + wgs_m1 + wgs_m2 + wgs_h;
    if (CW_AID_ENV("CW_DEBUG_DONE")) fprintf(stderr, "[debug] producer_wgs %u = S %u + M1 %u + M2 %u, L %u, linger %u\n", sc.producer_wgs, wgs_s, wgs_m1, wgs_m2, wgs_l, sc.linger_wgs);
    sc.tier_list[0] = (uint32_t*)(base + p.list[0]); sc.over_list[0] = (uint32_t*)(base + p.over[0]);
    sc.use_q = (CW_AID_ENV("CW_NO_TIER_Q") || CW_POA_AFFINE) ? 0u : 1u; /* (the affine gap model runs in the global-memory tier only: cw_poa_a.h) */
    sc.q_slab = base + p.qslab;
    sc.h_slab = base + p.hslab;
    for (int t = 0; t < CW_TIERS; ++t) {
        if (t) { sc.tier_list[t] = (uint32_t*)(base + p.list[t]); sc.over_list[t] = (uint32_t*)(base + p.over[t]); }
        sc.slab[t] = base + p.slab[t]; sc.slab_bytes[t] = p.tier[t].slab_bytes; sc.slots[t] = p.tier[t].slots;
        sc.slot_busy[t] = (uint32_t*)(base + p.sbusy[t]);
    }
    FinOut fo;
    fo.cons = res->cons; fo.cons_off = res->cons_off; fo.cons_len = res->cons_len; fo.win_status = res->win_status;
    fo.solid = res->solid; fo.solid_off = res->solid_off; fo.solid_len = res->solid_len;

    e->n_stages = 0;
    e->timings_valid = false;
#define M1_ARGS CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC, CW_POAM1_WAVES, 1
#define M2_ARGS CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC, CW_POAM2_WAVES, 2
#define L_ARGS CW_POAL_NC, CW_POAL_EC, CW_POAL_LC, CW_POAL_WAVES, 3
    const size_t lds_m1 = CW_POA_HOT2C_BYTES(CW_POAM1_NC, CW_POAM1_EC, CW_POAM1_LC) * CW_POAM1_WAVES;
    const size_t lds_m2 = CW_POA_HOT2T_BYTES(2, CW_POAM2_NC, CW_POAM2_EC, CW_POAM2_LC) * CW_POAM2_WAVES;
    const size_t lds_l = CW_POAL_LDS_BYTES;
    const uint32_t thr_l = 64 * CW_POAL_WAVES;
    int sid;
    CW_HIP(hipEventRecord(e->ev_begin, st));
    CW_HIP(hipMemsetAsync(sc.ctr, 0, sizeof(BatchCounters), st));
    CW_HIP(hipMemsetAsync(base + p.sbusy[0], 0, p.sbusy[CW_TIERS - 1] + (size_t)p.tier[CW_TIERS - 1].slots * 4 - p.sbusy[0], st)); /* every slab free */
    CW_HIP(hipMemsetAsync(sc.over_list[3], 0xFF, (size_t)p.task_cap * 4, st)); /* live queue: an entry is its own flag */
    sid = stage_begin(e, st, "setup");
    cw_setup_need_kernel<<<(batch->n_windows + 3) / 4, 256, 0, st>>>(db, sc, e->prm);
    cw_setup_kernel<<<1, 1024, 0, st>>>(db, sc, e->prm, p.solid_cap, p.seg_cap, p.arena_cap, p.arena_scale);
    stage_end(e, st, sid);
    sid = stage_begin(e, st, "index");
    {
        const uint32_t grid = batch->n_windows < (uint32_t)cus ? batch->n_windows : (uint32_t)cus;
        cw_index_kernel<<<grid, CW_IDX_THREADS, CW_IDX_LDS_BYTES, st>>>(db, sc, e->prm);
    }
    stage_end(e, st, sid);
    sid = stage_begin(e, st, "chain");
    {
        const uint32_t want = (batch->n_windows + CW_CH_WAVES - 1) / CW_CH_WAVES;
        const uint32_t cap = (uint32_t)cus * 2u; /* 2 work-groups x 4 waves x 20 KiB per CU */
        if (e->tmax_plan > 1024u) /* an engine configured for long templates (cw_configure): 32 KiB per wave, up to CW_TMAX anchors, one work-group per CU */
            cw_chain_kernel<CW_CH_SLAB_LONG><<<want < (uint32_t)cus ? want : (uint32_t)cus, 64 * CW_CH_WAVES, CW_CH_WAVES * CW_CH_SLAB_LONG, st>>>(db, sc, e->prm);
        else
            cw_chain_kernel<CW_CH_SLAB><<<want < cap ? want : cap, 64 * CW_CH_WAVES, CW_CH_WAVES * CW_CH_SLAB, st>>>(db, sc, e->prm);
    }
    stage_end(e, st, sid);
    auto knob_u = [](const char* name, uint32_t dflt, uint32_t lo, uint32_t hi) { const char* v = CW_AID_ENV(name); if (!v) return dflt; const long x = atol(v); return x >= (long)lo && x <= (long)hi ? (uint32_t)x : dflt; };
    /* Launch shapes chosen for two batches in flight on one GPU (two engines, or the driver's two workers): while the other batch's
       persistent tier kernels hold the LDS of every CU, a work-group that needs most of a CU waits for tens of milliseconds in the
       middle of this batch's chain (rocprofv3 timeline, depth 150: tier sort 0.45 -> 13-34 ms, tier Q 8.7 -> 55-94 ms with tier S
       queued behind it, the two overflow passes 0.05 -> 12-32 ms).  So: the sort keeps 16 KB of classes in LDS instead of 128 (the
       rest are recomputed), tier Q runs as two-wave work-groups (38 KB; the same 8 waves per CU when it has the machine to itself),
       and the overflow passes -- normally a handful of tasks -- are 64 / 16 work-groups instead of two per CU.  Alone on the GPU the
       step time is unchanged. */
    const uint32_t sort_lds = knob_u("CW_SORT_LDS", 16384, 0, CW_SORT_LDS_CLS);
    const uint32_t sort_thr = knob_u("CW_SORT_THREADS", 256, 128, 1024) / 64 * 64;
    const uint32_t q_waves = knob_u("CW_Q_WAVES", CW_POAQ_WAVES % 3 == 0 ? 3 : CW_POAQ_WAVES % 2 == 0 ? 2 : 1, 1, CW_POAQ_WAVES); /* waves per tier-Q work-group (four tasks per wave; round 5: three-wave work-groups of 40 KB, four per CU) */
    const uint32_t q_lds_cu = e->prop.sharedMemPerMultiprocessor ? (uint32_t)e->prop.sharedMemPerMultiprocessor : 160u * 1024u;
    const uint32_t q_fit = q_lds_cu / (((uint32_t)(CW_POAQ_TASK_BYTES * 4 * q_waves) + 1023u) / 1024u * 1024u);   /* work-groups whose LDS a CU holds (allocation granule: at most 1 KB) */
    const uint32_t q_most = q_fit < CW_POAQ_WAVES / q_waves ? (q_fit ? q_fit : 1u) : CW_POAQ_WAVES / q_waves;
    const uint32_t q_per_cu = knob_u("CW_Q_WGS_PER_CU", q_most, 1, q_most);
    const uint32_t q_grid = (uint32_t)cus * q_per_cu;
    const uint32_t pass1_wgs = knob_u("CW_PASS1_WGS", 64u < (uint32_t)cus * 2u ? 64u : (uint32_t)cus * 2u, 1, 64u < (uint32_t)cus * 2u ? 64u : (uint32_t)cus * 2u);
    const uint32_t big_cap = p.tier[4].slots / CW_POA_WAVES;
    const uint32_t big_wgs = knob_u("CW_BIG_WGS", big_cap < 16 ? big_cap : 16, 1, big_cap);
    cw_sort_tier_kernel<<<5, sort_thr, sort_lds, st>>>(sc, sort_lds); /* tiers M1, M2 and L: largest tasks first; tiers Q and H: like with like */
    /* pass 0: every tier works through its own routed list, all four concurrently; the long-running large tiers are
       dispatched first so that their tail overlaps the bulk of the small tasks */
    const char* ph_env = CW_AID_ENV("CW_PHASES");
    const int phases = ph_env ? atoi(ph_env) : 0;
    const uint32_t grid_l = wgs_l;
    if (phases == 2) {
        /* experiment: tier Q, then S + M1 + M2 side by side, then tier L alone with every hand-over already on its list */
        sid = stage_begin(e, st, "poa_q");
        cw_poa_q_kernel<<<q_grid, 64 * q_waves, CW_POAQ_TASK_BYTES * 4 * q_waves, st>>>(db, sc);
        stage_end(e, st, sid);
#if CW_Q_CODES && defined(CW_TEST_AIDS)
        if (wgs_h) {
            sid = stage_begin(e, st, "poa_h");
            cw_poa_h_kernel<<<wgs_h, 64 * h_waves, CW_POAH_TASK_BYTES * 2 * h_waves, st>>>(db, sc);
            stage_end(e, st, sid);
        }
#endif
        CW_HIP(hipEventRecord(e->ev_fork, st));
        for (int i = 0; i < 2; ++i) CW_HIP(hipStreamWaitEvent(e->side[i], e->ev_fork, 0));
        sid = stage_begin(e, e->side[1], "poa_m2");
        cw_poa_slab_kernel<M2_ARGS, 0><<<wgs_m2, 64 * CW_POAM2_WAVES, lds_m2, e->side[1]>>>(db, sc);
        stage_end(e, e->side[1], sid);
        sid = stage_begin(e, e->side[0], "poa_m1");
        cw_poa_slab_kernel<M1_ARGS, 0><<<wgs_m1, 64 * CW_POAM1_WAVES, lds_m1, e->side[0]>>>(db, sc);
        stage_end(e, e->side[0], sid);
        sid = stage_begin(e, st, "poa");
        cw_poa_kernel<<<wgs_s, 64 * CW_POA_WAVES, CW_POA_SLAB_BYTES * CW_POA_WAVES, st>>>(db, sc);
        stage_end(e, st, sid);
        for (int i = 0; i < 2; ++i) { CW_HIP(hipEventRecord(e->ev_join[i], e->side[i])); CW_HIP(hipStreamWaitEvent(st, e->ev_join[i], 0)); }
        sid = stage_begin(e, st, "poa_large");
        cw_poa_slab_kernel<L_ARGS, 0><<<grid_l, thr_l, lds_l, st>>>(db, sc);
        stage_end(e, st, sid);
    } else {
    CW_HIP(hipEventRecord(e->ev_fork, st));
    for (int i = 0; i < (wgs_lw ? 4 : 3); ++i) CW_HIP(hipStreamWaitEvent(e->side[i], e->ev_fork, 0));
#if CW_POA_LW
    if (wgs_lw) { /* tier LW first: its tasks are the longest of the batch (four waves each; cw_poa_w.h) */
        sid = stage_begin(e, e->side[3], "poa_lw");
        cw_poa_slab_kernel<L_ARGS, 0, CW_POAL_MW, 5><<<wgs_lw, 64 * CW_POAL_MW, CW_POALW_LDS_BYTES, e->side[3]>>>(db, sc);
        stage_end(e, e->side[3], sid);
    }
#endif
    /* tier L also consumes the live overflow queue; only sc.linger_wgs of its work-groups stay for that (far fewer than
       CUs, so they can never keep the producers they wait for off the machine) */
    sid = stage_begin(e, e->side[2], "poa_large");
    cw_poa_slab_kernel<L_ARGS, 0><<<grid_l, thr_l, lds_l, e->side[2]>>>(db, sc);
    stage_end(e, e->side[2], sid);
    sid = stage_begin(e, e->side[1], "poa_m2");
    cw_poa_slab_kernel<M2_ARGS, 0><<<wgs_m2, 64 * CW_POAM2_WAVES, lds_m2, e->side[1]>>>(db, sc);
    stage_end(e, e->side[1], sid);
    sid = stage_begin(e, e->side[0], "poa_m1");
    cw_poa_slab_kernel<M1_ARGS, 0><<<wgs_m1, 64 * CW_POAM1_WAVES, lds_m1, e->side[0]>>>(db, sc);
    stage_end(e, e->side[0], sid);
    /* Tiers Q and S go on the ENGINE's stream, also when the caller brought its own: the runtime multiplexes streams onto a few hardware
       queues, and a caller's stream that shares a queue with tier L's stream would make tier S wait for tier L's kernel to end --
       whose last work-groups wait for tier S to sign off (measured in the native driver: 131 ms per job until the bounded wait ran out).
       The engine's four streams were created together and first. */
    hipStream_t ms = e->stream;
    if (ms != st) CW_HIP(hipStreamWaitEvent(ms, e->ev_fork, 0));
    sid = stage_begin(e, ms, "poa_q");
    cw_poa_q_kernel<<<q_grid, 64 * q_waves, CW_POAQ_TASK_BYTES * 4 * q_waves, ms>>>(db, sc); /* four tasks per wave */
    stage_end(e, ms, sid);
#if CW_Q_CODES && defined(CW_TEST_AIDS)
    if (wgs_h) { /* two tasks per wave; hands what outgrows it to tier L's live queue, as tier S does */
        sid = stage_begin(e, ms, "poa_h");
        cw_poa_h_kernel<<<wgs_h, 64 * h_waves, CW_POAH_TASK_BYTES * 2 * h_waves, ms>>>(db, sc);
        stage_end(e, ms, sid);
    }
#endif
    sid = stage_begin(e, ms, "poa");
    cw_poa_kernel<<<wgs_s, 64 * CW_POA_WAVES, CW_POA_SLAB_BYTES * CW_POA_WAVES, ms>>>(db, sc); /* 11.3 KiB per wave: three work-groups per CU */
    stage_end(e, ms, sid);
    if (ms != st) { CW_HIP(hipEventRecord(e->ev_join_s, ms)); CW_HIP(hipStreamWaitEvent(st, e->ev_join_s, 0)); }
    for (int i = 0; i < (wgs_lw ? 4 : 3); ++i) { CW_HIP(hipEventRecord(e->ev_join[i], e->side[i])); CW_HIP(hipStreamWaitEvent(st, e->ev_join[i], 0)); }
    }
    /* pass 1: tasks that outgrew their tier (normally a handful) go straight to tier L, and from there to G */
    sid = stage_begin(e, st, "poa_overflow");
    cw_poa_slab_kernel<L_ARGS, 1><<<pass1_wgs, thr_l, lds_l, st>>>(db, sc);
    cw_poa_big_kernel<<<big_wgs, 64 * CW_POA_WAVES, 0, st>>>(db, sc);
    stage_end(e, st, sid);
    sid = stage_begin(e, st, "finish");
    {
        uint32_t grid = (batch->n_windows + CW_FIN_WAVES - 1) / CW_FIN_WAVES;
        if (grid > (uint32_t)cus * CW_FIN_WGS_PER_CU) grid = (uint32_t)cus * CW_FIN_WGS_PER_CU;
        cw_finish_kernel<CW_FIN_CB, CW_FIN_WAVES, false><<<grid, 64 * CW_FIN_WAVES, CW_FIN_SLAB * CW_FIN_WAVES, st>>>(db, sc, e->prm, fo);
        /* second pass: the windows whose strings outgrew the first pass's buffers (normally none: the kernel reads one counter and ends) */
        const uint32_t grid2 = batch->n_windows < 64u ? batch->n_windows : 64u;
        cw_finish_kernel<CW_FIN_CB_BIG, 1, true><<<grid2, 64, CW_FIN_SLAB_OF(0), st>>>(db, sc, e->prm, fo);
    }
    stage_end(e, st, sid);
    /* feedback for the next batch's linger_wgs (pinned destination: asynchronous) */
    CW_HIP(hipMemcpyAsync(&e->host_fb[0], &sc.ctr->n_over[3], 4, hipMemcpyDeviceToHost, st));
    CW_HIP(hipEventRecord(e->ev_end, st));
#undef M1_ARGS
#undef M2_ARGS
#undef L_ARGS
    CW_HIP(hipGetLastError());
    e->timings_valid = true;
    return CW_OK;
}

} // namespace

extern "C" {

int cw_run_device(cw_engine* e, const cw_batch* batch, const cw_result* res, void* hip_stream) {
    if (!e) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    return run_device_locked(e, batch, res, hip_stream);
}

/* Templates longer than the default plan provides for (include/consent_amd.h): the scratch plan and the chain kernel's instance follow; takes effect with
   the engine's next run. */
int cw_configure(cw_engine* e, uint32_t max_template_len) {
    if (!e) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    const uint32_t k = e->prm.k ? e->prm.k : 1u;
    uint32_t kmers = max_template_len >= k ? max_template_len - k + 1u : 1u;
    if (kmers > (uint32_t)CW_TMAX) return CW_E_INVALID; /* beyond what the index kernel holds (CW_WHY_TEMPLATE) */
    if (kmers < 128u) kmers = 128u; /* (a caller that knows its windows are shorter than the default plan assumes -- the native driver: `-l 500` is 492
                                       k-mers -- gets the smaller plan: anchor blocks, segment slots and the arena go by this number) */
    e->tmax_plan = kmers;
    return CW_OK;
}

int cw_poll(cw_engine* e) {
    if (!e) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->timings_valid) return 1;
    if (hipSetDevice(e->device) != hipSuccess) return CW_E_NO_DEVICE;
    hipError_t q = hipEventQuery(e->ev_end);
    if (q == hipSuccess) return 1;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); return CW_E_NO_DEVICE; }
    /* still running: has it left the part that fills the GPU?  (tiers S, M1 and M2 done; what is left is tier L's long tasks, a
       handful of waves, and the finish kernel) */
    int bulk = 0, done = 0;
    for (int i = 0; i < e->n_stages; ++i) {
        const char* n = e->stage_name[i];
        if (strcmp(n, "poa") && strcmp(n, "poa_m1") && strcmp(n, "poa_m2") && strcmp(n, "poa_h")) continue;
        ++bulk;
        q = hipEventQuery(e->ev1[i]);
        if (q == hipSuccess) ++done;
        else if (q != hipErrorNotReady) { (void)hipGetLastError(); return CW_E_NO_DEVICE; }
    }
    return bulk && done == bulk ? 2 : 0;
}

int cw_last_timings(cw_engine* e, float* ms, const char** names, int cap, int* n_stages) {
    if (!e || !n_stages) return CW_E_INVALID;
    *n_stages = 0;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->timings_valid) return CW_OK;
    CW_HIP(hipSetDevice(e->device));
    CW_HIP(hipEventSynchronize(e->ev_end));
    int k = 0;
    for (int i = 0; i < e->n_stages && k < cap; ++i, ++k) {
        float t = 0.f;
        CW_HIP(hipEventElapsedTime(&t, e->ev0[i], e->ev1[i]));
        e->stage_ms[i] = t;
        if (ms) ms[k] = t;
        if (names) names[k] = e->stage_name[i];
    }
    if (k < cap) { /* wall time of the whole run on the launch stream (tiers overlap, so this is not the sum) */
        float t = 0.f;
        CW_HIP(hipEventElapsedTime(&t, e->ev_begin, e->ev_end));
        if (ms) ms[k] = t;
        if (names) names[k] = "total";
        ++k;
    }
    *n_stages = k;
    return CW_OK;
}

/* Debug/inspection (cw_private.h): what plan_scratch would allocate, component by component (host arithmetic only). */
int cw_debug_plan(uint32_t k, uint32_t solid, uint32_t n_windows, uint32_t n_seqs, uint64_t n_words, int cus, uint32_t scale, uint32_t tmax, uint64_t* out15) {
    if (!out15 || !n_windows || !solid) return CW_E_INVALID;
    cw_params prm{k, solid, 8, 2, 150};
    const ScratchPlan p = plan_scratch(prm, n_windows, n_seqs, n_words, cus, 64, scale ? scale : 1, tmax ? tmax : 1024);
    auto slab = [&](int t) { return (uint64_t)p.tier[t].slots * p.tier[t].slab_bytes; };
    const uint64_t v[15] = {p.total, (uint64_t)n_windows * sizeof(WinInfo), p.solid_cap * 8, p.seg_cap * 8, p.arena_cap,
                            ((uint64_t)p.task_cap + 1) * sizeof(PoaTask) + (uint64_t)p.member_cap * sizeof(PoaMember), (uint64_t)p.task_cap * 4 * 2 * CW_TIERS,
                            slab(0), slab(1), slab(2), slab(3), slab(4), (uint64_t)p.hslab - p.qslab + ((uint64_t)p.ablock - p.hslab), p.ablock_units * 16,
                            (uint64_t)(p.exg - p.pfall) + (uint64_t)(p.tdbg - p.exg)};
    for (int i = 0; i < 15; ++i) out15[i] = v[i];
    return CW_OK;
}

/* Debug/inspection (cw_private.h): copy the engine's per-window bookkeeping of the last run (16 u32 per window) to host. */
int cw_debug_win_info(cw_engine* e, uint32_t n_windows, uint32_t* out16) {
    if (!e || !out16 || !e->scratch) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (n_windows > e->last_windows) return CW_E_INVALID;
    CW_HIP(hipSetDevice(e->device));
    CW_HIP(hipDeviceSynchronize());
    CW_HIP(hipMemcpy(out16, e->scratch, (size_t)n_windows * sizeof(WinInfo), hipMemcpyDeviceToHost));
    return CW_OK;
}

/* Debug/inspection (cw_private.h): the batch counters (u32) and the per-phase cycle totals (u64) of the last run; the caller says how many
 * words each of its buffers holds and gets min(capacity, available) of them (returned through the two *_n when given). */
int cw_debug_profile(cw_engine* e, uint32_t* counters, uint32_t counters_cap, unsigned long long* prof, uint32_t prof_cap, uint32_t* counters_n, uint32_t* prof_n) {
    if (!e || !e->scratch || !counters || !prof) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CW_HIP(hipSetDevice(e->device));
    CW_HIP(hipDeviceSynchronize());
    BatchCounters c;
    CW_HIP(hipMemcpy(&c, (uint8_t*)e->scratch + e->last_ctr_off, sizeof(c), hipMemcpyDeviceToHost));
    const uint32_t nc = counters_cap < 6u + 4u * CW_TIERS ? counters_cap : 6u + 4u * CW_TIERS; /* n_tasks .. any_overflow, then n_tier, next_tier, n_over, next_over [CW_TIERS] each */
    const uint32_t np = prof_cap < (uint32_t)CW_PROF_SLOTS ? prof_cap : (uint32_t)CW_PROF_SLOTS;
    memcpy(counters, &c, (size_t)nc * 4);
    memcpy(prof, c.prof, (size_t)np * sizeof(unsigned long long));
    if (counters_n) *counters_n = nc;
    if (prof_n) *prof_n = np;
    return CW_OK;
}

/* Debug/inspection (cw_private.h): with CW_TASK_TRACE set, 12 words per POA task of the last run: the PoaTask record (window, seg_slot,
 * member_off, n_members, max_len, out_off, out_cap, state) + start and duration in 10 ns units, tier | rc << 8 | pass << 16, wave */
int cw_debug_task_trace(cw_engine* e, uint32_t cap_tasks, uint32_t* out12, uint32_t* n_tasks) {
    if (!e || !e->scratch || !out12 || !n_tasks || !getenv("CW_TASK_TRACE")) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CW_HIP(hipSetDevice(e->device));
    CW_HIP(hipDeviceSynchronize());
    BatchCounters c;
    CW_HIP(hipMemcpy(&c, (uint8_t*)e->scratch + e->last_ctr_off, sizeof(c), hipMemcpyDeviceToHost));
    uint32_t n = c.n_tasks < e->last_task_cap ? c.n_tasks : e->last_task_cap;
    if (n > cap_tasks) n = cap_tasks;
    std::vector<uint32_t> a((size_t)n * 8), d((size_t)n * 4);
    CW_HIP(hipMemcpy(a.data(), (uint8_t*)e->scratch + e->last_tasks_off, (size_t)n * 32, hipMemcpyDeviceToHost));
    CW_HIP(hipMemcpy(d.data(), (uint8_t*)e->scratch + e->last_tdbg_off, (size_t)n * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) { memcpy(out12 + (size_t)i * 12, &a[(size_t)i * 8], 32); memcpy(out12 + (size_t)i * 12 + 8, &d[(size_t)i * 4], 16); }
    *n_tasks = n;
    return CW_OK;
}

/* getCoverages + getAlignmentWindowsPositions (src/alignmentWindows.cpp:5-85) on the host: per-base overlap depth over
 * [q_start, q_end] inclusive; a window [beg, beg+window_size-1] whenever window_size consecutive bases have depth >=
 * min_support, then rewind by window_overlap (:40-47); a base below min_support resets (:48-51); finally ONE trailing window
 * = the last window_size covered bases found scanning backwards from the end, never looking at base 0 (:58-79). */
int cw_window_positions(uint32_t tpl_len, const cw_overlap* overlaps, uint32_t n_overlaps, uint32_t min_support, uint32_t window_size,
                        int32_t window_overlap, uint32_t* out_beg_end, uint32_t cap_pairs, uint32_t* n_pairs) {
    if (!n_pairs || (n_overlaps && !overlaps) || (cap_pairs && !out_beg_end)) return CW_E_INVALID;
    *n_pairs = 0;
    /* the reference loops for ever on windowOverlap >= windowSize (the rewind undoes the whole window) and indexes out of bounds on a
       negative one: an error here */
    if (window_size == 0 || window_overlap < 0 || (uint32_t)window_overlap >= window_size) return CW_E_INVALID;
    if (tpl_len == 0) return CW_OK;
    /* getCoverages (:5-25) adds one to every base of every overlap; the same depths from a difference array and one prefix sum:
       O(overlaps + length) instead of O(sum of the overlap lengths), which was most of the host time per pile */
    std::vector<uint32_t> cov;
    try { cov.assign((size_t)tpl_len + 1, 0); } catch (...) { return CW_E_NOMEM; }
    for (uint32_t o = 0; o < n_overlaps; ++o) {
        if (overlaps[o].q_end < overlaps[o].q_start || overlaps[o].q_end >= tpl_len) return CW_E_INVALID; /* the reference would write out of bounds */
        cov[overlaps[o].q_start]++;
        cov[(size_t)overlaps[o].q_end + 1]--; /* wraps; the prefix sum below brings it back */
    }
    for (uint32_t i = 1; i < tpl_len; ++i) cov[i] += cov[i - 1];
    uint32_t count = 0;
    bool over = false;
    auto push = [&](uint32_t b, uint32_t e2) {
        if (count < cap_pairs) { out_beg_end[2 * count] = b; out_beg_end[2 * count + 1] = e2; } else over = true;
        ++count;
    };
    uint32_t cur = 0, beg = 0, i = 0;
    while (i < tpl_len) {
        if (cur >= window_size) {
            push(beg, beg + cur - 1);
            if (window_overlap) i = i - (uint32_t)window_overlap;
            beg = i;
            cur = 0;
        }
        if (cov[i] < min_support) { cur = 0; i++; beg = i; }
        else { cur++; i++; }
    }
    bool pushed = false;
    uint32_t end = tpl_len - 1;
    cur = 0;
    i = tpl_len - 1;
    while (i > 0 && !pushed) {
        if (cur >= window_size) { push(end - cur + 1, end); pushed = true; end = i; cur = 0; }
        if (cov[i] < min_support) { cur = 0; i--; end = i; }
        else { cur++; i--; }
    }
    *n_pairs = count;
    return over ? CW_E_CAPACITY : CW_OK;
}

int cw_extract_piles_device(cw_engine* e, const cw_read_set* reads, const cw_overlap* overlaps, uint64_t n_overlaps,
                            const cw_window_job* jobs, uint32_t n_jobs, uint32_t k, uint32_t* win_first_seq, uint32_t* seq_len,
                            uint64_t* seq_word_off, uint32_t* bases, uint32_t seq_cap, uint64_t word_cap, uint32_t* n_seqs,
                            uint64_t* n_words, void* hip_stream) {
    return cw_extract_impl(e, reads, overlaps, n_overlaps, jobs, nullptr, n_jobs, k, win_first_seq, seq_len, seq_word_off, bases, seq_cap, word_cap, n_seqs,
                           n_words, hip_stream);
}

/* jobs_host: the same jobs in host memory when the caller has them (the library's own driver does): saves the device-to-host copy */
int cw_extract_impl(cw_engine* e, const cw_read_set* reads, const cw_overlap* overlaps, uint64_t n_overlaps, const cw_window_job* jobs,
                    const cw_window_job* jobs_host, uint32_t n_jobs, uint32_t k, uint32_t* win_first_seq, uint32_t* seq_len, uint64_t* seq_word_off,
                    uint32_t* bases, uint32_t seq_cap, uint64_t word_cap, uint32_t* n_seqs, uint64_t* n_words, void* hip_stream) {
    if (!e || !reads || !jobs || !n_seqs || !n_words || (n_overlaps && !overlaps) || k < 1) return CW_E_INVALID;
    *n_seqs = 0; *n_words = 0;
    if (n_jobs == 0) return CW_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    CW_HIP(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    /* per-window descriptor offsets need the jobs' overlap counts: take them from the device once */
    std::vector<cw_window_job> hj_own;
    std::vector<uint64_t> doff;
    try { if (!jobs_host) hj_own.resize(n_jobs); doff.assign((size_t)n_jobs + 1, 0); } catch (...) { return CW_E_NOMEM; }
    if (!jobs_host) {
        CW_HIP(hipMemcpyAsync(hj_own.data(), jobs, (size_t)n_jobs * sizeof(cw_window_job), hipMemcpyDeviceToHost, st));
        CW_HIP(hipStreamSynchronize(st));
    }
    const cw_window_job* hj = jobs_host ? jobs_host : hj_own.data();
    for (uint32_t w = 0; w < n_jobs; ++w) {
        if ((uint64_t)hj[w].ovl_first + hj[w].ovl_count > n_overlaps || hj[w].tpl_read >= reads->n_reads || hj[w].q_end < hj[w].q_beg) return CW_E_INVALID;
        doff[w + 1] = doff[w] + hj[w].ovl_count;
    }
    size_t o = 0;
    auto put = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t o_desc = put((size_t)doff[n_jobs] * sizeof(ExtractDesc) + 16), o_doff = put((size_t)(n_jobs + 1) * 8),
                 o_ws = put((size_t)(n_jobs + 1) * 4), o_ww = put((size_t)(n_jobs + 1) * 8), o_st = put(16);
    int rc = ensure(&e->xscratch, &e->xscratch_bytes, o);
    if (rc) return rc;
    uint8_t* base = (uint8_t*)e->xscratch;
    CW_HIP(hipMemcpyAsync(base + o_doff, doff.data(), (size_t)(n_jobs + 1) * 8, hipMemcpyHostToDevice, st));
    CW_HIP(hipMemsetAsync(base + o_st, 0, 16, st));
    ExtractArgs a;
    a.reads = *reads; a.ovl = overlaps; a.jobs = jobs; a.n_jobs = n_jobs; a.k = k;
    a.desc = (ExtractDesc*)(base + o_desc); a.desc_off = (const uint64_t*)(base + o_doff);
    a.win_seqs = (uint32_t*)(base + o_ws); a.win_words = (uint64_t*)(base + o_ww);
    a.win_first_seq = win_first_seq; a.seq_len = seq_len; a.seq_word_off = seq_word_off; a.bases = bases;
    a.seq_cap = seq_cap; a.word_cap = word_cap; a.status = (uint32_t*)(base + o_st);
    cw_extract_count_kernel<<<(n_jobs + 3) / 4, 256, 0, st>>>(a);
    cw_extract_scan_kernel<<<1, 1024, 0, st>>>(a);
    uint32_t tot_s = 0, flags[4] = {0, 0, 0, 0}; uint64_t tot_w = 0;
    CW_HIP(hipMemcpyAsync(&tot_s, a.win_seqs + n_jobs, 4, hipMemcpyDeviceToHost, st));
    CW_HIP(hipMemcpyAsync(&tot_w, a.win_words + n_jobs, 8, hipMemcpyDeviceToHost, st));
    CW_HIP(hipMemcpyAsync(flags, a.status, 16, hipMemcpyDeviceToHost, st));
    CW_HIP(hipStreamSynchronize(st));
    *n_seqs = tot_s; *n_words = tot_w;
    if (flags[2]) return CW_E_INVALID;  /* an overlap names a target read outside the read set */
    if (flags[1]) return CW_E_CAPACITY; /* a piece longer than 65535 bases (the batch format's length limit): an error, never a silently shorter pile */
    if (tot_s > seq_cap || tot_w > word_cap || !win_first_seq || !seq_len || !seq_word_off || !bases) return CW_E_CAPACITY;
    cw_extract_fill_kernel<<<(n_jobs + 1 + 3) / 4, 256, 0, st>>>(a);
    CW_HIP(hipGetLastError());
    return CW_OK;
}

__global__ void __launch_bounds__(256) cw_add_offsets_kernel(uint32_t* a32, uint64_t n32, uint32_t add32, uint64_t* a64, uint64_t n64, uint64_t add64) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n32) a32[i] += add32;
    if (i < n64) a64[i] += add64;
}

int cw_add_offsets_device(uint32_t* a32, uint64_t n32, uint32_t add32, uint64_t* a64, uint64_t n64, uint64_t add64, void* hip_stream) {
    if ((n32 && !a32) || (n64 && !a64)) return CW_E_INVALID;
    const uint64_t n = n32 > n64 ? n32 : n64;
    if (n == 0 || (add32 == 0 && add64 == 0)) return CW_OK;
    if ((n + 255) / 256 > 0x7FFFFFFFull) return CW_E_INVALID;
    cw_add_offsets_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(a32, n32, add32, a64, n64, add64);
    CW_HIP(hipGetLastError());
    return CW_OK;
}

int cw_plan_results_device(cw_engine* e, const cw_batch* batch, uint64_t* cons_off, uint64_t* solid_off, uint64_t* cons_total, uint64_t* solid_total,
                           void* hip_stream) {
    if (!e || !batch || !cons_off || !solid_off || !cons_total || !solid_total) return CW_E_INVALID;
    *cons_total = 0; *solid_total = 0;
    if (batch->n_windows == 0) return CW_OK;
    if (!batch->win_first_seq || !batch->seq_len) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CW_HIP(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    int rc = ensure(&e->xscratch, &e->xscratch_bytes, 256);
    if (rc) return rc;
    PlanArgs a;
    a.n_windows = batch->n_windows; a.k = e->prm.k; a.solid = e->prm.solid;
    a.win_first_seq = batch->win_first_seq; a.seq_len = batch->seq_len;
    a.cons_off = cons_off; a.solid_off = solid_off; a.totals = (uint64_t*)e->xscratch;
    cw_plan_need_kernel<<<(batch->n_windows + 3) / 4, 256, 0, st>>>(a);
    cw_plan_scan_kernel<<<1, 1024, 0, st>>>(a);
    uint64_t tot[2] = {0, 0};
    CW_HIP(hipMemcpyAsync(tot, a.totals, 16, hipMemcpyDeviceToHost, st));
    CW_HIP(hipStreamSynchronize(st));
    *cons_total = tot[0]; *solid_total = tot[1];
    return CW_OK;
}

int cw_stitch_device(cw_engine* e, const cw_read_set* reads, const cw_stitch_read* jobs, uint32_t n_reads, const uint32_t* win_pos,
                     const cw_batch* batch, const cw_result* res, uint32_t window_size, uint32_t window_overlap, int32_t do_trim,
                     char* out, const uint64_t* out_off, uint32_t* out_len, uint8_t* read_status, void* hip_stream) {
    if (!e || !reads || !batch || !res || !out_off || !out_len || !read_status) return CW_E_INVALID;
    if (n_reads == 0) return CW_OK;
    if (!jobs || !win_pos || !out || !res->cons || !res->cons_off || !res->cons_len || !res->win_status || !res->solid || !res->solid_off ||
        !res->solid_len || !batch->win_first_seq || !batch->seq_len || !batch->seq_word_off || !batch->bases)
        return CW_E_INVALID;
    if ((uint64_t)window_size + 2ull * window_overlap > CW_ST_RMAX) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CW_HIP(hipSetDevice(e->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    const bool want_trace = CW_AID_ENV("CW_STITCH_TRACE") != nullptr;
    const size_t trace_bytes = want_trace ? (size_t)batch->n_windows * 32 : 0;
    int rc = ensure(&e->xscratch, &e->xscratch_bytes, 256 + trace_bytes + (size_t)n_reads * 4);
    if (rc) return rc;
    CW_HIP(hipMemsetAsync(e->xscratch, 0, 16, st));
    StitchArgs a;
    a.order = (uint32_t*)((uint8_t*)e->xscratch + 256 + trace_bytes);
    a.trace = want_trace ? (uint32_t*)((uint8_t*)e->xscratch + 256) : nullptr;
    if (want_trace) CW_HIP(hipMemsetAsync(a.trace, 0xFF, (size_t)batch->n_windows * 32, st));
    a.reads = *reads; a.jobs = jobs; a.n_reads = n_reads; a.win_pos = win_pos;
    a.batch.n_windows = batch->n_windows; a.batch.win_first_seq = batch->win_first_seq; a.batch.seq_len = batch->seq_len;
    a.batch.seq_word_off = batch->seq_word_off; a.batch.bases = batch->bases;
    a.cons = res->cons; a.cons_off = res->cons_off; a.cons_len = res->cons_len; a.win_status = res->win_status;
    a.solid = res->solid; a.solid_off = res->solid_off; a.solid_len = res->solid_len;
    a.window_size = window_size; a.window_overlap = window_overlap; a.mer_size = e->prm.k; a.do_trim = do_trim;
    a.out = out; a.out_off = out_off; a.out_len = out_len; a.read_status = read_status; a.cursor = (uint32_t*)e->xscratch;
    /* one work-group of the wide kernel per CU, i.e. one wave per SIMD: the launch lasts as long as its longest read, and that read's wave is
       faster alone on its SIMD (measured with the register count deciding it: 55.3 ms per job at one wave per SIMD, 60.5 at two).  The kernel
       fits two per CU by registers and LDS; asking for more than half of the LDS keeps the second one off (CW_STITCH_TWO_PER_CU=1: don't). */
    const size_t lds_one = CW_AID_ENV("CW_STITCH_TWO_PER_CU") ? 0 : stitch_lds_one(e->prop);
    const size_t lds_need = (size_t)CW_ST_WAVES * CW_ST_SLAB;
    const size_t lds = lds_need > lds_one ? lds_need : lds_one, lds_n = (size_t)CW_STN_WAVES * CW_ST_SLAB_OF(CW_STN_QMAX, CW_STN_RMAX);
    uint32_t wgs = (n_reads + CW_ST_WAVES - 1) / CW_ST_WAVES;
    if (wgs > CW_ST_MAX_WGS) wgs = CW_ST_MAX_WGS;
    /* the narrow kernel holds four work-groups of four waves per CU (29 KB of LDS each, <= 128 VGPRs) */
    const int cus_st = e->prop.multiProcessorCount > 0 ? e->prop.multiProcessorCount : 256;
    uint32_t wgs_n = (n_reads + CW_STN_WAVES - 1) / CW_STN_WAVES;
    if (wgs_n > (uint32_t)cus_st * 4u) wgs_n = (uint32_t)cus_st * 4u;
    /* Opt-in (CW_STITCH_NARROW=1).  Measured on the E. coli-scale ONT-profile set (10 jobs of 32768 windows, rocprofv3): the narrow kernel takes
       63 ms per job where the wide one takes ~75 -- a launch lasts as long as its longest read's serial chain of windows (a 30 kbp read: 66
       windows), which twice the resident waves do not shorten -- and the reads it hands on (a consensus above 640 in any of their windows)
       cost a second such tail, 51 ms: 114 ms per job against 75.  DESIGN.md "Round 3". */
    const bool narrow = CW_AID_ENV("CW_STITCH_NARROW") && (uint64_t)window_size + 2ull * window_overlap <= CW_STN_RMAX;
    /* several waves per read (cw_stitch.h, st_sweep_sys): one read per work-group of five waves */
    const bool sys = !narrow && CW_AID_ENV("CW_STITCH_SYS") && (uint64_t)window_size + 2ull * window_overlap <= CW_STS_RMAX;
    const size_t lds_s = (((size_t)CW_ST_SLAB_OF(CW_STS_QMAX, CW_STS_RMAX) + 15u) & ~(size_t)15u) + sizeof(StSys);
    uint32_t wgs_s = n_reads;
    if (wgs_s > (uint32_t)cus_st * 6u) wgs_s = (uint32_t)cus_st * 6u;
    uint32_t wgs_max = narrow && wgs_n > wgs ? wgs_n : wgs;
    if (sys && (wgs_s + CW_ST_WAVES - 1) / CW_ST_WAVES > wgs_max) wgs_max = (wgs_s + CW_ST_WAVES - 1) / CW_ST_WAVES;
    /* test aid: CW_STITCH_DIR_BYTES shrinks the banded-traceback scratch so that the capacity path can be exercised */
    a.dir_bytes = CW_ST_DIR_BYTES;
    if (const char* env = CW_AID_ENV("CW_STITCH_DIR_BYTES")) a.dir_bytes = (uint32_t)strtoul(env, nullptr, 10);
    if (a.dir_bytes < 64) a.dir_bytes = 64;
    /* the last launch (consensuses beyond CW_ST_QMAX characters; one wave per work-group, everything in global memory) has its slabs behind the traceback scratch */
    uint32_t wgs_h = n_reads < CW_STH_MAX_WGS ? n_reads : CW_STH_MAX_WGS;
    if (wgs_h > wgs_max * CW_ST_WAVES) wgs_h = wgs_max * CW_ST_WAVES; /* (its waves use the first wgs_h traceback scratches) */
    const size_t dir_total = (((size_t)wgs_max * CW_ST_WAVES * a.dir_bytes) + 255u) & ~(size_t)255u;
    rc = ensure(&e->stitch_scratch, &e->stitch_scratch_bytes, dir_total + (size_t)wgs_h * CW_STH_WAVE_BYTES);
    if (rc) return rc;
    a.dir_scratch = (int8_t*)e->stitch_scratch;
    a.huge = (uint8_t*)e->stitch_scratch + dir_total;
    a.prio = 1;
    if (const char* env = CW_AID_ENV("CW_STITCH_PRIO")) a.prio = atoi(env); /* measured: 69.6 -> 65.3 ms per job of 32768 windows (E. coli-scale ONT set) */
    cw_stitch_order_kernel<<<1, 1024, 0, st>>>(a);
#ifdef CW_TEST_AIDS /* the two opt-in re-assembly kernels (bit-identical, measured no faster: DESIGN.md) exist in the test-aid build only */
    if (sys) {
        cw_stitch_kernel<CW_STS_QMAX, CW_STS_RMAX, 5, 1, false, true><<<wgs_s, 64 * CW_STS_WAVES, lds_s, st>>>(a);
        cw_stitch_kernel<CW_ST_QMAX, CW_ST_RMAX, 16, CW_ST_WAVES, true><<<wgs, 64 * CW_ST_WAVES, lds, st>>>(a); /* the reads it marked (a consensus above 1280: normally none) */
    } else if (narrow) {
        cw_stitch_kernel<CW_STN_QMAX, CW_STN_RMAX, 5, CW_STN_WAVES, false><<<wgs_n, 64 * CW_STN_WAVES, lds_n, st>>>(a);
        cw_stitch_kernel<CW_ST_QMAX, CW_ST_RMAX, 16, CW_ST_WAVES, true><<<wgs, 64 * CW_ST_WAVES, lds, st>>>(a); /* the reads it marked (normally none) */
    } else
#endif
    {
        (void)lds_n; (void)lds_s; (void)wgs_n; (void)wgs_s; (void)narrow; (void)sys;
        cw_stitch_kernel<CW_ST_QMAX, CW_ST_RMAX, 16, CW_ST_WAVES, false><<<wgs, 64 * CW_ST_WAVES, lds, st>>>(a);
    }
    /* the reads the launches above marked (a window's consensus beyond CW_ST_QMAX characters: normally none, and the waves find nothing to do) */
    cw_stitch_kernel<CW_STH_QMAX, CW_ST_RMAX, 0, 1, true><<<wgs_h, 64, 0, st>>>(a);
    CW_HIP(hipGetLastError());
    return CW_OK;
}

int cw_debug_stitch_trace(cw_engine* e, uint32_t n_windows, uint32_t* out) {
    if (!e || !out || !e->xscratch || e->xscratch_bytes < 256 + (size_t)n_windows * 32) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CW_HIP(hipSetDevice(e->device));
    CW_HIP(hipDeviceSynchronize());
    CW_HIP(hipMemcpy(out, (uint8_t*)e->xscratch + 256, (size_t)n_windows * 32, hipMemcpyDeviceToHost));
    return CW_OK;
}

/* ---- host batches: cw_submit / cw_wait (and cw_run = both) -------------------------------------------------------------------
 * H2D on its own stream, the kernels on the compute stream, the results compacted on the device (cw_pack_*: only the bytes the
 * windows actually produced cross PCIe, not the capacities the caller reserved), D2H on a third stream into pinned staging, then
 * scattered into the caller's arrays.  With two batches in flight the copies of one overlap the kernels of the other. */
int cw_submit(cw_engine* e, const cw_batch* b, const cw_result* r, int* ticket) {
    if (!e || !b || !r || !ticket || !r->cons || !r->cons_off || !r->cons_len || !r->win_status) return CW_E_INVALID;
    *ticket = -1;
    if (b->n_windows && (!b->win_first_seq || !b->seq_len || !b->seq_word_off || !b->bases)) return CW_E_INVALID;
    if (b->n_windows > CW_MAX_BATCH_WINDOWS) return CW_E_INVALID;
    const uint32_t W = b->n_windows, S = b->n_seqs;
    /* light validation of the host batch */
    if (W) {
        if (b->win_first_seq[0] != 0 || b->win_first_seq[W] != S) return CW_E_INVALID;
        for (uint32_t w = 0; w < W; ++w) if (b->win_first_seq[w + 1] < b->win_first_seq[w]) return CW_E_INVALID;
        for (uint32_t s = 0; s < S; ++s)
            if (b->seq_word_off[s] + ((uint64_t)b->seq_len[s] + 15) / 16 > b->n_words || b->seq_len[s] > 65535u) return CW_E_INVALID;
    }
    const bool want_solid = r->solid != nullptr;
    if (want_solid && (!r->solid_off || !r->solid_len)) return CW_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    /* the copy streams exist only for callers of the host-batch path, and are created after the four kernel streams: the runtime
       multiplexes streams onto a handful of hardware queues in creation order, and two POA tiers that land on one queue run one after
       the other instead of side by side (measured: tier M1 started only when tier M2 had ended) */
    if (!e->copy_in) {
        CW_HIP(hipSetDevice(e->device));
        if (hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking) != hipSuccess) { e->copy_in = nullptr; return CW_E_NO_DEVICE; }
        if (hipStreamCreateWithFlags(&e->copy_out, hipStreamNonBlocking) != hipSuccess) { e->copy_out = nullptr; return CW_E_NO_DEVICE; }
    }
    int si = -1;
    for (int i = 0; i < CW_SLOTS; ++i) if (!e->slot[i].busy) { si = i; break; }
    if (si < 0) return CW_E_INVALID; /* CW_SLOTS batches are in flight already: cw_wait one of them first */
    cw_slot& sl = e->slot[si];
    sl.res = *r; sl.n_windows = W; sl.want_solid = want_solid;
    if (W == 0) { sl.busy = true; sl.waiting = false; *ticket = si; return CW_OK; }
    CW_HIP(hipSetDevice(e->device));

    size_t o = 0;
    auto put = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t i_wfs = put((size_t)(W + 1) * 4), i_len = put((size_t)S * 4), i_off = put((size_t)S * 8), i_bases = put((size_t)(b->n_words + 1) * 4);
    const size_t in_bytes = o;
    o = 0;
    sl.cons_cap = r->cons_off[W];
    sl.solid_cap = want_solid ? r->solid_off[W] : 0;
    const size_t o_cons = put(sl.cons_cap), o_coff = put((size_t)(W + 1) * 8);
    sl.o_clen = put((size_t)W * 4); sl.o_stat = put(W);
    const size_t o_solid = put(sl.solid_cap * 4), o_soff = put((size_t)(W + 1) * 8);
    sl.o_slen = put((size_t)W * 4);
    sl.o_pc = put(sl.cons_cap); sl.o_ps = put(sl.solid_cap * 4); /* the compacted copies */
    const size_t o_pco = put((size_t)(W + 1) * 8), o_pso = put((size_t)(W + 1) * 8);
    sl.o_tot = put(40);
    sl.o_cons = o_cons; sl.o_solid = o_solid;
    const size_t out_bytes = o;
    int rc = ensure(&sl.dev_in, &sl.dev_in_bytes, in_bytes);
    if (rc) return rc;
    rc = ensure(&sl.dev_out, &sl.dev_out_bytes, out_bytes);
    if (rc) return rc;
    uint8_t* din = (uint8_t*)sl.dev_in;
    uint8_t* dout = (uint8_t*)sl.dev_out;
    hipStream_t ci = e->copy_in;
    CW_HIP(hipMemcpyAsync(din + i_wfs, b->win_first_seq, (size_t)(W + 1) * 4, hipMemcpyHostToDevice, ci));
    CW_HIP(hipMemcpyAsync(din + i_len, b->seq_len, (size_t)S * 4, hipMemcpyHostToDevice, ci));
    CW_HIP(hipMemcpyAsync(din + i_off, b->seq_word_off, (size_t)S * 8, hipMemcpyHostToDevice, ci));
    CW_HIP(hipMemcpyAsync(din + i_bases, b->bases, (size_t)b->n_words * 4, hipMemcpyHostToDevice, ci));
    CW_HIP(hipMemsetAsync(din + i_bases + (size_t)b->n_words * 4, 0, 4, ci));
    CW_HIP(hipMemcpyAsync(dout + o_coff, r->cons_off, (size_t)(W + 1) * 8, hipMemcpyHostToDevice, ci));
    if (want_solid) CW_HIP(hipMemcpyAsync(dout + o_soff, r->solid_off, (size_t)(W + 1) * 8, hipMemcpyHostToDevice, ci));
    CW_HIP(hipEventRecord(sl.ev_in, ci));
    hipStream_t st = e->stream;
    CW_HIP(hipStreamWaitEvent(st, sl.ev_in, 0));

    cw_batch db = *b;
    db.win_first_seq = (const uint32_t*)(din + i_wfs); db.seq_len = (const uint32_t*)(din + i_len);
    db.seq_word_off = (const uint64_t*)(din + i_off); db.bases = (const uint32_t*)(din + i_bases);
    cw_result dr;
    dr.cons = (char*)(dout + o_cons); dr.cons_off = (const uint64_t*)(dout + o_coff); dr.cons_len = (uint32_t*)(dout + sl.o_clen);
    dr.win_status = dout + sl.o_stat;
    dr.solid = want_solid ? (uint32_t*)(dout + o_solid) : nullptr;
    dr.solid_off = want_solid ? (const uint64_t*)(dout + o_soff) : nullptr;
    dr.solid_len = want_solid ? (uint32_t*)(dout + sl.o_slen) : nullptr;
    rc = run_device_locked(e, &db, &dr, st);
    if (rc) return rc;
    PackArgs pa;
    pa.n_windows = W; pa.cons = dr.cons; pa.cons_off = dr.cons_off; pa.cons_len = dr.cons_len; pa.win_status = dr.win_status;
    pa.solid = dr.solid; pa.solid_off = dr.solid_off; pa.solid_len = dr.solid_len;
    pa.pc = (char*)(dout + sl.o_pc); pa.ps = (uint32_t*)(dout + sl.o_ps);
    pa.pc_off = (uint64_t*)(dout + o_pco); pa.ps_off = (uint64_t*)(dout + o_pso); pa.totals = (uint64_t*)(dout + sl.o_tot);
    pa.win = (const WinInfo*)e->scratch; /* this batch's: the pack kernels follow its finish kernel on the compute stream */
    pa.used = (const uint32_t*)((const uint8_t*)e->scratch + e->last_ctr_off + offsetof(BatchCounters, n_tasks));
    static_assert(offsetof(BatchCounters, n_members) == offsetof(BatchCounters, n_tasks) + 4, "cw_pack_scan_kernel reads the two counters as a pair");
    cw_pack_scan_kernel<<<1, 1024, 0, st>>>(pa);
    cw_pack_copy_kernel<<<(W + 3) / 4, 256, 0, st>>>(pa);
    CW_HIP(hipGetLastError());
    CW_HIP(hipEventRecord(sl.ev_done, st));
    sl.busy = true; sl.waiting = false;
    *ticket = si;
    return CW_OK;
}

static int wait_ticket(cw_engine* e, int ticket, uint64_t* why_tasks, uint32_t* used = nullptr);
int cw_wait(cw_engine* e, int ticket) { return wait_ticket(e, ticket, nullptr); }

/* why_tasks (may be NULL): how many windows of THIS ticket stopped on the batch's task / member / arena capacities */
/* used (may be NULL): this ticket's task and member counts */
static int wait_ticket(cw_engine* e, int ticket, uint64_t* why_tasks, uint32_t* used) {
    if (why_tasks) *why_tasks = 0;
    if (used) used[0] = used[1] = 0xFFFFFFFFu;
    if (!e || ticket < 0 || ticket >= CW_SLOTS) return CW_E_INVALID;
    cw_slot& sl = e->slot[ticket];
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!sl.busy || sl.waiting) return CW_E_INVALID; /* a ticket is waited for once: a second waiter would race on the staging buffers */
        if (sl.n_windows == 0) { sl.busy = false; return CW_OK; }
        sl.waiting = true;
    }
    /* the slot belongs to this ticket until busy is cleared; other threads may submit meanwhile */
    auto fail = [&](int code) { std::lock_guard<std::mutex> lk(e->mu); sl.busy = false; sl.waiting = false; return code; };
    if (hipSetDevice(e->device) != hipSuccess || hipEventSynchronize(sl.ev_done) != hipSuccess) return fail(CW_E_NO_DEVICE);
    const uint32_t W = sl.n_windows;
    const cw_result& r = sl.res;
    uint8_t* dout = (uint8_t*)sl.dev_out;
    hipStream_t co = e->copy_out;
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    int rc = CW_OK;
    if (hipMemcpyAsync(tot, dout + sl.o_tot, 40, hipMemcpyDeviceToHost, co) != hipSuccess ||
        hipMemcpyAsync(r.cons_len, dout + sl.o_clen, (size_t)W * 4, hipMemcpyDeviceToHost, co) != hipSuccess ||
        hipMemcpyAsync(r.win_status, dout + sl.o_stat, W, hipMemcpyDeviceToHost, co) != hipSuccess ||
        (sl.want_solid && hipMemcpyAsync(r.solid_len, dout + sl.o_slen, (size_t)W * 4, hipMemcpyDeviceToHost, co) != hipSuccess) ||
        hipStreamSynchronize(co) != hipSuccess)
        return fail(CW_E_NO_DEVICE);
    if (tot[0] > sl.cons_cap || tot[1] > sl.solid_cap) return fail(CW_E_INTERNAL);
    if (why_tasks) *why_tasks = tot[2];
    if (used) { used[0] = (uint32_t)tot[3]; used[1] = (uint32_t)tot[4]; }
    const size_t need = align_up(tot[0], 256) + tot[1] * 4 + 256;
    if (sl.pin_out_bytes < need) {
        if (sl.pin_out) (void)hipHostFree(sl.pin_out);
        sl.pin_out = nullptr; sl.pin_out_bytes = 0;
        const size_t want = need + need / 4; /* some slack: the next batch is rarely exactly as large */
        if (hipHostMalloc(&sl.pin_out, want, hipHostMallocDefault) != hipSuccess) { sl.pin_out = nullptr; return fail(CW_E_NOMEM); }
        sl.pin_out_bytes = want;
    }
    char* hc = (char*)sl.pin_out;
    uint32_t* hs = (uint32_t*)((uint8_t*)sl.pin_out + align_up(tot[0], 256));
    if ((tot[0] && hipMemcpyAsync(hc, dout + sl.o_pc, tot[0], hipMemcpyDeviceToHost, co) != hipSuccess) ||
        (tot[1] && hipMemcpyAsync(hs, dout + sl.o_ps, tot[1] * 4, hipMemcpyDeviceToHost, co) != hipSuccess) ||
        hipStreamSynchronize(co) != hipSuccess)
        return fail(CW_E_NO_DEVICE);
    uint64_t pc = 0, ps = 0;
    for (uint32_t w = 0; w < W; ++w) {
        const bool ok = r.win_status[w] != CW_WIN_OVERFLOW;
        if (!ok) rc = CW_E_CAPACITY;
        const uint32_t cl = ok ? r.cons_len[w] : 0u;
        if (cl) memcpy(r.cons + r.cons_off[w], hc + pc, cl);
        pc += cl;
        if (sl.want_solid) {
            const uint32_t n = ok ? r.solid_len[w] : 0u;
            if (n) memcpy(r.solid + r.solid_off[w], hs + ps, (size_t)n * 4);
            ps += n;
        }
    }
    if (pc != tot[0] || ps != tot[1]) rc = CW_E_INTERNAL;
    return fail(rc);
}

/* After a finished run: did windows stop on a capacity that a larger scratch plan cures (CW_WHY_TASKS: the task, member and arena slots are
   heuristics of the batch)?  Then the plan grows (x4, up to x64) and the caller runs the batch again -- if the larger plan is one the engine can
   actually run: inside the 32-bit offsets (plan_scratch clamps the arena's own scale) and inside the device's free memory.  A scale that is not
   is refused here, and the run that has finished stands with its overflow statuses (it used to end the batch with CW_E_INVALID or
   CW_E_NOMEM).  known_any: -1 = look at the last run's WinInfo in scratch (cw_run_device_sync: the caller has just waited for that very run);
   0 / 1 / 2 / 3 = the caller knows from its own ticket which kinds of such windows exist (bit 0 task / member slots, bit 1 arena slices) (cw_run: scratch may already belong to another thread's batch). */
static int grow_if_that_helps(cw_engine* e, bool* again, hipStream_t st, int known_any, uint32_t n_windows, uint32_t n_seqs, uint64_t n_words) {
    *again = false;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->scratch || n_windows == 0 || e->cap_scale >= 64u) return CW_OK;
    if (CW_AID_ENV("CW_TASK_CAP") || CW_AID_ENV("CW_MEMBER_CAP")) return CW_OK; /* (test aids that shrink exactly these capacities) */
    CW_HIP(hipSetDevice(e->device));
    int kinds = known_any > 0 ? known_any : 0; /* bit 0: windows stopped on task / member slots, bit 1: on their arena slices */
    if (known_any < 0) {
        if (e->last_windows != n_windows) return CW_OK;
        /* copies on the caller's stream, not hipMemcpy: the null stream would wait for every other stream of the device -- the other worker's job */
        uint32_t* flag = e->host_fb + 8; /* pinned */
        CW_HIP(hipMemcpyAsync(flag, (uint8_t*)e->scratch + e->last_ctr_off + offsetof(BatchCounters, any_overflow), 4, hipMemcpyDeviceToHost, st));
        CW_HIP(hipStreamSynchronize(st));
        if (!*flag) return CW_OK;
        std::vector<WinInfo> wi(e->last_windows);
        CW_HIP(hipMemcpyAsync(wi.data(), e->scratch, (size_t)e->last_windows * sizeof(WinInfo), hipMemcpyDeviceToHost, st));
        CW_HIP(hipStreamSynchronize(st));
        /* (windows stopped for other reasons stay stopped: the loop ends when no window names these capacities, or at x64) */
        for (const WinInfo& w : wi) if (w.status == CW_WIN_OVERFLOW) kinds |= w.pad_ == CW_WHY_TASKS ? 1 : w.pad_ == CW_WHY_ARENA ? 2 : 0;
    }
    if (!kinds) return CW_OK;
    const uint32_t next = e->cap_scale * 4u;
    const int cus = e->prop.multiProcessorCount > 0 ? e->prop.multiProcessorCount : 256;
    const ScratchPlan p = plan_scratch(e->prm, n_windows, n_seqs, n_words, cus, e->last_big_slots ? e->last_big_slots : 64, next, e->tmax_plan);
    {   /* does the larger plan grow what ran out?  (the arena's own scale is clamped to 32-bit offsets: a batch stopped on its arena slices
           at a clamped scale is the same batch with the same arena at x4, x16 and x64) */
        const ScratchPlan cur = plan_scratch(e->prm, n_windows, n_seqs, n_words, cus, e->last_big_slots ? e->last_big_slots : 64, e->cap_scale, e->tmax_plan);
        const bool helps = ((kinds & 1) && (p.task_cap > cur.task_cap || p.member_cap > cur.member_cap)) || ((kinds & 2) && p.arena_scale > cur.arena_scale);
        if (!helps) {
            fprintf(stderr, "[consent_amd] windows stopped on the batch's %s capacity; a plan x%u does not enlarge it: keeping the run's result\n", (kinds & 2) ? "arena" : "task / member", next);
            return CW_OK;
        }
    }
    size_t free_b = 0, total_b = 0;
    const bool mem_known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    if (!mem_known) (void)hipGetLastError();
    const size_t margin = (size_t)2 << 30; /* other engines of the device grow too */
    const bool fits = p.solid_cap <= 0xFFFFFFFFull && p.seg_cap <= 0xFFFFFFFFull && p.arena_cap <= 0xFFFFFFFFull &&
                      (!mem_known || p.total <= e->scratch_bytes || p.total - e->scratch_bytes + margin <= free_b);
    if (!fits) {
        fprintf(stderr, "[consent_amd] windows stopped on the batch's task / member / arena capacities; a plan x%u (%.1f GB) does not fit this device now: keeping the run's result\n",
                next, (double)p.total / 1e9);
        return CW_OK;
    }
    e->cap_scale = next; *again = true;
    fprintf(stderr, "[consent_amd] windows stopped on the batch's task / member / arena capacities: running the batch again with the plan x%u\n", e->cap_scale);
    return CW_OK;
}

/* A grown plan is not forever: after a run at scale > 1 that used less than half of what the next smaller plan offers, the
   scale goes back a step; scratch that has become three times what the smaller plan needs is given back to the device. */
/* known (may be NULL): the run's own task and member counts (cw_run: its ticket's totals -- the counters in scratch may by now be another thread's batch);
   NULL: read them from scratch (cw_run_device_sync: the caller has just waited for that very run).  Scratch is only given back when nothing of this
   engine is in flight. */
static void decay_scale(cw_engine* e, hipStream_t st, uint32_t n_windows, uint32_t n_seqs, uint64_t n_words, const uint32_t* known = nullptr) {
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->cap_scale <= 1u || !e->scratch || (!known && e->last_windows != n_windows)) return;
    if (hipSetDevice(e->device) != hipSuccess) return;
    uint32_t* used = e->host_fb + 10; /* pinned: n_tasks, n_members */
    if (known) { if (known[0] == 0xFFFFFFFFu) return; used[0] = known[0]; used[1] = known[1]; }
    else if (hipMemcpyAsync(used, (uint8_t*)e->scratch + e->last_ctr_off + offsetof(BatchCounters, n_tasks), 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return; }
    const int cus = e->prop.multiProcessorCount > 0 ? e->prop.multiProcessorCount : 256;
    const ScratchPlan lower = plan_scratch(e->prm, n_windows, n_seqs, n_words, cus, e->last_big_slots ? e->last_big_slots : 64, e->cap_scale / 4u, e->tmax_plan);
    if ((uint64_t)used[0] * 2u > lower.task_cap || (uint64_t)used[1] * 2u > lower.member_cap) return;
    e->cap_scale /= 4u;
    bool in_flight = false;
    for (int i = 0; i < CW_SLOTS; ++i) in_flight = in_flight || e->slot[i].busy; /* (cw_run: another thread's batch may be running on this scratch) */
    if (!in_flight && e->scratch_bytes > 3 * lower.total + ((size_t)1 << 30)) { /* nothing of this engine is in flight: the caller has just waited for its stream */
        if (hipFree(e->scratch) != hipSuccess) (void)hipGetLastError();
        e->scratch = nullptr; e->scratch_bytes = 0;
    }
}

/* cw_run_device, then wait for the stream, then -- when windows stopped on the batch's task / member / arena capacities only -- once more
   with a larger plan (cw_private.h; what the native driver and cw_run call: the asynchronous cw_run_device cannot look at its own result) */
int cw_run_device_sync(cw_engine* e, const cw_batch* batch, const cw_result* res, void* hip_stream) {
    if (!e || !batch) return CW_E_INVALID;
    bool grown = false;
    uint32_t scale_before = 1;
    for (;;) {
        int rc = cw_run_device(e, batch, res, hip_stream);
        if (rc != CW_OK) {
            if (grown && (rc == CW_E_INVALID || rc == CW_E_NOMEM)) { /* the larger plan could not be set up after all (memory went to another engine meanwhile):
                                                                         no kernel of the re-run was launched, the run before it stands, the scale goes back */
                std::lock_guard<std::mutex> lk(e->mu);
                e->cap_scale = scale_before;
                return CW_OK;
            }
            return rc;
        }
        CW_HIP(hipSetDevice(e->device));
        hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
        CW_HIP(hipStreamSynchronize(st));
        bool again = false;
        { std::lock_guard<std::mutex> lk(e->mu); scale_before = e->cap_scale; }
        if ((rc = grow_if_that_helps(e, &again, st, -1, batch->n_windows, batch->n_seqs, batch->n_words)) != CW_OK) return rc;
        if (!again) {
            if (!grown) decay_scale(e, st, batch->n_windows, batch->n_seqs, batch->n_words);
            return CW_OK;
        }
        grown = true;
    }
}

int cw_run(cw_engine* e, const cw_batch* b, const cw_result* r) {
    bool grown = false;
    uint32_t scale_before = 1;
    int first_rc = CW_OK;
    for (;;) {
        int t = -1;
        int rc = cw_submit(e, b, r, &t);
        if (rc != CW_OK) {
            if (grown && (rc == CW_E_INVALID || rc == CW_E_NOMEM)) { std::lock_guard<std::mutex> lk(e->mu); e->cap_scale = scale_before; return first_rc; } /* see cw_run_device_sync */
            return rc;
        }
        uint64_t why_tasks = 0;
        uint32_t used[2];
        const int wrc = wait_ticket(e, t, &why_tasks, used);
        if (wrc != CW_OK && wrc != CW_E_CAPACITY) return wrc;
        bool again = false;
        { std::lock_guard<std::mutex> lk(e->mu); scale_before = e->cap_scale; }
        /* the decision comes from this ticket's own totals (cw_pack_scan_kernel counts the windows stopped on CW_WHY_TASKS): other threads may have
           submitted since, and the WinInfo in scratch may be theirs */
        if (wrc == CW_E_CAPACITY && (rc = grow_if_that_helps(e, &again, e->stream, ((uint32_t)why_tasks ? 1 : 0) | ((why_tasks >> 32) ? 2 : 0), b->n_windows, b->n_seqs, b->n_words)) != CW_OK) return rc;
        if (!again) {
            if (!grown && wrc == CW_OK) decay_scale(e, e->stream, b->n_windows, b->n_seqs, b->n_words, used); /* the host path never lowered a grown scale */
            return wrc;
        }
        grown = true; first_rc = wrc;
    }
}

/* pinned host memory for batches and results: copies from and to it are real DMA transfers that overlap the kernels; plain
   (pageable) buffers work too, but the runtime stages them through its own bounce buffers and the overlap is lost */
int cw_host_alloc(void** ptr, size_t bytes) {
    if (!ptr) return CW_E_INVALID;
    *ptr = nullptr;
    if (hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { *ptr = nullptr; return CW_E_NOMEM; }
    return CW_OK;
}

void cw_host_free(void* ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

} // extern "C"
