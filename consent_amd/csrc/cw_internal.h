/* cw_internal.h -- engine object shared by the translation units of libconsent_amd.so. */
#ifndef CW_INTERNAL_H
#define CW_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/consent_amd.h"

#define CW_MAX_STAGES 16
#define CW_SLOTS 2 /* host batches in flight through cw_submit / cw_wait */

/* one host batch in flight (cw_submit .. cw_wait): its own device input/output buffers and pinned result staging; the kernels of
   all slots share the engine's scratch and run in submission order on the compute stream */
struct cw_slot {
    bool busy = false;
    bool waiting = false; /* a cw_wait is in progress on this ticket */
    void* dev_in = nullptr;
    size_t dev_in_bytes = 0;
    void* dev_out = nullptr;
    size_t dev_out_bytes = 0;
    void* pin_out = nullptr; /* pinned host staging of the compacted results */
    size_t pin_out_bytes = 0;
    hipEvent_t ev_in = nullptr, ev_done = nullptr;
    cw_result res{};         /* the caller's result arrays (host) */
    uint32_t n_windows = 0;
    bool want_solid = false;
    size_t o_cons = 0, o_clen = 0, o_stat = 0, o_solid = 0, o_slen = 0, o_pc = 0, o_ps = 0, o_tot = 0; /* offsets in dev_out */
    uint64_t cons_cap = 0, solid_cap = 0;
};

#include "cw_env.h"

struct cw_engine {
    std::mutex mu; /* every entry point that touches the engine takes it: one engine = one caller at a time (see consent_amd.h "Threading") */
    cw_params prm{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_in = nullptr, copy_out = nullptr; /* H2D / D2H of host batches, beside the compute stream */
    hipDeviceProp_t prop{};
    /* growable device scratch owned by the engine (never shrinks) */
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    cw_slot slot[CW_SLOTS];
    unsigned long long* step_clock = nullptr; /* see DevScratch */
    void* xscratch = nullptr; /* pile-extraction scratch */
    size_t xscratch_bytes = 0;
    void* stitch_scratch = nullptr; /* banded-traceback directions of cw_stitch_device, per wave */
    size_t stitch_scratch_bytes = 0;
    uint32_t* host_fb = nullptr; /* pinned: [0] tasks the last finished batch handed to tier L (feeds linger_wgs), [1] its sequence number */
    /* per-stage timing of the last run */
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr}; /* POA tiers run concurrently on their own streams (M1, M2, L; round 6: LW) */
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join_s = nullptr;
    hipEvent_t ev0[CW_MAX_STAGES] = {}, ev1[CW_MAX_STAGES] = {}; /* start/stop per stage, recorded on the stage's stream */
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int n_stages = 0;
    const char* stage_name[CW_MAX_STAGES] = {};
    float stage_ms[CW_MAX_STAGES] = {};
    bool timings_valid = false;
    uint32_t last_windows = 0, last_big_slots = 0, last_seqs = 0;
    uint32_t tmax_plan = 1024; /* template k-mers per window the scratch plan provides for (cw_configure) */
    uint32_t linger_wgs = 0; /* tier-L work-groups kept on the live overflow queue (adapted from the previous batch) */
    uint32_t cap_scale = 1;  /* multiplier of the batch's task / member / arena capacities: grows (x4) after a run that stopped on them (cw_run_device_sync) */
    uint64_t last_words = 0;
    size_t last_ctr_off = 0; /* where the last run's BatchCounters sit in scratch */
    size_t last_tasks_off = 0, last_tdbg_off = 0;
    uint32_t last_task_cap = 0;
};

extern "C" int cw_extract_impl(cw_engine* e, const cw_read_set* reads, const cw_overlap* overlaps, uint64_t n_overlaps, const cw_window_job* jobs,
                               const cw_window_job* jobs_host, uint32_t n_jobs, uint32_t k, uint32_t* win_first_seq, uint32_t* seq_len,
                               uint64_t* seq_word_off, uint32_t* bases, uint32_t seq_cap, uint64_t word_cap, uint32_t* n_seqs, uint64_t* n_words,
                               void* hip_stream);

#define CW_HIP(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return CW_E_NO_DEVICE;   \
    } while (0)

#endif
