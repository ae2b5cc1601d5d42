/* cw_internal.h -- engine object shared by the translation units of libconsent_amd.so. */
#ifndef CW_INTERNAL_H
#define CW_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/consent_amd.h"

#define CW_MAX_STAGES 14

struct cw_engine {
    cw_params prm;
    int device;
    hipStream_t stream;
    hipDeviceProp_t prop;
    /* growable device scratch owned by the engine (never shrinks) */
    void* scratch;
    size_t scratch_bytes;
    /* staging for cw_run (host buffers) */
    void* dev_in;
    size_t dev_in_bytes;
    void* dev_out;
    size_t dev_out_bytes;
    void* xscratch; /* pile-extraction scratch */
    size_t xscratch_bytes;
    void* stitch_scratch; /* banded-traceback directions of cw_stitch_device, per wave */
    size_t stitch_scratch_bytes;
    /* per-stage timing of the last run */
    hipStream_t side[3];          /* POA tiers run concurrently on their own streams */
    hipEvent_t ev_fork, ev_join[3];
    hipEvent_t ev0[CW_MAX_STAGES], ev1[CW_MAX_STAGES]; /* start/stop per stage, recorded on the stage's stream */
    hipEvent_t ev_begin, ev_end;
    int n_stages;
    const char* stage_name[CW_MAX_STAGES];
    float stage_ms[CW_MAX_STAGES];
    bool timings_valid;
    uint32_t last_windows, last_big_slots;
    uint32_t linger_wgs; /* tier-L work-groups kept on the live overflow queue (adapted from the previous batch) */
    uint64_t last_words;
};

#define CW_HIP(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return CW_E_NO_DEVICE;   \
    } while (0)

#endif
