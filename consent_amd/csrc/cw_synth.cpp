/* cw_synth.cpp -- synthetic pile generation on host and on device (see cw_synth.h). */
#include "cw_synth.h"
#include "cw_internal.h"
#include "cw_private.h"

static int check_spec(const cw_synth_spec* s) {
    if (!s || s->window_len < 16 || s->window_len > 2100) return CW_E_INVALID;
    if (s->sub_w + s->ins_w + s->del_w == 0 || s->err_permille > 500) return CW_E_INVALID;
    if ((uint64_t)s->seq_stride_words * 16 < (uint64_t)s->window_len + 24) return CW_E_INVALID;
    return CW_OK;
}

__global__ void synth_kernel(cw_synth_spec sp, uint32_t* win_first_seq, uint32_t* seq_len, uint64_t* seq_word_off,
                             uint32_t* bases) {
    const uint32_t per = sp.depth + 1;
    const uint64_t total = (uint64_t)sp.n_windows * per;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = (uint32_t)(g / per), s = (uint32_t)(g % per);
        const uint64_t off = g * sp.seq_stride_words;
        seq_word_off[g] = off;
        seq_len[g] = synth_sequence(sp.seed + sp.first_window + w, s, sp.window_len, sp.err_permille, sp.sub_w, sp.ins_w,
                                    sp.del_w, bases + off, sp.seq_stride_words * 16);
        if (s == 0) win_first_seq[w] = w * per;
        if (g == total - 1) win_first_seq[sp.n_windows] = (uint32_t)total;
    }
}

extern "C" {

int cw_synth_sizes(const cw_synth_spec* spec, uint32_t* n_seqs, uint64_t* n_words) {
    int rc = check_spec(spec);
    if (rc) return rc;
    uint64_t ns = (uint64_t)spec->n_windows * (spec->depth + 1);
    if (ns > 0xFFFFFFFFull) return CW_E_INVALID;
    if (n_seqs) *n_seqs = (uint32_t)ns;
    if (n_words) *n_words = ns * spec->seq_stride_words;
    return CW_OK;
}

int cw_synth_host(const cw_synth_spec* sp, uint32_t* win_first_seq, uint32_t* seq_len, uint64_t* seq_word_off,
                  uint32_t* bases) {
    int rc = check_spec(sp);
    if (rc) return rc;
    const uint32_t per = sp->depth + 1;
    for (uint32_t w = 0; w < sp->n_windows; ++w) {
        win_first_seq[w] = w * per;
        for (uint32_t s = 0; s < per; ++s) {
            const uint64_t g = (uint64_t)w * per + s, off = g * sp->seq_stride_words;
            seq_word_off[g] = off;
            for (uint32_t i = 0; i < sp->seq_stride_words; ++i) bases[off + i] = 0;
            seq_len[g] = synth_sequence(sp->seed + sp->first_window + w, s, sp->window_len, sp->err_permille, sp->sub_w,
                                        sp->ins_w, sp->del_w, bases + off, sp->seq_stride_words * 16);
        }
    }
    win_first_seq[sp->n_windows] = sp->n_windows * per;
    return CW_OK;
}

int cw_synth_device(cw_engine* e, const cw_synth_spec* sp, uint32_t* win_first_seq, uint32_t* seq_len,
                    uint64_t* seq_word_off, uint32_t* bases, void* hip_stream) {
    int rc = check_spec(sp);
    if (rc || !e) return rc ? rc : CW_E_INVALID;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    uint64_t nw = (uint64_t)sp->n_windows * (sp->depth + 1) * sp->seq_stride_words;
    CW_HIP(hipMemsetAsync(bases, 0, nw * 4, st));
    synth_kernel<<<2048, 256, 0, st>>>(*sp, win_first_seq, seq_len, seq_word_off, bases);
    CW_HIP(hipGetLastError());
    return CW_OK;
}

int64_t cw_pack_sequence(const char* seq, uint32_t len, uint32_t* out, uint64_t words_cap) {
    const uint64_t need = ((uint64_t)len + 15) / 16;
    if (!out || (!seq && len) || need > words_cap) return CW_E_INVALID;
    for (uint64_t i = 0; i < need; ++i) out[i] = 0;
    for (uint32_t j = 0; j < len; ++j) {
        uint32_t c;
        switch (seq[j]) { /* utils.cpp:24-28 (input is upper-cased first, utils.cpp:189) */
        case 'A': case 'a': c = 0; break;
        case 'C': case 'c': c = 1; break;
        case 'G': case 'g': c = 2; break;
        default: c = 3; break;
        }
        out[j >> 4] |= c << (30 - 2 * (j & 15));
    }
    return (int64_t)need;
}

} // extern "C"
