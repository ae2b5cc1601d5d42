/*
 * cw_synth.h -- synthetic window piles (SURVEY.md 8d generator), one routine for host and device.
 *
 * For window id W: truth = i.i.d. uniform ACGT from splitmix64(seed + W); the template and every
 * support sequence are independent noisy walks along the truth (error rate err_permille/1000 per
 * truth base, split sub:ins:del by the three weights; substituted base != truth base, inserted base
 * uniform).  The template starts at truth offset CW_SYNTH_FLANK and emits exactly window_len bases.
 * Support copies start at CW_SYNTH_FLANK + U[-20,20], aim for window_len + U[-20,20] bases, and 15 %
 * of them are cut to a random prefix or suffix of U[9, window_len-1] bases (mimics the partial
 * overlaps of alignmentWindows.cpp:110-131).
 *
 * Compiled with hipcc only (both the host filler and the device kernel call synth_sequence()).
 */
#ifndef CW_SYNTH_H
#define CW_SYNTH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CW_SYNTH_FLANK 64
#define CW_SYNTH_TRUTH_MAX 4096 /* (1024 through round 5: a 500-base window's walk ends near 650, so the piles of the bench and of the tests are what they were) */

struct SynthRng {
    uint64_t s;
    __host__ __device__ inline uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    /* uniform in [0, n) -- multiply-shift on the high 32 bits (n < 2^31) */
    __host__ __device__ inline uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

__host__ __device__ inline uint32_t synth_truth_base(uint64_t wseed, uint32_t pos) {
    /* counter-based so that every sequence of the window sees the same truth without storing it */
    SynthRng r{wseed ^ (0xD1B54A32D192ED03ull * (uint64_t)(pos / 32 + 1))};
    uint64_t bits = r.next();
    return (uint32_t)(bits >> (2 * (pos % 32))) & 3u;
}

/*
 * Emit sequence `s` (0 = template) of window seed `wseed` as 2-bit codes into words[] (16 per word).
 * Returns its length in bases.  cap_bases is the room at `words`.
 */
__host__ __device__ inline uint32_t synth_sequence(uint64_t wseed, uint32_t s, uint32_t window_len, uint32_t err_permille,
                                                   uint32_t sub_w, uint32_t ins_w, uint32_t del_w, uint32_t* words,
                                                   uint32_t cap_bases) {
    SynthRng r{wseed * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull * (uint64_t)(s + 1)};
    int32_t start = CW_SYNTH_FLANK;
    uint32_t want = window_len;
    uint32_t cut_mode = 0, cut_len = 0; /* 1 = keep prefix, 2 = keep suffix */
    if (s != 0) {
        start += (int32_t)r.below(41) - 20;
        want = (uint32_t)((int32_t)window_len + (int32_t)r.below(41) - 20);
        if (r.below(100) < 15) {
            cut_mode = 1 + r.below(2);
            uint32_t span = window_len > 10 ? window_len - 9 : 1;
            cut_len = 9 + r.below(span);
        }
    }
    if (want > cap_bases) want = cap_bases;
    const uint32_t wsum = sub_w + ins_w + del_w;
    /* a suffix cut needs the full walk first; walk once, remember how many bases to drop at the front */
    uint32_t drop = 0, keep = want;
    if (cut_mode == 1 && cut_len < want) keep = cut_len;
    if (cut_mode == 2 && cut_len < want) { drop = want - cut_len; keep = cut_len; }

    uint32_t emitted = 0, out = 0, cur = 0;
    uint32_t tp = (uint32_t)start;
    const uint32_t total = drop + keep;
    while (emitted < total && tp < CW_SYNTH_TRUTH_MAX) {
        uint32_t tb = synth_truth_base(wseed, tp);
        uint32_t b = tb;
        bool emit = true;
        if (r.below(1000) < err_permille) {
            uint32_t e = r.below(wsum);
            if (e < sub_w) { b = (tb + 1 + r.below(3)) & 3u; tp++; }
            else if (e < sub_w + ins_w) { b = r.below(4); }
            else { emit = false; tp++; }
        } else {
            tp++;
        }
        if (!emit) continue;
        if (emitted >= drop) {
            cur |= b << (30 - 2 * (out & 15u));
            if ((out & 15u) == 15u) { words[out >> 4] = cur; cur = 0; }
            out++;
        }
        emitted++;
    }
    if (out & 15u) words[out >> 4] = cur;
    return out;
}

#endif /* CW_SYNTH_H */
