/*
 * consent_amd_adapter.hpp -- C++ host side over the C ABI, shaped like CONSENT's operator
 * (computeConsensusReadCorrection / computeConsensusAssemblyPolishing, src/correctionMSA.h:8,10) but batched over windows.
 *
 * Header-only; needs only consent_amd.h and the shared library.  It mirrors the reference interface for this path:
 * same parameter names and meaning; the `(consensus, merCounts)` pair of the reference becomes
 * `(consensus, solid k-mers)` -- downstream code reads merCounts only through `count >= solidThresh`
 * (src/correctionAlignment.cpp:6-15).  Errors: std::runtime_error on a library error; a window whose capacity overflowed
 * comes back with `overflow = true` and an empty consensus (the caller decides what to do with it).
 */
#ifndef CONSENT_AMD_ADAPTER_HPP
#define CONSENT_AMD_ADAPTER_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "consent_amd.h"

namespace consent_amd {

struct WindowConsensus {
    std::string consensus;             /* case-annotated, as the reference returns it                       */
    std::vector<uint32_t> solid_kmers; /* ascending, str2num order, count >= solidThresh in the window pile */
    bool template_fallback = false;    /* MSA empty: raw template returned (correctionMSA.cpp:34-36)        */
    bool overflow = false;
};

class Engine {
  public:
    Engine(unsigned merSize, unsigned solidThresh, unsigned commonKMers, unsigned minAnchors, unsigned maxMSA, int device = 0) : solid_(solidThresh) {
        cw_params p{merSize, solidThresh, commonKMers, minAnchors, maxMSA};
        const int rc = cw_create(&p, device, &eng_);
        if (rc != CW_OK) throw std::runtime_error(std::string("consent_amd: cw_create: ") + cw_strerror(rc));
    }
    ~Engine() { cw_destroy(eng_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    /* piles[w] = what getAlignmentWindowsSequences returned for window w (template first, alignmentWindows.cpp:100) */
    std::vector<WindowConsensus> computeConsensus(const std::vector<std::vector<std::string>>& piles) {
        std::vector<uint32_t> wfs{0}, len, bases;
        std::vector<uint64_t> off;
        for (const auto& pile : piles) {
            for (const auto& s : pile) {
                const uint64_t words = (s.size() + 15) / 16;
                off.push_back(bases.size());
                len.push_back((uint32_t)s.size());
                bases.resize(bases.size() + words);
                if (words && cw_pack_sequence(s.data(), (uint32_t)s.size(), bases.data() + off.back(), words) < 0)
                    throw std::runtime_error("consent_amd: cw_pack_sequence failed");
            }
            wfs.push_back((uint32_t)len.size());
        }
        bases.push_back(0);
        const uint32_t W = (uint32_t)piles.size();
        std::vector<uint64_t> coff(W + 1, 0), soff(W + 1, 0);
        for (uint32_t w = 0; w < W; ++w) {
            uint64_t tpl = piles[w].empty() ? 0 : piles[w][0].size(), tot = 0;
            for (const auto& s : piles[w]) tot += s.size();
            coff[w + 1] = coff[w] + 3 * tpl + 256;
            soff[w + 1] = soff[w] + tot / (solid_ ? solid_ : 1) + 16;
        }
        std::vector<char> cons(coff[W] + 1);
        std::vector<uint32_t> clen(W), solid(soff[W] + 1), slen(W);
        std::vector<uint8_t> st(W);
        std::vector<WindowConsensus> out(W);
        if (W == 0) return out;
        cw_batch b{W, (uint32_t)len.size(), (uint64_t)bases.size() - 1, wfs.data(), len.data(), off.data(), bases.data()};
        cw_result r{cons.data(), coff.data(), clen.data(), st.data(), solid.data(), soff.data(), slen.data()};
        const int rc = cw_run(eng_, &b, &r);
        if (rc != CW_OK && rc != CW_E_CAPACITY) throw std::runtime_error(std::string("consent_amd: cw_run: ") + cw_strerror(rc));
        for (uint32_t w = 0; w < W; ++w) {
            out[w].overflow = st[w] == CW_WIN_OVERFLOW;
            out[w].template_fallback = st[w] == CW_WIN_TEMPLATE;
            if (out[w].overflow) continue;
            out[w].consensus.assign(cons.data() + coff[w], clen[w]);
            out[w].solid_kmers.assign(solid.begin() + soff[w], solid.begin() + soff[w] + slen[w]);
        }
        return out;
    }

  private:
    cw_engine* eng_ = nullptr;
    unsigned solid_;
};

} // namespace consent_amd
#endif
