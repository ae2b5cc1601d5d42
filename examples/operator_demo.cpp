// Minimal C++ caller of the engine through the adapter: one synthetic window, prints the consensus.
// Build: g++ -std=c++17 -Iinclude examples/operator_demo.cpp -Lconsent_amd -lconsent_amd -Wl,-rpath,$PWD/consent_amd -o /tmp/operator_demo
#include <cstdio>
#include <random>

#include "consent_amd_adapter.hpp"

int main() {
    std::mt19937 rng(7);
    std::string truth;
    for (int i = 0; i < 300; ++i) truth.push_back("ACGT"[rng() & 3]);
    auto noisy = [&](double rate) {
        std::string s;
        for (char c : truth) {
            double x = (rng() & 0xFFFF) / 65536.0;
            if (x < rate * 0.3) continue;
            if (x < rate * 0.6) s.push_back("ACGT"[rng() & 3]);
            s.push_back(x < rate ? "ACGT"[rng() & 3] : c);
        }
        return s;
    };
    std::vector<std::vector<std::string>> piles(1);
    for (int i = 0; i < 25; ++i) piles[0].push_back(noisy(0.12));
    try {
        consent_amd::Engine eng(/*merSize*/ 9, /*solidThresh*/ 4, /*commonKMers*/ 8, /*minAnchors*/ 2, /*maxMSA*/ 150);
        auto res = eng.computeConsensus(piles);
        std::printf("status: %s\nconsensus (%zu): %s\nsolid k-mers: %zu\n", res[0].template_fallback ? "template" : "consensus",
                    res[0].consensus.size(), res[0].consensus.c_str(), res[0].solid_kmers.size());
        size_t same = 0;
        for (size_t i = 0; i < std::min(truth.size(), res[0].consensus.size()); ++i) same += std::toupper(res[0].consensus[i]) == truth[i];
        std::printf("positions equal to the truth (no realignment): %zu / %zu\n", same, truth.size());
    } catch (const std::exception& e) {
        std::printf("engine unavailable: %s\n", e.what());
        return 2;
    }
    return 0;
}
