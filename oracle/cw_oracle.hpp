/*
 * cw_oracle.hpp -- CPU restatement of CONSENT's per-window consensus path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link, load or execute anything under oracle/.  The product (consent_amd/) never does.
 *
 * Parity status, row by row:
 *   A1/A2  window positions + pile extraction  : restated from alignmentWindows.cpp; PINNED against the
 *          reference's own alignmentWindows.cpp compiled unmodified into oracle/_ref (tests/test_oracle_ref.py).
 *   A3     computeConsensus* glue              : restated from correctionMSA.cpp:29-71.
 *   A4     MSABMAAC (BMEAN + spoa)             : sources absent from /root/reference (un-vendored submodule
 *          Malfoy/BMEAN, pin unknown) -> restatement of the published algorithm, policies in cw_policy.h.
 *          PARITY UNPINNED.
 *   A5-A10 weightConsensus / polishCorrection / DBG : restated statement-by-statement from
 *          correctionMSA.cpp:6-27, correctionDBG.cpp, DBG.cpp.  Those TUs include "../BMEAN/utils.h"
 *          (absent) so they cannot be compiled here without inventing that header; the reference has no
 *          tests or golden vectors.  PARITY UNPINNED (restated from visible source).
 */
#ifndef CW_ORACLE_HPP
#define CW_ORACLE_HPP

#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace cwo {

typedef uint64_t kmer_t;
typedef std::unordered_map<kmer_t, uint32_t> KmerCounts;

struct Params {
    unsigned k;            /* merSize      (main.cpp:20, -k)  */
    unsigned solid;        /* solidThresh  (main.cpp:23, -f)  */
    unsigned common_kmers; /* commonKMers  (main.cpp:21, -c)  */
    unsigned min_anchors;  /* minAnchors   (main.cpp:22, -A)  */
    unsigned max_msa;      /* maxMSA       (main.cpp:25, -M)  */
};

/* Work counters the oracle can emit beside a result (SURVEY 8d "secondary algorithmic work"). */
struct Stats {
    uint64_t kmers = 0, tpl_anchors = 0, chain_len = 0, pair_tests = 0;
    uint64_t segments = 0, poa_segments = 0, alignments = 0, dp_cells = 0, max_nodes = 0, max_seg_len = 0;
    uint64_t alignments_routed = 0, dp_cells_routed = 0; /* ... of the segments the engine sends to a POA tier (not: a single piece; equal pieces no longer than k) */
    uint64_t link_calls = 0, nbr_calls = 0;
};

/* A11: 2-bit codes, MSB-first, A0 C1 G2 T3 (pinned by DBG.cpp:5-16 vs :30-32 and utils.cpp:21-32). */
kmer_t str2num(const std::string& s);
std::string kmer2str(kmer_t v, unsigned k);
std::string revcomp(const std::string& s);            /* reverseComplement.cpp:6-24,33-40 */

/* A4: returns true and the concatenated segment consensus, or false when fewer than min_anchors chained. */
bool segmented_poa(const std::vector<std::string>& pile, unsigned k, double anchor_support, unsigned min_anchors,
                   unsigned max_msa, std::string& consensus, KmerCounts& counts, Stats* st);

/* A4d alone (one segment): POA over `seqs` in order + the column-majority consensus of cw_policy.h (CW_POA_CONSENSUS). */
std::string poa_consensus(const std::vector<std::string>& seqs, Stats* st);

/* A5 */ std::string weight_consensus(std::string cons, const KmerCounts& counts, unsigned k, unsigned solid);
/* A8 */ std::vector<std::string> neighbours(std::string kmer, unsigned k, int left, const KmerCounts& counts, unsigned solid);
/* A9 */ unsigned extend_left(const KmerCounts& counts, unsigned k, unsigned ext_len, std::string& lr, unsigned solid);
/* A9 */ unsigned extend_right(const KmerCounts& counts, unsigned k, unsigned ext_len, std::string& lr, unsigned solid);
/* A6 */ std::string polish(std::string read, const KmerCounts& counts, unsigned k, unsigned solid, Stats* st);
/* A6a */ int next_src(const std::string& s, unsigned beg, unsigned m);
/* A6a */ int next_dst(const std::string& s, unsigned beg, unsigned m);

/* A3: status 0 = consensus, 1 = template fallback (MSA empty). */
struct WindowResult {
    std::string consensus;
    int status;
    KmerCounts counts;
};
WindowResult window_consensus(const std::vector<std::string>& pile, const Params& p, Stats* st);

/* A1/A2 (host feeder).  Overlap fields as Overlap.h:8-20 after parsing (ends inclusive). */
struct Ovl {
    uint32_t q_len, q_start, q_end;
    int strand; /* 0 '+', 1 '-' */
    uint32_t t_len, t_start, t_end;
    uint32_t t_id; /* index into the sequence list handed to window_pile */
};
std::vector<std::pair<uint32_t, uint32_t>> window_positions(uint32_t tpl_len, const std::vector<Ovl>& ovl,
                                                            unsigned min_support, unsigned window_size, int window_overlap);
std::vector<std::string> window_pile(const std::vector<Ovl>& ovl, const std::string& tpl,
                                     const std::vector<std::string>& targets, uint32_t q_beg, uint32_t q_end, unsigned k);

/* ---- read re-assembly -------------------------------------------------------------------------------- */
/* Result of the local alignment the reference obtains from StripedSmithWaterman::Aligner::Align (policy: cw_policy.h). */
struct SwResult {
    int score;
    int ref_begin, ref_end, query_begin, query_end; /* inclusive, 0-based */
    unsigned ins, del;                              /* inserted / deleted bases of the banded traceback */
};
SwResult ssw_align(const std::string& query, const std::string& ref);

/* alignConsensus (correctionAlignment.cpp:47-140); solid[i] = ascending solid k-mers of window i (what merCounts[i] is read for). */
std::string align_consensus(const std::string& sequence, const std::vector<std::string>& consensuses,
                            const std::vector<std::vector<uint32_t>>& solid, const std::vector<std::pair<uint32_t, uint32_t>>& piles_pos,
                            const std::vector<std::string>& templates, int start_pos, unsigned window_size, unsigned window_overlap,
                            unsigned mer_size);
/* trimRead / dropRead (utils.cpp:96-128, :71-73) -- pinned against oracle/_ref */
std::string trim_read(const std::string& corrected, unsigned mer_size);
bool drop_read(const std::string& corrected);

} // namespace cwo
#endif
