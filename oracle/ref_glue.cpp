/*
 * ref_glue.cpp -- extern "C" doorway into the reference's OWN translation units, compiled where they lie
 * under /root/reference/src (alignmentWindows.cpp, alignmentPiles.cpp, utils.cpp, reverseComplement.cpp --
 * the four TUs of the path that build with nothing but the files present).  This file contains no
 * reference code: it only converts plain arrays to the reference's argument types and back.
 * TEST INFRASTRUCTURE: builds into oracle/_ref/libconsent_ref.so (git-ignored), used by tests only.
 *
 * NOT built this way, and why: DBG.cpp, correctionDBG.cpp, correctionMSA.cpp, correctionAlignment.cpp and both
 * drivers include "../BMEAN/..." or "../CTPL/..." headers that are absent (un-vendored submodules);
 * building them would need invented stand-ins, so they are treated as unbuildable.
 */
#include "alignmentPiles.h"
#include "alignmentWindows.h"
#include "utils.h"

#include <cstring>
#include <fstream>

static std::vector<Overlap> to_overlaps(const uint32_t* ovl, uint32_t n) {
    std::vector<Overlap> v(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t* o = ovl + 8 * i;
        v[i].qName = "q";
        v[i].qLength = o[0]; v[i].qStart = o[1]; v[i].qEnd = o[2]; v[i].strand = o[3] != 0;
        v[i].tLength = o[4]; v[i].tStart = o[5]; v[i].tEnd = o[6];
        v[i].tName = "t" + std::to_string(o[7]);
        v[i].resMatches = 0; v[i].alBlockLen = 0; v[i].mapQual = 0;
    }
    return v;
}

extern "C" {

int ref_window_positions(uint32_t tpl_len, const uint32_t* ovl, uint32_t n_ovl, uint32_t min_support, uint32_t window_size,
                         int32_t window_overlap, uint32_t* out_pairs, uint32_t cap_pairs) {
    std::vector<Overlap> v = to_overlaps(ovl, n_ovl);
    auto w = getAlignmentWindowsPositions(tpl_len, v, min_support, 0, window_size, window_overlap);
    if (w.size() > cap_pairs) return -4;
    for (size_t i = 0; i < w.size(); ++i) { out_pairs[2 * i] = w[i].first; out_pairs[2 * i + 1] = w[i].second; }
    return (int)w.size();
}

int ref_window_pile(const uint32_t* ovl, uint32_t n_ovl, const char* tpl, uint32_t tpl_len, const char* const* targets,
                    const uint32_t* target_lens, uint32_t n_targets, uint32_t q_beg, uint32_t q_end, uint32_t k, char* out,
                    uint64_t cap, uint32_t* lens_out, uint32_t cap_members) {
    std::vector<Overlap> v = to_overlaps(ovl, n_ovl);
    robin_hood::unordered_map<std::string, std::string> seqs;
    seqs["q"] = std::string(tpl, tpl_len);
    for (uint32_t i = 0; i < n_targets; ++i) seqs["t" + std::to_string(i)] = std::string(targets[i], target_lens[i]);
    auto pile = getAlignmentWindowsSequences(v, 0, q_end - q_beg + 1, 0, seqs, q_beg, q_end, k, 0, 0);
    if (pile.size() > cap_members) return -4;
    uint64_t off = 0;
    for (size_t i = 0; i < pile.size(); ++i) {
        if (off + pile[i].size() > cap) return -4;
        memcpy(out + off, pile[i].data(), pile[i].size());
        off += pile[i].size();
        lens_out[i] = (uint32_t)pile[i].size();
    }
    return (int)pile.size();
}

/* Reads every pile of a PAF file through getNextReadPile; writes, per kept overlap, its 0-based line number
 * within its pile's block (so tests see the reference's sort/truncate order).  Returns overlaps written. */
int ref_paf_pile_order(const char* paf_path, uint32_t max_support, uint32_t* pile_id, uint32_t* res_matches,
                       char* tnames, uint32_t tname_stride, uint32_t cap) {
    std::ifstream f(paf_path);
    uint32_t n = 0, pile = 0;
    std::vector<Overlap> cur = getNextReadPile(f, max_support);
    while (!cur.empty()) {
        for (auto& o : cur) {
            if (n >= cap) return -4;
            pile_id[n] = pile;
            res_matches[n] = o.resMatches;
            strncpy(tnames + (size_t)n * tname_stride, o.tName.c_str(), tname_stride - 1);
            tnames[(size_t)n * tname_stride + tname_stride - 1] = 0;
            n++;
        }
        pile++;
        cur = getNextReadPile(f, max_support);
    }
    return (int)n;
}

/* indexReads (utils.cpp:166-205) on a file; for the read called `name`: its length and decoded bases (fullnum2str).
 * Returns the number of reads in the index, or -1 when `name` is not in it. */
int ref_index_reads_lookup(const char* path, const char* name, char* out, uint32_t cap, uint32_t* len) {
    robin_hood::unordered_map<std::string, std::vector<bool>> index;
    indexReads(index, path);
    auto it = index.find(name);
    if (it == index.end()) return -1;
    std::string s = fullnum2str(it->second);
    *len = (uint32_t)s.size();
    if (s.size() > cap) return -4;
    memcpy(out, s.data(), s.size());
    return (int)index.size();
}

/* Every pile of a PAF file through getNextReadPile: per kept overlap the parsed fields (qStart qEnd tStart tEnd strand resMatches
 * qLength tLength), 8 words each, and the pile index; names go to qnames/tnames.  Returns overlaps written. */
int ref_paf_piles(const char* paf_path, uint32_t max_support, uint32_t* pile_id, uint32_t* fields8, char* qnames, char* tnames,
                  uint32_t name_stride, uint32_t cap) {
    std::ifstream f(paf_path);
    uint32_t n = 0, pile = 0;
    std::vector<Overlap> cur = getNextReadPile(f, max_support);
    while (!cur.empty() || !f.eof()) {
        for (auto& o : cur) {
            if (n >= cap) return -4;
            pile_id[n] = pile;
            uint32_t* w = fields8 + 8 * (size_t)n;
            w[0] = o.qStart; w[1] = o.qEnd; w[2] = o.tStart; w[3] = o.tEnd; w[4] = o.strand ? 1u : 0u; w[5] = o.resMatches;
            w[6] = o.qLength; w[7] = o.tLength;
            strncpy(qnames + (size_t)n * name_stride, o.qName.c_str(), name_stride - 1);
            qnames[(size_t)n * name_stride + name_stride - 1] = 0;
            strncpy(tnames + (size_t)n * name_stride, o.tName.c_str(), name_stride - 1);
            tnames[(size_t)n * name_stride + name_stride - 1] = 0;
            n++;
        }
        if (!cur.empty()) pile++;
        if (f.eof()) break;
        cur = getNextReadPile(f, max_support);   /* as the driver does: ask again after an empty pile (CONSENT-correction.cpp:88-91) */
    }
    return (int)n;
}

int ref_revcomp(const char* s, uint32_t len, char* out) {
    std::string r = rev_comp::run(std::string(s, len));
    memcpy(out, r.data(), r.size());
    return (int)r.size();
}

/* 2-bit round trip of utils.cpp:21-54 */
int ref_pack_unpack(const char* s, uint32_t len, char* out) {
    std::string r = fullnum2str(fullstr2num(std::string(s, len)));
    memcpy(out, r.data(), r.size());
    return (int)r.size();
}

int ref_trim_read(const char* s, uint32_t len, uint32_t mer, char* out) {
    std::string r = trimRead(std::string(s, len), mer);
    memcpy(out, r.data(), r.size());
    return (int)r.size();
}

int ref_drop_read(const char* s, uint32_t len) { return dropRead(std::string(s, len)) ? 1 : 0; }

} // extern "C"
