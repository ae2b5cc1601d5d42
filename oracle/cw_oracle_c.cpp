/*
 * cw_oracle_c.cpp -- C entry points of the CPU oracle (liboracle.so), shaped like the product's C ABI
 * (include/consent_amd.h) so that tests can hand both the same buffers.
 * TEST INFRASTRUCTURE: loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#include "../include/consent_amd.h"
#include "cw_oracle.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

using namespace cwo;

static std::string unpack(const cw_batch* b, uint32_t s) {
    std::string out(b->seq_len[s], 'A');
    const uint32_t* w = b->bases + b->seq_word_off[s];
    for (uint32_t j = 0; j < b->seq_len[s]; ++j) out[j] = "ACGT"[(w[j >> 4] >> (30 - 2 * (j & 15))) & 3];
    return out;
}

static void add_stats(uint64_t* dst, const Stats& s) {
    const uint64_t v[14] = {s.kmers, s.tpl_anchors, s.chain_len, s.pair_tests, s.segments, s.poa_segments, s.alignments,
                            s.dp_cells, s.max_nodes, s.max_seg_len, s.link_calls, s.nbr_calls, s.alignments_routed, s.dp_cells_routed};
    for (int i = 0; i < 14; ++i) {
        if (i == 8 || i == 9) dst[i] = std::max(dst[i], v[i]);
        else dst[i] += v[i];
    }
}

extern "C" {

/* Same contract as cw_run (host buffers).  stats (nullable) receives 14 counters, see add_stats. */
int cwo_run(const cw_params* p, const cw_batch* b, const cw_result* r, uint64_t* stats, int n_threads) {
    if (!p || !b || !r) return CW_E_INVALID;
    Params prm{p->k, p->solid, p->common_kmers, p->min_anchors, p->max_msa};
    std::atomic<uint32_t> next(0);
    std::atomic<int> rc(CW_OK);
    if (n_threads < 1) n_threads = 1;
    std::vector<Stats> tstats(n_threads);
    auto worker = [&](int tid) {
        for (;;) {
            uint32_t w = next.fetch_add(1);
            if (w >= b->n_windows) break;
            std::vector<std::string> pile;
            for (uint32_t s = b->win_first_seq[w]; s < b->win_first_seq[w + 1]; ++s) pile.push_back(unpack(b, s));
            WindowResult res = window_consensus(pile, prm, &tstats[tid]);
            uint64_t cap = r->cons_off[w + 1] - r->cons_off[w];
            bool over = res.consensus.size() > cap;
            std::vector<uint32_t> solid;
            if (r->solid) {
                for (auto& kv : res.counts)
                    if (kv.second >= prm.solid) solid.push_back((uint32_t)kv.first);
                std::sort(solid.begin(), solid.end());
                if (solid.size() > r->solid_off[w + 1] - r->solid_off[w]) over = true;
            }
            if (over) {
                r->win_status[w] = CW_WIN_OVERFLOW;
                r->cons_len[w] = 0;
                if (r->solid) r->solid_len[w] = 0;
                rc = CW_E_CAPACITY;
                continue;
            }
            if (!res.consensus.empty()) memcpy(r->cons + r->cons_off[w], res.consensus.data(), res.consensus.size());
            r->cons_len[w] = (uint32_t)res.consensus.size();
            r->win_status[w] = (uint8_t)res.status;
            if (r->solid) {
                if (!solid.empty()) memcpy(r->solid + r->solid_off[w], solid.data(), solid.size() * 4); /* (an empty vector's data() may be null: UB for memcpy, found by `make asan`) */
                r->solid_len[w] = (uint32_t)solid.size();
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (stats) {
        memset(stats, 0, 14 * sizeof(uint64_t));
        for (auto& s : tstats) add_stats(stats, s);
    }
    return rc;
}

/* One segment's POA consensus over ASCII strings (A4d). */
int cwo_poa(const char* const* seqs, const uint32_t* lens, uint32_t n, char* out, uint32_t cap, uint32_t* out_len) {
    std::vector<std::string> v;
    for (uint32_t i = 0; i < n; ++i) v.emplace_back(seqs[i], lens[i]);
    std::string c = poa_consensus(v, nullptr);
    if (c.size() > cap) return CW_E_CAPACITY;
    memcpy(out, c.data(), c.size());
    *out_len = (uint32_t)c.size();
    return CW_OK;
}

/* weightConsensus (+ optionally polishCorrection) on an ASCII consensus with explicit counts (A5-A10). */
int cwo_weight_polish(const char* cons, uint32_t len, const uint64_t* keys, const uint32_t* cnts, uint32_t n_keys, uint32_t k,
                      uint32_t solid, int do_weight, int do_polish, char* out, uint32_t cap, uint32_t* out_len) {
    KmerCounts m;
    for (uint32_t i = 0; i < n_keys; ++i) m[keys[i]] = cnts[i];
    std::string s(cons, len);
    if (do_weight) s = weight_consensus(s, m, k, solid);
    if (do_polish) s = polish(s, m, k, solid, nullptr);
    if (s.size() > cap) return CW_E_CAPACITY;
    memcpy(out, s.data(), s.size());
    *out_len = (uint32_t)s.size();
    return CW_OK;
}

/* A1: returns the number of windows written as (beg,end) pairs, or negative on overflow.
 * ovl rows: q_len q_start q_end strand t_len t_start t_end t_id (8 x u32), ends inclusive. */
int cwo_window_positions(uint32_t tpl_len, const uint32_t* ovl, uint32_t n_ovl, uint32_t min_support, uint32_t window_size,
                         int32_t window_overlap, uint32_t* out_pairs, uint32_t cap_pairs) {
    std::vector<Ovl> v(n_ovl);
    for (uint32_t i = 0; i < n_ovl; ++i) {
        const uint32_t* o = ovl + 8 * i;
        v[i] = Ovl{o[0], o[1], o[2], (int)o[3], o[4], o[5], o[6], o[7]};
    }
    auto w = window_positions(tpl_len, v, min_support, window_size, window_overlap);
    if (w.size() > cap_pairs) return CW_E_CAPACITY;
    for (size_t i = 0; i < w.size(); ++i) { out_pairs[2 * i] = w[i].first; out_pairs[2 * i + 1] = w[i].second; }
    return (int)w.size();
}

/* A2: pile of one window as concatenated ASCII; lens_out[i] per member.  Returns member count. */
int cwo_window_pile(const uint32_t* ovl, uint32_t n_ovl, const char* tpl, uint32_t tpl_len, const char* const* targets,
                    const uint32_t* target_lens, uint32_t n_targets, uint32_t q_beg, uint32_t q_end, uint32_t k, char* out,
                    uint64_t cap, uint32_t* lens_out, uint32_t cap_members) {
    std::vector<Ovl> v(n_ovl);
    for (uint32_t i = 0; i < n_ovl; ++i) {
        const uint32_t* o = ovl + 8 * i;
        v[i] = Ovl{o[0], o[1], o[2], (int)o[3], o[4], o[5], o[6], o[7]};
    }
    std::vector<std::string> tg;
    for (uint32_t i = 0; i < n_targets; ++i) tg.emplace_back(targets[i], target_lens[i]);
    auto pile = window_pile(v, std::string(tpl, tpl_len), tg, q_beg, q_end, k);
    if (pile.size() > cap_members) return CW_E_CAPACITY;
    uint64_t off = 0;
    for (size_t i = 0; i < pile.size(); ++i) {
        if (off + pile[i].size() > cap) return CW_E_CAPACITY;
        memcpy(out + off, pile[i].data(), pile[i].size());
        off += pile[i].size();
        lens_out[i] = (uint32_t)pile[i].size();
    }
    return (int)pile.size();
}

} // extern "C"

/* ---- SURVEY 8f-1 entry points ---------------------------------------------------------------------------------------- */
extern "C" {

/* ssw_align restatement: out7 = score, ref_begin, ref_end, query_begin, query_end, ins, del */
int cwo_ssw(const char* query, uint32_t qlen, const char* ref, uint32_t rlen, int32_t* out7) {
    SwResult r = ssw_align(std::string(query, qlen), std::string(ref, rlen));
    out7[0] = r.score; out7[1] = r.ref_begin; out7[2] = r.ref_end; out7[3] = r.query_begin; out7[4] = r.query_end;
    out7[5] = (int32_t)r.ins; out7[6] = (int32_t)r.del;
    return CW_OK;
}

/* alignConsensus + trimRead(…,1) + dropRead for one read (CONSENT-correction.cpp:47-56).
 * cons: concatenated consensus strings, cons_len[n]; templates likewise; solid: concatenated ascending k-mers, solid_len[n];
 * pos: n (beg,end) pairs.  out receives the final read ("" when dropped); *stitched_len / stitched (nullable) the string
 * before trimming. */
int cwo_stitch(const char* seq, uint32_t seq_len, uint32_t n, const char* cons, const uint32_t* cons_len, const char* tpls,
               const uint32_t* tpl_len, const uint32_t* solid, const uint32_t* solid_len, const uint32_t* pos, uint32_t window_size,
               uint32_t window_overlap, uint32_t mer_size, int do_trim, char* out, uint32_t cap, uint32_t* out_len, char* stitched,
               uint32_t* stitched_len) {
    std::vector<std::string> cs(n), ts(n);
    std::vector<std::vector<uint32_t>> sl(n);
    std::vector<std::pair<uint32_t, uint32_t>> pp(n);
    size_t co = 0, to = 0, so = 0;
    for (uint32_t i = 0; i < n; ++i) {
        cs[i].assign(cons + co, cons_len[i]); co += cons_len[i];
        ts[i].assign(tpls + to, tpl_len[i]); to += tpl_len[i];
        sl[i].assign(solid + so, solid + so + solid_len[i]); so += solid_len[i];
        pp[i] = {pos[2 * i], pos[2 * i + 1]};
    }
    std::string r = align_consensus(std::string(seq, seq_len), cs, sl, pp, ts, n ? (int)pp[0].first : 0, window_size, window_overlap, mer_size);
    if (stitched && stitched_len) {
        if (r.size() > cap) return CW_E_CAPACITY;
        memcpy(stitched, r.data(), r.size());
        *stitched_len = (uint32_t)r.size();
    }
    if (do_trim) {
        r = trim_read(r, 1);
        if (!r.empty() && drop_read(r)) r.clear();
    }
    if (r.size() > cap) return CW_E_CAPACITY;
    memcpy(out, r.data(), r.size());
    *out_len = (uint32_t)r.size();
    return CW_OK;
}

int cwo_trim_read(const char* s, uint32_t len, uint32_t mer, char* out) {
    std::string r = trim_read(std::string(s, len), mer);
    memcpy(out, r.data(), r.size());
    return (int)r.size();
}
int cwo_drop_read(const char* s, uint32_t len) { return drop_read(std::string(s, len)) ? 1 : 0; }

} // extern "C"
