/*
 * cw_oracle.cpp -- CPU restatement of CONSENT's per-window consensus path (see cw_oracle.hpp for
 * the parity status of every row).  TEST INFRASTRUCTURE: never linked into the product.
 *
 * Written for clarity, not speed: strings and node-based containers, one window at a time.
 */
#include "cw_oracle.hpp"
#include "../include/cw_policy.h"

#include <algorithm>
#if defined(CWO_SIMD) && defined(__AVX2__)
#include <immintrin.h>
#endif
#include <cassert>
#include <climits>
#include <set>

namespace cwo {

/* ------------------------------------------------------------------------------------------------
 * A11 -- alphabet helpers
 * ---------------------------------------------------------------------------------------------- */
static inline unsigned base_code(char c) {
    switch (c) { /* utils.cpp:24-28: everything that is not A/C/G becomes T */
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    default: return 3;
    }
}

kmer_t str2num(const std::string& s) {
    kmer_t v = 0;
    for (char c : s) v = (v << 2) | base_code(c);
    return v;
}

std::string kmer2str(kmer_t v, unsigned k) {
    std::string s(k, 'A');
    for (unsigned i = 0; i < k; ++i) {
        s[k - 1 - i] = "ACGT"[v & 3];
        v >>= 2;
    }
    return s;
}

std::string revcomp(const std::string& s) {
    /* reverseComplement.cpp:33-40 fills ACGT/acgt only; the static table is zero elsewhere. */
    std::string r(s.size(), '\0');
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i], o = '\0';
        switch (c) {
        case 'A': o = 'T'; break; case 'T': o = 'A'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break;
        case 'a': o = 't'; break; case 't': o = 'a'; break; case 'c': o = 'g'; break; case 'g': o = 'c'; break;
        default: break;
        }
        r[i] = o;
    }
    return r;
}

static inline bool is_upper(char c) { return 'A' <= c && c <= 'Z'; } /* utils.cpp:56-58 */
static inline uint32_t count_of(const KmerCounts& m, kmer_t key) {
    auto it = m.find(key);
    return it == m.end() ? 0u : it->second;
}
static std::string upper_copy(std::string s) {
    for (char& c : s)
        if ('a' <= c && c <= 'z') c = (char)(c - 'a' + 'A');
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * A4d -- partial-order alignment of one segment pile (POA, Lee 2002; alignment engine and graph update as
 *        in spoa's scalar engine; rank order and consensus rule per cw_policy.h)
 * ---------------------------------------------------------------------------------------------- */
namespace {

struct PoaEdge { int from, to; int weight; /* sequences whose path uses the edge (heaviest-bundle consensus) */ };
struct PoaNode {
    char base;
    int coverage;                         /* sequences whose path runs through this node */
    std::vector<int> in_edges, out_edges; /* indices into edges, insertion order */
    std::vector<int> aligned;             /* other nodes of the same MSA column   */
};

struct PoaGraph {
    std::vector<PoaNode> nodes;
    std::vector<PoaEdge> edges;
    std::vector<int> rank2node, node2rank;
    int n_sequences = 0;
    int template_nodes = 0; /* nodes 0 .. template_nodes-1 are the first sequence's chain */

    int add_node(char b) {
        nodes.push_back(PoaNode{b, 1, {}, {}, {}});
        return (int)nodes.size() - 1;
    }
    void add_edge(int from, int to) {
        for (int e : nodes[from].out_edges)
            if (edges[e].to == to) { edges[e].weight++; return; }
        edges.push_back(PoaEdge{from, to, 1});
        nodes[from].out_edges.push_back((int)edges.size() - 1);
        nodes[to].in_edges.push_back((int)edges.size() - 1);
    }
    /* Rank order policy (cw_policy.h "rank order"): the order is maintained incrementally, never re-sorted.
       Columns (a node + its aligned nodes) always occupy consecutive ranks.
       place_after_column(v, p): v joins p's column -> goes right after the last member of that column.
       place_before_column(v, q): v is an insertion whose next ranked path node is q -> goes right before the
       first member of q's column; q == -1 -> appended at the end. */
    int column_first(int v) const {
        int r = node2rank[v];
        for (int a : nodes[v].aligned) r = std::min(r, node2rank[a]);
        return r;
    }
    int column_last(int v) const {
        int r = node2rank[v];
        for (int a : nodes[v].aligned) r = std::max(r, node2rank[a]);
        return r;
    }
    void insert_rank(int v, int at) {
        rank2node.insert(rank2node.begin() + at, v);
        node2rank.resize(nodes.size(), -1);
        for (int r = at; r < (int)rank2node.size(); ++r) node2rank[rank2node[r]] = r;
    }
    void place_before_column(int v, int q) { insert_rank(v, q == -1 ? (int)rank2node.size() : column_first(q)); }
    void place_after_column(int v, int p) {
        /* called after v was linked into p's column: exclude v itself when looking for the block end */
        int r = node2rank[p];
        for (int a : nodes[p].aligned) if (a != v) r = std::max(r, node2rank[a]);
        insert_rank(v, r + 1);
    }
    bool order_is_valid() const {
        for (const PoaEdge& e : edges) if (node2rank[e.from] >= node2rank[e.to]) return false;
        for (int v = 0; v < (int)nodes.size(); ++v)
            if (column_last(v) - column_first(v) != (int)nodes[v].aligned.size()) return false;
        return true;
    }

    /* Global alignment, linear gaps.  Returns (node or -1, seq index or -1) pairs in order. */
    std::vector<std::pair<int, int>> align(const std::string& seq, Stats* st) const {
        std::vector<std::pair<int, int>> path;
        const int n = (int)nodes.size(), L = (int)seq.size();
        if (n == 0 || L == 0) return path;
        const int cols = L + 1;
        const int g = CW_POA_GAP;
        auto pred_row = [&](const PoaNode& nd, size_t p) { return node2rank[edges[nd.in_edges[p]].from] + 1; };
#if CW_POA_AFFINE
        /* Affine gaps (cw_policy.h CW_POA_GAP_MODEL_AFFINE): three layers, a walk back that keeps its layer. */
        {
            (void)g;
            const int32_t go = CW_POA_GAP_OPEN, ge = CW_POA_GAP_EXT, NEG = -(1 << 28);
            static thread_local std::vector<int32_t> Hh, Ff, Ee;
            const size_t cells = (size_t)(n + 1) * cols;
            if (Hh.size() < cells) { Hh.resize(cells); Ff.resize(cells); Ee.resize(cells); }
            auto h = [&](int i, int j) -> int32_t& { return Hh[(size_t)i * cols + j]; };
            auto f = [&](int i, int j) -> int32_t& { return Ff[(size_t)i * cols + j]; };
            auto e = [&](int i, int j) -> int32_t& { return Ee[(size_t)i * cols + j]; };
            h(0, 0) = 0; f(0, 0) = NEG; e(0, 0) = NEG;
            for (int j = 1; j < cols; ++j) { e(0, j) = go + (j - 1) * ge; h(0, j) = e(0, j); f(0, j) = NEG; }
            for (int i = 1; i <= n; ++i) {
                const PoaNode& nd = nodes[rank2node[i - 1]];
                const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
                for (int j = 0; j < cols; ++j) {
                    int32_t fv = NEG, dv = NEG;
                    for (size_t p = 0; p < np; ++p) {
                        const int pi = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                        fv = std::max(fv, std::max(h(pi, j) + go, f(pi, j) + ge));
                        if (j >= 1) dv = std::max(dv, h(pi, j - 1) + ((seq[j - 1] == nd.base) ? CW_POA_MATCH : CW_POA_MISMATCH));
                    }
                    f(i, j) = fv;
                    if (j == 0) { e(i, 0) = NEG; h(i, 0) = fv; }
                    else {
                        e(i, j) = std::max(h(i, j - 1) + go, e(i, j - 1) + ge);
                        h(i, j) = std::max(std::max(dv, fv), e(i, j));
                    }
                }
            }
            if (st) { st->dp_cells += (uint64_t)n * L; st->alignments++; }
            int bi = -1;
            int32_t bs = INT_MIN;
            for (int i = 1; i <= n; ++i) {
                if (!nodes[rank2node[i - 1]].out_edges.empty()) continue;
                if (bi == -1 || bs < h(i, L)) { bs = h(i, L); bi = i; }
            }
            int i = bi, j = L, layer = 0; /* 0 H, 1 F, 2 E */
            while (!(i == 0 && j == 0 && layer == 0)) {
                if (layer == 0) {
                    bool found = false;
                    if (i != 0 && j != 0) {
                        const PoaNode& nd = nodes[rank2node[i - 1]];
                        const int32_t sc = (seq[j - 1] == nd.base) ? CW_POA_MATCH : CW_POA_MISMATCH;
                        const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
                        for (size_t p = 0; p < np && !found; ++p) {
                            const int r = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                            if (h(i, j) == h(r, j - 1) + sc) { path.emplace_back(rank2node[i - 1], j - 1); i = r; j = j - 1; found = true; }
                        }
                    }
                    if (!found) {
                        if (i != 0 && h(i, j) == f(i, j)) layer = 1;
                        else { assert(j != 0 && h(i, j) == e(i, j)); layer = 2; }
                    }
                } else if (layer == 1) {
                    const PoaNode& nd = nodes[rank2node[i - 1]];
                    const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
                    bool found = false;
                    for (size_t p = 0; p < np && !found; ++p) {
                        const int r = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                        if (f(i, j) == h(r, j) + go) { path.emplace_back(rank2node[i - 1], -1); i = r; layer = 0; found = true; }
                    }
                    for (size_t p = 0; p < np && !found; ++p) {
                        const int r = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                        if (f(i, j) == f(r, j) + ge) { path.emplace_back(rank2node[i - 1], -1); i = r; found = true; }
                    }
                    assert(found);
                } else {
                    path.emplace_back(-1, j - 1);
                    if (e(i, j) == h(i, j - 1) + go) layer = 0;
                    else assert(e(i, j) == e(i, j - 1) + ge);
                    j = j - 1;
                }
            }
            std::reverse(path.begin(), path.end());
            return path;
        }
#endif
#if defined(CWO_SIMD) && defined(__AVX2__) && CW_POA_MODE != CW_POA_MODE_SW && !CW_POA_AFFINE
        /* Row-vectorised fill for the CPU BASELINE leg of bench.py only (the real reference's POA, spoa, is SIMD code, so a
           scalar port flatters the GPU/CPU ratio).  Eight int32 columns per AVX2 register; rows are kept as W[i][j] = H[i][j] - j * gap,
           in which a horizontal move costs nothing: the horizontal recurrence is a prefix max (three shift-and-max steps inside a
           register, a carry between registers), the candidates of a row are plain vector loads of the predecessor rows at j and j - 1.
           Exact integer arithmetic: every H value, hence the traceback and the consensus, is identical to the scalar code's
           (tests/test_oracle_units.py builds both and compares). */
        const int colsP = (cols + 7) & ~7, stride = colsP + 8; /* 8 cells of padding in front of column 0: the diagonal of column 0 reads "minus infinity" */
        const int32_t NEGW = -(1 << 28);
        static thread_local std::vector<int32_t> Wm, sc_tab; /* reused from alignment to alignment: most are a few dozen cells */
        if (Wm.size() < (size_t)(n + 1) * stride + 8) Wm.resize((size_t)(n + 1) * stride + 8);
        auto wrow = [&](int i) -> int32_t* { return Wm.data() + (size_t)i * stride + 8; };
        for (int i = 0; i <= n; ++i) for (int j = -8; j < 0; ++j) wrow(i)[j] = NEGW;
        for (int j = 0; j < colsP; ++j) wrow(0)[j] = 0;
        if (sc_tab.size() < (size_t)4 * colsP) sc_tab.resize((size_t)4 * colsP); /* per base code: substitution score - gap for every column */
        for (int b = 0; b < 4; ++b) for (int j = 0; j < colsP; ++j) sc_tab[(size_t)b * colsP + j] = (j >= 1 && j < cols && seq[j - 1] == "ACGT"[b]) ? CW_POA_MATCH - g : CW_POA_MISMATCH - g;
        const __m256i vneg = _mm256_set1_epi32(NEGW), vg = _mm256_set1_epi32(g);
        const __m256i sh1 = _mm256_setr_epi32(0, 0, 1, 2, 3, 4, 5, 6), sh2 = _mm256_setr_epi32(0, 0, 0, 1, 2, 3, 4, 5), sh4 = _mm256_setr_epi32(0, 0, 0, 0, 0, 1, 2, 3),
                      last = _mm256_set1_epi32(7);
        for (int i = 1; i <= n; ++i) {
            const PoaNode& nd = nodes[rank2node[i - 1]];
            const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
            const int32_t* srow = sc_tab.data() + (size_t)base_code(nd.base) * colsP;
            int32_t* out = wrow(i);
            __m256i carry = vneg;
            for (int j = 0; j < colsP; j += 8) {
                __m256i acc = vneg;
                const __m256i s8 = _mm256_loadu_si256((const __m256i*)(srow + j));
                for (size_t p = 0; p < np; ++p) {
                    const int32_t* pr = wrow(nd.in_edges.empty() ? 0 : pred_row(nd, p));
                    const __m256i up = _mm256_loadu_si256((const __m256i*)(pr + j)), dg = _mm256_loadu_si256((const __m256i*)(pr + j - 1));
                    acc = _mm256_max_epi32(acc, _mm256_max_epi32(_mm256_add_epi32(dg, s8), _mm256_add_epi32(up, vg)));
                }
#if CW_POA_MODE == CW_POA_MODE_OV
                if (j == 0) acc = _mm256_blend_epi32(acc, _mm256_setzero_si256(), 0x01); /* overlap mode: column 0 costs nothing */
#endif
                __m256i t = _mm256_blend_epi32(_mm256_permutevar8x32_epi32(acc, sh1), vneg, 0x01); acc = _mm256_max_epi32(acc, t);
                t = _mm256_blend_epi32(_mm256_permutevar8x32_epi32(acc, sh2), vneg, 0x03); acc = _mm256_max_epi32(acc, t);
                t = _mm256_blend_epi32(_mm256_permutevar8x32_epi32(acc, sh4), vneg, 0x0F); acc = _mm256_max_epi32(acc, t);
                acc = _mm256_max_epi32(acc, carry);
                carry = _mm256_permutevar8x32_epi32(acc, last);
                _mm256_storeu_si256((__m256i*)(out + j), acc);
            }
        }
        struct AtW { int32_t* base; int stride; int g; int32_t operator()(int i, int j) const { return base[(size_t)i * stride + 8 + j] + j * g; } };
        const AtW at{Wm.data(), stride, g};
#else
        static thread_local std::vector<int32_t> H; /* reused from alignment to alignment; every cell is written before it is read */
        if (H.size() < (size_t)(n + 1) * cols) H.resize((size_t)(n + 1) * cols);
        auto at = [&](int i, int j) -> int32_t& { return H[(size_t)i * cols + j]; };

        for (int j = 0; j < cols; ++j) at(0, j) = CW_POA_MODE == CW_POA_MODE_SW ? 0 : j * g; /* local mode: the sequence's prefix is free too */
        for (int i = 1; i <= n; ++i) {
            const PoaNode& nd = nodes[rank2node[i - 1]];
            int32_t best = nd.in_edges.empty() ? 0 : INT_MIN;
            for (size_t p = 0; p < nd.in_edges.size(); ++p) best = std::max(best, at(pred_row(nd, p), 0));
            at(i, 0) = CW_POA_MODE != CW_POA_MODE_NW ? 0 : best + g; /* overlap and local mode: the graph's prefix is free */
        }
        for (int i = 1; i <= n; ++i) {
            const PoaNode& nd = nodes[rank2node[i - 1]];
            const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
            for (size_t p = 0; p < np; ++p) {
                const int pi = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                for (int j = 1; j < cols; ++j) {
                    int32_t s = (seq[j - 1] == nd.base) ? CW_POA_MATCH : CW_POA_MISMATCH;
                    int32_t v = std::max(at(pi, j - 1) + s, at(pi, j) + g);
                    at(i, j) = (p == 0) ? v : std::max(at(i, j), v);
                }
            }
            for (int j = 1; j < cols; ++j) at(i, j) = std::max(at(i, j - 1) + g, at(i, j));
#if CW_POA_MODE == CW_POA_MODE_SW
            for (int j = 1; j < cols; ++j) at(i, j) = std::max(at(i, j), 0); /* local mode: no cell below 0 (a clamped cell never feeds a positive horizontal move: gaps cost) */
#endif
        }
#endif
        if (st) { st->dp_cells += (uint64_t)n * L; st->alignments++; }

        int bi = -1, bj = L;
        int32_t bs = INT_MIN;
#if CW_POA_MODE == CW_POA_MODE_SW
        /* local mode: the best cell anywhere, columns 1..L: lowest rank, then lowest column on ties; the bases beyond it are insertions */
        for (int i = 1; i <= n; ++i)
            for (int j = 1; j <= L; ++j) if (bi == -1 || bs < at(i, j)) { bs = at(i, j); bi = i; bj = j; }
        for (int j = L; j > bj; --j) path.emplace_back(-1, j - 1);
#elif CW_POA_MODE == CW_POA_MODE_OV
        /* overlap mode (spoa kOV as published: first row gap-penalised, first column free, the alignment ends in the best cell of a node
           without out-edges, columns 1..L, and stops where it reaches the first row or the first column): lowest rank, then lowest column on ties.
           The sequence's bases beyond the end cell and before the stop are on no graph node: insertions, as in the global mode's first row. */
        for (int i = 1; i <= n; ++i) {
            if (!nodes[rank2node[i - 1]].out_edges.empty()) continue;
            for (int j = 1; j <= L; ++j) if (bi == -1 || bs < at(i, j)) { bs = at(i, j); bi = i; bj = j; }
        }
        for (int j = L; j > bj; --j) path.emplace_back(-1, j - 1);
#else
        for (int i = 1; i <= n; ++i) {
            if (!nodes[rank2node[i - 1]].out_edges.empty()) continue;
            if (bi == -1 || bs < at(i, L)) { bs = at(i, L); bi = i; }
        }
#endif
        int i = bi, j = bj;
        while (CW_POA_MODE != CW_POA_MODE_NW ? (i != 0 && j != 0) : !(i == 0 && j == 0)) {
            const int32_t h = at(i, j);
            if (CW_POA_MODE == CW_POA_MODE_SW && h == 0) break; /* local mode: the alignment starts where the score does */
            int pi = i, pj = j;
            bool found = false;
            if (i != 0 && j != 0) {
                const PoaNode& nd = nodes[rank2node[i - 1]];
                int32_t s = (seq[j - 1] == nd.base) ? CW_POA_MATCH : CW_POA_MISMATCH;
                const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
                for (size_t p = 0; p < np && !found; ++p) {
                    int r = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                    if (h == at(r, j - 1) + s) { pi = r; pj = j - 1; found = true; }
                }
            }
            if (!found && i != 0) {
                const PoaNode& nd = nodes[rank2node[i - 1]];
                const size_t np = nd.in_edges.empty() ? 1 : nd.in_edges.size();
                for (size_t p = 0; p < np && !found; ++p) {
                    int r = nd.in_edges.empty() ? 0 : pred_row(nd, p);
                    if (h == at(r, j) + g) { pi = r; pj = j; found = true; }
                }
            }
            if (!found && j != 0) {
                if (h == at(i, j - 1) + g) { pi = i; pj = j - 1; found = true; }
            }
            assert(found);
            path.emplace_back(i == pi ? -1 : rank2node[i - 1], j == pj ? -1 : j - 1);
            i = pi; j = pj;
        }
#if CW_POA_MODE != CW_POA_MODE_NW
        for (; j > 0; --j) path.emplace_back(-1, j - 1); /* stopped in the first row (or, local mode, at a cell of value 0): the bases before are insertions */
#endif
        std::reverse(path.begin(), path.end());
        return path;
    }

    void add_sequence(const std::string& seq, Stats* st) {
        if (seq.empty()) return;
        n_sequences++;
        if (nodes.empty()) {
            for (size_t t = 0; t < seq.size(); ++t) {
                int v = add_node(seq[t]);
                if (t) add_edge(v - 1, v);
                place_before_column(v, -1);
            }
            template_nodes = (int)nodes.size();
            return;
        }
        std::vector<std::pair<int, int>> aln = align(seq, st);
        /* global mode: every sequence base is on the path, so there is no unaligned head or tail */
        int head = -1;
        for (size_t t = 0; t < aln.size(); ++t) {
            const std::pair<int, int>& pr = aln[t];
            if (pr.second == -1) continue;                 /* graph node against a gap */
            const char b = seq[pr.second];
            int cur;
            if (pr.first == -1) {                          /* insertion: fresh node before the next ranked path node */
                int q = -1;
                for (size_t u = t + 1; u < aln.size(); ++u)
                    if (aln[u].second != -1 && aln[u].first != -1) { q = aln[u].first; break; }
                cur = add_node(b);
                place_before_column(cur, q);
            } else if (nodes[pr.first].base == b) {
                cur = pr.first;
                nodes[cur].coverage++;
            } else {
                cur = -1;
                for (int a : nodes[pr.first].aligned)
                    if (nodes[a].base == b) { cur = a; nodes[a].coverage++; break; }
                if (cur == -1) {                           /* new member of pr.first's column */
                    cur = add_node(b);
                    std::vector<int> column = nodes[pr.first].aligned;
                    for (int a : column) {
                        nodes[cur].aligned.push_back(a);
                        nodes[a].aligned.push_back(cur);
                    }
                    nodes[cur].aligned.push_back(pr.first);
                    nodes[pr.first].aligned.push_back(cur);
                    place_after_column(cur, pr.first);
                }
            }
            if (head != -1) add_edge(head, cur);
            head = cur;
        }
        assert(order_is_valid());
    }

    /* Column-majority consensus over the MSA the graph encodes (one column per aligned group, in
       topological order).  A column whose gap count strictly exceeds every base count is dropped;
       otherwise the most frequent base is emitted; ties between bases go to the template's base when
       it is among the tied, else to the smallest code (A<C<G<T).  See cw_policy.h. */
    /* cw_policy.h CW_POA_CONSENSUS_HEAVIEST_BUNDLE */
    std::string consensus_heaviest_bundle() const {
        const int n = (int)nodes.size();
        std::vector<long> score(n, 0);
        std::vector<int> pred(n, -1);
        int end = -1;
        for (int r = 0; r < n; ++r) {
            const int v = rank2node[r];
            long w_best = -1;
            for (int e : nodes[v].in_edges) {
                const int u = edges[e].from;
                if (w_best < edges[e].weight || (w_best == edges[e].weight && score[pred[v]] <= score[u])) { w_best = edges[e].weight; pred[v] = u; }
            }
            score[v] = pred[v] == -1 ? 0 : w_best + score[pred[v]];
            if (nodes[v].out_edges.empty() && (end == -1 || score[end] < score[v])) end = v;
        }
        std::string out;
        for (int v = end; v != -1; v = pred[v]) out.push_back(nodes[v].base);
        std::reverse(out.begin(), out.end());
        return out;
    }

    std::string consensus() const {
        if (CW_CONS_HEAVIEST_BUNDLE) return consensus_heaviest_bundle();
        std::string out;
        const int n = (int)nodes.size();
        for (int r = 0; r < n;) {
            const int lead = rank2node[r];
            const int width = 1 + (int)nodes[lead].aligned.size();
            int cnt[4] = {0, 0, 0, 0};
            int tpl_code = -1;
            for (int c = 0; c < width; ++c) {
                const int v = rank2node[r + c];
                const int code = (int)base_code(nodes[v].base);
                cnt[code] += nodes[v].coverage;
                if (v < template_nodes) tpl_code = code;
            }
            const int gaps = n_sequences - (cnt[0] + cnt[1] + cnt[2] + cnt[3]);
            int top = 0;
            for (int c = 1; c < 4; ++c) if (cnt[c] > cnt[top]) top = c;
            if (!CW_CONS_DROPS(gaps, cnt[top])) { /* cw_policy.h "switches" */
                if (CW_CONS_TEMPLATE_WINS_TIES && tpl_code != -1 && cnt[tpl_code] == cnt[top]) top = tpl_code;
                out.push_back("ACGT"[top]);
            }
            r += width;
        }
        return out;
    }
};

} // namespace

std::string poa_consensus(const std::vector<std::string>& seqs, Stats* st) {
    PoaGraph g;
    for (const std::string& s : seqs) g.add_sequence(s, st);
    if (st) st->max_nodes = std::max<uint64_t>(st->max_nodes, g.nodes.size());
    return g.consensus();
}

/* ------------------------------------------------------------------------------------------------
 * A4 -- k-mer index, anchor chain, segmentation, per-segment POA (published algorithm; see cw_policy.h)
 * ---------------------------------------------------------------------------------------------- */
bool segmented_poa(const std::vector<std::string>& pile, unsigned k, double anchor_support, unsigned min_anchors,
                   unsigned max_msa, std::string& consensus, KmerCounts& counts, Stats* st) {
    struct Occ { uint32_t seq; int32_t pos; };
    std::unordered_map<kmer_t, std::vector<Occ>> index;
    std::set<kmer_t> repeated;
    const kmer_t mask = (k >= 32) ? ~(kmer_t)0 : (((kmer_t)1 << (2 * k)) - 1);
    std::vector<kmer_t> tpl;
#ifdef CWO_FAST
    /* CPU-baseline build (bench.py's cpu_baseline leg; never the checker): the same index with direct-addressed tables instead of three hash
       maps per k-mer -- per-key count, "seen in this sequence" stamp and repeat flag in flat arrays over all 4^k keys (k <= 11), occurrence
       lists only for the template's surviving k-mers (all the chain and the segmentation ever look up).  Same counts, same anchors, same
       lists in the same order: tests/test_oracle_units.py compares this build with the plain one. */
    const bool flat = k <= 11;
    static thread_local std::vector<uint32_t> f_cnt, f_stamp;
    static thread_local std::vector<int32_t> f_slot;
    static thread_local std::vector<uint8_t> f_rep;
    std::vector<kmer_t> touched;
    std::vector<std::vector<Occ>> occs;
    if (flat) {
        const size_t K4 = (size_t)1 << (2 * k);
        if (f_cnt.size() < K4) { f_cnt.assign(K4, 0); f_stamp.assign(K4, 0); f_slot.assign(K4, -1); f_rep.assign(K4, 0); }
        for (uint32_t s = 0; s < pile.size(); ++s) {
            const std::string& r = pile[s];
            if (r.size() < k) continue;
            kmer_t v = 0;
            for (size_t i = 0; i < r.size(); ++i) {
                v = ((v << 2) | base_code(r[i])) & mask;
                if (i + 1 < k) continue;
                if (f_cnt[v]++ == 0) touched.push_back(v);
                if (f_stamp[v] == s + 1) f_rep[v] = 1; else f_stamp[v] = s + 1;
            }
            if (st) st->kmers += r.size() - k + 1;
        }
        counts.reserve(counts.size() + touched.size());
        for (kmer_t v : touched) counts[v] += f_cnt[v];
        if (!pile.empty() && pile[0].size() >= k) {
            kmer_t v = 0;
            for (size_t i = 0; i < pile[0].size(); ++i) {
                v = ((v << 2) | base_code(pile[0][i])) & mask;
                if (i + 1 >= k && !f_rep[v] && (double)f_cnt[v] >= anchor_support) tpl.push_back(v);
            }
        }
        occs.resize(tpl.size());
        for (size_t a = 0; a < tpl.size(); ++a) { f_slot[tpl[a]] = (int32_t)a; occs[a].reserve(f_cnt[tpl[a]]); }
        for (uint32_t s = 0; s < pile.size(); ++s) {
            const std::string& r = pile[s];
            if (r.size() < k) continue;
            kmer_t v = 0;
            for (size_t i = 0; i < r.size(); ++i) {
                v = ((v << 2) | base_code(r[i])) & mask;
                if (i + 1 >= k && f_slot[v] >= 0) occs[(size_t)f_slot[v]].push_back(Occ{s, (int32_t)(i + 1 - k)});
            }
        }
    } else
#endif
    {
    /* A4a: pile-wide counts + occurrence lists; a k-mer seen twice in one sequence is repeated. */
    for (uint32_t s = 0; s < pile.size(); ++s) {
        const std::string& r = pile[s];
        if (r.size() < k) continue;
        std::unordered_map<kmer_t, uint32_t> local;
        kmer_t v = 0;
        for (size_t i = 0; i < r.size(); ++i) {
            v = ((v << 2) | base_code(r[i])) & mask;
            if (i + 1 < k) continue;
            int32_t pos = (int32_t)(i + 1 - k);
            index[v].push_back(Occ{s, pos});
            counts[v]++;
            if (++local[v] > 1) repeated.insert(v);
            if (st) st->kmers++;
        }
    }
    for (kmer_t v : repeated) index.erase(v);
    for (auto it = index.begin(); it != index.end();) {
        if ((double)it->second.size() < anchor_support) it = index.erase(it);
        else ++it;
    }

    /* Template anchors in template order. */
    if (!pile.empty() && pile[0].size() >= k) {
        kmer_t v = 0;
        for (size_t i = 0; i < pile[0].size(); ++i) {
            v = ((v << 2) | base_code(pile[0][i])) & mask;
            if (i + 1 >= k && index.count(v)) tpl.push_back(v);
        }
    }
    }
    /* the occurrence list of a surviving template k-mer, in (sequence, position) order */
    auto occ_of = [&](kmer_t a) -> const std::vector<Occ>& {
#ifdef CWO_FAST
        if (flat) return occs[(size_t)f_slot[a]];
#endif
        return index[a];
    };
#ifdef CWO_FAST
    struct Cleaner { std::vector<kmer_t>& t; bool on; ~Cleaner() { if (on) for (kmer_t v : t) { f_cnt[v] = 0; f_stamp[v] = 0; f_rep[v] = 0; f_slot[v] = -1; } } } cleaner{touched, flat};
#endif
    if (st) st->tpl_anchors += tpl.size();

    /* A4b: longest ordered chain. */
    auto pair_score = [&](kmer_t a, kmer_t b) -> int {
        const std::vector<Occ>& va = occ_of(a);
        const std::vector<Occ>& vb = occ_of(b);
        size_t i = 0, j = 0;
        int n = 0;
        while (i < va.size() && j < vb.size()) {
            if (va[i].seq == vb[j].seq) { if (va[i].pos < vb[j].pos) ++n; ++i; ++j; }
            else if (va[i].seq < vb[j].seq) ++i;
            else ++j;
        }
        if (st) st->pair_tests++;
        return n;
    };
    const int A = (int)tpl.size();
    std::vector<int> len(A, 0), nxt(A, -1);
    std::vector<long> sc(A, 0);
#ifdef CWO_FAST
    std::vector<int> smax(A + 1, -1); /* smax[b] = longest chain from any anchor >= b (baseline build only, see below) */
#endif
    for (int a = A - 1; a >= 0; --a) {
        int best_len = -1, best_next = -1;
        long best_sc = 0;
        for (int b = a + 1; b < A; ++b) {
#ifdef CWO_FAST
            /* CPU-baseline build: an exact shortcut the straightforward restatement does without -- a successor replaces the best link only
               with a longer chain, or an equally long one of higher score, so once no anchor from b on has a chain as long as the best
               found, the scan is over.  Same links, same chain (tests/test_oracle_units.py compares the two builds); ~50x fewer pair tests. */
            if (smax[b] < best_len) break;
#endif
            int s = pair_score(tpl[a], tpl[b]);
            if ((double)s >= anchor_support) {
                if (len[b] > best_len) { best_len = len[b]; best_sc = sc[b] + s; best_next = b; }
                else if (len[b] == best_len && (CW_CHAIN_TIE == CW_CHAIN_TIE_LARGEST_SUCCESSOR ? sc[b] + s >= best_sc : sc[b] + s > best_sc)) { best_sc = sc[b] + s; best_next = b; } /* cw_policy.h CW_CHAIN_TIE: on equal length and score the scan upward keeps the first (smallest) successor, or takes the later (largest) one */
            }
        }
        len[a] = best_len + 1; sc[a] = best_sc; nxt[a] = best_next;
#ifdef CWO_FAST
        smax[a] = std::max(smax[a + 1], len[a]);
#endif
    }
    int start = -1, top_len = 0;
    long top_sc = 0;
    for (int a = A - 1; a >= 0; --a) {
        if (len[a] > top_len) { top_len = len[a]; top_sc = sc[a]; start = a; }
        else if (len[a] == top_len && sc[a] > top_sc) { top_sc = sc[a]; start = a; }
    }
    std::vector<kmer_t> chain;
    for (int a = start; a != -1; a = nxt[a]) chain.push_back(tpl[a]);
    if (st) st->chain_len += chain.size();
    if (chain.size() < min_anchors || chain.empty()) return false;

    /* A4c + A4d */
    auto pos_in = [&](kmer_t a, uint32_t s) -> int32_t {
#ifdef CWO_FAST
        { /* (baseline build: the list is sorted by sequence and an anchor occurs at most once per sequence) */
            const std::vector<Occ>& v = occ_of(a);
            auto it = std::lower_bound(v.begin(), v.end(), s, [](const Occ& o, uint32_t q) { return o.seq < q; });
            return it != v.end() && it->seq == s ? it->pos : -1;
        }
#endif
        for (const Occ& o : occ_of(a)) if (o.seq == s) return o.pos;
        return -1;
    };
    const size_t m = chain.size();
    consensus.clear();
    /* cw_policy.h CW_SEG_MISSING_ANCHOR: where sequence s holds chain anchor i, or -1.  Under EXTRAPOLATE an anchor the sequence lacks is placed
       where the template's spacing puts it, counted from the nearest anchor the sequence does hold (the one before it, else the one after),
       inside [0, min(length, 65534)]; a sequence that holds no chain anchor at all stays out of every segment. */
    std::vector<std::vector<int32_t>> qpos(pile.size(), std::vector<int32_t>(m, -1));
    for (uint32_t s = 0; s < pile.size(); ++s) {
        for (size_t i = 0; i < m; ++i) qpos[s][i] = pos_in(chain[i], s);
#if CW_SEG_MISSING_ANCHOR == CW_SEG_MISSING_ANCHOR_EXTRAPOLATE
        const int32_t top = (int32_t)std::min<size_t>(pile[s].size(), 65534);
        std::vector<int32_t> held = qpos[s];
        int last = -1, first = -1;
        for (size_t i = 0; i < m; ++i) {
            if (held[i] != -1) { last = (int)i; if (first == -1) first = (int)i; }
            else if (last != -1) qpos[s][i] = std::min(top, held[last] + (pos_in(chain[i], 0) - pos_in(chain[last], 0)));
        }
        for (int i = 0; i < first; ++i) qpos[s][i] = std::min(top, std::max(0, held[first] - (pos_in(chain[first], 0) - pos_in(chain[i], 0))));
#endif
    }
    for (size_t seg = 0; seg <= m; ++seg) {
        std::vector<std::string> members;
        for (uint32_t s = 0; s < pile.size() && members.size() < max_msa; ++s) {
            const std::string& r = pile[s];
            std::string piece;
            if (seg == 0) {
                int32_t p = qpos[s][0];
                if (p == -1) continue;
                piece = r.substr(0, (size_t)p);
            } else if (seg == m) {
                int32_t p = qpos[s][m - 1];
                if (p == -1) continue;
                piece = r.substr((size_t)p);
            } else {
                int32_t p1 = qpos[s][seg - 1], p2 = qpos[s][seg];
                if (p1 == -1 || p2 == -1 || p1 >= p2) continue;
                piece = r.substr((size_t)p1, (size_t)(p2 - p1));
            }
            if (!piece.empty()) members.push_back(piece);
        }
        if (st) {
            st->segments++;
            if (!members.empty()) st->poa_segments++;
            for (auto& x : members) st->max_seg_len = std::max<uint64_t>(st->max_seg_len, x.size());
        }
        /* (statistics only: what the engine's chain kernel writes directly -- a single piece, or equal pieces no longer than k, which are a
           prefix of the left anchor -- never reaches one of its POA tiers; bench.py's poa_gcups_routed counts the other segments' cells) */
        bool trivial = members.size() <= 1;
        if (!trivial && members[0].size() <= k) { trivial = true; for (auto& x : members) if (x != members[0]) { trivial = false; break; } }
        const uint64_t c0 = st ? st->dp_cells : 0, a0 = st ? st->alignments : 0;
        consensus += poa_consensus(members, st);
        if (st && !trivial) { st->dp_cells_routed += st->dp_cells - c0; st->alignments_routed += st->alignments - a0; }
    }
    return true;
}

/* ------------------------------------------------------------------------------------------------
 * A5 -- weightConsensus (correctionMSA.cpp:6-27)
 * ---------------------------------------------------------------------------------------------- */
std::string weight_consensus(std::string cons, const KmerCounts& counts, unsigned k, unsigned solid) {
    /* :14 the loop bound is length-k+1 in unsigned arithmetic; callers guarantee length >= k (:43). */
    for (unsigned i = 0; i + k <= cons.size(); ++i) {
        std::string word = upper_copy(cons.substr(i, k));           /* :15-16 */
        bool strong = count_of(counts, str2num(word)) >= solid;      /* :17 */
        for (unsigned j = i; j < i + k; ++j) {                        /* :18 / :20, inclusive end i+k-1 */
            char c = cons[j];
            if (strong) { if ('a' <= c && c <= 'z') c = (char)(c - 32); }
            else        { if ('A' <= c && c <= 'Z') c = (char)(c + 32); }
            cons[j] = c;
        }
    }
    return cons;
}

/* ------------------------------------------------------------------------------------------------
 * A8 -- getNeighbours (DBG.cpp:18-54)
 * ---------------------------------------------------------------------------------------------- */
std::vector<std::string> neighbours(std::string kmer, unsigned k, int left, const KmerCounts& counts, unsigned solid) {
    std::vector<std::string> out;
    kmer = upper_copy(kmer);                                   /* :22 */
    if (left == 1) kmer = revcomp(kmer);                       /* :24-26 */
    const std::string stem = kmer.substr(1);                   /* :27 */
    for (int nuc = 0; nuc < 4; ++nuc) {                        /* :29 */
        kmer_t key = (str2num(stem) << 2) + (kmer_t)nuc;       /* :30-32 */
        std::string cand;
        if (left == 1) {                                       /* :33-37 */
            cand = revcomp(kmer2str(key, k));
            key = str2num(cand);
        } else {
            cand = stem + "ACGT"[nuc];                         /* :42 via concatNucR :5-16 */
        }
        if (count_of(counts, key) >= solid) out.push_back(cand); /* :38-44 */
    }
    /* :48-52 std::sort on <=4 elements == insertion sort == stable; descending count. */
    std::stable_sort(out.begin(), out.end(), [&](const std::string& a, const std::string& b) {
        return count_of(counts, str2num(a)) > count_of(counts, str2num(b));
    });
    return out;
}

/* ------------------------------------------------------------------------------------------------
 * A9 -- extendLeft / extendRight (DBG.cpp:56-75, :77-96)
 * ---------------------------------------------------------------------------------------------- */
unsigned extend_left(const KmerCounts& counts, unsigned k, unsigned ext_len, std::string& lr, unsigned solid) {
    unsigned dist = 0;
    std::vector<std::string> nb = neighbours(lr.substr(0, k), k, 1, counts, solid);   /* :62 */
    while (nb.size() == 1 && dist < ext_len) {                                        /* :66 */
        lr = nb[0].substr(0, nb[0].size() - (k - 1)) + lr;                            /* :67 */
        dist += (unsigned)nb[0].size() - (k - 1);                                     /* :68 */
        nb = neighbours(lr.substr(0, k), k, 1, counts, solid);                        /* :70 */
    }
    return dist;
}

unsigned extend_right(const KmerCounts& counts, unsigned k, unsigned ext_len, std::string& lr, unsigned solid) {
    unsigned dist = 0;
    std::vector<std::string> nb = neighbours(lr.substr(lr.size() - k), k, 0, counts, solid);  /* :83 */
    while (!nb.empty() && dist < ext_len) {                                                   /* :87 */
        lr = lr + nb[0].substr(k - 1);                                                        /* :88 */
        dist += (unsigned)nb[0].size() - (k - 1);                                             /* :89 */
        nb = neighbours(lr.substr(lr.size() - k), k, 0, counts, solid);                       /* :91 */
    }
    return dist;
}

/* ------------------------------------------------------------------------------------------------
 * A10 -- link (DBG.cpp:99-169).  cur_k == mer_size == min_order on every call the reference makes
 * (correctionDBG.cpp:164, DBG.cpp:149), so they are folded into k.
 * ---------------------------------------------------------------------------------------------- */
static int link_path(const KmerCounts& counts, const std::string& dst_seed, unsigned k, std::set<std::string>& visited,
                     unsigned* branches, unsigned dist, const std::string& cur_ext, std::string& found_path,
                     unsigned max_len, unsigned max_branches, unsigned solid, Stats* st) {
    if (st) st->link_calls++;
    if (*branches > max_branches || dist > max_len) {          /* :100-103 */
        found_path = std::string();
        return 0;
    }
    std::string src_anchor = cur_ext.substr(cur_ext.size() - k);   /* :105 */
    const std::string tgt = dst_seed.substr(0, k);                  /* :106 (and :123,:143: candidates have length k) */
    bool found = (src_anchor == tgt);                               /* :109 */
    std::string path = cur_ext;                                     /* :111 */

    std::vector<std::string> nb = neighbours(src_anchor, k, 0, counts, solid);   /* :115 */
    if (st) st->nbr_calls++;
    size_t it = 0;

    while (!found && nb.size() == 1 && it < nb.size() && dist <= max_len) {      /* :119 */
        const std::string cand = nb[it];
        bool seen = visited.count(cand) != 0;                                    /* :121 */
        found = (cand == tgt);                                                   /* :123 */
        if (!found && !seen) {                                                   /* :124 */
            visited.insert(cand);
            path += cand[k - 1];
            dist += (unsigned)cand.size() - (k - 1);
            src_anchor = path.substr(path.size() - k);                           /* :130 */
            nb = neighbours(src_anchor, k, 0, counts, solid);                    /* :131 */
            if (st) st->nbr_calls++;
            it = 0;
        } else if (found) {
            path += cand[k - 1];                                                 /* :134 */
        } else {
            ++it;                                                                /* :136 */
        }
    }
    while (!found && nb.size() > 1 && it < nb.size() && dist <= max_len) {       /* :141 */
        const std::string cand = nb[it];
        bool seen = visited.count(cand) != 0;
        found = (cand == tgt);
        if (!found && !seen) {                                                   /* :146 */
            visited.insert(cand);
            (*branches)++;
            found = link_path(counts, dst_seed, k, visited, branches, dist + (unsigned)cand.size() - (k - 1),
                              path + cand[k - 1], found_path, max_len, max_branches, solid, st) != 0;   /* :149 */
            if (!found) ++it;
            else return 1;                                                       /* :153 */
        } else if (found) {
            path += cand[k - 1];                                                 /* :156 */
        } else {
            ++it;
        }
    }
    if (!found) return 0;                                                        /* :163 */
    found_path = path + dst_seed.substr(k);                                      /* :166 (empty suffix) */
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * A6a/A6b -- getNextSrc/getNextDst/getAnchors (correctionDBG.cpp:13-91)
 * ---------------------------------------------------------------------------------------------- */
int next_src(const std::string& s, unsigned beg, unsigned m) {
    unsigned run = 0, i = beg;
    while (i < s.size() && (is_upper(s[i]) || run < m)) {
        if (is_upper(s[i])) run++; else run = 0;
        i++;
    }
    return run >= m ? (int)i - 1 : -1;
}

int next_dst(const std::string& s, unsigned beg, unsigned m) {
    unsigned run = 0, i = beg;
    while (i < s.size() && run < m) {
        if (is_upper(s[i])) run++; else run = 0;
        i++;
    }
    return run >= m ? (int)i - 1 : -1;
}

namespace {
struct AnchorPair { std::string src, dst; unsigned src_off, dst_off; };
}

static std::vector<AnchorPair> zone_anchors(const KmerCounts& counts, const std::string& src_zone, const std::string& dst_zone,
                                            unsigned k, unsigned keep) {
    auto unique_words = [&](const std::string& zone) {
        std::vector<std::pair<std::string, unsigned>> out;      /* (word, its only offset), zone order */
        const unsigned n = (unsigned)zone.size() - k + 1;
        for (unsigned i = 0; i < n; ++i) {
            std::string w = zone.substr(i, k);
            unsigned occ = 0;
            for (unsigned j = 0; j < n; ++j) if (zone.compare(j, k, w) == 0) occ++;
            if (occ == 1) out.emplace_back(w, i);               /* :67,:69 */
        }
        return out;
    };
    auto srcs = unique_words(src_zone), dsts = unique_words(dst_zone);
    std::vector<AnchorPair> all;
    for (auto& s : srcs) for (auto& d : dsts) all.push_back(AnchorPair{s.first, d.first, s.second, d.second});  /* :66-74 */
    /* :77-83 -- at most 16 pairs => libstdc++ std::sort is a plain insertion sort => stable. */
    std::stable_sort(all.begin(), all.end(), [&](const AnchorPair& a, const AnchorPair& b) {
        int oa = (int)(count_of(counts, str2num(a.src)) + count_of(counts, str2num(a.dst)));
        int ob = (int)(count_of(counts, str2num(b.src)) + count_of(counts, str2num(b.dst)));
        return oa > ob;
    });
    if (all.size() > keep) all.resize(keep);                     /* :85-88 */
    return all;
}

/* ------------------------------------------------------------------------------------------------
 * A6 -- polishCorrection (correctionDBG.cpp:93-205)
 * ---------------------------------------------------------------------------------------------- */
std::string polish(std::string read, const KmerCounts& counts, unsigned k, unsigned solid, Stats* st) {
    std::set<std::string> visited;                               /* :94 -- never cleared */
    const unsigned zone = CW_DBG_ZONE, max_branches = CW_DBG_MAX_BRANCHES;
    unsigned tmp_src_beg = 0, tmp_src_end = 0, tmp_dst_beg = 0, tmp_dst_end = 0;   /* :104, live across iterations */

    unsigned i = 0;
    while (i < read.size() && !is_upper(read[i])) i++;           /* :116-119 */
    if (i > 0 && i < read.size() && read.size() - i >= k) {      /* :121 */
        int ext_len = (int)i;
        std::string old = read;
        read = read.substr(i);
        int ext = (int)extend_left(counts, k, (unsigned)ext_len, read, solid);     /* :125 */
        if (ext < ext_len) {
            read = old.substr(0, (size_t)(ext_len - ext)) + read;                   /* :127 */
            i = i - (unsigned)(ext_len - ext);                                      /* :128 */
        }
    }

    while (i < read.size()) {                                    /* :133 */
        int src_end = next_src(read, i, k + zone);
        int dst_end = next_dst(read, (unsigned)(src_end + 1), k + zone);
        int src_beg = src_end - (int)k - (int)zone + 1;
        int dst_beg = dst_end - (int)k - (int)zone + 1;
        if (src_end != -1 && dst_end != -1) {                    /* :140 */
            std::string region;
            std::string src_zone = read.substr((size_t)src_beg, k + zone);
            std::string dst_zone = read.substr((size_t)dst_beg, k + zone);
            std::vector<AnchorPair> anchors = zone_anchors(counts, src_zone, dst_zone, k, CW_DBG_MAX_ANCHORS);
            size_t a = 0;
            while (a < anchors.size() && region.empty()) {       /* :150 */
                const AnchorPair& ap = anchors[a];
                tmp_src_beg = (unsigned)src_beg + ap.src_off;    /* :153-156 */
                tmp_src_end = tmp_src_beg + k - 1;
                tmp_dst_beg = (unsigned)dst_beg + ap.dst_off;
                tmp_dst_end = tmp_dst_beg + k - 1;
                if (ap.src != ap.dst) {                          /* :158 */
                    unsigned branches = 0;
                    region.clear();
                    /* :163 -- double arithmetic, evaluated left to right, truncated to unsigned. */
                    unsigned gap = tmp_dst_beg - tmp_src_end - 1;
                    volatile double t0 = 15.0 / 100.0 * 2.0;
                    volatile double t1 = t0 * (double)gap;
                    volatile double t2 = t1 + (double)gap;
                    volatile double t3 = t2 + (double)k;
                    unsigned max_size = (unsigned)t3;
                    link_path(counts, ap.dst, k, visited, &branches, 0, ap.src, region, max_size, max_branches, solid, st);
                }
                a++;
            }
            if (!region.empty()) {                               /* :169 */
                std::string r = read.substr(tmp_src_beg, tmp_dst_end - tmp_src_beg + 1);
                int b = (int)read.find(r);                       /* :173 first occurrence */
                if (b != -1) {
                    read.replace((size_t)b, r.size(), region);
                    i = (unsigned)b;
                } else {
                    i = tmp_dst_beg > i ? tmp_dst_beg : (unsigned)dst_beg;
                }
            } else {
                i = tmp_dst_beg > i ? tmp_dst_beg : (unsigned)dst_beg;   /* :182 */
            }
        } else {
            i = (unsigned)read.size();                           /* :185 */
        }
    }

    i = (unsigned)read.size() - 1;                               /* :189 */
    while (i > 0 && !is_upper(read[i])) i--;
    if (i > 0 && i < read.size() - 1 && i + 1 >= k) {            /* :194 */
        int ext_len = (int)(read.size() - 1 - i);
        std::string old = read;
        read = read.substr(0, i + 1);
        int ext = (int)extend_right(counts, k, (unsigned)ext_len, read, solid);
        if (ext < ext_len) read += old.substr(old.size() - (size_t)(ext_len - ext), (size_t)(ext_len - ext));   /* :200 */
    }
    return read;
}

/* ------------------------------------------------------------------------------------------------
 * A3 -- computeConsensusReadCorrection / computeConsensusAssemblyPolishing (correctionMSA.cpp:29-71)
 * ---------------------------------------------------------------------------------------------- */
WindowResult window_consensus(const std::vector<std::string>& pile, const Params& p, Stats* st) {
    WindowResult out;
    int sup = std::min((int)p.common_kmers, (int)pile.size() / 2);                 /* :31 */
    std::string cons;
    bool ok = segmented_poa(pile, p.k, (double)sup, p.min_anchors, p.max_msa, cons, out.counts, st);   /* :32 */
    if (!ok) {                                                                      /* :34-36 */
        out.consensus = pile.empty() ? std::string() : pile[0];
        out.status = 1;
        return out;
    }
    if (cons.size() >= p.k) {                                                       /* :43-46 */
        cons = weight_consensus(cons, out.counts, p.k, p.solid);
        cons = polish(cons, out.counts, p.k, p.solid, st);
    }
    out.consensus = cons;
    out.status = 0;
    return out;
}

/* ------------------------------------------------------------------------------------------------
 * A1 -- getCoverages + getAlignmentWindowsPositions (alignmentWindows.cpp:5-85)
 * ---------------------------------------------------------------------------------------------- */
std::vector<std::pair<uint32_t, uint32_t>> window_positions(uint32_t tpl_len, const std::vector<Ovl>& ovl,
                                                            unsigned min_support, unsigned window_size, int window_overlap) {
    std::vector<uint32_t> cov(tpl_len, 0);
    for (const Ovl& o : ovl)
        for (uint32_t i = o.q_start; i <= o.q_end && i < tpl_len; ++i) cov[i]++;   /* :15-22, ends inclusive */
    std::vector<std::pair<uint32_t, uint32_t>> out;
    uint32_t cur = 0, beg = 0, i = 0;
    while (i < tpl_len) {                                       /* :39 */
        if (cur >= window_size) {
            out.emplace_back(beg, beg + cur - 1);
            if (window_overlap) i = i - (uint32_t)window_overlap;
            beg = i;
            cur = 0;
        }
        if (cov[i] < min_support) { cur = 0; i++; beg = i; }
        else { cur++; i++; }
    }
    bool pushed = false;                                        /* :59-79 trailing window */
    uint32_t end = tpl_len - 1;
    cur = 0;
    i = tpl_len - 1;
    while (i > 0 && !pushed) {
        if (cur >= window_size) {
            out.emplace_back(end - cur + 1, end);
            pushed = true;
            end = i;
            cur = 0;
        }
        if (cov[i] < min_support) { cur = 0; i--; end = i; }
        else { cur++; i--; }
    }
    return out;
}

/* ------------------------------------------------------------------------------------------------
 * A2 -- getAlignmentWindowsSequences (alignmentWindows.cpp:87-149)
 * ---------------------------------------------------------------------------------------------- */
static std::string clamp_substr(const std::string& s, long pos, long n) {
    /* std::string::substr semantics for pos <= size (the reference never passes pos > size on
       well-formed PAF; it would throw).  n is taken modulo 2^64 like the size_t conversion. */
    if (pos < 0 || (size_t)pos > s.size()) return std::string();
    return s.substr((size_t)pos, (size_t)n);
}

std::vector<std::string> window_pile(const std::vector<Ovl>& ovl, const std::string& tpl,
                                     const std::vector<std::string>& targets, uint32_t q_beg, uint32_t q_end, unsigned k) {
    std::vector<std::string> pile;
    uint32_t length = q_end - q_beg + 1;
    if ((uint64_t)q_beg + length - 1 >= tpl.size()) return pile;   /* :95-97 */
    pile.push_back(tpl.substr(q_beg, length));                      /* :100 */
    for (const Ovl& al : ovl) {
        uint32_t t_beg = al.t_start, t_end = al.t_end, shift;
        length = q_end - q_beg + 1;
        shift = (q_beg > al.q_start) ? q_beg - al.q_start : 0;      /* :110-114 */
        bool spans = ((al.q_start <= q_beg && al.q_end > q_beg) || (q_end <= al.q_end && al.q_start < q_end));
        if (!(spans && al.t_start + shift <= al.t_end)) continue;   /* :117 */
        if (q_beg < al.q_start && al.q_end < q_end) {               /* :119-123 */
            shift = 0;
            t_beg = (uint32_t)std::max(0, (int)al.t_start - ((int)al.q_start - (int)q_beg));
            t_end = (uint32_t)std::min((int)al.t_len - 1, (int)al.t_end + ((int)q_end - (int)al.q_end));
            length = t_end - t_beg + 1;
        } else if (q_beg < al.q_start) {                            /* :124-127 */
            shift = 0;
            t_beg = (uint32_t)std::max(0, (int)al.t_start - ((int)al.q_start - (int)q_beg));
            length = (uint32_t)std::min((int)length, std::min((int)al.t_len - 1, (int)t_beg + (int)length - 1) - (int)t_beg + 1);
        } else if (al.q_end < q_end) {                              /* :128-130 */
            t_end = (uint32_t)std::min((int)al.t_len - 1, (int)al.t_end + ((int)q_end - (int)al.q_end));
            length = (uint32_t)std::min((int)length, (int)t_end - std::max(0, (int)t_end - (int)length + 1) + 1);
        }
        std::string seq = clamp_substr(targets[al.t_id], (long)t_beg, (long)t_end - (long)t_beg + 1);   /* :133 */
        if (al.strand) seq = revcomp(seq);                                                              /* :134-136 */
        seq = clamp_substr(seq, (long)shift, (long)length);                                             /* :138 */
        if (seq.size() >= k) pile.push_back(seq);                                                       /* :141 */
    }
    return pile;
}

} // namespace cwo

/* ==================================================================================================
 * SURVEY 8f-1 -- read re-assembly: alignConsensus (correctionAlignment.cpp:47-140) over a restated
 * striped-Smith-Waterman (library absent: policy in include/cw_policy.h, PARITY UNPINNED), trimRead / dropRead
 * (utils.cpp:96-128, :71-73; pinned against oracle/_ref).
 * ================================================================================================== */
namespace cwo {

namespace {

inline int ssw_code(char c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
inline int ssw_score(int a, int b) { return (a == 4 || b == 4) ? 0 : (a == b ? CW_SSW_MATCH : -CW_SSW_MISMATCH); }

/* One exact local-alignment sweep over the columns of `ref` in the given order; returns the best score, the first column
 * reaching it (strict '>' while sweeping; stops early once `terminate` is reached, -1 = never) and the smallest query index
 * of that column holding it. */
struct Sweep { int score, col, row; };
Sweep ssw_sweep(const std::vector<int>& q, const std::vector<int>& r, int r_first, int r_last_excl, int step, int terminate) {
    const int m = (int)q.size();
    std::vector<int> H(m, 0), E(m, 0), Hn(m, 0);
    Sweep best{0, -1, 0};
    for (int i = r_first; i != r_last_excl; i += step) {
        int f = 0, diag = 0, col_max = 0, col_row = 0;
        for (int j = 0; j < m; ++j) {
            /* E: gap along the reference (column to column), F: gap along the query (inside the column) */
            int e = std::max(E[j] - CW_SSW_GAP_EXT, H[j] - CW_SSW_GAP_OPEN);
            if (e < 0) e = 0;
            int h = diag + ssw_score(q[j], r[i]);
            if (h < e) h = e;
            if (h < f) h = f;
            if (h < 0) h = 0;
            diag = H[j];
            Hn[j] = h;
            E[j] = e;
            f = std::max(f - CW_SSW_GAP_EXT, h - CW_SSW_GAP_OPEN);
            if (f < 0) f = 0;
            if (h > col_max) { col_max = h; col_row = j; }
        }
        H.swap(Hn);
        if (col_max > best.score) {
            best.score = col_max; best.col = i; best.row = col_row;
            if (best.score == terminate) break;
        }
    }
    return best;
}

/* banded traceback between the found ends: totals of inserted (query-only) and deleted (reference-only) bases */
void ssw_banded_indels(const std::vector<int>& ref, const std::vector<int>& read, int score, unsigned* ins, unsigned* del) {
    const int refLen = (int)ref.size(), readLen = (int)read.size();
    *ins = 0; *del = 0;
    if (refLen == 0 || readLen == 0) return;
    int band = std::abs(refLen - readLen) + 1;
    std::vector<int8_t> dir;
    int width_d = 0;
    for (;;) {
        const int width = band * 2 + 3;
        width_d = band * 2 + 1;
        std::vector<int> h_b(width, 0), e_b(width, 0), h_c(width, 0);
        dir.assign((size_t)width_d * readLen * 3, 0);
        int max = 0;
        auto set_u = [&](int w, int i, int j) { int x = i - w; x = x > 0 ? x : 0; return j - x + 1; };
        auto set_d = [&](int w, int i, int j, int p) { int x = i - w; x = x > 0 ? x : 0; x = j - x; return x * 3 + p; };
        for (int i = 0; i < readLen; ++i) {
            int beg = 0, end = refLen - 1, u = 0;
            int j = i - band; beg = beg > j ? beg : j;
            j = i + band; end = end < j ? end : j;
            const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
            int f = 0;
            h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
            int8_t* line = dir.data() + (size_t)width_d * i * 3;
            for (j = beg; j <= end; ++j) {
                u = set_u(band, i, j);
                const int e = set_u(band, i - 1, j), b = set_u(band, i, j - 1), d = set_u(band, i - 1, j - 1);
                const int de = set_d(band, i, j, 0), df = set_d(band, i, j, 1), dh = set_d(band, i, j, 2);
                int t1 = i == 0 ? -CW_SSW_GAP_OPEN : h_b[e] - CW_SSW_GAP_OPEN;
                int t2 = i == 0 ? -CW_SSW_GAP_EXT : e_b[e] - CW_SSW_GAP_EXT;
                e_b[u] = t1 > t2 ? t1 : t2;
                line[de] = t1 > t2 ? 3 : 2;
                t1 = h_c[b] - CW_SSW_GAP_OPEN;
                t2 = f - CW_SSW_GAP_EXT;
                f = t1 > t2 ? t1 : t2;
                line[df] = t1 > t2 ? 5 : 4;
                const int e1 = e_b[u] > 0 ? e_b[u] : 0, f1 = f > 0 ? f : 0;
                t1 = e1 > f1 ? e1 : f1;
                t2 = h_b[d] + ssw_score(ref[j], read[i]);
                h_c[u] = t1 > t2 ? t1 : t2;
                if (h_c[u] > max) max = h_c[u];
                if (t1 <= t2) line[dh] = 1;
                else line[dh] = e1 > f1 ? line[de] : line[df];
            }
            for (j = 1; j <= u; ++j) h_b[j] = h_c[j];
        }
        if (max >= score || band > refLen + readLen) break;
        band *= 2;
    }
    /* trace back from the last cell */
    int i = readLen - 1, j = refLen - 1, state = 2;
    auto set_d = [&](int w, int ii, int jj, int p) { int x = ii - w; x = x > 0 ? x : 0; x = jj - x; return x * 3 + p; };
    while (i > 0 && j >= 0) {
        const int8_t* line = dir.data() + (size_t)width_d * i * 3;
        const int idx = set_d(band, i, j, state);
        if (idx < 0 || idx >= width_d * 3) break;
        switch (line[idx]) {
        case 1: --i; --j; state = 2; break;
        case 2: --i; state = 0; ++*ins; break;
        case 3: --i; state = 2; ++*ins; break;
        case 4: --j; state = 1; ++*del; break;
        case 5: --j; state = 2; ++*del; break;
        default: return; /* outside the band */
        }
    }
}

} // namespace

SwResult ssw_align(const std::string& query, const std::string& ref) {
    SwResult out{0, 0, -1, 0, -1, 0, 0};
    std::vector<int> q(query.size()), r(ref.size());
    for (size_t i = 0; i < query.size(); ++i) q[i] = ssw_code(query[i]);
    for (size_t i = 0; i < ref.size(); ++i) r[i] = ssw_code(ref[i]);
    if (q.empty() || r.empty()) return out;
    const Sweep fw = ssw_sweep(q, r, 0, (int)r.size(), 1, -1);
    out.score = fw.score;
    if (fw.score <= 0) return out;
    out.ref_end = fw.col; out.query_end = fw.row;
    std::vector<int> qrev(q.begin(), q.begin() + fw.row + 1);
    std::reverse(qrev.begin(), qrev.end());
    const Sweep bw = ssw_sweep(qrev, r, fw.col, -1, -1, fw.score);
    out.ref_begin = bw.col; out.query_begin = fw.row - bw.row;
    std::vector<int> rsub(r.begin() + out.ref_begin, r.begin() + out.ref_end + 1), qsub(q.begin() + out.query_begin, q.begin() + out.query_end + 1);
    ssw_banded_indels(rsub, qsub, fw.score, &out.ins, &out.del);
    return out;
}

static int nb_solid_mers(const std::string& seq, const std::vector<uint32_t>& solid, unsigned k) { /* correctionAlignment.cpp:6-15 */
    int nb = 0;
    for (size_t i = 0; i + k <= seq.size(); ++i) {
        kmer_t v = 0;
        for (unsigned j = 0; j < k; ++j) {
            const char c = seq[i + j];
            v = (v << 2) | (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u); /* policy: anything else, lower case included, is T */
        }
        if (std::binary_search(solid.begin(), solid.end(), (uint32_t)v)) nb++;
    }
    return nb;
}
static int nb_upper(const std::string& s) { int n = 0; for (char c : s) n += is_upper(c) ? 1 : 0; return n; }

std::string align_consensus(const std::string& sequence, const std::vector<std::string>& consensuses,
                            const std::vector<std::vector<uint32_t>>& solid, const std::vector<std::pair<uint32_t, uint32_t>>& piles_pos,
                            const std::vector<std::string>& templates, int start_pos, unsigned window_size, unsigned window_overlap,
                            unsigned mer_size) {
    std::string out = sequence;                                                    /* :56-57 */
    for (char& c : out) if ('A' <= c && c <= 'Z') c = (char)(c + 32);
    unsigned beg, end, old_end = 0;
    int cur_pos = start_pos;
    std::string cur, old_cons;
    const std::vector<uint32_t>* old_mers = nullptr;
    for (size_t i = 0; i < consensuses.size(); ++i) {                              /* :74 */
        cur = consensuses[i].size() < mer_size ? templates[i] : consensuses[i];    /* :75-80 */
        const std::vector<uint32_t>* cur_mers = &solid[i];
        const int al_pos = std::max(0, cur_pos - (int)window_overlap);             /* :83 */
        int size_al;
        if ((size_t)al_pos + window_size + 2 * window_overlap >= out.size()) size_al = (int)out.size() - al_pos;   /* :84-88 */
        else size_al = (int)(window_size + 2 * window_overlap);
        if (size_al <= 0 || cur.empty()) continue;                                 /* Align() refuses an empty query */
        const SwResult al = ssw_align(cur, out.substr((size_t)al_pos, (size_t)size_al));   /* :90 */
        if (al.score <= 0) continue;
        beg = (unsigned)(al.ref_begin + al_pos);                                   /* :91-93 */
        end = (unsigned)(al.ref_end + al_pos);
        cur = cur.substr((size_t)al.query_begin, (size_t)(al.query_end - al.query_begin + 1));
        if (i != 0 && old_end >= beg) {                                            /* :96 */
            const unsigned overlap = old_end - beg + 1;
            if (consensuses[i].size() >= mer_size && old_cons.size() >= overlap && cur.size() >= overlap) {
                const std::string seq1 = old_cons.substr(old_cons.size() - overlap, overlap);   /* :99-100 */
                const std::string seq2 = cur.substr(0, overlap);
                if (upper_copy(seq1) != upper_copy(seq2)) {
                    int s1, s2;
                    if (overlap >= mer_size) { s1 = nb_solid_mers(seq1, *old_mers, mer_size); s2 = nb_solid_mers(seq2, *cur_mers, mer_size); }
                    else { s1 = nb_upper(seq1); s2 = nb_upper(seq2); }
                    if (s1 > s2) {                                                 /* :109-119 */
                        const size_t rl = std::min(seq1.size(), seq2.size());
                        const SwResult sub = ssw_align(seq1, seq2.substr(0, rl));
                        const unsigned cut = overlap - sub.ins + sub.del;
                        if (cut < cur.size()) cur = seq1 + cur.substr(cut);
                        else cur.clear();
                    }
                }
            }
        }
        if (!cur.empty()) {                                                        /* :124 */
            if (consensuses[i].size() >= mer_size) out.replace(beg, end - beg + 1, upper_copy(cur));   /* :125-129 */
            if (i + 1 < consensuses.size()) {                                      /* :130-135 */
                const long long np = (long long)cur_pos + (long long)piles_pos[i + 1].first - (long long)piles_pos[i].first -
                                     (long long)(end - beg + 1) + (long long)cur.size();
                cur_pos = (int)(int32_t)(uint32_t)np;
                old_cons = cur;
                old_mers = &solid[i];
                old_end = beg + (unsigned)cur.size() - 1;
            }
        }
    }
    return out;
}

std::string trim_read(const std::string& s, unsigned mer_size) {                  /* utils.cpp:96-128 */
    unsigned i = 0, n = 0;
    while (i < s.size() && n < mer_size) { n = is_upper(s[i]) ? n + 1 : 0; i++; }
    const unsigned beg = i - mer_size;
    long long k = (long long)s.size() - 1;
    n = 0;
    /* the reference loops on an unsigned `i >= 0`; it terminates only because an upper-case run exists whenever beg is valid.
       A read without such a run is undefined behaviour there; here it yields the empty string. */
    while (k >= 0 && n < mer_size) { n = is_upper(s[(size_t)k]) ? n + 1 : 0; k--; }
    if (n < mer_size) return std::string();
    const unsigned end = (unsigned)(k + mer_size);
    if (end > beg) return s.substr(beg, end - beg + 1);
    return std::string();
}

bool drop_read(const std::string& s) {                                            /* utils.cpp:60-73 */
    int n = 0;
    for (char c : s) n += ('A' <= c && c <= 'Z') ? 1 : 0;
    return (float)n / s.size() < 0.1;
}

} // namespace cwo
