/*
 * cw_hostio.cpp -- host feeders of the path: the read indexer and the PAF pile reader.
 *
 *   cw_index_reads      <- indexReads      (src/utils.cpp:166-205): FASTA/FASTQ -> 2-bit reads keyed by name
 *   cw_paf_next_pile    <- getNextReadPile (src/alignmentPiles.cpp:22-58) + Overlap(std::string) (src/Overlap.h:26-58)
 *   cw_paf_reformat / cw_paf_explode / cw_paf_merge <- the wrapper tools reformatPAF.cpp, explode.cpp, merge.cpp
 *
 * Plain host C++ (no device code); results are handed over as the same cw_read_set / cw_overlap structures the device
 * entry points take.  Behaviours kept on purpose (each pinned against the reference's own TUs in tests/test_hostio_ref.py):
 *   - names are the header up to the first blank; a later record with the same name replaces the earlier one
 *   - bases are upper-cased, then anything but A, C, G packs as T (utils.cpp:21-32, :189)
 *   - multi-line FASTA/FASTQ: sequence lines are joined until a line starting with '>' or '+'; for FASTQ the quality block is
 *     skipped by line count; the loop ends at the first empty header line
 *   - PAF ends are made inclusive on parse; a pile = consecutive lines with the same query name; it is sorted with the
 *     reference's own expression -- std::sort over reverse iterators with operator< on resMatches (unstable: the permutation
 *     of ties is whatever libstdc++'s introsort produces, reproduced here by running the same algorithm on the same keys) --
 *     and cut to max_support
 *   - a blank line closes the pile being read and is skipped, as the reference's driver loop ends up doing
 *   Files whose last line lacks a newline make the reference spin (a failed getline leaves its string untouched); here they
 *   simply end.
 */
#include "../../include/consent_amd.h"
#include "cw_env.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

struct cw_read_index {
    std::vector<std::string> names;
    std::vector<uint32_t> len;
    std::vector<uint64_t> word_off;
    std::vector<uint32_t> words;
    std::unordered_map<std::string, uint32_t> by_name;
};

namespace {

void pack_into(const std::string& seq, std::vector<uint32_t>& words) {
    const size_t nw = (seq.size() + 15) / 16;
    const size_t base = words.size();
    words.resize(base + nw, 0u);
    for (size_t j = 0; j < seq.size(); ++j) {
        const int c = toupper((unsigned char)seq[j]);
        const uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u;
        words[base + (j >> 4)] |= code << (30 - 2 * (j & 15));
    }
}

/* std::stoi as the reference uses it: leading blanks, optional sign, digits; anything after is ignored; no digits = error */
bool parse_int(const std::string& tok, long* out) {
    const char* s = tok.c_str();
    char* end = nullptr;
    const long v = strtol(s, &end, 10);
    if (end == s) return false;
    *out = v;
    return true;
}

/* One PAF line as the pile reader needs it: numbers parsed, names as spans of the mapped file, the target already looked up.
 * kind: 0 = an overlap, 1 = a blank line, 2 = malformed (reported when the reader reaches it, as a sequential parse would). */
struct PafRec {
    const char* q_name;
    uint32_t q_name_len;
    int32_t t_id;         /* index of the target name in the read index, -1 = unknown name */
    uint32_t q_len, q_start, q_end, strand, t_len, t_start, t_end, res_matches;
    uint32_t kind;
    bool operator<(const PafRec& o) const { return res_matches < o.res_matches; } /* Overlap.h:90-96 */
};

/* std::stoi as the reference uses it (strtol: leading blanks, optional sign, digits; anything after is ignored; no digits = error)
 * on a token of the mapped file, which is not NUL-terminated */
bool parse_int_span(const char* b, const char* e, long* out) {
    char buf[64];
    const size_t n = (size_t)(e - b);
    if (n >= 1 && n <= 9) {                      /* the usual token: nothing but digits (same value as strtol gives) */
        long v = 0;
        size_t i = 0;
        for (; i < n && (unsigned)(b[i] - '0') <= 9u; ++i) v = v * 10 + (b[i] - '0');
        if (i == n) { *out = v; return true; }
    }
    if (n < sizeof buf) {
        memcpy(buf, b, n);
        buf[n] = 0;
        char* end = nullptr;
        const long v = strtol(buf, &end, 10);
        if (end == buf) return false;
        *out = v;
        return true;
    }
    return parse_int(std::string(b, e), out);
}

/* Overlap(std::string) (src/Overlap.h:26-58): twelve tab-separated fields; ends made inclusive (:39, :49) */
void parse_paf_line(const char* b, const char* e, const cw_read_index* idx, PafRec* o) {
    o->kind = 2;
    const char* tb[12];
    const char* te[12];
    const char* p = b;
    for (int f = 0; f < 12; ++f) {
        if (p > e) return;                       /* fewer than twelve fields */
        const char* t = (const char*)memchr(p, '\t', (size_t)(e - p));
        tb[f] = p; te[f] = t ? t : e;
        p = t ? t + 1 : e + 1;
    }
    long v[9];
    const int num[9] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
    for (int i = 0; i < 9; ++i)
        if (!parse_int_span(tb[num[i]], te[num[i]], &v[i])) return;
    o->q_name = tb[0]; o->q_name_len = (uint32_t)(te[0] - tb[0]);
    o->q_len = (uint32_t)v[0]; o->q_start = (uint32_t)v[1]; o->q_end = (uint32_t)(v[2] - 1);
    o->strand = (te[4] - tb[4] == 1 && *tb[4] == '+') ? 0u : 1u;
    o->t_len = (uint32_t)v[3]; o->t_start = (uint32_t)v[4]; o->t_end = (uint32_t)(v[5] - 1);
    o->res_matches = (uint32_t)v[6];
    auto it = idx->by_name.find(std::string(tb[5], te[5]));
    o->t_id = it == idx->by_name.end() ? -1 : (int32_t)it->second;
    o->kind = 0;
}

} // namespace

/* The PAF is mapped and cut into blocks at line ends; parser threads take blocks in file order and stay at most `window` blocks
 * ahead of the reader, which walks the parsed records strictly in file order (so piles, blank-line handling and the place where a
 * malformed line is reported are those of a sequential getline loop). */
struct cw_paf_reader {
    const cw_read_index* idx = nullptr;
    uint32_t max_support = 0;
    bool bad = false;
    const char* data = nullptr;
    size_t size = 0;
    bool mapped = false;
    std::string owned;                       /* the whole input when it cannot be mapped (a pipe) */
    std::vector<size_t> cut;                 /* block b = [cut[b], cut[b+1]) */
    std::vector<std::vector<PafRec>> blocks;
    std::vector<char> ready;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_ready, cv_room;
    size_t next_block = 0, consumed = 0, window = 4;
    bool stop = false, failed = false;
    size_t cur_block = 0, cur_rec = 0;       /* reader position */
    std::vector<PafRec> pile;

    void parse_block(size_t b, std::vector<PafRec>& out) const {
        const char* p = data + cut[b];
        const char* end = data + cut[b + 1];
        out.reserve((size_t)(end - p) / 48 + 4);
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* le = nl ? nl : end;      /* a last line without a newline simply ends */
            PafRec r{};
            if (le == p) r.kind = 1;
            else parse_paf_line(p, le, idx, &r);
            out.push_back(r);
            p = nl ? nl + 1 : end;
        }
    }
    void parser_main() {
        for (;;) {
            size_t b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_room.wait(lk, [&] { return stop || next_block >= blocks.size() || next_block < consumed + window; });
                if (stop || next_block >= blocks.size()) return;
                b = next_block++;
            }
            std::vector<PafRec> recs;
            bool ok = true;
            try { parse_block(b, recs); } catch (...) { ok = false; }
            {
                std::lock_guard<std::mutex> lk(mu);
                blocks[b].swap(recs);
                ready[b] = 1;
                if (!ok) failed = true;
            }
            cv_ready.notify_all();
        }
    }
    /* the next record in file order, or nullptr at the end of the file */
    const PafRec* peek() {
        for (;;) {
            if (cur_block >= blocks.size()) return nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_ready.wait(lk, [&] { return ready[cur_block] != 0; });
                if (failed) throw std::bad_alloc();
            }
            if (cur_rec < blocks[cur_block].size()) return &blocks[cur_block][cur_rec];
            {
                std::lock_guard<std::mutex> lk(mu);
                std::vector<PafRec>().swap(blocks[cur_block]);
                consumed = ++cur_block;
                cur_rec = 0;
            }
            cv_room.notify_all();
        }
    }
    ~cw_paf_reader() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_room.notify_all();
        for (auto& t : threads) t.join();
        if (mapped && data) munmap((void*)data, size);
    }
};

extern "C" {

/* indexReads (src/utils.cpp:166-205), fast: the file is mapped; one sequential pass finds the records with the reference's own line
 * rules (below), then the bases are packed by all cores, each record into the words its place in the file gives it.
 *   - a record = a header line (first character dropped, cut at the first blank), one sequence line, and every following line that is
 *     not empty and does not start with '>' or '+'; a '+' line is followed by as many quality lines as there were sequence lines, then
 *     the next header; the loop ends at the first empty header line (or the end of the file)
 *   - a later record with the same name replaces the earlier one (its words stay where they were written, unreferenced)
 * Only '\n' ends a line: a '\r' is a base like any other non-ACG byte (T), as with std::getline. */
struct RecSpan { size_t seq_beg, seq_end; uint32_t len, id; uint64_t word_off; };

static void pack_span(const char* p, size_t beg, size_t end, uint32_t* words) {
    static const uint8_t* lut = [] {
        static uint8_t t[256];
        for (int c = 0; c < 256; ++c) { const int u = toupper(c); t[c] = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : 3; }
        return (const uint8_t*)t;
    }();
    uint32_t acc = 0, nb = 0;
    size_t w = 0;
    for (size_t i = beg; i < end; ++i) {
        const unsigned char c = (unsigned char)p[i];
        if (c == '\n') continue;
        acc |= (uint32_t)lut[c] << (30 - 2 * nb);
        if (++nb == 16) { words[w++] = acc; acc = 0; nb = 0; }
    }
    if (nb) words[w] = acc;
}

static int index_file(cw_read_index* ix, const char* path) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return CW_E_INVALID;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return CW_E_INVALID; }
    const size_t n = (size_t)sb.st_size;
    const char* p = n ? (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (n && p == (const char*)MAP_FAILED) return CW_E_INVALID;
    int rc = CW_OK;
    try {
        if (!ix->words.empty()) ix->words.pop_back(); /* the guard word of an earlier file */
        std::vector<RecSpan> recs;
        /* a "line" = [pos, eol); getline semantics: the last line may lack its newline; past the end every line is empty */
        auto eol_of = [&](size_t pos) { if (pos >= n) return n; const void* q = memchr(p + pos, '\n', n - pos); return q ? (size_t)((const char*)q - p) : n; };
        auto next_of = [&](size_t eol) { return eol < n ? eol + 1 : n; };
        size_t pos = 0, eol = eol_of(0);
        uint64_t woff = ix->words.size();
        while (eol > pos) { /* header.length() > 0 */
            size_t hb = pos + 1, he = eol;
            const void* sp = hb < he ? memchr(p + hb, ' ', he - hb) : nullptr;
            if (sp) he = (size_t)((const char*)sp - p);
            std::string header(p + hb, he > hb ? he - hb : 0);
            pos = next_of(eol); eol = eol_of(pos);
            RecSpan r;
            r.seq_beg = pos;
            size_t len = eol - pos;
            int nb_lines = 1;
            r.seq_end = eol;
            pos = next_of(eol); eol = eol_of(pos);
            while (eol > pos && p[pos] != '>' && p[pos] != '+') {
                len += eol - pos; nb_lines++;
                r.seq_end = eol;
                pos = next_of(eol); eol = eol_of(pos);
            }
            if (len > 0xFFFFFFFFull) { rc = CW_E_INVALID; break; }
            r.len = (uint32_t)len;
            auto it = ix->by_name.find(header);
            if (it == ix->by_name.end()) {
                r.id = (uint32_t)ix->names.size();
                ix->by_name.emplace(header, r.id);
                ix->names.push_back(header);
                ix->len.push_back(0);
                ix->word_off.push_back(0);
            } else r.id = it->second;                            /* index[header] = ... replaces */
            r.word_off = woff;
            woff += (len + 15) / 16;
            ix->len[r.id] = r.len;
            ix->word_off[r.id] = r.word_off;
            recs.push_back(r);
            if (eol > pos && p[pos] == '+') {                    /* FASTQ: skip the quality block by line count */
                pos = next_of(eol); eol = eol_of(pos);
                for (int i = 1; i < nb_lines; ++i) { pos = next_of(eol); eol = eol_of(pos); }
                pos = next_of(eol); eol = eol_of(pos);
            }
        }
        if (rc == CW_OK) {
            ix->words.resize((size_t)woff + 1, 0u);              /* + the guard word: readers may look one word past a sequence's last word */
            uint32_t* words = ix->words.data();
            unsigned nt = std::thread::hardware_concurrency();
            nt = nt < 1 ? 1 : nt > 16 ? 16 : nt;
            if (recs.size() < 64 || n < (1u << 20)) nt = 1;
            std::vector<std::thread> th;
            std::atomic<size_t> cursor{0};
            auto work = [&]() {
                for (;;) {
                    const size_t i0 = cursor.fetch_add(64);
                    if (i0 >= recs.size()) break;
                    const size_t i1 = i0 + 64 < recs.size() ? i0 + 64 : recs.size();
                    for (size_t i = i0; i < i1; ++i) pack_span(p, recs[i].seq_beg, recs[i].seq_end, words + recs[i].word_off);
                }
            };
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto& t : th) t.join();
        }
    } catch (...) {
        rc = CW_E_NOMEM;
    }
    if (n) munmap((void*)p, n);
    return rc;
}

int cw_index_reads(const char* path, cw_read_index** out) {
    if (!path || !out) return CW_E_INVALID;
    *out = nullptr;
    cw_read_index* ix = new (std::nothrow) cw_read_index();
    if (!ix) return CW_E_NOMEM;
    const int rc = index_file(ix, path);
    if (rc != CW_OK) { delete ix; return rc; }
    *out = ix;
    return CW_OK;
}

/* a second file into the same index, as runCorrection does with the proof file (CONSENT-correction.cpp:69-73): same names replace */
int cw_index_reads_append(cw_read_index* ix, const char* path) {
    if (!ix || !path) return CW_E_INVALID;
    return index_file(ix, path);
}

void cw_read_index_free(cw_read_index* ix) { delete ix; }

uint32_t cw_read_index_count(const cw_read_index* ix) { return ix ? (uint32_t)ix->names.size() : 0u; }

int cw_read_index_view(const cw_read_index* ix, cw_read_set* view, uint64_t* n_words) {
    if (!ix || !view) return CW_E_INVALID;
    view->n_reads = (uint32_t)ix->names.size();
    view->read_len = ix->len.data();
    view->read_word_off = ix->word_off.data();
    view->bases = ix->words.data();
    if (n_words) *n_words = ix->words.size();
    return CW_OK;
}

int32_t cw_read_index_find(const cw_read_index* ix, const char* name) {
    if (!ix || !name) return -1;
    auto it = ix->by_name.find(name);
    return it == ix->by_name.end() ? -1 : (int32_t)it->second;
}

const char* cw_read_index_name(const cw_read_index* ix, uint32_t id) {
    return (ix && id < ix->names.size()) ? ix->names[id].c_str() : nullptr;
}

int cw_paf_open(const char* path, const cw_read_index* idx, uint32_t max_support, cw_paf_reader** out) {
    if (!path || !idx || !out) return CW_E_INVALID;
    *out = nullptr;
    if (max_support < 1) return CW_E_INVALID; /* the reference cuts the pile to maxSupport and then reads its first overlap (alignmentPiles.cpp:52-57): -S 0 is undefined there */
    cw_paf_reader* r = new (std::nothrow) cw_paf_reader();
    if (!r) return CW_E_NOMEM;
    r->idx = idx; r->max_support = max_support;
    try {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) { delete r; return CW_E_INVALID; }
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            r->size = (size_t)st.st_size;
            if (r->size) {
                void* m = mmap(nullptr, r->size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) { close(fd); delete r; return CW_E_INVALID; }
                (void)madvise(m, r->size, MADV_SEQUENTIAL);
                r->data = (const char*)m; r->mapped = true;
            }
        } else {                                  /* a pipe or a device: read it to its end */
            char buf[1 << 16];
            ssize_t k;
            while ((k = read(fd, buf, sizeof buf)) > 0) r->owned.append(buf, (size_t)k);
            r->data = r->owned.data(); r->size = r->owned.size();
        }
        close(fd);
        /* blocks of ~4 MiB, cut after a newline */
        size_t block = 4u << 20;
        if (const char* env = getenv("CW_PAF_BLOCK")) { const long v = atol(env); if (v >= 64) block = (size_t)v; }
        r->cut.push_back(0);
        while (r->cut.back() < r->size) {
            size_t e = r->cut.back() + block;
            if (e >= r->size) e = r->size;
            else {
                const char* nl = (const char*)memchr(r->data + e, '\n', r->size - e);
                e = nl ? (size_t)(nl - r->data) + 1 : r->size;
            }
            r->cut.push_back(e);
        }
        const size_t nb = r->cut.size() - 1;
        r->blocks.resize(nb);
        r->ready.assign(nb, 0);
        unsigned nt = std::thread::hardware_concurrency();
        if (nt > 8) nt = 8;
        if (const char* env = getenv("CW_HOST_THREADS")) { const int v = atoi(env); if (v >= 1 && v <= 64) nt = (unsigned)v; }
        if (nt < 1) nt = 1;
        if (nt > nb) nt = (unsigned)nb;
        r->window = 2 * (size_t)nt + 1;
        for (unsigned t = 0; t < nt; ++t) r->threads.emplace_back([r] { r->parser_main(); });
    } catch (...) { delete r; return CW_E_NOMEM; }
    *out = r;
    return CW_OK;
}

void cw_paf_close(cw_paf_reader* r) { delete r; }

int cw_paf_next_pile(cw_paf_reader* r, uint32_t* tpl_read, uint32_t* tpl_len, cw_overlap* out, uint32_t* res_matches, uint32_t cap,
                     uint32_t* n) {
    if (!r || !tpl_read || !n) return CW_E_INVALID;
    *n = 0;
    if (r->bad) return CW_E_INVALID;
    try {
        std::vector<PafRec>& cur = r->pile;
        cur.clear();
        /* One record of look-ahead instead of the reference's seekg(-len-1): the same piles for any file whose lines end in a
           newline.  A blank line closes the current pile and is skipped (the reference returns an empty pile there and its
           driver asks again, CONSENT-correction.cpp:88-91). */
        for (;;) {
            const PafRec* al = r->peek();
            if (!al) break;
            if (al->kind == 1) { ++r->cur_rec; if (!cur.empty()) break; else continue; }
            if (al->kind == 2) { r->bad = true; return CW_E_INVALID; }
            if (cur.empty() || (al->q_name_len == cur[0].q_name_len && memcmp(al->q_name, cur[0].q_name, al->q_name_len) == 0)) { cur.push_back(*al); ++r->cur_rec; }
            else break;
        }
        if (cur.empty()) return CW_OK; /* end of the stream */
        std::sort(cur.rbegin(), cur.rend());                   /* alignmentPiles.cpp:43 / :53 */
        if (cur.size() > r->max_support) cur.resize(r->max_support);
        const int32_t q = cw_read_index_find(r->idx, std::string(cur[0].q_name, cur[0].q_name_len).c_str());
        if (q < 0) { r->bad = true; return CW_E_INVALID; }
        *tpl_read = (uint32_t)q;
        if (tpl_len) *tpl_len = cur[0].q_len;
        *n = (uint32_t)cur.size();
        if (cur.size() > cap || !out) return CW_E_CAPACITY;  /* the pile is consumed: size the buffer for max_support */
        for (size_t i = 0; i < cur.size(); ++i) {
            const int32_t t = cur[i].t_id;
            if (t < 0) { r->bad = true; return CW_E_INVALID; }
            /* the reference clamps target coordinates with the PAF's tLength (alignmentWindows.cpp:121-129) and slices the indexed read;
               downstream (cw_extract_piles_device) there is only the indexed length, so a PAF that disagrees with the read file is
               refused here instead of being corrected differently */
            if (cur[i].t_len != r->idx->len[(size_t)t]) { r->bad = true; return CW_E_INVALID; }
            out[i].q_start = cur[i].q_start; out[i].q_end = cur[i].q_end;
            out[i].t_read = (uint32_t)t; out[i].t_start = cur[i].t_start; out[i].t_end = cur[i].t_end;
            out[i].strand = cur[i].strand;
            if (res_matches) res_matches[i] = cur[i].res_matches;
        }
        return CW_OK;
    } catch (...) {
        r->bad = true;
        return CW_E_NOMEM;
    }
}

/* ---- wrapper plumbing: what CONSENT-correct / CONSENT-polish run around the binary ------------------------------ */

/* reformatPAF (src/reformatPAF.cpp:22-46): swap query and target columns of every line (fields 6-9, 5, 1-4, then the rest). */
int cw_paf_reformat(const char* in_path, const char* out_path) {
    if (!in_path || !out_path) return CW_E_INVALID;
    std::ifstream in(in_path);
    std::ofstream out(out_path);
    if (!in || !out) return CW_E_INVALID;
    try {
        std::string line;
        while (std::getline(in, line)) {
            std::vector<std::string> v;
            size_t p = 0;
            while (p <= line.size()) {
                const size_t e = line.find('\t', p);
                if (e == std::string::npos) { if (p < line.size()) v.push_back(line.substr(p)); break; }   /* getline-style split: no trailing empty field */
                v.push_back(line.substr(p, e - p));
                p = e + 1;
            }
            if (v.size() < 9) return CW_E_INVALID;
            out << v[5] << '\t' << v[6] << '\t' << v[7] << '\t' << v[8] << '\t' << v[4] << '\t' << v[0] << '\t' << v[1] << '\t' << v[2] << '\t' << v[3];
            for (size_t i = 9; i < v.size(); ++i) out << '\t' << v[i];
            out << '\n';
        }
    } catch (...) { return CW_E_NOMEM; }
    return out.good() ? CW_OK : CW_E_INTERNAL;
}

/* explode (what src/explode.cpp:14-51 produces): split a PAF into prefix_1, prefix_2, ... so that inside one file every query name forms a
 * single run of lines; a run whose name already ended a run of the current file opens the next file.  Input ends at the first empty line.
 *
 * One pass over the mapped file: runs are contiguous in the input, so a run is a byte range [run_beg, run_end) of the mapping and goes to the
 * current file with one write -- no line is copied or re-assembled; the names that have closed a run in the current file are views into the
 * mapping.  One behaviour of the reference program is kept because its output files are the contract (tests/test_wrappers_ref.py): a line whose
 * name is empty takes the next line into its run whatever that line's name (the reference uses the empty string as "no previous name").  A
 * last line without a newline gets one (the reference program does not end on such a file: explode.cpp:25,30 keep reading at end-of-file). */
int cw_paf_explode(const char* in_path, const char* out_prefix, uint32_t* n_files) {
    if (!in_path || !out_prefix) return CW_E_INVALID;
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return CW_E_INVALID;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return CW_E_INVALID; }
    const size_t n = (size_t)sb.st_size;
    const char* const base = n ? (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (n && base == (const char*)MAP_FAILED) return CW_E_INVALID;
    int rc = CW_OK;
    FILE* out = nullptr;
    uint32_t nb = 0;
    auto open_next = [&]() -> bool {
        if (out && fclose(out) != 0) rc = CW_E_INTERNAL;
        out = fopen((std::string(out_prefix) + "_" + std::to_string(++nb)).c_str(), "wb");
        return out != nullptr;
    };
    auto put = [&](const char* b, const char* e) { /* a run's bytes; the file's last line may lack its newline */
        if (e > b && fwrite(b, 1, (size_t)(e - b), out) != (size_t)(e - b)) rc = CW_E_INTERNAL;
        if (e > b && e[-1] != '\n' && fputc('\n', out) == EOF) rc = CW_E_INTERNAL;
    };
    try {
        if (!open_next()) { if (n) munmap((void*)base, n); return CW_E_INVALID; }
        std::unordered_set<std::string_view> closed; /* names whose run in the current file has ended */
        const char* const end = base + n;
        const char* run_beg = base;     /* first byte of the run being collected */
        std::string_view run_name;      /* its name; empty: none yet (or an empty one: see above) */
        const char* p = base;
        while (p < end && *p != '\n') { /* an empty line ends the input */
            const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* const line_end = eol ? eol : end;
            const char* tab = (const char*)memchr(p, '\t', (size_t)(line_end - p));
            const std::string_view name(p, (size_t)((tab ? tab : line_end) - p));
            if (!run_name.empty() && name != run_name) { /* the run in front of this line is complete */
                put(run_beg, p);
                closed.insert(run_name);
                run_beg = p;
                if (closed.count(name)) { /* this name came back: its lines start the next file */
                    closed.clear();
                    if (!open_next()) { rc = CW_E_INVALID; break; }
                }
            }
            run_name = name;
            p = eol ? eol + 1 : end;
        }
        if (rc == CW_OK) put(run_beg, p);
    } catch (...) { rc = CW_E_NOMEM; }
    if (out && fclose(out) != 0 && rc == CW_OK) rc = CW_E_INTERNAL;
    if (n) munmap((void*)base, n);
    if (rc == CW_OK && n_files) *n_files = nb;
    return rc;
}

/* merge (src/merge.cpp:29-65): for every header line (its first character dropped) copy, file after file, the run of lines whose
 * first column equals it -- all overlaps of one read end up contiguous, in header order. */
int cw_paf_merge(const char* out_path, const char* headers_path, const char* const* in_paths, uint32_t n_in) {
    if (!out_path || !headers_path || (n_in && !in_paths)) return CW_E_INVALID;
    std::ofstream out(out_path);
    std::ifstream headers(headers_path);
    if (!out || !headers) return CW_E_INVALID;
    try {
        std::vector<std::ifstream> files(n_in);
        std::vector<std::string> pending(n_in);
        std::vector<char> has_pending(n_in, 0);
        for (uint32_t i = 0; i < n_in; ++i) { files[i].open(in_paths[i]); if (!files[i]) return CW_E_INVALID; }
        std::string header, line;
        while (std::getline(headers, header)) {
            header = header.empty() ? header : header.substr(1);
            for (uint32_t i = 0; i < n_in; ++i) {
                for (;;) {
                    if (has_pending[i]) { line.swap(pending[i]); has_pending[i] = 0; }
                    else { line.clear(); if (!std::getline(files[i], line)) break; }
                    if (line.substr(0, line.find('\t')) == header) out << line << '\n';
                    else { pending[i].swap(line); has_pending[i] = 1; break; }   /* the reference seeks back one line */
                }
            }
        }
    } catch (...) { return CW_E_NOMEM; }
    return out.good() ? CW_OK : CW_E_INTERNAL;
}

} // extern "C"
