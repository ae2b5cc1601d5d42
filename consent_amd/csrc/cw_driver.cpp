/*
 * cw_driver.cpp -- the loop of CONSENT's two drivers behind one C entry point, cw_run_correction:
 *
 *   runCorrection / processRead    (src/CONSENT-correction.cpp:62-135, :19-58)   polishing = 0
 *   runCorrection / processContig  (src/CONSENT-polishing.cpp:107-135, :21-105)  polishing = 1 (never trims, :19)
 *
 * Same inputs (the flags of src/main.cpp:29-76), same output: FASTA records ">name\nsequence\n" in PAF order, nothing for a read
 * without windows or dropped by the 10 % rule (CONSENT-correction.cpp:23-25, :52-56, :101-103).
 *
 * Shape: ONE host process; the calling thread reads the PAF pile by pile (cw_paf_next_pile), cuts the windows
 * (cw_window_positions) and packs piles into jobs of ~32k windows; one worker thread per device owns an engine and a copy of the
 * 2-bit read set and pulls jobs from a bounded queue (dynamic: a device that finishes early takes the next job); per job the piles
 * never leave the device: cw_extract_piles_device -> cw_plan_results_device -> cw_run_device -> cw_stitch_device, and only the
 * corrected reads come back.  An emitter thread writes finished jobs in submission order, so the output does not depend on the
 * number of devices.  No collective: jobs are independent.
 *
 * Host code only (device memory through the HIP runtime API); everything that computes is behind the C ABI of consent_amd.h.
 */
#include "cw_internal.h"
#include "cw_private.h"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Job {
    uint64_t seq = 0;
    std::vector<cw_overlap> ovl;
    std::vector<cw_window_job> wj;
    std::vector<uint32_t> win_pos;   /* (beg, end) per window */
    std::vector<cw_stitch_read> sr;  /* one per pile */
    uint64_t cost = 0;               /* sum over windows of (overlaps + 1): what the shard balance goes by */
    /* results */
    std::vector<std::string> out;    /* per pile; "" = no record */
    int rc = CW_OK;
    std::string err;
    int device = -1;
    double ms_extract = 0, ms_consensus = 0, ms_stitch = 0;
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return CW_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 2 + 256; /* half again: a regrowth is a hipFree + hipMalloc, both of which wait for the device (and for the other worker's kernels on it) */
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return CW_E_NOMEM; }
        cap = want;
        return CW_OK;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T* as() const { return (T*)p; }
};

struct Shared {
    const cw_driver_args* a = nullptr;
    const cw_read_index* index = nullptr;
    cw_read_set host_reads{};
    uint64_t read_words = 0;
    double t_start = 0; /* now_ms() when the run began (inspection output) */
    bool do_trim = true;
    bool skip_on_capacity = false;
    bool dry = false; /* CW_DRIVER_DRY: workers take the jobs and throw them away -- what the host side alone can feed (no device is touched) */
    /* job queue (producer -> workers) */
    std::mutex mu;
    std::condition_variable cv_work, cv_room, cv_done;
    std::deque<Job*> queue;
    size_t queue_cap = 4;
    bool closed = false;
    /* finished jobs, keyed by sequence number (workers -> emitter) */
    std::map<uint64_t, Job*> finished;
    std::atomic<bool> abort{false}; /* written under mu, also read outside it by the producer loop */
    int first_error = CW_OK;
    std::string first_error_msg;
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* An error leaves process() while copies may still be queued on `st` whose host side are locals of process() and vectors of the Job:
   wait for the stream before they go out of scope (a late copy would write into freed memory -- exactly when the device misbehaves). */
#define DRV_HIP(expr, job, what)                                                        \
    do {                                                                                \
        if ((expr) != hipSuccess) { (job).rc = CW_E_NO_DEVICE; (job).err = what; if (st) (void)hipStreamSynchronize(st); return; } \
    } while (0)
#define DRV_RC(expr, job, what)                                                         \
    do {                                                                                \
        const int _rc = (expr);                                                         \
        if (_rc != CW_OK) { (job).rc = _rc; (job).err = what; if (st) (void)hipStreamSynchronize(st); return; } \
    } while (0)

struct Worker {
    int device = 0; /* the device this worker serves: read-set ownership, statistics */
    int phys = 0;   /* the HIP device behind it (== device, except under the test aid CW_VIRTUAL_DEVICES: several logical devices on one GPU) */
    cw_engine* eng = nullptr;
    hipStream_t st = nullptr;
    cw_read_set dev_reads{};
    DevBuf rd_len, rd_off, rd_bases;
    void* pin = nullptr; /* pinned landing buffer for the corrected reads */
    size_t pin_cap = 0;
    DevBuf ovl, wj, pos, sr, b_wfs, b_len, b_off, b_bases, r_cons, r_coff, r_clen, r_stat, r_solid, r_soff, r_slen, o_buf, o_off, o_len, o_stat;
    uint64_t windows = 0, reads = 0, jobs = 0;
    double ms_extract = 0, ms_consensus = 0, ms_stitch = 0;

    /* The 2-bit read set is uploaded once per DEVICE: the first worker on a device owns the copy and publishes it, the others on that
       device (two workers per GPU by default) wait for it and use the same buffers. */
    std::promise<int> reads_ready;            /* set by an owner, in every path out of init() */
    std::shared_future<int> reads_of_owner;   /* valid in a worker that borrows another worker's copy */
    const Worker* owner = nullptr;

    int upload_reads(const Shared& sh) {
        const uint32_t n = sh.host_reads.n_reads;
        int rc;
        if ((rc = rd_len.ensure((size_t)n * 4 + 4)) || (rc = rd_off.ensure((size_t)n * 8 + 8)) || (rc = rd_bases.ensure((size_t)sh.read_words * 4 + 8))) return rc;
        if (hipMemcpy(rd_len.p, sh.host_reads.read_len, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(rd_off.p, sh.host_reads.read_word_off, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(rd_bases.p, sh.host_reads.bases, (size_t)sh.read_words * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset((uint8_t*)rd_bases.p + (size_t)sh.read_words * 4, 0, 8) != hipSuccess)
            return CW_E_NO_DEVICE;
        dev_reads.n_reads = n;
        dev_reads.read_len = rd_len.as<uint32_t>();
        dev_reads.read_word_off = rd_off.as<uint64_t>();
        dev_reads.bases = rd_bases.as<uint32_t>();
        return CW_OK;
    }

    int init(const Shared& sh) {
        int rc = CW_OK;
        if (sh.dry) { if (!owner) reads_ready.set_value(CW_OK); return CW_OK; }
        if (hipSetDevice(phys) != hipSuccess) rc = CW_E_NO_DEVICE;
        if (!owner) { /* the copy first: the borrowers' engines are being created meanwhile */
            if (rc == CW_OK) rc = upload_reads(sh);
            reads_ready.set_value(rc);
        }
        if (rc != CW_OK) return rc;
        const cw_driver_args& a = *sh.a;
        cw_params prm{a.mer_size, a.solid_thresh, a.common_kmers, a.min_anchors, a.max_msa};
        rc = cw_create(&prm, phys, &eng);
        if (rc != CW_OK) return rc;
        /* templates are windows: at most window_size bases -- longer than the default plan assumes, or shorter (beyond what the engine takes at all, every
           window reports its capacity as before: CW_WHY_TEMPLATE) */
        (void)cw_configure(eng, a.window_size);
        if (hipStreamCreate(&st) != hipSuccess) return CW_E_NO_DEVICE;
        if (owner) {
            if ((rc = reads_of_owner.get()) != CW_OK) return rc;
            dev_reads = owner->dev_reads;
        }
        return CW_OK;
    }

    void close() {
        if (!eng && !st && !pin) return; /* nothing was created (dry run, or init failed early) */
        (void)hipSetDevice(phys);
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
        if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
        if (eng) { cw_destroy(eng); eng = nullptr; }
    }

    /* one job, start to end, on this worker's device */
    void process(const Shared& sh, Job& j) {
        const cw_driver_args& a = *sh.a;
        const uint32_t n_win = (uint32_t)j.wj.size(), n_piles = (uint32_t)j.sr.size();
        j.out.assign(n_piles, std::string());
        j.device = device;
        if (n_win == 0 || n_piles == 0) return;
        if (sh.dry) { windows += n_win; reads += n_piles; jobs++; return; }
        const double t0 = now_ms();
        DRV_HIP(hipSetDevice(phys), j, "hipSetDevice");
        DRV_RC(ovl.ensure(j.ovl.size() * sizeof(cw_overlap) + 64), j, "device memory (overlaps)");
        DRV_RC(wj.ensure((size_t)n_win * sizeof(cw_window_job)), j, "device memory (window jobs)");
        DRV_RC(pos.ensure((size_t)n_win * 8), j, "device memory (window positions)");
        DRV_RC(sr.ensure((size_t)n_piles * sizeof(cw_stitch_read)), j, "device memory (reads)");
        if (!j.ovl.empty()) DRV_HIP(hipMemcpyAsync(ovl.p, j.ovl.data(), j.ovl.size() * sizeof(cw_overlap), hipMemcpyHostToDevice, st), j, "H2D overlaps");
        DRV_HIP(hipMemcpyAsync(wj.p, j.wj.data(), (size_t)n_win * sizeof(cw_window_job), hipMemcpyHostToDevice, st), j, "H2D jobs");
        DRV_HIP(hipMemcpyAsync(pos.p, j.win_pos.data(), (size_t)n_win * 8, hipMemcpyHostToDevice, st), j, "H2D positions");
        DRV_HIP(hipMemcpyAsync(sr.p, j.sr.data(), (size_t)n_piles * sizeof(cw_stitch_read), hipMemcpyHostToDevice, st), j, "H2D reads");

        /* ---- piles, cut on the device ----
           A job's windows are extracted in slices (the extraction keeps one descriptor per window and overlap of its pile: a slice stays
           under 2^26 of them) into ONE batch in device memory -- every slice writes window -> sequence and sequence -> word offsets that
           start at zero, which cw_add_offsets_device then moves to the slice's place -- and corrected in runs of at most CW_MAX_BATCH_WINDOWS
           windows over that batch (a run names its windows by pointer; sequence indices and word offsets are the batch's), results into
           one set of arrays planned for the whole job.  The re-assembly then walks each read's windows over the whole job, so a read --
           or a contig of any length -- may span slices and runs.  (Usually: one slice, one run.) */
        uint64_t desc_budget = 1ull << 26;
        uint32_t run_windows = CW_MAX_BATCH_WINDOWS;
        if (const char* env = CW_AID_ENV("CW_DRIVER_SLICE_DESC")) { const long long v = atoll(env); if (v >= 1) desc_budget = (uint64_t)v; }           /* test aids */
        if (const char* env = CW_AID_ENV("CW_DRIVER_RUN_WINDOWS")) { const long v = atol(env); if (v >= 1 && v < (long)run_windows) run_windows = (uint32_t)v; }
        struct Slice { uint32_t w0, w1, n_seqs; uint64_t n_words, seq_base, word_base; };
        std::vector<Slice> slices;
        uint64_t tot_seqs = 0, tot_words = 0;
        for (uint32_t w0 = 0; w0 < n_win;) {
            uint64_t d = 0;
            uint32_t w1 = w0;
            while (w1 < n_win && w1 - w0 < run_windows && (w1 == w0 || d + j.wj[w1].ovl_count + 1 <= desc_budget)) { d += j.wj[w1].ovl_count + 1; ++w1; }
            Slice sl{w0, w1, 0, 0, tot_seqs, tot_words};
            const int rc = cw_extract_impl(eng, &dev_reads, ovl.as<cw_overlap>(), j.ovl.size(), wj.as<cw_window_job>() + w0, j.wj.data() + w0, w1 - w0, a.mer_size, nullptr, nullptr,
                                           nullptr, nullptr, 0, 0, &sl.n_seqs, &sl.n_words, st);
            if (rc != CW_OK && rc != CW_E_CAPACITY) { j.rc = rc; j.err = "cw_extract_piles_device (sizing)"; (void)hipStreamSynchronize(st); return; }
            tot_seqs += sl.n_seqs; tot_words += sl.n_words;
            slices.push_back(sl);
            w0 = w1;
        }
        if (tot_seqs > 0xFFFFFFF0ull) { j.rc = CW_E_CAPACITY; j.err = "more than 2^32 pile sequences in one job"; (void)hipStreamSynchronize(st); return; }
        DRV_RC(b_wfs.ensure((size_t)(n_win + 1) * 4), j, "device memory (batch)");
        DRV_RC(b_len.ensure((size_t)tot_seqs * 4 + 4), j, "device memory (batch)");
        DRV_RC(b_off.ensure((size_t)tot_seqs * 8 + 8), j, "device memory (batch)");
        DRV_RC(b_bases.ensure((size_t)tot_words * 4 + 8), j, "device memory (batch)");
        DRV_HIP(hipMemsetAsync((uint8_t*)b_bases.p + (size_t)tot_words * 4, 0, 8, st), j, "memset");
        for (const Slice& sl : slices) {
            uint32_t ns = 0; uint64_t nw = 0;
            DRV_RC(cw_extract_impl(eng, &dev_reads, ovl.as<cw_overlap>(), j.ovl.size(), wj.as<cw_window_job>() + sl.w0, j.wj.data() + sl.w0, sl.w1 - sl.w0, a.mer_size,
                                   b_wfs.as<uint32_t>() + sl.w0, b_len.as<uint32_t>() + sl.seq_base, b_off.as<uint64_t>() + sl.seq_base, b_bases.as<uint32_t>() + sl.word_base,
                                   sl.n_seqs, sl.n_words, &ns, &nw, st),
                   j, "cw_extract_piles_device");
            DRV_RC(cw_add_offsets_device(b_wfs.as<uint32_t>() + sl.w0, (uint64_t)(sl.w1 - sl.w0) + 1, (uint32_t)sl.seq_base, b_off.as<uint64_t>() + sl.seq_base, sl.n_seqs, sl.word_base, st),
                   j, "cw_add_offsets_device");
        }
        const uint32_t n_seqs = (uint32_t)tot_seqs;
        const uint64_t n_words = tot_words;
        cw_batch batch{n_win, n_seqs, n_words, b_wfs.as<uint32_t>(), b_len.as<uint32_t>(), b_off.as<uint64_t>(), b_bases.as<uint32_t>()};
        const double t1 = now_ms();

        /* ---- consensus per window ---- */
        const double t_a0 = now_ms();
        DRV_RC(r_coff.ensure((size_t)(n_win + 1) * 8), j, "device memory (results)");
        DRV_RC(r_soff.ensure((size_t)(n_win + 1) * 8), j, "device memory (results)");
        uint64_t cons_total = 0, solid_total = 0;
        DRV_RC(cw_plan_results_device(eng, &batch, r_coff.as<uint64_t>(), r_soff.as<uint64_t>(), &cons_total, &solid_total, st), j, "cw_plan_results_device");
        DRV_RC(r_cons.ensure(cons_total + 16), j, "device memory (results)");
        DRV_RC(r_solid.ensure(solid_total * 4 + 16), j, "device memory (results)");
        DRV_RC(r_clen.ensure((size_t)n_win * 4), j, "device memory (results)");
        DRV_RC(r_slen.ensure((size_t)n_win * 4), j, "device memory (results)");
        DRV_RC(r_stat.ensure(n_win), j, "device memory (results)");
        DRV_HIP(hipMemsetAsync(r_stat.p, 0xFF, n_win, st), j, "memset");
        cw_result res{r_cons.as<char>(), r_coff.as<uint64_t>(), r_clen.as<uint32_t>(), r_stat.as<uint8_t>(), r_solid.as<uint32_t>(), r_soff.as<uint64_t>(), r_slen.as<uint32_t>()};
        const double t_alloc = now_ms() - t_a0; /* result planning + allocation (inspection) */
        for (size_t s0 = 0; s0 < slices.size();) { /* runs = whole slices, at most run_windows windows */
            size_t s1 = s0 + 1;
            while (s1 < slices.size() && slices[s1].w1 - slices[s0].w0 <= run_windows) ++s1;
            const uint32_t w0 = slices[s0].w0, w1 = slices[s1 - 1].w1;
            uint64_t rs = 0, rw = 0;
            for (size_t x = s0; x < s1; ++x) { rs += slices[x].n_seqs; rw += slices[x].n_words; }
            cw_batch run{w1 - w0, (uint32_t)rs, rw, b_wfs.as<uint32_t>() + w0, b_len.as<uint32_t>(), b_off.as<uint64_t>(), b_bases.as<uint32_t>()};
            cw_result rres{r_cons.as<char>(), r_coff.as<uint64_t>() + w0, r_clen.as<uint32_t>() + w0, r_stat.as<uint8_t>() + w0, r_solid.as<uint32_t>(), r_soff.as<uint64_t>() + w0,
                           r_slen.as<uint32_t>() + w0};
            #ifdef CW_DRIVER_NO_SYNC /* experiment: without the wait + capacity check between consensus and re-assembly */
            DRV_RC(cw_run_device(eng, &run, &rres, st), j, "cw_run_device");
#else
            DRV_RC(cw_run_device_sync(eng, &run, &rres, st), j, "cw_run_device");
#endif
            s0 = s1;
        }

        /* ---- re-assembly per read ---- */
        std::vector<uint64_t> out_off((size_t)n_piles + 1, 0);
        for (uint32_t i = 0; i < n_piles; ++i) out_off[i + 1] = out_off[i] + 2ull * sh.host_reads.read_len[j.sr[i].read] + 1024ull;
        DRV_RC(o_off.ensure((size_t)(n_piles + 1) * 8), j, "device memory (output)");
        DRV_RC(o_buf.ensure(out_off[n_piles] + 16), j, "device memory (output)");
        DRV_RC(o_len.ensure((size_t)n_piles * 4), j, "device memory (output)");
        DRV_RC(o_stat.ensure(n_piles), j, "device memory (output)");
        DRV_HIP(hipMemcpyAsync(o_off.p, out_off.data(), (size_t)(n_piles + 1) * 8, hipMemcpyHostToDevice, st), j, "H2D offsets");
        DRV_HIP(hipMemsetAsync(o_len.p, 0, (size_t)n_piles * 4, st), j, "memset");
        DRV_HIP(hipMemsetAsync(o_stat.p, 0xFF, n_piles, st), j, "memset");
        double t2 = 0;
        if (getenv("CW_DRIVER_TIMING")) { DRV_HIP(hipStreamSynchronize(st), j, "sync"); t2 = now_ms(); }
        DRV_RC(cw_stitch_device(eng, &dev_reads, sr.as<cw_stitch_read>(), n_piles, pos.as<uint32_t>(), &batch, &res, a.window_size, a.window_overlap, sh.do_trim ? 1 : 0,
                                o_buf.as<char>(), o_off.as<uint64_t>(), o_len.as<uint32_t>(), o_stat.as<uint8_t>(), st),
               j, "cw_stitch_device");
        std::vector<uint32_t> h_len(n_piles);
        std::vector<uint8_t> h_st(n_piles), h_wst(n_win);
        DRV_HIP(hipMemcpyAsync(h_len.data(), o_len.p, (size_t)n_piles * 4, hipMemcpyDeviceToHost, st), j, "D2H lengths");
        DRV_HIP(hipMemcpyAsync(h_st.data(), o_stat.p, n_piles, hipMemcpyDeviceToHost, st), j, "D2H read status");
        DRV_HIP(hipMemcpyAsync(h_wst.data(), r_stat.p, n_win, hipMemcpyDeviceToHost, st), j, "D2H window status");
        DRV_HIP(hipStreamSynchronize(st), j, "kernels");
        const double t3 = now_ms();
        if (t2 == 0) t2 = t3;
        if (getenv("CW_DRIVER_PROFILE")) { /* inspection: how the job's POA tasks spread over the tiers, and where tier L's time went */
            uint32_t c[30] = {0}; unsigned long long pr[72] = {0}; /* cw_private.h: words 6.. = n_tier[6], 18.. = n_over[6] */
            if (cw_debug_profile(eng, c, 30, pr, 72, nullptr, nullptr) == CW_OK)
                fprintf(stderr, "[job %llu] windows %u tasks %u members %u | routed Q %u M1 %u M2 %u L %u, outgrew into S %u L %u G %u | L Mcycles meta %.0f fill %.0f trace %.0f merge %.0f, rows %llu chunk-rows %llu, longest task M1 %.1f M2 %.1f L %.1f Mcycles\n",
                        (unsigned long long)j.seq, n_win, c[0], c[1], c[6], c[7], c[8], c[9], c[18], c[21], c[22], pr[23] / 1e6, pr[24] / 1e6, pr[25] / 1e6, pr[26] / 1e6,
                        pr[47], pr[46], pr[37] / 1e6, pr[38] / 1e6, pr[39] / 1e6);
            if (getenv("CW_TASK_TRACE")) { /* timeline of the slab tiers: when their tasks ran (10 ns units since the tier sort) */
                std::vector<uint32_t> tr((size_t)c[0] * 12 + 12);
                uint32_t nt = 0;
                if (cw_debug_task_trace(eng, c[0], tr.data(), &nt) == CW_OK) {
                    for (uint32_t tier = 1; tier <= 3; ++tier) {
                        double busy = 0, first = 1e30, last = 0, longest = 0; uint32_t cnt = 0;
                        for (uint32_t i = 0; i < nt; ++i) {
                            const uint32_t* q = &tr[(size_t)i * 12];
                            if ((q[10] & 0xFF) != tier || q[9] == 0) continue;
                            const double st_ = q[8] * 1e-5, du = q[9] * 1e-5;
                            busy += du; first = st_ < first ? st_ : first; last = st_ + du > last ? st_ + du : last; longest = du > longest ? du : longest; ++cnt;
                        }
                        if (!cnt) continue;
                        int occ[9] = {0};
                        for (uint32_t i = 0; i < nt; ++i) {
                            const uint32_t* q = &tr[(size_t)i * 12];
                            if ((q[10] & 0xFF) != tier || q[9] == 0) continue;
                            const double st_ = q[8] * 1e-5, en = st_ + q[9] * 1e-5;
                            for (int k = 1; k <= 9; ++k) { const double at = last * k / 10.0; if (st_ < at && en > at) occ[k - 1]++; }
                        }
                        if (tier == 3) { /* the three runs that ended last */
                            for (int rep = 0; rep < 3; ++rep) {
                                double best = -1; uint32_t bi = 0;
                                for (uint32_t i = 0; i < nt; ++i) {
                                    const uint32_t* q = &tr[(size_t)i * 12];
                                    if ((q[10] & 0xFF) != tier || q[9] == 0) continue;
                                    const double en = (q[8] + (double)q[9]) * 1e-5;
                                    if (en > best && en < last + 1 - rep * 1e-9 && (rep == 0 || en < last)) { best = en; bi = i; }
                                }
                                const uint32_t* q = &tr[(size_t)bi * 12];
                                fprintf(stderr, "[job %llu]    ends %.2f start %.2f dur %.2f ms members %u max_len %u mean_len %u rc %u pass %u wave %u\n", (unsigned long long)j.seq, best,
                                        q[8] * 1e-5, q[9] * 1e-5, q[3], q[4] & 0xFFFF, q[4] >> 16, (q[10] >> 8) & 0xFF, q[10] >> 16, q[11]);
                                last = best;
                            }
                        }
                        fprintf(stderr, "[job %llu] tier %u: %u runs, busy %.0f wave-ms, first start %.2f, last end %.2f ms, longest %.2f ms; running at 10..90 %% of its span: %d %d %d %d %d %d %d %d %d\n",
                                (unsigned long long)j.seq, tier, cnt, busy, first, last, longest, occ[0], occ[1], occ[2], occ[3], occ[4], occ[5], occ[6], occ[7], occ[8]);
                    }
                }
            }
        }
        /* a read whose re-assembly, or one of whose windows, exceeded a documented capacity is never silently different from the reference */
        std::vector<char> tainted(n_piles, 0);
        std::string names;
        uint32_t n_bad = 0;
        for (uint32_t i = 0; i < n_piles; ++i) {
            bool bad = h_st[i] == CW_READ_CAPACITY;
            for (uint32_t w = j.sr[i].win_first; !bad && w < j.sr[i].win_first + j.sr[i].win_count; ++w) bad = h_wst[w] == CW_WIN_OVERFLOW;
            if (bad) {
                tainted[i] = 1;
                if (n_bad++ < 64) { names += names.empty() ? "" : ", "; names += cw_read_index_name(sh.index, j.sr[i].read); }
            }
        }
        if (n_bad) { /* say which capacity: the engine keeps a reason per window (CW_WHY_*, cw_device.h) */
            std::vector<uint32_t> wi((size_t)n_win * 16);
            uint32_t n_st = 0, hist[16] = {0};
            for (uint32_t i = 0; i < n_piles; ++i) n_st += h_st[i] == CW_READ_CAPACITY;
            if (cw_debug_win_info(eng, n_win, wi.data()) == CW_OK)
                for (uint32_t w = 0; w < n_win; ++w) if (h_wst[w] == CW_WIN_OVERFLOW) hist[wi[(size_t)w * 16 + 15] & 15u]++;
            names += " [";
            for (int q = 0; q < 16; ++q) if (hist[q]) names += "window reason " + std::to_string(q) + " x" + std::to_string(hist[q]) + "; ";
            names += std::to_string(n_st) + " read(s) beyond a re-assembly limit]";
        }
        if (n_bad && !sh.skip_on_capacity) {
            j.rc = CW_E_CAPACITY;
            j.err = "engine capacity exceeded for " + std::to_string(n_bad) + " read(s): " + names + (n_bad > 64 ? ", ..." : "") + " (CW_ON_CAPACITY=skip leaves them out)";
            return;
        }
        if (n_bad) fprintf(stderr, "[consent_amd] left out (engine capacity): %s%s\n", names.c_str(), n_bad > 64 ? ", ..." : "");
        /* only the corrected reads come back: one transfer of the output slots into pinned memory, then one string per read */
        uint64_t used_end = 0;
        for (uint32_t i = 0; i < n_piles; ++i)
            if (!tainted[i] && h_st[i] == CW_READ_OK && h_len[i]) used_end = out_off[i] + h_len[i];
        if (used_end) {
            if (pin_cap < used_end) {
                if (pin) (void)hipHostFree(pin);
                pin = nullptr; pin_cap = 0;
                if (hipHostMalloc(&pin, used_end + used_end / 4, hipHostMallocDefault) != hipSuccess) { pin = nullptr; j.rc = CW_E_NOMEM; j.err = "pinned host memory"; return; }
                pin_cap = used_end + used_end / 4;
            }
            DRV_HIP(hipMemcpyAsync(pin, o_buf.p, used_end, hipMemcpyDeviceToHost, st), j, "D2H reads");
            DRV_HIP(hipStreamSynchronize(st), j, "D2H reads");
            for (uint32_t i = 0; i < n_piles; ++i)
                if (!tainted[i] && h_st[i] == CW_READ_OK && h_len[i]) j.out[i].assign((const char*)pin + out_off[i], h_len[i]);
        }
        const double t4 = now_ms();
        j.ms_extract = t1 - t0; j.ms_consensus = t2 - t1; j.ms_stitch = t4 - t2;
        windows += n_win; reads += n_piles; jobs++;
        ms_extract += j.ms_extract; ms_consensus += j.ms_consensus; ms_stitch += j.ms_stitch;
        if (const char* tv = getenv("CW_DRIVER_TIMING")) if (tv[0] == '2') /* inspection: when each job's phases ran, per worker */
            fprintf(stderr, "[job %llu] worker %p windows %u: start %.1f extract %.1f consensus %.1f (alloc %.1f) stitch %.1f ms\n", (unsigned long long)j.seq, (void*)this, n_win, t0 - sh.t_start, t1 - t0, t2 - t1, t_alloc, t4 - t2);
    }
};

void worker_main(Shared* sh, Worker* w) {
    for (;;) {
        Job* j = nullptr;
        {
            std::unique_lock<std::mutex> lk(sh->mu);
            sh->cv_work.wait(lk, [&] { return !sh->queue.empty() || sh->closed || sh->abort; });
            if (sh->abort || (sh->queue.empty() && sh->closed)) return;
            j = sh->queue.front();
            sh->queue.pop_front();
            sh->cv_room.notify_all();
        }
        w->process(*sh, *j);
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            if (j->rc != CW_OK && sh->first_error == CW_OK) { sh->first_error = j->rc; sh->first_error_msg = j->err; sh->abort = true; sh->cv_work.notify_all(); sh->cv_room.notify_all(); }
            sh->finished[j->seq] = j;
            sh->cv_done.notify_all();
        }
    }
}

bool write_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t k = write(fd, p, n);
        if (k < 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

/* writes finished jobs in submission order: the FASTA does not depend on which device took which job */
void emitter_main(Shared* sh, int out_fd, const uint64_t* n_jobs_total, const bool* producer_done, uint64_t* records, uint64_t* bases) {
    uint64_t next = 0;
    std::string buf;
    for (;;) {
        Job* j = nullptr;
        {
            std::unique_lock<std::mutex> lk(sh->mu);
            sh->cv_done.wait(lk, [&] { return sh->finished.count(next) || sh->abort || (*producer_done && next >= *n_jobs_total); });
            if (sh->abort) return;
            if (!sh->finished.count(next)) return; /* everything written */
            j = sh->finished[next];
            sh->finished.erase(next);
        }
        buf.clear();
        for (size_t i = 0; i < j->out.size(); ++i) {
            if (j->out[i].empty()) continue;
            buf += '>'; buf += cw_read_index_name(sh->index, j->sr[i].read); buf += '\n';
            buf += j->out[i]; buf += '\n';
            ++*records; *bases += j->out[i].size();
        }
        const bool ok = buf.empty() || write_all(out_fd, buf.data(), buf.size());
        delete j;
        ++next;
        if (!ok) {
            std::lock_guard<std::mutex> lk(sh->mu);
            if (sh->first_error == CW_OK) { sh->first_error = CW_E_INTERNAL; sh->first_error_msg = "write to the output failed"; }
            sh->abort = true; sh->cv_work.notify_all(); sh->cv_room.notify_all();
            return;
        }
    }
}

} // namespace

extern "C" int cw_run_correction(const cw_driver_args* a, int out_fd, cw_driver_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!a || !a->alignment_file || !a->reads_file || out_fd < 0) return CW_E_INVALID;
    if (a->window_size == 0 || a->window_overlap >= a->window_size || a->mer_size < 2 || a->mer_size > 16 || a->solid_thresh < 1 || a->max_msa < 1 || a->max_support < 1) return CW_E_INVALID;
    const double t_begin = now_ms();
    Shared sh;
    sh.a = a;
    sh.t_start = now_ms();
    const bool has_proof = a->proof_file && a->proof_file[0];
    sh.do_trim = !a->polishing && !has_proof; /* CONSENT-correction.cpp:17,69-73; CONSENT-polishing.cpp:19 */
    if (const char* oc = getenv("CW_ON_CAPACITY")) sh.skip_on_capacity = strcmp(oc, "skip") == 0;
    /* GPU_MAX_HW_QUEUES (hardware queues HIP spreads a process's streams over, default 4, read when the runtime starts) is the HOST
       PROCESS's business: bin/CONSENT-* set it to 12 in main() before anything touches HIP, consent_amd/pipeline.py before the library is
       loaded; an embedding application does the same (include/consent_amd.h "Threading").  A library call does not change its caller's
       environment. */
    /* the HIP runtime takes ~0.1 s to start (first call of the process): let it start while the reads are indexed */
    if (const char* dr = CW_AID_ENV("CW_DRIVER_DRY")) sh.dry = dr[0] && dr[0] != '0';
    const bool dry = sh.dry;
    struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } hip_warm{std::thread([dry] { if (dry || CW_AID_ENV("CW_NO_WARM")) return; int n = 0; (void)hipGetDeviceCount(&n); if (n > 0) (void)hipFree(nullptr); })};

    /* ---- indexReads (+ the proof file into the same index) ---- */
    cw_read_index* index = nullptr;
    int rc = cw_index_reads(a->reads_file, &index);
    if (rc != CW_OK) { fprintf(stderr, "[consent_amd] cannot index %s: %s\n", a->reads_file, cw_strerror(rc)); return rc; }
    uint64_t tpl_bases = 0; /* bases of the templates (the sequences of reads_file: every one of them may be a pile's query): sizes the jobs */
    {
        cw_read_set v0; uint64_t w0 = 0;
        cw_read_index_view(index, &v0, &w0);
        for (uint32_t i = 0; i < v0.n_reads; ++i) tpl_bases += v0.read_len[i];
    }
    if (has_proof && (rc = cw_index_reads_append(index, a->proof_file)) != CW_OK) {
        fprintf(stderr, "[consent_amd] cannot index %s: %s\n", a->proof_file, cw_strerror(rc));
        cw_read_index_free(index);
        return rc;
    }
    sh.index = index;
    cw_read_index_view(index, &sh.host_reads, &sh.read_words);
    const double t_indexed = now_ms();
    hip_warm.t.join();

    /* ---- devices: explicit list, or CW_DEVICES="0,1,..." (an id may repeat: several engines on one GPU), or the first
            min(nb_threads, device count) devices ---- */
    std::vector<int> devs;
    if (a->devices && a->n_devices > 0) devs.assign(a->devices, a->devices + a->n_devices);
    else if (const char* env = getenv("CW_DEVICES")) {
        for (const char* p = env; *p;) { char* e = nullptr; const long v = strtol(p, &e, 10); if (e == p) break; devs.push_back((int)v); p = *e == ',' ? e + 1 : e; }
    }
    int n_dev = 0;
    if (dry) { n_dev = a->nb_threads < 1 ? 1 : (int)a->nb_threads; for (int d : devs) n_dev = d + 1 > n_dev ? d + 1 : n_dev; } /* as many "devices" as were asked for */
    else if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { cw_read_index_free(index); fprintf(stderr, "[consent_amd] no HIP device: the engine has no CPU path\n"); return CW_E_NO_DEVICE; }
    /* test aid (-DCW_TEST_AIDS build only): CW_VIRTUAL_DEVICES=8 makes the one GPU of a test box eight logical devices -- eight read-set uploads,
       the workers, job size and queue an 8-GPU node gets (tests/test_gpu_driver.py); logical device v runs on HIP device v % (physical count) */
    const int n_phys = n_dev > 0 ? n_dev : 1;
    if (!dry) if (const char* env = CW_AID_ENV("CW_VIRTUAL_DEVICES")) { const int v = atoi(env); if (v >= 1 && v <= 64) n_dev = v; }
    if (devs.empty()) {
        /* two workers (engine + buffers each) per device: while one job is in its re-assembly -- one wave per read, the longest read sets
           the time, most of the GPU idle -- the other worker's consensus kernels run (measured on one GPU: 717 -> 550 ms for 112 k windows) */
        const int want = a->nb_threads < 1 ? 1 : (int)a->nb_threads;
        /* A run that gives a device fewer than ~1e5 windows (the E. coli-scale set on eight GPUs: 4e4 each) is cut into jobs of a few
           thousand windows, whose fixed costs -- the longest POA task, the longest read of the re-assembly, the synchronisation points of a
           run -- no longer hide behind one other job: four workers per device then (measured on one GPU with jobs of 5000 windows:
           1.97e5 windows/s with two workers, 2.36e5 with three, 2.77e5 with four; with jobs of 32768: 3.3e5 / 2.8e5 / 2.8e5) */
        const uint64_t est_per_dev = (tpl_bases / (a->window_size - a->window_overlap) + 1) / (uint64_t)(n_dev < want ? n_dev : want);
        /* round 6: two workers also for a small per-device load (four from round 4 on, when such a run was cut into jobs of ~5000 windows).  With three larger jobs
           per device (below) two workers are as fast (a device's 4.2e4 windows: 0.134-0.138 s with two workers on two or three jobs, 0.133-0.134 s with four
           on two or four; profiles/r06_job_size_sweep.txt) and set up half the engines: an engine's scratch is ~10 GB, and obtaining that much new device
           memory is where a fresh process can stall for a second or more (hipMalloc: 0.2 ms or 0.5-1.9 s a call) */
        int per_dev = 2; /* (never more workers than jobs is implied: three jobs per device at least) */
        if (est_per_dev >= 1000000ull) per_dev = 3;
        if (const char* env = getenv("CW_WORKERS_PER_DEVICE")) { const int v = atoi(env); if (v >= 1 && v <= 8) per_dev = v; }
        for (int k = 0; k < per_dev; ++k) for (int d = 0; d < n_dev && d < want; ++d) devs.push_back(d);
    }
    for (int d : devs) if (d < 0 || d >= n_dev) { cw_read_index_free(index); return CW_E_INVALID; }

    std::vector<Worker> workers(devs.size());
    for (size_t i = 0; i < devs.size(); ++i) {
        workers[i].device = devs[i];
        workers[i].phys = devs[i] % n_phys;
        for (size_t o = 0; o < i; ++o)
            if (!workers[o].owner && workers[o].device == devs[i]) { workers[i].owner = &workers[o]; break; }
    }
    {   /* one future per owner, shared by its borrowers */
        std::vector<std::shared_future<int>> fut(devs.size());
        for (size_t i = 0; i < devs.size(); ++i) if (!workers[i].owner) fut[i] = workers[i].reads_ready.get_future().share();
        for (size_t i = 0; i < devs.size(); ++i) if (workers[i].owner) workers[i].reads_of_owner = fut[(size_t)(workers[i].owner - &workers[0])];
    }
    sh.queue_cap = 2 * devs.size() + 1;
    /* Engines and read-set uploads start on the worker threads themselves (all devices at once) while this thread already parses the
       alignments: creating two engines with their scratch takes 0.2-0.3 s, during which the producer fills the queue */
    std::vector<double> ms_init(devs.size(), 0.0);
    double ms_parse = 0, ms_windows = 0;

    uint64_t n_jobs_total = 0, records = 0, bases_out = 0;
    bool producer_done = false;
    std::vector<std::thread> threads;
    for (size_t i = 0; i < workers.size(); ++i)
        threads.emplace_back([&, i] {
            const double t0 = now_ms();
            const int irc = workers[i].init(sh);
            ms_init[i] = now_ms() - t0;
            if (irc != CW_OK) {
                std::lock_guard<std::mutex> lk(sh.mu);
                if (sh.first_error == CW_OK) { sh.first_error = irc; sh.first_error_msg = "device " + std::to_string(workers[i].device); }
                sh.abort = true;
                sh.cv_work.notify_all(); sh.cv_room.notify_all(); sh.cv_done.notify_all();
                return;
            }
            worker_main(&sh, &workers[i]);
        });
    std::thread emitter(emitter_main, &sh, out_fd, &n_jobs_total, &producer_done, &records, &bases_out);

    /* ---- producer: getNextReadPile -> getAlignmentWindowsPositions -> jobs ----
       Piles are read in PAF order on this thread (the reader parses blocks of the file ahead on its own threads, cw_hostio.cpp) a
       few hundred at a time; the window positions of one such round are computed on helper threads -- piles are independent -- while
       this thread already reads the next round; jobs are then put together in pile order, so their contents do not depend on the
       number of helpers.  One thread did all three before: 3 M windows/s, within 2x of what eight GPUs take. */
    /* Windows per job.  The caller's figure, else 32768 -- unless the run is too short for that many workers: a job is the unit the workers
       share, and with fewer than about eight jobs per device the last ones leave most engines idle (the E. coli-scale set is 3.2e5 windows:
       ten jobs of 32768 for an 8-GPU node).  The number of windows is not known before the alignments are read, but
       it is close to template bases / (window size - overlap); floor 4096 windows (below that a job no longer fills a GPU). */
    size_t n_distinct_devs;
    { std::vector<int> distinct(devs); std::sort(distinct.begin(), distinct.end()); distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end()); n_distinct_devs = distinct.size(); }
    uint32_t per_job = a->windows_per_batch ? (a->windows_per_batch > CW_MAX_BATCH_WINDOWS ? CW_MAX_BATCH_WINDOWS : a->windows_per_batch) : 32768u;
    if (!a->windows_per_batch) {
        const uint64_t est_windows = tpl_bases / (a->window_size - a->window_overlap) + 1;
        /* jobs per device.  Eight (four per worker with two workers) while a device gets 1e5 windows or more; THREE below that (round 6): a device that gets
           4e4 windows -- the E. coli-scale set on eight GPUs -- ran its nine jobs of 5 200 windows on four workers in 0.234 s, and four jobs of 10 400 on the
           same four workers in 0.134 s (two jobs of 20 800 on two: 0.138; tools/job_size_model.py JSM_MODE=sweep, profiles/r06_job_size_sweep.txt): a job's
           fixed costs -- its longest POA task, its longest read, its launches and synchronisation points -- are paid once per job and worker, and small
           jobs do not fill the GPU while they are paid */
        const uint64_t jobs_per_dev = est_windows / n_distinct_devs < 100000ull ? 3ull : 8ull;
        const uint64_t want = est_windows / (jobs_per_dev * n_distinct_devs) + 1;
        if (want < per_job) per_job = (uint32_t)(want < 4096 ? 4096 : want);
    }
    if (const char* env = CW_AID_ENV("CW_JOB_WINDOWS")) { const long v = atol(env); if (v >= 1 && v <= (long)CW_MAX_BATCH_WINDOWS) per_job = (uint32_t)v; } /* test aid: jobs of a few windows, so that a small data set reaches every worker */
    cw_paf_reader* paf = nullptr;
    rc = cw_paf_open(a->alignment_file, index, a->max_support, &paf);
    uint64_t n_piles = 0, n_windows = 0, n_overlaps = 0;
    unsigned n_help = std::thread::hardware_concurrency() / 2;
    if (n_help > 8) n_help = 8;
    if (const char* env = getenv("CW_PRODUCER_THREADS")) { const int v = atoi(env); if (v >= 1 && v <= 64) n_help = (unsigned)v; }
    if (n_help < 1) n_help = 1;
    if (rc == CW_OK) {
        struct PileRec { uint32_t tpl, tpl_len, n, np; size_t ov_off; int rc; std::vector<uint32_t> wp; };
        struct Round { std::vector<PileRec> piles; std::vector<cw_overlap> ov; size_t used = 0; bool last = false; };
        Round rounds[2];
        const size_t round_piles = 512, round_overlaps = 1u << 17;
        std::vector<cw_overlap> ov(a->max_support ? a->max_support : 1);
        Job* cur = new Job();
        uint64_t job_cost_cap = 1ull << 26;
        if (const char* env = CW_AID_ENV("CW_JOB_COST_CAP")) { const long long v = atoll(env); if (v >= 1) job_cost_cap = (uint64_t)v; } /* test aid */
        auto push_job = [&]() {
            if (cur->wj.empty()) return true;
            cur->seq = n_jobs_total;
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.cv_room.wait(lk, [&] { return sh.queue.size() < sh.queue_cap || sh.abort; });
            if (sh.abort) return false;
            sh.queue.push_back(cur);
            ++n_jobs_total;
            sh.cv_work.notify_one();
            cur = new Job();
            return true;
        };
        /* stage 1 (this thread): the next piles of the file */
        auto read_round = [&](Round& r) -> int {
            r.used = 0; r.ov.clear(); r.last = false;
            while (r.used < round_piles && r.ov.size() < round_overlaps) {
                uint32_t tpl = 0, tpl_len = 0, n = 0;
                const double tp0 = now_ms();
                const int prc = cw_paf_next_pile(paf, &tpl, &tpl_len, ov.data(), nullptr, (uint32_t)ov.size(), &n);
                ms_parse += now_ms() - tp0;
                if (prc != CW_OK) { fprintf(stderr, "[consent_amd] %s: %s (malformed line, a name missing from the read file, or a length that disagrees with it)\n", a->alignment_file, cw_strerror(prc)); return prc; }
                if (n == 0) { r.last = true; break; }
                if (tpl_len != sh.host_reads.read_len[tpl]) {
                    fprintf(stderr, "[consent_amd] %s states length %u for %s, the read file has %u\n", a->alignment_file, tpl_len, cw_read_index_name(index, tpl), sh.host_reads.read_len[tpl]);
                    return CW_E_INVALID;
                }
                if (r.used == r.piles.size()) r.piles.emplace_back();
                PileRec& p = r.piles[r.used++];
                p.tpl = tpl; p.tpl_len = tpl_len; p.n = n; p.np = 0; p.rc = CW_OK; p.ov_off = r.ov.size();
                r.ov.insert(r.ov.end(), ov.begin(), ov.begin() + n);
            }
            return CW_OK;
        };
        /* stage 2 (helpers): window positions of every pile of a round */
        auto positions = [&](Round& r) {
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (;;) {
                    const size_t i = next.fetch_add(1, std::memory_order_relaxed);
                    if (i >= r.used) return;
                    PileRec& p = r.piles[i];
                    p.wp.resize(2 * ((size_t)p.tpl_len / (a->window_size - a->window_overlap) + 8));
                    uint32_t np = 0;
                    int wrc = cw_window_positions(p.tpl_len, r.ov.data() + p.ov_off, p.n, a->min_support, a->window_size, (int32_t)a->window_overlap, p.wp.data(), (uint32_t)(p.wp.size() / 2), &np);
                    if (wrc == CW_E_CAPACITY) { p.wp.resize(2 * (size_t)np); wrc = cw_window_positions(p.tpl_len, r.ov.data() + p.ov_off, p.n, a->min_support, a->window_size, (int32_t)a->window_overlap, p.wp.data(), np, &np); }
                    p.np = np; p.rc = wrc;
                }
            };
            const unsigned nt = (unsigned)std::min<size_t>(n_help, (r.used + 15) / 16);
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto& t : th) t.join();
        };
        /* stage 3 (this thread): jobs, in pile order */
        auto assemble = [&](Round& r) -> int {
            for (size_t i = 0; i < r.used; ++i) {
                PileRec& p = r.piles[i];
                if (p.rc != CW_OK) return p.rc;
                ++n_piles;
                const uint32_t np = p.np, n = p.n;
                if (np == 0) continue; /* processRead returns (readId, "") before anything else (CONSENT-correction.cpp:22-25) */
                /* (a read or contig with more windows than one engine call takes -- CW_MAX_BATCH_WINDOWS -- becomes a job of its own: the worker
                   corrects it in several runs and re-assembles it once, Worker::process) */
                /* a job's extraction scratch is one descriptor per (window, overlap of its pile): polishing with -S 20000 looks at every
                   overlap of the contig for every window (alignmentWindows.cpp:105), so jobs are also cut by that product -- 2^26
                   descriptors = 1 GiB; only a single pile larger than that still makes a larger job (a read's windows are never split) */
                if (!cur->wj.empty() && (cur->wj.size() + np > CW_MAX_BATCH_WINDOWS || cur->cost + (uint64_t)np * (n + 1) > job_cost_cap) && !push_job()) return CW_E_INTERNAL;
                cw_stitch_read sr_{p.tpl, (uint32_t)cur->wj.size(), np};
                const uint32_t ovl_first = (uint32_t)cur->ovl.size();
                cur->ovl.insert(cur->ovl.end(), r.ov.begin() + (ptrdiff_t)p.ov_off, r.ov.begin() + (ptrdiff_t)(p.ov_off + n));
                for (uint32_t w = 0; w < np; ++w) {
                    cur->wj.push_back(cw_window_job{p.tpl, p.wp[2 * w], p.wp[2 * w + 1], ovl_first, n});
                    cur->win_pos.push_back(p.wp[2 * w]); cur->win_pos.push_back(p.wp[2 * w + 1]);
                }
                cur->sr.push_back(sr_);
                cur->cost += (uint64_t)np * (n + 1);
                n_windows += np; n_overlaps += n;
                if (cur->wj.size() >= per_job && !push_job()) return CW_E_INTERNAL;
            }
            return CW_OK;
        };
        int k = 0;
        rc = read_round(rounds[0]);
        while (rc == CW_OK) {
            Round& r = rounds[k];
            int rrc = CW_OK;
            const double tw0 = now_ms();
            if (r.last && r.used == 0) break;
            if (r.last) positions(r); /* nothing left to read beside it */
            else {
                std::thread helper([&] { positions(r); });
                rrc = read_round(rounds[k ^ 1]);
                helper.join();
            }
            ms_windows += now_ms() - tw0; /* includes the reading that ran beside it */
            rc = assemble(r);
            if (rc == CW_E_INTERNAL && sh.abort) { rc = CW_OK; break; } /* a worker failed: its error is reported below */
            if (rc == CW_OK) rc = rrc;
            if (r.last) break;
            k ^= 1;
        }
        if (rc == CW_OK && !sh.abort) push_job();
        delete cur;
        cw_paf_close(paf);
    } else {
        fprintf(stderr, "[consent_amd] cannot open %s\n", a->alignment_file);
    }
    const double t_produced = now_ms();
    {
        std::lock_guard<std::mutex> lk(sh.mu);
        producer_done = true;
        sh.closed = true;
        if (rc != CW_OK) { if (sh.first_error == CW_OK) { sh.first_error = rc; sh.first_error_msg = "reading the alignments"; } sh.abort = true; }
        sh.cv_work.notify_all(); sh.cv_done.notify_all(); sh.cv_room.notify_all();
    }
    for (auto& t : threads) t.join();
    {
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.cv_done.notify_all();
    }
    emitter.join();
    for (auto& kv : sh.finished) delete kv.second;
    for (Job* j : sh.queue) delete j;
    const double t_end = now_ms();
    if (stats) {
        stats->n_devices = (uint32_t)devs.size();
        stats->piles = n_piles; stats->windows = n_windows; stats->overlaps = n_overlaps; stats->jobs = n_jobs_total;
        stats->records = records; stats->bases_out = bases_out;
        stats->ms_index = t_indexed - t_begin; stats->ms_total = t_end - t_begin;
        for (size_t i = 0; i < workers.size() && i < 16; ++i) {
            stats->dev_windows[i] = workers[i].windows;
            stats->dev_ms_extract[i] = workers[i].ms_extract; stats->dev_ms_consensus[i] = workers[i].ms_consensus; stats->dev_ms_stitch[i] = workers[i].ms_stitch;
        }
    }
    if (getenv("CW_DRIVER_STATS")) { /* counters on stderr; stdout stays pure FASTA */
        fprintf(stderr, "{\"dry\": %s, \"producer_threads\": %u, \"ms_producer\": %.1f, \"workers\": %zu, \"workers_per_device\": %zu, \"windows_per_job\": %u, \"piles\": %llu, \"windows\": %llu, \"jobs\": %llu, \"records\": %llu, \"bases_out\": %llu, \"ms_index\": %.1f, \"ms_engines\": %.1f, \"ms_paf_parse\": %.1f, \"ms_window_positions\": %.1f, \"ms_total\": %.1f, \"windows_per_s\": %.1f, \"per_device\": [",
                dry ? "true" : "false", n_help, t_produced - t_indexed, devs.size(), devs.size() / (size_t)std::max<size_t>(1, n_distinct_devs), per_job, (unsigned long long)n_piles, (unsigned long long)n_windows, (unsigned long long)n_jobs_total, (unsigned long long)records, (unsigned long long)bases_out,
                t_indexed - t_begin, *std::max_element(ms_init.begin(), ms_init.end()), ms_parse, ms_windows, t_end - t_begin, n_windows / ((t_end - t_indexed) * 1e-3 + 1e-9));
        for (size_t i = 0; i < workers.size(); ++i)
            fprintf(stderr, "%s{\"device\": %d, \"windows\": %llu, \"jobs\": %llu, \"ms_extract\": %.1f, \"ms_consensus\": %.1f, \"ms_stitch\": %.1f}", i ? ", " : "", workers[i].device,
                    (unsigned long long)workers[i].windows, (unsigned long long)workers[i].jobs, workers[i].ms_extract, workers[i].ms_consensus, workers[i].ms_stitch);
        fprintf(stderr, "]}\n");
    }
    for (auto& w : workers) w.close();
    cw_read_index_free(index);
    if (sh.first_error != CW_OK) fprintf(stderr, "[consent_amd] %s: %s\n", sh.first_error_msg.c_str(), cw_strerror(sh.first_error));
    return sh.first_error;
}
