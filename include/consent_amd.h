/*
 * consent_amd.h -- C ABI of the MI355X window-correction engine (libconsent_amd.so).
 *
 * Drop-in boundary for CONSENT's per-window consensus operator.  Each entry point states the
 * reference interface it stands in for (paths are under the CONSENT source tree, src/):
 *
 *   cw_run / cw_run_device   <- computeConsensusReadCorrection      (correctionMSA.h:8,  correctionMSA.cpp:29-49)
 *                               computeConsensusAssemblyPolishing   (correctionMSA.h:10, correctionMSA.cpp:51-71)
 *                               i.e. MSABMAAC (call sites correctionMSA.cpp:32,54) + weightConsensus
 *                               (correctionMSA.cpp:6-27) + polishCorrection (correctionDBG.h:11)
 *                               -- batched over windows, because one window per call cannot feed a GPU.
 *   cw_window_positions      <- getAlignmentWindowsPositions (alignmentWindows.cpp:27-85), host
 *   cw_extract_piles_device  <- getAlignmentWindowsSequences (alignmentWindows.cpp:87-149) evaluated on the device
 *   cw_stitch_device         <- alignConsensus + trimRead + dropRead (correctionAlignment.cpp:47-140, utils.cpp:96-128, :71-73)
 *   cw_index_reads / cw_paf_next_pile <- indexReads (utils.cpp:166-205) / getNextReadPile (alignmentPiles.cpp:22-58), host
 *   cw_pack_sequence         <- the vector<string> pile handed to those operators
 *                               (CONSENT-correction.cpp:35-37, CONSENT-polishing.cpp:46-49): 2-bit packing
 *                               with the reference's own alphabet (utils.cpp:21-32: A=00 C=01 G=10 else=11).
 *
 * Plain pointers and sizes only; the caller owns every buffer; nothing is retained past return.
 * No C++ exception crosses this boundary.  All functions return 0 on success or a negative cw_status.
 */
#ifndef CONSENT_AMD_H
#define CONSENT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cw_status {
    CW_OK = 0,
    CW_E_INVALID = -1,   /* bad argument / malformed batch                      */
    CW_E_NO_DEVICE = -2, /* no HIP device, or the HIP runtime reported an error  */
    CW_E_NOMEM = -3,     /* device or host allocation failed                     */
    CW_E_CAPACITY = -4,  /* at least one window overflowed an output or scratch capacity: its win_status says which */
    CW_E_INTERNAL = -5
} cw_status;

/* Per-window outcome (cw_result.win_status). */
enum {
    CW_WIN_CONSENSUS = 0, /* consensus computed (correctionMSA.cpp:38-48)                        */
    CW_WIN_TEMPLATE = 1,  /* MSA empty -> raw template returned (correctionMSA.cpp:34-36)        */
    CW_WIN_OVERFLOW = 2   /* a capacity was exceeded; no output for this window                  */
};

/* The five parameters of the operator that its body actually uses (correctionMSA.cpp:29-49). */
typedef struct cw_params {
    uint32_t k;            /* merSize     -k (main.cpp:20) */
    uint32_t solid;        /* solidThresh -f (main.cpp:23) */
    uint32_t common_kmers; /* commonKMers -c (main.cpp:21) */
    uint32_t min_anchors;  /* minAnchors  -A (main.cpp:22) */
    uint32_t max_msa;      /* maxMSA      -M (main.cpp:25) */
} cw_params;

/*
 * A batch of window piles.  Sequence s of window w has global index win_first_seq[w] + s; index
 * win_first_seq[w] itself is the template (pile[0], alignmentWindows.cpp:100).
 * Bases are 2-bit codes (A=0 C=1 G=2 T=3), 16 per 32-bit word, most significant first: base j of a
 * sequence sits in word seq_word_off[s] + j/16 at bits [30-2*(j%16), 31-2*(j%16)], so that a k-mer read
 * off the words is already in str2num order.  Every sequence starts on a word boundary; unused low
 * bits of its last word are zero.
 */
typedef struct cw_batch {
    uint32_t n_windows;
    uint32_t n_seqs;
    uint64_t n_words;
    const uint32_t* win_first_seq; /* [n_windows + 1] */
    const uint32_t* seq_len;       /* [n_seqs] bases  */
    const uint64_t* seq_word_off;  /* [n_seqs]        */
    const uint32_t* bases;         /* [n_words]       */
} cw_batch;

/*
 * Outputs.  Window w may write cons[cons_off[w] .. cons_off[w+1]) and solid[solid_off[w] .. solid_off[w+1]).
 * cons holds ASCII ACGT/acgt (lower case = weakly supported, correctionMSA.cpp:18-22).
 * solid holds the window's k-mers (2-bit codes, MSB-first, str2num order) whose pile-wide count is
 * >= params.solid, ascending: everything the caller's downstream code asks of merCounts
 * (correctionAlignment.cpp:6-15).  solid/solid_off/solid_len may all be NULL to skip that output.
 */
typedef struct cw_result {
    char* cons;
    const uint64_t* cons_off; /* [n_windows + 1] */
    uint32_t* cons_len;       /* [n_windows]     */
    uint8_t* win_status;      /* [n_windows]     */
    uint32_t* solid;
    const uint64_t* solid_off; /* [n_windows + 1] */
    uint32_t* solid_len;       /* [n_windows]     */
} cw_result;

typedef struct cw_engine cw_engine;

/* Library / build identification: returns a static string such as "consent_amd 0.1 gfx950". */
const char* cw_version(void);
const char* cw_strerror(int status);

/* Engine bound to one HIP device.
 *
 * Threading.  The reference calls its operator from nbThreads pool threads at once (CONSENT-correction.cpp:77, one thread per read;
 * CONSENT-polishing.cpp:37,49 one task per window), with no shared mutable state.  Here an engine owns one scratch arena and one set
 * of streams, and every entry point that takes a cw_engine* locks the engine's mutex for the time it enqueues work: calls from
 * several host threads on ONE engine are safe and are serialised; they do not run concurrently.  For concurrency use one engine per
 * host thread or per device (any number of engines may share a GPU; results do not depend on how windows are spread over engines,
 * batches or GPUs).  cw_run_device is asynchronous: the caller's device buffers must stay valid, and the engine's next call on
 * another stream must be ordered by the caller, until that stream has been synchronised.  The host feeders (cw_index_reads,
 * cw_paf_*) keep their state in their own handles: one thread per handle (cw_paf_open starts parser threads of its own; they end in
 * cw_paf_close, which must come before cw_read_index_free of the index it was opened with).
 * Several engines on one GPU: an engine launches on four streams of its own.  The HIP runtime maps a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (default 4), read when the runtime starts; with more engine streams than queues one engine's
 * kernels queue behind another's long-running ones.  Set GPU_MAX_HW_QUEUES >= 4 x engines per GPU in the process environment (8 for
 * the usual two; cw_run_correction and bench.py do so themselves unless the variable is already set). */
int cw_create(const cw_params* params, int device, cw_engine** out);
void cw_destroy(cw_engine* e);

/* Templates longer than 1024 + k - 1 bases (windows beyond the wrappers' `-l 500`: the reference takes any `-l`, src/main.cpp:46-47).  The engine takes
 * templates of up to 2048 + k - 1 bases; its scratch plan provides for 1024 k-mers per template unless told otherwise -- call this once, before the
 * first run, with the longest template the engine will see (the window size): anchor blocks, segment slots and the chain kernel's work-groups are then
 * sized for it (up to twice the scratch).  Without it, windows with longer templates may stop on a capacity (status 2: CW_WHY_SETUP / CW_WHY_ANCHORS),
 * never with a wrong result.  CW_E_INVALID beyond 2048 + k - 1 (such a window's status is 2, CW_WHY_TEMPLATE).  A caller whose windows are all SHORTER than
 * 1024 + k - 1 bases may say so too: the plan shrinks with the number (anchor blocks, segment slots, arena).  cw_run_correction calls it with its window size. */
int cw_configure(cw_engine* e, uint32_t max_template_len);

/* Largest batch one call accepts (per-window offsets into the engine's scratch are 32-bit); CW_E_INVALID beyond it.  Split larger
 * inputs: results do not depend on the batch composition. */
#define CW_MAX_BATCH_WINDOWS 131072u

/* Host buffers in, host buffers out: H2D, kernels, D2H of the bytes the windows actually produced (not of the reserved
 * capacities), synchronous.  Equivalent to cw_submit + cw_wait. */
int cw_run(cw_engine* e, const cw_batch* batch, const cw_result* result);

/* The same in two halves, so that a caller can keep the GPU busy: cw_submit enqueues the H2D copies and the kernels of a batch and
 * returns a ticket; cw_wait blocks until that batch is done and fills `result`.  Up to two batches may be in flight per engine
 * (submit n+1, then wait n: the copies of one overlap the kernels of the other); a third cw_submit returns CW_E_INVALID.  Every
 * array named by `batch` and `result` must stay valid and untouched until cw_wait has returned.  Arrays from cw_host_alloc (pinned)
 * are moved by real DMA that overlaps the kernels; plain pageable arrays work too, the runtime then stages them itself. */
int cw_submit(cw_engine* e, const cw_batch* batch, const cw_result* result, int* ticket);
int cw_wait(cw_engine* e, int ticket);

/* Pinned host memory for batches / results (hipHostMalloc). */
int cw_host_alloc(void** ptr, size_t bytes);
void cw_host_free(void* ptr);

/* Every pointer inside batch/result is a DEVICE pointer; asynchronous on `hip_stream` (a hipStream_t,
 * NULL = the engine's own stream); statuses are checked by the caller after synchronising.  */
int cw_run_device(cw_engine* e, const cw_batch* batch, const cw_result* result, void* hip_stream);

/* 1 when everything the last cw_run_device on this engine launched has completed (or nothing was launched yet), 0 while it is
 * still running, 2 while it is running but past the part that fills the GPU (what is left is the tail: a few long alignment tasks on
 * one wave each, then the finish kernel); never blocks.  For callers that keep several engines busy: hand the next batch to an
 * engine that returns 1.  (Waiting for the other engine to reach 2 before starting -- batches staggered instead of side by side --
 * halves a batch's latency and measured 4 % less throughput at depth 150: bench.py, CW_BENCH_STAGGER.) */
int cw_poll(cw_engine* e);

/* Milliseconds spent in each device stage of the last cw_run / cw_run_device on this engine, measured
 * with HIP events on the launch stream.  n_stages entries are written (at most cap); names are static. */
int cw_last_timings(cw_engine* e, float* ms, const char** names, int cap, int* n_stages);

/* Pack one ASCII window pile (n strings, lens[i] bytes each, not NUL-terminated) at the tail of host
 * arrays laid out as cw_batch.  words_cap counts 32-bit words available at bases_out.  Returns the
 * number of words written, or a negative cw_status.  Non-ACGT bytes pack as T (utils.cpp:28). */
int64_t cw_pack_sequence(const char* seq, uint32_t len, uint32_t* bases_out, uint64_t words_cap);

/* ---- device-side pile extraction ------------------------------------------------------------------
 * Stands in for getAlignmentWindowsSequences (src/alignmentWindows.cpp:87-149) + the read decoding of getSequencesMap
 * (src/alignmentPiles.cpp:5-20): piles are cut on the GPU from the 2-bit read set and the overlap tuples. */
typedef struct cw_read_set {      /* the indexed reads (utils.cpp:166-205), 2-bit packed like cw_batch.bases */
    uint32_t n_reads;
    const uint32_t* read_len;      /* [n_reads] */
    const uint64_t* read_word_off; /* [n_reads] */
    const uint32_t* bases;
} cw_read_set;

typedef struct cw_overlap {        /* one PAF record after Overlap(std::string) (src/Overlap.h:26-58): ends INCLUSIVE */
    uint32_t q_start, q_end;
    uint32_t t_read;               /* index of the target read in the read set (its length is tLength)             */
    uint32_t t_start, t_end;
    uint32_t strand;               /* 0 '+', 1 '-' */
} cw_overlap;

typedef struct cw_window_job {     /* one window of one template (pilesPos[i], CONSENT-correction.cpp:35)            */
    uint32_t tpl_read;
    uint32_t q_beg, q_end;         /* inclusive */
    uint32_t ovl_first, ovl_count; /* the template's pile in `overlaps`, in getNextReadPile order                    */
} cw_window_job;

/* HOST function: window positions of one template, getAlignmentWindowsPositions (src/alignmentWindows.cpp:27-85, with
 * getCoverages :5-25).  `overlaps` is the template's pile (only q_start/q_end are read).  Writes (beg,end) pairs, inclusive,
 * in the reference's order (forward windows, then the one trailing window); *n_pairs = how many exist; CW_E_CAPACITY if
 * cap_pairs was too small (call again).  CW_E_INVALID if an overlap ends beyond the template. */
int cw_window_positions(uint32_t tpl_len, const cw_overlap* overlaps, uint32_t n_overlaps, uint32_t min_support, uint32_t window_size,
                        int32_t window_overlap, uint32_t* out_beg_end, uint32_t cap_pairs, uint32_t* n_pairs);

/* All pointers are DEVICE pointers (read set, overlaps, jobs, and the four output arrays laid out as cw_batch:
 * win_first_seq[n_jobs+1], seq_len[seq_cap], seq_word_off[seq_cap], bases[word_cap]).  On return *n_seqs / *n_words hold
 * the totals (the call synchronises once to read them); CW_E_CAPACITY if they exceed the capacities -- call again with
 * larger arrays (capacities 0 just size the batch).  A window beyond its template (alignmentWindows.cpp:95-97) or a
 * piece shorter than k (:141) yields no member, as in the reference. */
int cw_extract_piles_device(cw_engine* e, const cw_read_set* reads, const cw_overlap* overlaps, uint64_t n_overlaps,
                            const cw_window_job* jobs, uint32_t n_jobs, uint32_t k, uint32_t* win_first_seq, uint32_t* seq_len,
                            uint64_t* seq_word_off, uint32_t* bases, uint32_t seq_cap, uint64_t word_cap, uint32_t* n_seqs,
                            uint64_t* n_words, void* hip_stream);

/* Result slots for a device-resident batch (what a caller of cw_run_device has to size before the call): writes the exclusive offsets
 * cons_off[n_windows+1] (CW_CONS_SLOT_BYTES(k, template length) per window) and solid_off[n_windows+1] (k-mers in the window's pile /
 * solidThresh + 16 entries), both DEVICE arrays, and returns their totals (one synchronisation).
 *
 * The consensus slot.  A consensus is about as long as its template; the polish may lengthen it; with k < 9, chance anchors make consensuses of
 * several templates (the finish kernel holds 32768 characters).  Slots are real device memory (and, through cw_submit, twice that per batch in
 * flight), so the rule follows k: k >= 9 -- no window of the randomised testers (5e5 windows, k 9..16) ever went beyond three templates + 256 --
 * gets four templates + 1024, at least 3072; k = 8 sixteen templates + 1024; k < 8 the full 32768.  A consensus that outgrows its slot is a
 * reported capacity (window status 2, CW_WHY_OUT_CONS), never a truncation; a caller that sizes its own slots may give every window 32768. */
#define CW_CONS_SLOT_MAX 32768u
#define CW_CONS_SLOT_BYTES(k, tpl_len) \
    ((k) >= 9u ? (4u * (uint32_t)(tpl_len) + 1024u < 3072u ? 3072u : (4u * (uint32_t)(tpl_len) + 1024u > CW_CONS_SLOT_MAX ? CW_CONS_SLOT_MAX : 4u * (uint32_t)(tpl_len) + 1024u)) \
     : (k) == 8u ? (16u * (uint32_t)(tpl_len) + 1024u > CW_CONS_SLOT_MAX ? CW_CONS_SLOT_MAX : 16u * (uint32_t)(tpl_len) + 1024u) : CW_CONS_SLOT_MAX)
int cw_plan_results_device(cw_engine* e, const cw_batch* batch, uint64_t* cons_off, uint64_t* solid_off, uint64_t* cons_total, uint64_t* solid_total,
                           void* hip_stream);

/* ---- host feeders: read indexer and PAF pile reader, plain host code ------------------------------------
 * cw_index_reads stands in for indexReads (src/utils.cpp:166-205): FASTA or FASTQ (multi-line allowed) -> 2-bit reads keyed by
 * name (header up to the first blank; a later record with the same name replaces the earlier one; bases upper-cased, anything
 * but A/C/G packs as T).  The view's pointers are HOST pointers laid out like the device read set: upload them once. */
typedef struct cw_read_index cw_read_index;
int cw_index_reads(const char* path, cw_read_index** out);
int cw_index_reads_append(cw_read_index* idx, const char* path);   /* a second file into the same index (the proof file, CONSENT-correction.cpp:69-73) */
void cw_read_index_free(cw_read_index* idx);
uint32_t cw_read_index_count(const cw_read_index* idx);
int cw_read_index_view(const cw_read_index* idx, cw_read_set* host_view, uint64_t* n_words);
int32_t cw_read_index_find(const cw_read_index* idx, const char* name);   /* read id or -1 */
const char* cw_read_index_name(const cw_read_index* idx, uint32_t id);

/* cw_paf_next_pile stands in for getNextReadPile (src/alignmentPiles.cpp:22-58) with Overlap(std::string) (src/Overlap.h:26-58):
 * the next run of consecutive PAF lines sharing a query name, ends made inclusive, sorted by residue matches descending with the
 * reference's own std::sort expression (ties in libstdc++'s order), cut to max_support.  *n = overlaps in the pile, 0 at the end
 * of the file; *tpl_read / *tpl_len = the query's id in the index and its length as the PAF states it; res_matches may be NULL.
 * CW_E_CAPACITY when cap < *n (the pile is consumed: size the buffer for max_support); CW_E_INVALID on a malformed line or a
 * name missing from the index. */
typedef struct cw_paf_reader cw_paf_reader;
int cw_paf_open(const char* path, const cw_read_index* idx, uint32_t max_support, cw_paf_reader** out);
int cw_paf_next_pile(cw_paf_reader* r, uint32_t* tpl_read, uint32_t* tpl_len, cw_overlap* out, uint32_t* res_matches, uint32_t cap,
                     uint32_t* n);
void cw_paf_close(cw_paf_reader* r);

/* ---- wrapper plumbing: the small tools CONSENT-correct / CONSENT-polish run around the binary, as host functions.
 * cw_paf_reformat <- src/reformatPAF.cpp:22-46 (swap query and target columns); cw_paf_explode <- src/explode.cpp:14-51 (split so
 * that every query name is one run of lines per file; files are out_prefix_1 .. out_prefix_N); cw_paf_merge <- src/merge.cpp:29-65
 * (gather, in the order of a header file, every read's lines from the exploded chunks). */
int cw_paf_reformat(const char* in_path, const char* out_path);
int cw_paf_explode(const char* in_path, const char* out_prefix, uint32_t* n_files);
int cw_paf_merge(const char* out_path, const char* headers_path, const char* const* in_paths, uint32_t n_in);

/* ---- read re-assembly on the device -------------------------------------------------
 * Stands in for alignConsensus (src/correctionAlignment.cpp:47-140) followed, when do_trim != 0, by trimRead(.,1) and
 * dropRead (src/CONSENT-correction.cpp:47-58, src/utils.cpp:96-128, :71-73): every window consensus of a read is
 * aligned back onto the lower-cased read (local alignment, scores and position rules in include/cw_policy.h) and
 * replaces the aligned stretch in upper case; overlapping windows are reconciled by their solid k-mers. */
typedef struct cw_stitch_read {
    uint32_t read;                 /* the template in the read set (sequences[alignments[0].qName])                  */
    uint32_t win_first, win_count; /* its windows in the batch / result arrays, in pilesPos order                     */
} cw_stitch_read;

#define CW_READ_OK       0
#define CW_READ_DROPPED  1  /* dropRead: fewer than 10 % corrected bases after trimming; out_len = 0                    */
#define CW_READ_CAPACITY 2  /* output slot, consensus (> 32768) or aligned slice (> 2048) too large; out_len = 0         */

/* All pointers are DEVICE pointers.  `win_pos` = (beg,end) per window as cw_window_positions wrote them; `batch` = the piles
 * the consensuses were computed from (the first sequence of a window is its template, CONSENT-correction.cpp:37);
 * `res` = what cw_run_device filled (cons, cons_off, cons_len, win_status, solid, solid_off, solid_len; solid is required).
 * out_off[n_reads+1] gives every read a slot (2*read_len + 1024 is ample); the corrected read is written at its start,
 * out_len[r] characters, upper case = corrected.  Windows with CW_WIN_OVERFLOW are left uncorrected.  Asynchronous on
 * `hip_stream` (NULL = the engine's stream). */
int cw_stitch_device(cw_engine* e, const cw_read_set* reads, const cw_stitch_read* jobs, uint32_t n_reads, const uint32_t* win_pos,
                     const cw_batch* batch, const cw_result* res, uint32_t window_size, uint32_t window_overlap, int32_t do_trim,
                     char* out, const uint64_t* out_off, uint32_t* out_len, uint8_t* read_status, void* hip_stream);

/* ---- the drivers' loop (SURVEY 8b "who calls it", 8e, 8f-4) -----------------------------------------------------------------------
 * cw_run_correction stands in for runCorrection of BOTH reference drivers -- src/CONSENT-correction.cpp:62-135 (with processRead :19-58)
 * when polishing == 0, src/CONSENT-polishing.cpp:107-135 (with processContig :21-105) when polishing != 0 -- called by src/main.cpp:78
 * with the values of its getopt loop (:29-76).  bin/CONSENT-correction and bin/CONSENT-polishing are that main() over this function.
 * FASTA records (">name\nsequence\n") go to the file descriptor out_fd in PAF order; a read without windows, or dropped by the 10 % rule,
 * produces none.  Trimming and dropping happen only for correction without a proof file (CONSENT-correction.cpp:17,69-73).
 * nb_threads (-j) = how many GPUs to use: the first min(nb_threads, visible devices); `devices` / the environment variable CW_DEVICES
 * ("0,1,..", an id may repeat: one worker = one engine per entry) override that; by default every device gets two workers, so that
 * one job's re-assembly overlaps the other's consensus kernels (CW_WORKERS_PER_DEVICE changes the two).  Piles are handed to the devices job by job from one queue, there is no collective, and
 * the output does not depend on the number of devices.  paf_index (-i) and path (-p) are accepted and, as in the reference, never read. */
typedef struct cw_driver_args {
    const char* paf_index;      /* -i */
    const char* alignment_file; /* -a */
    const char* reads_file;     /* -r */
    const char* proof_file;     /* -R, NULL or "" = none */
    const char* path;           /* -p */
    uint32_t min_support, max_support, window_size, mer_size, common_kmers, min_anchors, solid_thresh, window_overlap, nb_threads, max_msa;
    int32_t polishing;
    const int32_t* devices;     /* NULL = choose as described above */
    int32_t n_devices;
    uint32_t windows_per_batch; /* windows per job; 0 = 32768, fewer (down to 4096) when the templates are too few for four such jobs per worker */
} cw_driver_args;

typedef struct cw_driver_stats {
    uint32_t n_devices;
    uint64_t piles, windows, overlaps, jobs, records, bases_out;
    double ms_index, ms_total;
    uint64_t dev_windows[16];
    double dev_ms_extract[16], dev_ms_consensus[16], dev_ms_stitch[16];
} cw_driver_stats;

int cw_run_correction(const cw_driver_args* args, int out_fd, cw_driver_stats* stats /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* CONSENT_AMD_H */
