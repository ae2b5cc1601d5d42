/*
 * consent_main.cpp -- main() of bin/CONSENT-correction and bin/CONSENT-polishing (one source built twice, as the reference links one
 * main.o against either driver object, Makefile:26,32).
 *
 * The drop-in contract is the command line of src/main.cpp:29-76: the getopt string "a:A:d:k:s:S:M:l:f:e:p:c:m:j:w:m:r:R:n:i:", the
 * defaults of :17-26, -i and -p accepted and never read; -d -e -w -n are letters of the string that have no case in the reference's
 * switch (:77 `default`), so there, and here, they end in a usage message and a failure exit like any unknown letter.  The unmodified
 * wrappers (CONSENT-correct:202, CONSENT-polish:197) can therefore call these binaries:
 *   CONSENT-correction -a aln.paf -s 3 -S 150 -l 500 -k 9 -c 8 -A 2 -f 4 -m 50 -j $nproc -r reads.fa -M 150 -p $dir >> out
 *   CONSENT-polishing  -a aln.paf -s 1 -S 20000 ... -j $nproc -r contigs.fa -R reads.fa -M 150 -p $dir >> out
 * Everything after the flags is cw_run_correction (include/consent_amd.h); -j is the number of GPUs to use.
 * The usage text is this program's own: it lists the flags that are honoured, with what they mean here.
 */
#include <getopt.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "consent_amd.h"

#ifndef CW_DRIVER_POLISHING
#define CW_DRIVER_POLISHING 0
#endif

namespace {

const char kOptString[] = "a:A:d:k:s:S:M:l:f:e:p:c:m:j:w:m:r:R:n:i:"; /* src/main.cpp:29 */

struct NumFlag { char letter; uint32_t cw_driver_args::*field; const char* what; };
const NumFlag kNumFlags[] = {
    {'s', &cw_driver_args::min_support, "overlaps a base needs to be inside a window"},
    {'S', &cw_driver_args::max_support, "overlaps kept per read, best first"},
    {'M', &cw_driver_args::max_msa, "sequences aligned per segment"},
    {'l', &cw_driver_args::window_size, "window length"},
    {'m', &cw_driver_args::window_overlap, "overlap of consecutive windows"},
    {'k', &cw_driver_args::mer_size, "k-mer size"},
    {'c', &cw_driver_args::common_kmers, "sequences that must share an anchor k-mer"},
    {'A', &cw_driver_args::min_anchors, "anchors a window needs for a consensus"},
    {'f', &cw_driver_args::solid_thresh, "occurrences that make a k-mer solid"},
    {'j', &cw_driver_args::nb_threads, "GPUs to use"},
};

int usage(const char* prog) {
    fprintf(stderr, "Usage: %s -a overlaps.paf -r %s [-R reads.fasta] [options] > corrected.fasta\n", prog, CW_DRIVER_POLISHING ? "contigs.fasta" : "reads.fasta");
    for (const NumFlag& f : kNumFlags) fprintf(stderr, "  -%c N   %s\n", f.letter, f.what);
    fprintf(stderr, "  -i, -p  accepted and ignored (kept for the CONSENT wrappers)\n\n");
    return EXIT_FAILURE;
}

} // namespace

int main(int argc, char* argv[]) {
    if (argc < 2) return usage(argv[0]);
    /* HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime starts.  The
       driver's two workers per device have a dozen kernel and copy streams; with four queues one worker's stream lands behind the other's
       long-running tier-L kernel (E. coli-scale run: 1.68-2.84 s with 4 queues, 1.57-1.58 s with 12; sixteen since a long run takes three
       workers per device -- measured equal to twelve on the eight-copy set).  Set here, in the executable and
       before anything touches HIP -- a library call must not change its host process's environment; a caller's own value is kept. */
    setenv("GPU_MAX_HW_QUEUES", "24", 0);
    std::string alignment_file, reads_file, proof_file, ignored;
    cw_driver_args a{};
    a.min_support = 3; a.max_support = 1000; a.max_msa = 150; a.window_size = 500; a.mer_size = 9; a.common_kmers = 8;  /* src/main.cpp:17-26 */
    a.min_anchors = 10; a.solid_thresh = 4; a.window_overlap = 50; a.nb_threads = 1;
    a.polishing = CW_DRIVER_POLISHING;
    for (int opt; (opt = getopt(argc, argv, kOptString)) != -1;) {
        bool known = false;
        for (const NumFlag& f : kNumFlags)
            if (f.letter == opt) { a.*(f.field) = (uint32_t)atoi(optarg); known = true; break; }
        if (known) continue;
        if (opt == 'a') alignment_file = optarg;
        else if (opt == 'r') reads_file = optarg;
        else if (opt == 'R') proof_file = optarg;
        else if (opt == 'i' || opt == 'p') ignored = optarg;
        else return usage(argv[0]);
    }
    a.paf_index = ""; a.path = "";
    a.alignment_file = alignment_file.c_str(); a.reads_file = reads_file.c_str(); a.proof_file = proof_file.c_str();
    const int rc = cw_run_correction(&a, STDOUT_FILENO, nullptr);
    if (rc != CW_OK) {
        fprintf(stderr, "%s: %s\n", argv[0], cw_strerror(rc));
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
