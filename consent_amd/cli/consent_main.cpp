/*
 * consent_main.cpp -- main() of bin/CONSENT-correction and bin/CONSENT-polishing (the same source built twice, as the reference
 * links the same src/main.o against one driver object or the other, Makefile:26,32).
 *
 * The command line is the reference's (src/main.cpp:29-76): getopt string "a:A:d:k:s:S:M:l:f:e:p:c:m:j:w:m:r:R:n:i:", the same
 * defaults (:17-26), -i and -p stored and never read; -d -e -w -n are in the getopt string but have no case in the reference's switch
 * (:77 `default`), so there, and here, they print the usage text and fail like any unknown letter.  So the unmodified wrappers (CONSENT-correct:202, CONSENT-polish:197) can call these binaries:
 *   CONSENT-correction -a aln.paf -s 3 -S 150 -l 500 -k 9 -c 8 -A 2 -f 4 -m 50 -j $nproc -r reads.fa -M 150 -p $dir >> out
 *   CONSENT-polishing  -a aln.paf -s 1 -S 20000 ... -j $nproc -r contigs.fa -R reads.fa -M 150 -p $dir >> out
 * Everything after the flags is cw_run_correction (include/consent_amd.h); -j is the number of GPUs to use.
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

int main(int argc, char* argv[]) {
    if (argc < 2) {
        fprintf(stderr, "Usage: %s [-a alignmentFile.paf] [-s minSupportForGoodRegions] [-l minLengthForGoodRegions] [-j threadsNb] \n\n", argv[0]);
        exit(EXIT_FAILURE);
    }
    std::string paf_index, alignment_file, reads_file, proof_file, path;
    cw_driver_args a{};
    a.min_support = 3; a.max_support = 1000; a.max_msa = 150; a.window_size = 500; a.mer_size = 9; a.common_kmers = 8;
    a.min_anchors = 10; a.solid_thresh = 4; a.window_overlap = 50; a.nb_threads = 1;
    a.polishing = CW_DRIVER_POLISHING;
    int opt;
    while ((opt = getopt(argc, argv, "a:A:d:k:s:S:M:l:f:e:p:c:m:j:w:m:r:R:n:i:")) != -1) {
        switch (opt) {
        case 'i': paf_index = optarg; break;
        case 'a': alignment_file = optarg; break;
        case 's': a.min_support = (uint32_t)atoi(optarg); break;
        case 'S': a.max_support = (uint32_t)atoi(optarg); break;
        case 'M': a.max_msa = (uint32_t)atoi(optarg); break;
        case 'l': a.window_size = (uint32_t)atoi(optarg); break;
        case 'k': a.mer_size = (uint32_t)atoi(optarg); break;
        case 'c': a.common_kmers = (uint32_t)atoi(optarg); break;
        case 'A': a.min_anchors = (uint32_t)atoi(optarg); break;
        case 'f': a.solid_thresh = (uint32_t)atoi(optarg); break;
        case 'm': a.window_overlap = (uint32_t)atoi(optarg); break;
        case 'r': reads_file = optarg; break;
        case 'R': proof_file = optarg; break;
        case 'p': path = optarg; path += "/BMEAN/BOA/blosum80.mat"; break;
        case 'j': a.nb_threads = (uint32_t)atoi(optarg); break;
        default:
            fprintf(stderr, "Usage: %s [-a alignmentFile.paf] [-k merSize] [-s minSupportForGoodRegions] [-l minLengthForGoodRegions] [-f freqThresholdForKMers] [-e maxError] [-p freqThresholdForKPersFreqs] [-c freqThresholdForKPersCons] [-m mode (0 for regions, 1 for cluster)] [-j threadsNb] \n\n", argv[0]);
            exit(EXIT_FAILURE);
        }
    }
    a.paf_index = paf_index.c_str(); a.alignment_file = alignment_file.c_str(); a.reads_file = reads_file.c_str();
    a.proof_file = proof_file.c_str(); a.path = path.c_str();
    const int rc = cw_run_correction(&a, STDOUT_FILENO, nullptr);
    if (rc != CW_OK) {
        fprintf(stderr, "%s: %s\n", argv[0], cw_strerror(rc));
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
