/*
 * paf_tools.cpp -- bin/explode, bin/merge, bin/reformatPAF: the three small programs the wrappers run around the driver
 * (CONSENT-correct:194-196, CONSENT-polish:193), each a main() over the library's host function of the same job:
 *   explode  in.paf prefix                 (src/explode.cpp:53-60)     -> cw_paf_explode
 *   merge    out.paf headers chunk...      (src/merge.cpp:29-65)       -> cw_paf_merge
 *   reformatPAF in.paf out.paf             (src/reformatPAF.cpp:35-49) -> cw_paf_reformat
 * Built three times with -DCW_TOOL=1|2|3.  Like the reference's programs they take their arguments positionally and say nothing.
 */
#include <cstdio>
#include <cstdlib>

#include "consent_amd.h"

int main(int argc, char* argv[]) {
#if CW_TOOL == 1
    if (argc < 3) { fprintf(stderr, "Usage: %s alignments.paf outputPrefix\n", argv[0]); return EXIT_FAILURE; }
    const int rc = cw_paf_explode(argv[1], argv[2], nullptr);
#elif CW_TOOL == 2
    if (argc < 3) { fprintf(stderr, "Usage: %s merged.paf headersFile chunk_1 [chunk_2 ...]\n", argv[0]); return EXIT_FAILURE; }
    const int rc = cw_paf_merge(argv[1], argv[2], argv + 3, (uint32_t)(argc - 3));
#elif CW_TOOL == 3
    if (argc < 3) { fprintf(stderr, "Usage: %s in.paf out.paf\n", argv[0]); return EXIT_FAILURE; }
    const int rc = cw_paf_reformat(argv[1], argv[2]);
#else
#error "CW_TOOL must be 1 (explode), 2 (merge) or 3 (reformatPAF)"
#endif
    if (rc != CW_OK) { fprintf(stderr, "%s: %s\n", argv[0], cw_strerror(rc)); return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}
