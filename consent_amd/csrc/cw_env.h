/* cw_env.h -- environment switches.  The product library reads ten of them (devices, workers and host threads,
   the PAF block size, what to do on a capacity, statistics / timing / profile / task trace on stderr; tests/test_docs_env.py counts them); the test aids (shrunken capacities, dry run, ...) and the experiment knobs
   (work-groups per CU and tier, routing bounds, opt-in kernels) exist only in a -DCW_TEST_AIDS build of the same sources (consent_amd/aids/,
   built beside the product by consent_amd/_build.py): in the product they are constants, and a stray variable in a user's environment changes
   nothing. */
#ifndef CW_ENV_H
#define CW_ENV_H
#include <cstdlib>
#ifdef CW_TEST_AIDS
#define CW_AID_ENV(name) getenv(name)
#else
#define CW_AID_ENV(name) ((const char*)nullptr)
#endif
#endif
