// PGS-only twin of smj_kernels_mid.hip (the three-envs-per-CU build of the tall variant: 128 rows, 44 contacts).
#define SMJ_ONLY_PGS 1
#define SMJ_TALL 1
#define SMJ_TALL_ROWS 128
#define SMJ_TALL_CONTACTS 44
#define SMJ_VARIANT_TAG midp
#define SMJ_PROFILING 0
#include "smj_step_tu.h"
