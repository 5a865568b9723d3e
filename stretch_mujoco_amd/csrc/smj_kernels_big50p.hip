// PGS-only twin of smj_kernels_big50.hip (the big variant with 50 dof columns, two envs per CU).
#define SMJ_ONLY_PGS 1
#define SMJ_BIG 1
#define SMJ_NVS 50
#define SMJ_VARIANT_TAG big50p
#define SMJ_PROFILING 0
#include "smj_step_tu.h"
