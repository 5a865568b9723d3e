// PGS-only twin of smj_kernels_big38.hip (the big variant with 38 dof columns, two envs per CU).
#define SMJ_ONLY_PGS 1
#define SMJ_BIG 1
#define SMJ_NVS 38
#define SMJ_VARIANT_TAG big38p
#define SMJ_PROFILING 0
#include "smj_step_tu.h"
