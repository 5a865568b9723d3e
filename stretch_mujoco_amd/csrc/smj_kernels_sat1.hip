// The 16-satellite build with the PGS solver only and ONE wavefront per env: what option pgs_two_waves = 0 selects (the comparator of
// smj_kernels_satp.hip in the tests).  smj_kernels_sat.hip carries the Newton solver only.
#define SMJ_ONLY_PGS 1
#define SMJ_SAT 16
#define SMJ_SAT_ROWS 208
#define SMJ_SAT_CONTACTS 56
#define SMJ_SAT_DENSE 96
#define SMJ_SAT_ITEMS 16
#define SMJ_SAT_EXT 3
#define SMJ_VARIANT_TAG sat1
#define SMJ_PROFILING 0
#include "smj_step_tu.h"
