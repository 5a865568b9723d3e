// The standard variant of the step kernel with the PGS solver only (smj_kernels.hip carries the Newton-only twin): smj_step launches
// this one when DevModel::solver is PGS.  Same stages, same arithmetic; the split exists because a kernel's register allocation pays
// for every path compiled into it.
#define SMJ_PROFILING 0
#define SMJ_ONLY_PGS 1
#include "smj_step_tu.h"
