// The big capacity variant of the step kernel (64 dof lanes, 160 constraint rows, 48 contacts; smj_model.h): scenes with several
// free objects -- the reference's own scene.xml (table + 2 objects, models/scene.xml:21-35) and the kitchens.  This build: 50 dof columns
// (the robot + four free objects: the kitchen of SURVEY.md 8(d)), under 80 KB of LDS per env: two envs per CU.
#define SMJ_BIG 1
#define SMJ_NVS 50
#define SMJ_VARIANT_TAG big50
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0   // the per-stage cycle counters cost registers; tools build a profiling copy with -DSMJ_PROFILING=1
#endif
#if !SMJ_PROFILING
#define SMJ_ONLY_NEWTON 1   // product build: the Newton solver only; PGS launches go to smj_kernels_big50p.hip (smj_step_impl.h newton()); the profiling copy keeps both
#endif
#include "smj_step_tu.h"
