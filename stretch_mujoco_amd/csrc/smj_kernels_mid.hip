// The three-envs-per-CU build of the tall capacity variant (32 dofs, 128 constraint rows, 44 contacts; smj_model.h): 53.3 KB of
// LDS per env instead of 59.8 KB -- 42 allocation granules of 1280 B, so that three envs fit a CU (163 840 B).  It is the PRIMARY kernel of contact-rich scenes around the robot (kitchen fixtures); a step
// that needs more rows is finished by the 160-row build (smj_kernels_tall.hip), like a step of the standard variant.
#define SMJ_TALL 1
#define SMJ_TALL_ROWS 128
#define SMJ_TALL_CONTACTS 44
#define SMJ_VARIANT_TAG mid
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#if !SMJ_PROFILING
#define SMJ_ONLY_NEWTON 1   // product build: the Newton solver only; PGS launches go to smj_kernels_midp.hip (smj_step_impl.h newton()); the profiling copy keeps both
#endif
#include "smj_step_tu.h"

void smj_mid_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats) {
  *nvp = NVP; *nbp = NBP; *nent = NENT; *nefc = NEFC; *ncon = NCON; *debug_floats = SMJ_DEBUG_FLOATS;
}
