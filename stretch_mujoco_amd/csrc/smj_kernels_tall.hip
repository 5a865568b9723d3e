// The tall capacity variant of the step kernel (32 dofs, 160 constraint rows, 48 contacts; smj_model.h): contact-rich scenes
// around the robot, and the escalation target of the standard variant.
#define SMJ_TALL 1
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0   // the per-stage cycle counters cost this variant ~0.5 KB of scratch per lane; only the standard variant has a profiling build (smj_kernels_prof.hip)
#endif
#include "smj_step_tu.h"

void smj_tall_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats) {
  *nvp = NVP; *nbp = NBP; *nent = NENT; *nefc = NEFC; *ncon = NCON; *debug_floats = SMJ_DEBUG_FLOATS;
}
