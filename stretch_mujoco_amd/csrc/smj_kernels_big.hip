// The big capacity variant of the step kernel (64 dofs, 160 constraint rows, 48 contacts; smj_model.h): scenes with several
// free objects -- the reference's own scene.xml (table + 2 objects, models/scene.xml:21-35) and the kitchens.
#define SMJ_BIG 1
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0   // the per-stage cycle counters cost this variant ~0.5 KB of scratch per lane; only the standard variant has a profiling build (smj_kernels_prof.hip)
#endif
#include "smj_step_tu.h"

// capacities and layouts of this variant for the host side (smj_capi.hip is compiled for the standard variant)
void smj_big_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats) {
  *nvp = NVP; *nbp = NBP; *nent = NENT; *nefc = NEFC; *ncon = NCON; *debug_floats = SMJ_DEBUG_FLOATS;
}
