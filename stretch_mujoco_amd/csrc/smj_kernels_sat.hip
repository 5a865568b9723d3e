// The satellite build of the step kernel (smj_sat.h): the main tree with the standard variant's mapping (32 dof lanes / columns)
// plus up to 16 satellites -- free objects, doors, drawers, knobs -- one lane each.  208 constraint rows (96 of them with a dense
// Jacobian row: the rows that touch the main tree), 56 contacts, up to 3 satellites coupled to the robot / to each other per step:
// 79 KB of LDS per env, two envs per CU.  The kernel of kitchens: the reference's scene.xml (table + 2 free objects), the kitchen
// stand-ins with free objects, exported Robocasa kitchens (robocasa_gen.py:129-239).  Larger models / steps: smj_kernels_sat32.hip.
#define SMJ_SAT 16
#ifndef SMJ_SAT_ROWS
#define SMJ_SAT_ROWS 208
#endif
#define SMJ_SAT_CONTACTS 56
#ifndef SMJ_SAT_DENSE
#define SMJ_SAT_DENSE 96
#endif
#define SMJ_SAT_ITEMS 16
#define SMJ_SAT_EXT 3
#define SMJ_VARIANT_TAG sat
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#define SMJ_ONLY_NEWTON 1   // the product build of this translation unit carries the Newton solver only (smj_step_impl.h newton()): PGS launches go to smj_kernels_satp.hip (two wavefronts per env) or smj_kernels_sat1.hip (one); the profiling build (csrc/Makefile bigprof) likewise -- PGS stage cycles come from the profiling copy of smj_kernels_sat1.hip
#include "smj_step_tu.h"

void smj_sat_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nsat) {
  *nvp = NVP; *nbp = NBP; *nent = NENT; *nefc = NEFC; *ncon = NCON; *debug_floats = SMJ_DEBUG_FLOATS; *nsat = NSAT;
}
