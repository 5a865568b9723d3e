// The Newton kernel of the 16-satellite build with TWO wavefronts per env (smj_wave.h SMJ_TWO_WAVES).  The build's 81.9 KB of LDS
// put two envs on a CU, so with one wavefront per env two of the CU's four SIMDs idle; here the env's second wavefront works the
// moving-moving pairs of every collision stage (bounding spheres, oriented boxes, MPR / multiccd / box-box) while the first one
// works the pairs with the static world (smj_sat.h collision_static), two workgroup barriers per step; the first wavefront then
// appends the second one's contacts to its own (smj_step_impl.h collision_convex).  Same capacities, same LDS, same contact list
// contact for contact -- and so the same states bit for bit -- as the one-wavefront kernel smj_kernels_sat.hip (option
// newton_two_waves = 0 selects that one).
#define SMJ_TWO_WAVES 1
#define SMJ_ONLY_NEWTON 1
#define SMJ_SAT 16
#define SMJ_SAT_ROWS 208
#define SMJ_SAT_CONTACTS 56
#define SMJ_SAT_DENSE 96
#define SMJ_SAT_ITEMS 16
#define SMJ_SAT_EXT 3
#define SMJ_VARIANT_TAG sat2
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#include "smj_step_tu.h"
