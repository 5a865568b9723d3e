// The Newton kernel of the 16-satellite build with TWO wavefronts per env (smj_wave.h SMJ_TWO_WAVES).  The build's 80 KB of LDS put
// two envs on a CU, so with one wavefront per env two of the CU's four SIMDs idle.  Here the env's second wavefront takes jobs off
// the first one's critical path (smj_step_impl.h helper(): a mailbox in the last 16 bytes of the LDS, two workgroup barriers per job):
//   * the satellites' forward pass (sat_forward) beside the main tree's kinematics;
//   * the moving-moving pairs of the collision stage (bounding spheres, oriented boxes, MPR / multiccd / box-box) beside the pairs
//     with the static world (smj_sat.h collision_static) -- its contacts come back in the slots NCON - 1, NCON - 2, ... and the first
//     wavefront appends them to its own (collision_convex);
//   * in every Newton iteration the satellites' 6 x 6 blocks and the search direction of the uncoupled ones (sat_hessian,
//     sat_solve_own) beside the main block's H = M + J' W J on the matrix cores;
//   * the satellites' integration (sat_integrate) beside the main tree's.
// Every job is work the first wavefront does itself in the one-wavefront kernel (smj_kernels_sat.hip), in the same arithmetic order on
// the same data: same capacities, same contact list contact for contact, the same states BIT FOR BIT (tests/test_satellites.py
// test_gpu_newton_two_wavefronts_per_env_equal_one_bit_for_bit; option newton_two_waves = 0 selects the one-wavefront kernel).
#define SMJ_TWO_WAVES 1
#define SMJ_ONLY_NEWTON 1
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
#define SMJ_VARIANT_TAG sat2
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#include "smj_step_tu.h"

int smj_sat2_profiling() { return SMJ_PROFILING; }   // 1 in a tools build (csrc/Makefile bigprof): the first wavefront's per-stage cycle counters are compiled in
