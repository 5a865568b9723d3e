// The PGS kernel of the 16-satellite build: TWO wavefronts per env (smj_wave.h SMJ_TWO_WAVES).  Wavefront 0 runs the step exactly
// as smj_kernels_sat.hip does; during the PGS sweeps wavefront 1 sweeps the satellite islands (one lane each, smj_sat_pgs.h
// pgs_helper) BESIDE wavefront 0's sweeps of the dense system -- constraint islands do not see each other's rows, so the two
// run concurrently on two of the CU's four SIMDs (the build's 81.9 KB of LDS put two envs on a CU: with one wavefront per env
// the other two SIMDs idle), meet at one workgroup barrier per sweep, add their improvements and take the same decision.
// Same capacities, same LDS, same results as the one-wavefront kernel (option pgs_two_waves = 0 selects that one).
#define SMJ_TWO_WAVES 1
#define SMJ_ONLY_PGS 1
#define SMJ_SAT 16
#define SMJ_SAT_ROWS 208
#define SMJ_SAT_CONTACTS 56
#define SMJ_SAT_DENSE 96
#define SMJ_SAT_ITEMS 16
#define SMJ_SAT_EXT 3
#define SMJ_VARIANT_TAG satp
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#include "smj_step_tu.h"
