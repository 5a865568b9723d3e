// The Newton kernels of the large satellite build (smj_kernels_sat32.hip: 32 satellites, 320 rows, 64 contacts, one env per CU) with
// TWO wavefronts per env, as smj_kernels_sat2.hip is to smj_kernels_sat.hip: the second wavefront takes the satellites' forward
// pass, the moving-moving pairs of the collision stage, the satellites' Newton blocks and their integration (smj_step_impl.h
// helper()).  Primary kernel (models with more than 16 satellites) and escalation worker of the 16-satellite build -- in the worker the
// second wavefront follows the first one from env to env (the env travels with the first job of every step).  Same states bit for bit
// as the one-wavefront kernels (option newton_two_waves = 0 selects those).
#define SMJ_TWO_WAVES 1
#define SMJ_ONLY_NEWTON 1
#define SMJ_SAT 32
#define SMJ_SAT_ROWS 320
#define SMJ_SAT_CONTACTS 64
#define SMJ_SAT_DENSE 256
#define SMJ_SAT_EXT 4
#define SMJ_SAT_ITEMS 40
#define NCH 64
#define SMJ_VARIANT_TAG sat32n
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#include "smj_step_tu.h"
