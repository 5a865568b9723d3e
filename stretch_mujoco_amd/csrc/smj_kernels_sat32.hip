// The large satellite build (smj_sat.h): up to 32 satellites, 320 rows (256 of them dense), 64 contacts, 4 coupled satellites per step --
// one env per CU.  Models with more than 16 satellites, and the escalation target of the 16-satellite build: an env whose step
// needs more rows / contacts / coupled satellites than that build holds is finished here (DevState::redo, as standard -> tall).
#define SMJ_SAT 32
#define SMJ_SAT_ROWS 320
#define SMJ_SAT_CONTACTS 64
#define SMJ_SAT_DENSE 256
#define SMJ_SAT_EXT 4
#define SMJ_SAT_ITEMS 40
#define NCH 64   // a cone-Hessian block for every contact (one env per CU: the LDS is there)
#define SMJ_VARIANT_TAG sat32
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 0
#endif
#include "smj_step_tu.h"

void smj_sat32_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nsat) {
  *nvp = NVP; *nbp = NBP; *nent = NENT; *nefc = NEFC; *ncon = NCON; *debug_floats = SMJ_DEBUG_FLOATS; *nsat = NSAT;
}
