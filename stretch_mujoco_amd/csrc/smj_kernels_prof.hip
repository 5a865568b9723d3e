// The standard variant of the step kernel WITH the per-stage shader-cycle counters (DevState::prof, SMJ_SLOT_PROF): what
// smj_step launches when the profiling slot is bound (tools/gpu_diag.py).  Same code, same arithmetic.
#define SMJ_PROFILING 1
#define SMJ_PROF_TU 1
#include "smj_step_tu.h"
