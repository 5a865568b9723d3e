// Launchers of the gfx950 kernels (smj_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smj_model.h"

void smj_launch_step(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);
void smj_launch_reset(const DevModel& m, const DevState& s, const uint8_t* mask, hipStream_t stream);
