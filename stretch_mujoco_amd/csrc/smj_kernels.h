// Launchers of the gfx950 kernels (smj_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
// a launcher's answer when its build does not carry the solver asked for (nothing was launched): its own value, not a HIP error code --
// a genuine hipErrorInvalidValue of a launch (a bad LDS size ...) must not be mistaken for it (smj_capi.hip by_solver)
#define SMJ_LAUNCH_REFUSED_SOLVER (-7001)
#include <stdint.h>

#include "smj_model.h"

// One batch-major array [rows][ld] (4-byte words) <-> words off..off+rows of every env's staging row
// (a run of at most 128 rows: one LDS tile per block)
struct StageSeg { void* ptr; int rows, off; };
struct StagePlan {
  StageSeg seg[24];
  int row0[24] = {};   // first row of the run inside its array
  int nseg = 0;
  void add(void* p, int rows, int off) {
    for (int r0 = 0; p && r0 < rows && nseg < 24; r0 += 128) {
      seg[nseg] = StageSeg{p, rows - r0 < 128 ? rows - r0 : 128, off + r0};
      row0[nseg++] = r0;
    }
  }
};

// the capacity variants of the step kernel (smj_model.h); return 0 or a hipError_t
int smj_launch_step(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);
int smj_launch_step_pgs(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);    // standard, PGS-only build (smj_launch_step: Newton-only)
int smj_launch_step_prof(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);   // standard + cycle counters
int smj_launch_step_tall(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);
int smj_launch_step_mid(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);      // tall with 128 rows: three envs per CU
int smj_launch_step_midp(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);     // PGS-only twins of mid / big38 / big50 (those carry the Newton solver only)
int smj_launch_step_big38p(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);
int smj_launch_step_big50p(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);
int smj_launch_step_big(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);     // 64 dof columns
int smj_launch_step_big38(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);   // 38
int smj_launch_step_big50(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);   // 50
int smj_launch_step_satp(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);    // the same build with two wavefronts per env: PGS, satellite islands beside the dense system (smj_kernels_satp.hip)
int smj_launch_step_sat2(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);    // 16 satellites, Newton only, two wavefronts per env (option newton_two_waves = 1)
int smj_sat2_profiling();   // whether that build carries the per-stage cycle counters (DevState::prof)
int smj_launch_step_sat1(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);    // 16 satellites, PGS only, one wavefront per env (option pgs_two_waves = 0)
int smj_launch_step_sat(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);     // main tree + satellites (smj_sat.h)
int smj_launch_step_sat32n(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);   // the same, Newton only, two wavefronts per env (option newton_two_waves = 1): primary kernel and escalation worker
int smj_launch_step_sat32(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream);   // up to 32 satellites, one env per CU
void smj_sat_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nsat);
void smj_sat32_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nsat);
void smj_tall_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats);
void smj_mid_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats);
void smj_big_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nvs);
void smj_big38_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nvs);
void smj_big50_caps(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nvs);
void smj_launch_reset(const DevModel& m, const DevState& s, const uint8_t* mask, hipStream_t stream);
// batch-major -> env-major staging rows (import) and back (export); tiles of 64 envs transposed through LDS
void smj_launch_stage(const StagePlan& plan, float* stage, int stride, int B, long ld, bool is_export, hipStream_t stream);
// launch order for the next step launch: envs by descending cost (256 buckets relative to the maximum), one workgroup
void smj_launch_order(const int* cost, int* order, int B, hipStream_t stream);
// one BaseController.update() on the bound BASE_POSE / BASECTL / CTRL arrays (lane = env); the same device function the
// step kernel runs after every step
void smj_launch_base_tick(const DevState& s, hipStream_t stream);
