// gfx950 kernels of the batched Stretch physics path.  One wavefront (64 lanes) per environment, one
// workgroup per wavefront; the per-environment working set (body tree, mass-matrix factor, constraint
// Jacobian, A = J M^-1 J' + R) lives in LDS for the whole launch, model constants stream from L2.
#include "smj_kernels.h"
#include "smj_step_impl.h"

__global__ __launch_bounds__(64) void smj_step_kernel(const DevModel M, const DevState S, int nsteps, unsigned read_flags) {
  // dynamic LDS: a Newton launch asks for sizeof(Smem), a PGS launch for the extra tail that holds A (smj_lds_bytes)
  extern __shared__ __align__(16) unsigned char smj_lds[];
  Smem& smem = *reinterpret_cast<Smem*>(smj_lds);
  const int env = blockIdx.x;
  if (env >= S.B) return;
  StepKernel k(M, S, smem, env);
  k.run(nsteps, read_flags);
}

// mj_resetData for masked envs: batch-major, lanes = envs (coalesced)
__global__ __launch_bounds__(256) void smj_reset_kernel(const DevModel M, const DevState S, const uint8_t* mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.B) return;
  if (mask && !mask[e]) return;
  for (int k = 0; k < M.nq; k++) S.qpos[k * S.ld + e] = M.qpos0[k];
  for (int k = 0; k < M.nv; k++) { S.qvel[k * S.ld + e] = 0.f; S.warm[k * S.ld + e] = 0.f; }
  for (int k = 0; k < M.nu; k++) S.ctrl[k * S.ld + e] = 0.f;
  S.nstep[e] = 0;
  for (int k = 0; k < 4; k++) S.info[k * S.ld + e] = 0;
}

void smj_launch_step(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream) {
  hipLaunchKernelGGL(smj_step_kernel, dim3(s.B), dim3(64), smj_lds_bytes(m.solver != 2), stream, m, s, nsteps, read_flags);
}
void smj_launch_reset(const DevModel& m, const DevState& s, const uint8_t* mask, hipStream_t stream) {
  hipLaunchKernelGGL(smj_reset_kernel, dim3((s.B + 255) / 256), dim3(256), 0, stream, m, s, mask);
}
