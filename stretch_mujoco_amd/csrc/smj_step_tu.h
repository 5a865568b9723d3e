// The step kernel and its launcher for ONE capacity variant (smj_model.h): included by smj_kernels.hip (standard) and by
// smj_kernels_tall.hip (SMJ_TALL) and smj_kernels_big.hip (SMJ_BIG).  Device code is compiled per translation unit, so the two instantiations of StepKernel / Smem
// never meet.
#pragma once
#include "smj_kernels.h"
#include "smj_step_impl.h"

#if defined(SMJ_BIG)
#define SMJ_STEP_KERNEL smj_step_kernel_big
#define SMJ_LAUNCH_STEP smj_launch_step_big
#elif defined(SMJ_TALL)
#define SMJ_STEP_KERNEL smj_step_kernel_tall
#define SMJ_LAUNCH_STEP smj_launch_step_tall
#else
#define SMJ_STEP_KERNEL smj_step_kernel
#define SMJ_LAUNCH_STEP smj_launch_step
#endif

__global__ __launch_bounds__(64) void SMJ_STEP_KERNEL(const DevModel M, const DevState S, int nsteps, unsigned read_flags) {
  // dynamic LDS: a Newton launch asks for sizeof(Smem), a PGS launch for the extra tail that holds A (smj_lds_bytes)
  extern __shared__ __align__(16) unsigned char smj_lds[];
  Smem& smem = *reinterpret_cast<Smem*>(smj_lds);
#if defined(SMJ_TALL) || defined(SMJ_BIG)
  // One call site of run() (the whole step pipeline is inlined into it): a normal launch takes env = blockIdx.x, steps = nsteps;
  // an escalation launch works the list of envs the standard variant parked (DevState::redo).
  int i = blockIdx.x, last = blockIdx.x;
  if (S.redo_worker) last = S.redo[0] - 1;
  for (; i <= last; i += gridDim.x) {
    if (i >= S.B) return;
    int env = S.order ? S.order[i] : i, steps = nsteps;
    if (S.redo_worker) { env = S.redo[1 + 2 * i]; steps = nsteps - S.redo[2 + 2 * i]; }
    StepKernel k(M, S, smem, env);
    k.run(steps, read_flags);
    __syncthreads();
  }
#else
  if ((int)blockIdx.x >= S.B) return;
  const int env = S.order ? S.order[blockIdx.x] : (int)blockIdx.x;
  StepKernel k(M, S, smem, env);
  k.run(nsteps, read_flags);
#endif
}

int SMJ_LAUNCH_STEP(const DevModel& m, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream) {
  const size_t lds = smj_lds_bytes(m.solver != 2);
  static size_t lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {   // beyond the default 64 KB per workgroup: raise the kernel's dynamic-LDS limit once (gfx950: 160 KB per CU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(SMJ_STEP_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_allowed = lds;
  }
  hipLaunchKernelGGL(SMJ_STEP_KERNEL, dim3(s.redo_worker ? (s.B < 128 ? s.B : 128) : s.B), dim3(64), lds, stream, m, s, nsteps, read_flags);
  return 0;
}
