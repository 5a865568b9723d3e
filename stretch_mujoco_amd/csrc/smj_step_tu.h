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
  int slot = blockIdx.x, chunk = 0, steps = nsteps;
  if (S.pipe_len) {   // pipelined chunks: workgroup = (chunk, slot), chunk-major (DevState::pipe_len)
    chunk = (int)blockIdx.x / S.B;
    slot = (int)blockIdx.x - chunk * S.B;
    const int base = chunk * S.pipe_len;
    steps = nsteps - base < S.pipe_len ? nsteps - base : S.pipe_len;
    if (base + steps < nsteps) read_flags = 0;   // readouts go with the env's last chunk
  }
  if (slot >= S.B) return;
  const int env = S.order ? S.order[slot] : slot;
  if (chunk > 0) {
    // wait for the env's previous chunk (normally long finished: it was dispatched B workgroups earlier).  Bounded: a wait of
    // seconds means the in-order dispatch this scheme leans on did not hold -- flag the env instead of hanging the device.
    const long long t0 = wall_clock64();
    int seen;
    while ((seen = __hip_atomic_load(&S.progress[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < chunk) {
      __builtin_amdgcn_s_sleep(32);
      if (wall_clock64() - t0 > 300000000LL) {   // 3 s of the 100 MHz constant clock
        if (threadIdx.x == 0) atomicOr(&reinterpret_cast<int*>(S.stage + (size_t)env * S.lay.stride)[S.lay.info + SMJ_INFO_FLAGS], SMJ_FLAG_PIPE_TIMEOUT);
        return;
      }
    }
    if (seen >= SMJ_PIPE_PARKED) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  StepKernel k(M, S, smem, env);
  k.step_base = chunk * S.pipe_len;
  k.run(steps, read_flags);
  if (S.pipe_len) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) __hip_atomic_store(&S.progress[env], k.parked ? (int)SMJ_PIPE_PARKED : chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
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
  unsigned grid = s.redo_worker ? (s.B < 128 ? s.B : 128) : s.B;
#if !defined(SMJ_TALL) && !defined(SMJ_BIG)
  if (s.pipe_len) grid = (unsigned)s.B * (unsigned)((nsteps + s.pipe_len - 1) / s.pipe_len);
#endif
  hipLaunchKernelGGL(SMJ_STEP_KERNEL, dim3(grid), dim3(64), lds, stream, m, s, nsteps, read_flags);
  return 0;
}
