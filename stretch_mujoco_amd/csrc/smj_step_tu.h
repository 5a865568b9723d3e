// The step kernel and its launcher for ONE build of ONE capacity variant (smj_model.h): included by smj_kernels.hip (standard),
// smj_kernels_prof.hip (standard with the per-stage cycle counters), smj_kernels_tall.hip (SMJ_TALL, which also gets the
// escalation worker kernel) and smj_kernels_big.hip (SMJ_BIG).  Device code is compiled per translation unit, so the
// instantiations of StepKernel / Smem never meet.
#pragma once
#include "smj_kernels.h"
#include "smj_step_impl.h"

#define SMJ_CAT2(a, b) a##b
#define SMJ_CAT(a, b) SMJ_CAT2(a, b)
#if defined(SMJ_SAT)   // the satellite build (smj_sat.h)
#define SMJ_STEP_KERNEL SMJ_CAT(smj_step_kernel_, SMJ_VARIANT_TAG)
#define SMJ_LAUNCH_STEP SMJ_CAT(smj_launch_step_, SMJ_VARIANT_TAG)
#if SMJ_SAT == 32   // the escalation target of the 16-satellite build (smj_step_kernel_sat32_worker; the two-wavefront Newton build: _sat32n_worker)
#define SMJ_WORKER_KERNEL SMJ_CAT(SMJ_CAT(smj_step_kernel_, SMJ_VARIANT_TAG), _worker)
#endif
#elif defined(SMJ_BIG)   // three builds: SMJ_VARIANT_TAG = big38 / big50 / big (column capacity SMJ_NVS, smj_model.h)
#define SMJ_STEP_KERNEL SMJ_CAT(smj_step_kernel_, SMJ_VARIANT_TAG)
#define SMJ_LAUNCH_STEP SMJ_CAT(smj_launch_step_, SMJ_VARIANT_TAG)
#if SMJ_NVS == 64   // the escalation target of the 38- / 50-column builds
#define SMJ_WORKER_KERNEL smj_step_kernel_big_worker
#endif
#elif defined(SMJ_TALL) && defined(SMJ_TALL_ROWS) && defined(SMJ_ONLY_PGS)   // the 128-row build, PGS-only twin (smj_kernels_midp.hip)
#define SMJ_STEP_KERNEL smj_step_kernel_midp
#define SMJ_LAUNCH_STEP smj_launch_step_midp
#elif defined(SMJ_TALL) && defined(SMJ_TALL_ROWS)   // the 128-row build: primary kernel only, its steps escalate to the 160-row build
#define SMJ_STEP_KERNEL smj_step_kernel_mid
#define SMJ_LAUNCH_STEP smj_launch_step_mid
#elif defined(SMJ_TALL)
#define SMJ_STEP_KERNEL smj_step_kernel_tall
#define SMJ_LAUNCH_STEP smj_launch_step_tall
#define SMJ_WORKER_KERNEL smj_step_kernel_tall_worker   // the escalation target of the standard variant
#elif defined(SMJ_PROF_TU)
#define SMJ_STEP_KERNEL smj_step_kernel_prof
#define SMJ_LAUNCH_STEP smj_launch_step_prof
#elif defined(SMJ_ONLY_PGS)   // the standard variant's PGS-only build (smj_kernels_pgs.hip)
#define SMJ_STEP_KERNEL smj_step_kernel_pgs
#define SMJ_LAUNCH_STEP smj_launch_step_pgs
#else
#define SMJ_STEP_KERNEL smj_step_kernel
#define SMJ_LAUNCH_STEP smj_launch_step
#endif

// agent-scope relaxed atomics on the scheduling words (coherent across the XCDs' L2s); data handed over with them is fenced
#define SMJ_ALOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMJ_ASTORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SMJ_WAIT_TICKS 300000000LL   // 3 s of the 100 MHz constant clock: the bound of every wait loop below

// Pipelined chunks (DevState::pipe_len): what workgroup blockIdx.x of a normal launch runs.  go = false: nothing (the env
// is on the escalation list and nobody finishes its chunk now, or the wait ran out).
struct SmjTicket { int env, steps, chunk; unsigned read_flags; bool go; };
static __device__ __forceinline__ SmjTicket smj_take_ticket(const DevState& S, int nsteps, unsigned read_flags) {
  SmjTicket t;
  int slot = blockIdx.x;
  t.chunk = 0; t.steps = nsteps; t.read_flags = read_flags; t.go = true; t.env = 0;
  if (S.pipe_len) {   // workgroup = (chunk, slot), chunk-major
    t.chunk = (int)blockIdx.x / S.B;
    slot = (int)blockIdx.x - t.chunk * S.B;
    const int base = t.chunk * S.pipe_len;
    t.steps = nsteps - base < S.pipe_len ? nsteps - base : S.pipe_len;
    if (base + t.steps < nsteps) t.read_flags = 0;   // readouts go with the env's last chunk
  }
  if (slot >= S.B) { t.go = false; return t; }
  t.env = S.order ? S.order[slot] : slot;
  if (t.chunk > 0) {
    // wait for the env's previous chunk (normally long finished: it was dispatched B workgroups earlier).  Bounded: a wait of
    // seconds means the in-order dispatch this scheme leans on did not hold -- flag the env instead of hanging the device.
    const long long t0 = wall_clock64();
    for (;;) {
      const int seen = SMJ_ALOAD(&S.progress[t.env]);
      if (seen >= t.chunk) { t.go = seen < SMJ_PIPE_SWEPT; break; }
      const bool late = wall_clock64() - t0 > SMJ_WAIT_TICKS;
      if (seen < 0 && (late || SMJ_ALOAD(&S.sched[SMJ_SCHED_POLLERS]) == 0)) {
        // parked and nobody to finish the chunk now: the sweep after this kernel takes the env from where it stands
        if (threadIdx.x == 0) atomicCAS(&S.progress[t.env], seen, (int)SMJ_PIPE_ABANDONED);
        t.go = false;
        break;
      }
      if (late) {
        if (threadIdx.x == 0) atomicOr(&reinterpret_cast<int*>(S.stage + (size_t)t.env * S.lay.stride)[S.lay.info + SMJ_INFO_FLAGS], SMJ_FLAG_PIPE_TIMEOUT);
        t.go = false;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    coh_acquire();
  }
  return t;
}

#ifndef SMJ_KERNEL_ATTR
#define SMJ_KERNEL_ATTR
#endif
// The primary kernel of a variant: workgroup = one env (or one chunk of one env), no loop around run() -- a loop at this level
// keeps the whole step pipeline's live ranges alive across its back edge and costs hundreds of bytes of scratch per lane.
#ifdef SMJ_TWO_WAVES
#define SMJ_WG_THREADS 128
#else
#define SMJ_WG_THREADS 64
#endif
__global__ __launch_bounds__(SMJ_WG_THREADS) SMJ_KERNEL_ATTR void SMJ_STEP_KERNEL(const DevModel M, const DevState S, int nsteps, unsigned read_flags) {
  // dynamic LDS: a Newton launch asks for sizeof(Smem), a PGS launch for the extra tail that holds A (smj_lds_bytes)
  extern __shared__ __align__(16) unsigned char smj_lds[];
  Smem& smem = *reinterpret_cast<Smem*>(smj_lds);
#ifdef SMJ_TWO_WAVES
  if (threadIdx.x >= 64) {
#if SMJ_SPLIT_COLLIDE
    // the env's second wavefront under Newton: jobs the first one hands it (smj_step_impl.h helper()); the first wavefront names the
    // env with its first job -- or ends the launch
    WG_BARRIER();
    if (uni(smem.mbox[0]) == StepKernel::W2_EXIT) return;
    StepKernel h(M, S, smem, uni(smem.mbox[1]));
    h.helper();
#else
    // the env's second wavefront under PGS: the satellite islands' sweeps (smj_sat_pgs.h pgs_helper), nothing else
    StepKernel h(M, S, smem, 0);
    h.pgs_helper();
#endif
    return;
  }
#endif
  const SmjTicket t = smj_take_ticket(S, nsteps, read_flags);
  if (t.go) {
    StepKernel k(M, S, smem, t.env);
    k.step_base = t.chunk * S.pipe_len;
    k.pipe_chunk = t.chunk;
    k.run(t.steps, t.read_flags);
    if (S.pipe_len && !k.parked) {
      coh_release();
      if (threadIdx.x == 0) SMJ_ASTORE(&S.progress[t.env], t.chunk + 1);
    }
  }
  if (S.sched && threadIdx.x == 0) atomicAdd(&S.sched[SMJ_SCHED_EXITED], 1);
#ifdef SMJ_TWO_WAVES
  // release the second wavefront (every path of the first one ends here)
#if SMJ_SPLIT_COLLIDE
  if (threadIdx.x == 0) smem.mbox[0] = StepKernel::W2_EXIT;
#else
  if (threadIdx.x == 0) smem.sat.x[SX_MV][0][4] = __builtin_bit_cast(float, (int)StepKernel::PGS2_EXIT);
#endif
  WG_BARRIER();
#endif
}

#if defined(SMJ_WORKER_KERNEL)
// The escalation worker (the tall variant for the standard one, the 64-column big build for the 38- / 50-column ones): works the
// list of envs the smaller variant parked (DevState::redo) -- as the sweep after its kernel (redo_worker 1) or as a poller beside
// it (redo_worker 2, DevState::sched; standard variant only).  One call site of run().
static __device__ __forceinline__ void smj_worker_body(const DevModel& M, const DevState& S, Smem& smem, int nsteps, unsigned read_flags) {
  const int mode = S.redo_worker;
  long long t0 = mode == 2 ? wall_clock64() : 0;   // of the last sign of life of the standard kernel
  int exited_seen = -1;
  if (mode == 2) {
    if (S.pollers > 0 && *S.hot == 0) return;   // a quiet run: no second queue beside the standard kernel
    if (threadIdx.x == 0) atomicAdd(&S.sched[SMJ_SCHED_POLLERS], 1);
  }
  for (int i = blockIdx.x;; i += gridDim.x) {
    int env, steps = nsteps, chunk = 0;
    unsigned fl = read_flags;
    if (mode == 1) {
      const int cnt = SMJ_ALOAD(&S.sched[SMJ_SCHED_COUNT]);
      if (i == 0 && threadIdx.x == 0) *S.hot = cnt > 0 ? (int)SMJ_HOT_LAUNCHES : *S.hot > 0 ? *S.hot - 1 : 0;
      if (i >= cnt) return;
      env = S.redo[i];
      int old = 0;
      if (threadIdx.x == 0) old = atomicExch(&S.progress[env], (int)SMJ_PIPE_SWEPT);   // an env can be on the list more than once
      old = __builtin_amdgcn_readfirstlane(old);
      steps = nsteps - ld_coh(&S.done_steps[env]);
      if (old == SMJ_PIPE_SWEPT || steps <= 0) continue;
    } else {
      // poller: claim the next published entry; leave when the standard kernel is through and the list is drained
      int at = -1;
      for (;;) {
        const int exited = SMJ_ALOAD(&S.sched[SMJ_SCHED_EXITED]);   // read BEFORE the count: exited == total => the count is final
        const int cnt = SMJ_ALOAD(&S.sched[SMJ_SCHED_COUNT]), cl = SMJ_ALOAD(&S.sched[SMJ_SCHED_CLAIMED]);
        if (cl < cnt) {
          int got = 0;
          if (threadIdx.x == 0) got = atomicCAS(&S.sched[SMJ_SCHED_CLAIMED], cl, cl + 1);
          if (__builtin_amdgcn_readfirstlane(got) == cl) { at = cl; break; }
          continue;
        }
        if (exited != exited_seen) { exited_seen = exited; t0 = wall_clock64(); }
        if (exited >= S.pipe_total || wall_clock64() - t0 > SMJ_WAIT_TICKS) {
          if (threadIdx.x == 0) atomicSub(&S.sched[SMJ_SCHED_POLLERS], 1);   // waiters stop counting on pollers
          return;
        }
        __builtin_amdgcn_s_sleep(64);
      }
      t0 = wall_clock64();
      while ((env = SMJ_ALOAD(&S.redo[at])) < 0)   // the count is bumped before the entry is written
        if (wall_clock64() - t0 > SMJ_WAIT_TICKS) {
          if (threadIdx.x == 0) atomicSub(&S.sched[SMJ_SCHED_POLLERS], 1);
          return;
        }
      coh_acquire();
      const int done = ld_coh(&S.done_steps[env]);
      chunk = done / S.pipe_len;
      const int end = (chunk + 1) * S.pipe_len < nsteps ? (chunk + 1) * S.pipe_len : nsteps;
      steps = end - done;
      if (end < nsteps) fl = 0;
    }
    StepKernel k(M, S, smem, env);
    k.run(steps, fl);
    SYNC();
    if (mode == 2) {   // hand the env back to the standard kernel's next chunk (unless that one has given the env up)
      coh_release();
      if (threadIdx.x == 0) atomicCAS(&S.progress[env], -(chunk + 1), chunk + 1);
    }
  }
}
__global__ __launch_bounds__(SMJ_WG_THREADS) void SMJ_WORKER_KERNEL(const DevModel M, const DevState S, int nsteps, unsigned read_flags) {
  extern __shared__ __align__(16) unsigned char smj_lds[];
  Smem& smem = *reinterpret_cast<Smem*>(smj_lds);
#if SMJ_SPLIT_COLLIDE
  if (threadIdx.x >= 64) {   // the second wavefront: jobs of whichever env the first one is working (helper() takes the env from every forward-pass job)
    WG_BARRIER();
    if (uni(smem.mbox[0]) == StepKernel::W2_EXIT) return;
    StepKernel h(M, S, smem, uni(smem.mbox[1]));
    h.helper();
    return;
  }
#endif
  smj_worker_body(M, S, smem, nsteps, read_flags);
#if SMJ_SPLIT_COLLIDE
  if (threadIdx.x == 0) smem.mbox[0] = StepKernel::W2_EXIT;   // (every path of the first wavefront ends here)
  WG_BARRIER();
#endif
}
#endif

int SMJ_LAUNCH_STEP(const DevModel& m_in, const DevState& s, int nsteps, unsigned read_flags, hipStream_t stream) {
  // a build that carries one solver only (smj_step_impl.h newton()) refuses a launch for the other instead of running its own
#if defined(SMJ_ONLY_NEWTON)
  if (m_in.solver != 2) return SMJ_LAUNCH_REFUSED_SOLVER;
#elif defined(SMJ_ONLY_PGS)
  if (m_in.solver == 2) return SMJ_LAUNCH_REFUSED_SOLVER;
#endif
  DevModel m = m_in;
  size_t lds = smj_lds_bytes(m.solver != 2);
  // A PGS launch of a primary kernel that can hand steps over (DevState::redo) keeps to the rows whose A fits the struct
  // (smj_pgs_rows_static) and asks for no dynamic LDS: the variant's Newton occupancy instead of one env per CU.
  if (m.solver != 2 && s.redo && !s.redo_worker && lds > sizeof(Smem) && smj_pgs_rows_static() > 0) {
    m.pgs_cap = smj_pgs_rows_static();
    lds = sizeof(Smem);
  }
  static size_t lds_allowed_dev[64] = {};   // per device: the attribute belongs to the device's copy of the kernel
  int dev_now = 0;
  (void)hipGetDevice(&dev_now);
  size_t& lds_allowed = lds_allowed_dev[dev_now & 63];
  if (lds_allowed == 0) lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {   // beyond the default 64 KB per workgroup: raise the kernels' dynamic-LDS limit once (gfx950: 160 KB per CU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(SMJ_STEP_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#if defined(SMJ_WORKER_KERNEL)
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(SMJ_WORKER_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
    if (e != hipSuccess) return (int)e;
    lds_allowed = lds;
  }
#if defined(SMJ_WORKER_KERNEL)
  if (s.redo_worker) {
    const unsigned wg = s.redo_worker == 2 ? (unsigned)(s.pollers < 0 ? -s.pollers : s.pollers) : (unsigned)(nsteps <= 2 ? (s.B < 64 ? s.B : 64) : s.B < 512 ? s.B : 512);   // sweep: two envs per CU fit (one under PGS); surplus workgroups find the list drained and leave (a one-step launch: 64 -- the empty sweep is pure launch cost there)
    hipLaunchKernelGGL(SMJ_WORKER_KERNEL, dim3(wg), dim3(SMJ_WG_THREADS), lds, stream, m, s, nsteps, read_flags);
    return 0;
  }
#endif
  unsigned grid = s.B;
  if (s.pipe_len) grid = (unsigned)s.B * (unsigned)((nsteps + s.pipe_len - 1) / s.pipe_len);
  hipLaunchKernelGGL(SMJ_STEP_KERNEL, dim3(grid), dim3(SMJ_WG_THREADS), lds, stream, m, s, nsteps, read_flags);
  return 0;
}

#if defined(SMJ_BIG)
// capacities and layouts of this build for the host side (smj_capi.hip is compiled for the standard variant)
void SMJ_CAT(SMJ_CAT(smj_, SMJ_VARIANT_TAG), _caps)(int* nvp, int* nbp, int* nent, int* nefc, int* ncon, int* debug_floats, int* nvs) {
  *nvp = NVP; *nbp = NBP; *nent = NENT; *nefc = NEFC; *ncon = NCON; *debug_floats = SMJ_DEBUG_FLOATS; *nvs = NVS;
}
#endif
