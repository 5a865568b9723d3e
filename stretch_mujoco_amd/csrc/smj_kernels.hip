// gfx950 kernels of the batched Stretch physics path.  One wavefront (64 lanes) per environment, one
// workgroup per wavefront; the per-environment working set (body tree, mass-matrix factor, constraint
// Jacobian, A = J M^-1 J' + R) lives in LDS for the whole launch, model constants stream from L2.
#include "smj_kernels.h"
// the standard variant of the step kernel, without the per-stage cycle counters: they are runtime-optional but cost the
// kernel registers it does not have (scratch 144 -> 48 B per lane); smj_kernels_prof.hip compiles the same kernel with them
#define SMJ_PROFILING 0
#define SMJ_ONLY_NEWTON 1   // this translation unit's step kernel carries the Newton solver only; smj_kernels_pgs.hip is its PGS twin (smj_step_impl.h newton())
#include "smj_step_tu.h"

// mj_resetData for masked envs: batch-major, lanes = envs (coalesced)
__global__ __launch_bounds__(256) void smj_reset_kernel(const DevModel M, const DevState S, const uint8_t* mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.B) return;
  if (mask && !mask[e]) return;
  for (int k = 0; k < M.nq_all; k++) S.qpos[k * S.ld + e] = M.qpos0[k];
  for (int k = 0; k < M.nv_all; k++) { S.qvel[k * S.ld + e] = 0.f; S.warm[k * S.ld + e] = 0.f; }
  for (int k = 0; k < M.nu; k++) S.ctrl[k * S.ld + e] = 0.f;
  if (S.bctl)
    for (int k = 0; k < SMJ_BC_ROWS; k++) S.bctl[k * S.ld + e] = 0.f;
  S.nstep[e] = 0;
  for (int k = 0; k < 4; k++) S.info[k * S.ld + e] = 0;
  // what the library keeps between steps beside the state must not outlive the episode (mj_resetData leaves nothing behind): the PGS
  // second start (row count 0 = none), the kept contact manifolds and separating directions (tag word 0 = empty entry)
  if (S.pgsprev) S.pgsprev[(size_t)e * SMJ_PGSPREV_STRIDE] = 0.f;
  if (S.mcache)
    for (int k = 0; k < SMJ_MC_SLOTS; k++) S.mcache[((size_t)e * SMJ_MC_SLOTS + k) * SMJ_MC_WORDS] = 0.f;
  if (S.sepcache)
    for (int k = 0; k < SMJ_SEP_SLOTS; k++) S.sepcache[((size_t)e * SMJ_SEP_SLOTS + k) * 4 + 3] = 0.f;
}

// Staging transposes.  A block moves one tile: 64 envs x up to 128 rows of one batch-major array (blockIdx.y = tile).  Rows of
// the batch-major array are read / written with lanes = envs (256-byte coalesced runs), the env-major rows with lanes =
// consecutive words; the tile turns in LDS (row stride 65: conflict-free).
__global__ __launch_bounds__(256) void smj_stage_kernel(const StagePlan P, float* stage, int stride, int B, long ld, int is_export) {
  __shared__ float tile[128][65];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, env0 = blockIdx.x * 64, sg = blockIdx.y;
  float* arr = static_cast<float*>(P.seg[sg].ptr) + (long)P.row0[sg] * ld;
  const int nr = P.seg[sg].rows, off = P.seg[sg].off;
  if (!is_export) {
    for (int k = w; k < nr; k += 4) tile[k][l] = env0 + l < B ? arr[(long)k * ld + env0 + l] : 0.f;
    __syncthreads();
    for (int idx = t; idx < 64 * nr; idx += 256) {
      const int e = idx / nr, k = idx - e * nr;
      if (env0 + e < B) stage[(size_t)(env0 + e) * stride + off + k] = tile[k][e];
    }
  } else {
    for (int idx = t; idx < 64 * nr; idx += 256) {
      const int e = idx / nr, k = idx - e * nr;
      if (env0 + e < B) tile[k][e] = stage[(size_t)(env0 + e) * stride + off + k];
    }
    __syncthreads();
    for (int k = w; k < nr; k += 4)
      if (env0 + l < B) arr[(long)k * ld + env0 + l] = tile[k][l];
  }
}

// BaseController.update() for every env on the bound arrays (lane = env): same arithmetic as StepKernel::base_controller
__global__ __launch_bounds__(256) void smj_base_tick_kernel(const DevState S) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.B || !S.bctl) return;
  const long ld = S.ld;
  const int mode = (int)S.bctl[SMJ_BC_MODE * ld + e];
  if (mode == 0) return;
  const float x = S.base[e], y = S.base[ld + e], th = S.base[2 * ld + e];
  const float inc = S.bctl[SMJ_BC_INC * ld + e], sign = inc > 0.f ? 1.f : -1.f;
  float v = 0.f, w = 0.f;
  int next = mode;
  if (mode == 1) {
    const float dx = x - S.bctl[SMJ_BC_X0 * ld + e], dy = y - S.bctl[SMJ_BC_Y0 * ld + e];
    if (!(sqrtf(dx * dx + dy * dy) <= fabsf(inc))) next = 0; else v = SMJ_BASE_X_VEL * sign;
  } else if (mode == 2) {
    if (!(fabsf(S.bctl[SMJ_BC_TH0 * ld + e] - th) <= fabsf(inc))) next = 0; else w = SMJ_BASE_R_VEL * sign;
  } else { v = S.bctl[SMJ_BC_V * ld + e]; w = S.bctl[SMJ_BC_W * ld + e]; }
  S.ctrl[e] = (v - (w * SMJ_WHEEL_SEPARATION / 2.f)) / SMJ_WHEEL_RADIUS;
  S.ctrl[ld + e] = (v + (w * SMJ_WHEEL_SEPARATION / 2.f)) / SMJ_WHEEL_RADIUS;
  S.bctl[SMJ_BC_MODE * ld + e] = (float)next;
}

// Envs sorted by the shader time of their last launch, longest first: a counting sort on 256 cost buckets relative to the
// maximum (order inside a bucket does not matter: envs are independent, only the dispatch order changes).
__global__ __launch_bounds__(1024) void smj_order_kernel(const int* cost, int* order, int B) {
  __shared__ int hist[256], mx;
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  if (t == 0) mx = 1;
  __syncthreads();
  int m = 1;
  for (int e = t; e < B; e += 1024) m = cost[e] > m ? cost[e] : m;
  atomicMax(&mx, m);
  __syncthreads();
  const float sc = 255.f / (float)mx;
  for (int e = t; e < B; e += 1024) atomicAdd(&hist[255 - (int)((float)(cost[e] > 0 ? cost[e] : 0) * sc)], 1);
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < 256; k++) { const int n = hist[k]; hist[k] = acc; acc += n; }
  }
  __syncthreads();
  for (int e = t; e < B; e += 1024) order[atomicAdd(&hist[255 - (int)((float)(cost[e] > 0 ? cost[e] : 0) * sc)], 1)] = e;
}
void smj_launch_order(const int* cost, int* order, int B, hipStream_t stream) {
  hipLaunchKernelGGL(smj_order_kernel, dim3(1), dim3(1024), 0, stream, cost, order, B);
}
void smj_launch_stage(const StagePlan& plan, float* stage, int stride, int B, long ld, bool is_export, hipStream_t stream) {
  if (plan.nseg == 0) return;
  hipLaunchKernelGGL(smj_stage_kernel, dim3((B + 63) / 64, plan.nseg), dim3(256), 0, stream, plan, stage, stride, B, ld, is_export ? 1 : 0);
}
void smj_launch_base_tick(const DevState& s, hipStream_t stream) {
  hipLaunchKernelGGL(smj_base_tick_kernel, dim3((s.B + 255) / 256), dim3(256), 0, stream, s);
}
void smj_launch_reset(const DevModel& m, const DevState& s, const uint8_t* mask, hipStream_t stream) {
  hipLaunchKernelGGL(smj_reset_kernel, dim3((s.B + 255) / 256), dim3(256), 0, stream, m, s, mask);
}
