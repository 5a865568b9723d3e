// Wave-level programming model used by the physics kernels.
//
// Kernels are written as a sequence of *lane regions* (code every lane of the 64-wide wavefront runs on its
// own lane-private data) separated by wave-uniform code and LDS synchronisation points:
//
//     LANES { x[lane] = ...; }        // per-lane work, no lane may read LDS another lane wrote in this region
//     SYNC();                         // LDS writes of the region become visible to all lanes
//     float s = wave_sum(x);          // cross-lane ops live OUTSIDE regions, in uniform control flow
//
// On gfx950 a region is plain straight-line code (`lane` = threadIdx.x, PL<T> is a register), SYNC() is the
// single-wave workgroup barrier (a compiler/LDS fence; one wave per workgroup, see __launch_bounds__(64)),
// and the cross-lane ops lower to DPP/readlane/ballot.  With SMJ_EMUL defined the same source compiles with
// g++: a region is a loop over 64 lanes, PL<T> is an array.  The emulator is TEST INFRASTRUCTURE (tests/emul):
// it lets the kernel logic be checked against the fp64 oracle on a box without a GPU.  It is never built into
// the product library.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef SMJ_EMUL
#define SMJ_DEV
#define LANES for (int lane = 0; lane < 64; ++lane)
#define SYNC() ((void)0)
template <class T>
struct PL {
  T v[64];
#ifdef SMJ_EMUL_POISON   // debug build of the emulator: lane-private values start as this byte pattern (0xFF = NaN / -1), so a
  PL() { memset(v, SMJ_EMUL_POISON, sizeof v); }   // read of a lane that was never written changes the results
#endif
  T& operator[](int l) { return v[l]; }
  const T& operator[](int l) const { return v[l]; }
};
static inline float wave_sum(const PL<float>& x) {
  // same association as the butterfly used on the GPU so that fp32 results match bit for bit
  float t[64];
  for (int i = 0; i < 64; i++) t[i] = x.v[i];
  for (int off = 32; off >= 1; off >>= 1) {
    float u[64];
    for (int i = 0; i < 64; i++) u[i] = t[i] + t[i ^ off];
    for (int i = 0; i < 64; i++) t[i] = u[i];
  }
  return t[0];
}
// sum over the caller's group of G = 2 or 4 consecutive lanes, left in every lane of the group (xor-1, then xor-2: the GPU's quad permutes)
template <int G>
static inline void wave_group_sum(PL<float>& x) {
  for (int off = 1; off < G; off <<= 1) {
    float u[64];
    for (int i = 0; i < 64; i++) u[i] = x.v[i] + x.v[i ^ off];
    for (int i = 0; i < 64; i++) x.v[i] = u[i];
  }
}
static inline float wave_min(const PL<float>& x) {
  float m = x.v[0];
  for (int i = 1; i < 64; i++) m = fminf(m, x.v[i]);
  return m;
}
static inline float wave_max(const PL<float>& x) {
  float m = x.v[0];
  for (int i = 1; i < 64; i++) m = fmaxf(m, x.v[i]);
  return m;
}
static inline uint64_t wave_ballot(const PL<int>& p) {
  uint64_t b = 0;
  for (int i = 0; i < 64; i++)
    if (p.v[i]) b |= 1ull << i;
  return b;
}
template <class T>
static inline T wave_read(const PL<T>& x, int l) { return x.v[l]; }
static inline int popc64(uint64_t x) { return __builtin_popcountll(x); }
// value of lane N of the caller's 16-lane row (DPP row_newbcast on the GPU)
template <int N>
static inline float wave_bcast16(const PL<float>& x, int lane) { return x.v[(lane & ~15) + N]; }
static inline int ffs64(uint64_t x) { return __builtin_ffsll((long long)x) - 1; }
static inline long long smj_clock() { return 0; }
static inline int opaque(int x) { return x; }
static inline float ld_coh(const float* p) { return *p; }
static inline int ld_coh(const int* p) { return *p; }
static inline void st_coh(float* p, float v) { *p = v; }
static inline void st_coh(int* p, int v) { *p = v; }
static inline int uni(int x) { return x; }
static inline float uni(float x) { return x; }
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline int lds_atomic_inc(int* p) { const int v = *p; *p = v + 1; return v; }   // (lanes run one after the other here)
static inline int lds_atomic_add(int* p, int n) { const int v = *p; *p = v + n; return v; }
#else
#include <hip/hip_runtime.h>
#define SMJ_DEV __device__ __forceinline__
// a region is a one-trip scope that names the lane id
#ifdef SMJ_TWO_WAVES
// Two wavefronts per env (smj_kernels_satp.hip: the PGS kernel of the satellite build -- wavefront 0 runs the step, wavefront 1
// sweeps the satellite islands beside the dense system's sweeps): `lane` is the lane inside the wavefront, SYNC() orders the LDS
// accesses of ONE wavefront (its DS operations execute in order; the fence keeps the compiler from moving them), and the two
// wavefronts meet at WG_BARRIER() only.
#define LANES for (int lane = (int)(threadIdx.x & 63u), _k = 0; _k < 1; ++_k)
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#define WG_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#else
#define LANES for (int lane = (int)threadIdx.x, _k = 0; _k < 1; ++_k)
#define SYNC() __syncthreads()
#endif
template <class T>
struct PL {
  T v;
  __device__ __forceinline__ T& operator[](int) { return v; }
  __device__ __forceinline__ const T& operator[](int) const { return v; }
};
// Wave-wide reductions on the DPP data path (no LDS round trips): xor-1 / xor-2 quad permutes, half-row and row mirrors
// leave every lane of a 16-lane row with the row total; row_bcast:15 / row_bcast:31 (gfx9 DPP controls) then carry the
// totals across rows so that lane 63 holds the wave total, which v_readlane returns as a scalar.  Must be called with all
// 64 lanes active (reductions sit between lane regions, never inside divergent code).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float smj_dpp(float ident, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL,
                                                                ROWMASK, 0xF, false));
}
#define SMJ_WAVE_REDUCE(OP, IDENT)                \
  t = OP(t, smj_dpp<0xB1, 0xF>(IDENT, t));        \
  t = OP(t, smj_dpp<0x4E, 0xF>(IDENT, t));        \
  t = OP(t, smj_dpp<0x141, 0xF>(IDENT, t));       \
  t = OP(t, smj_dpp<0x140, 0xF>(IDENT, t));       \
  t = OP(t, smj_dpp<0x142, 0xA>(IDENT, t));       \
  t = OP(t, smj_dpp<0x143, 0xC>(IDENT, t));       \
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
__device__ __forceinline__ float smj_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(const PL<float>& x) {
  float t = x.v;
  SMJ_WAVE_REDUCE(smj_addf, 0.0f)
}
// sum over the caller's group of G = 2 or 4 consecutive lanes, left in every lane of the group (quad permutes [1,0,3,2] and [2,3,0,1])
template <int G>
__device__ __forceinline__ void wave_group_sum(PL<float>& x) {
  x.v += smj_dpp<0xB1, 0xF>(0.0f, x.v);
  if (G > 2) x.v += smj_dpp<0x4E, 0xF>(0.0f, x.v);
}
__device__ __forceinline__ float wave_min(const PL<float>& x) {
  float t = x.v;
  SMJ_WAVE_REDUCE(fminf, __builtin_inff())
}
__device__ __forceinline__ float wave_max(const PL<float>& x) {
  float t = x.v;
  SMJ_WAVE_REDUCE(fmaxf, -__builtin_inff())
}
#undef SMJ_WAVE_REDUCE
// value of lane N of the caller's 16-lane row: one DPP move (row_newbcast), no SGPR round trip
template <int N>
__device__ __forceinline__ float wave_bcast16(const PL<float>& x, int) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x.v), 0x150 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ uint64_t wave_ballot(const PL<int>& p) { return __ballot(p.v != 0); }
__device__ __forceinline__ float wave_read(const PL<float>& x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x.v), l));
}
__device__ __forceinline__ int wave_read(const PL<int>& x, int l) { return __builtin_amdgcn_readlane(x.v, l); }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int ffs64(uint64_t x) { return __ffsll((long long)x) - 1; }
__device__ __forceinline__ long long smj_clock() { return (long long)__builtin_readcyclecounter(); }
// Device-coherent accesses (agent scope, relaxed: sc1 loads / write-through stores that do not live in an XCD's L2) for the
// words one workgroup hands to another inside a launch -- the staged state between the chunks of an env, scheduling words.
// With them the hand-over needs no L2 write-back / invalidate (an agent-scope release fence writes back EVERY dirty line of
// the XCD's L2, scratch spills of the other resident waves included: measured 193 MB of HBM writes per launch).
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_coh(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all of this wave's stores have reached the coherent level (then the flag that publishes them may be stored)
__device__ __forceinline__ void coh_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void coh_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// hides a value from the optimiser: table loads indexed through it are not loop invariant, so the stage-local constant
// tables are re-fetched from L2 in every step instead of being hoisted out of the step loop and kept alive (spilled)
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
// uni(): assert to the compiler that a value loaded from memory is wave-uniform (v_readfirstlane -> SGPR), so that
// loops / branches on it are scalar and readlane selectors need no waterfall loop
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uni(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
// 1-ulp hardware reciprocal / reciprocal square root (v_rcp_f32 / v_rsq_f32): the serial solver math is latency
// bound, an IEEE divide costs ~10 dependent instructions
__device__ __forceinline__ int lds_atomic_inc(int* p) { return atomicAdd(p, 1); }   // a counter in LDS bumped from divergent lanes (ds_add_rtn)
__device__ __forceinline__ int lds_atomic_add(int* p, int n) { return atomicAdd(p, n); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#endif

// ---------------------------------------------------------------------------------------------- small math
struct alignas(16) Vec4 { float x, y, z, w; };
struct F3 { float x, y, z; };
struct F4 { float w, x, y, z; };
struct F6 { float a[6]; };
struct M3 { float m[9]; };

SMJ_DEV float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
SMJ_DEV void cross3(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
SMJ_DEV float normalize3(float* a) {
  float n = sqrtf(dot3(a, a));
  if (n < 1e-15f) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  float i = 1.0f / n;
  a[0] *= i; a[1] *= i; a[2] *= i;
  return n;
}
SMJ_DEV void quat_mul(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
SMJ_DEV void quat_normalize(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15f) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  float i = 1.0f / n;
  q[0] *= i; q[1] *= i; q[2] *= i; q[3] *= i;
}
SMJ_DEV void quat2mat(float* R, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
SMJ_DEV void mulmat3vec(float* r, const float* R, const float* v) {
  float x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
        z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
SMJ_DEV void mulmat3Tvec(float* r, const float* R, const float* v) {
  float x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
        z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
SMJ_DEV void mulmat3(float* r, const float* A, const float* B) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = t[i];
}
// spatial algebra on [angular; linear] vectors (same conventions as oracle/smj_oracle.c)
SMJ_DEV void mul_inert_vec(float* r, const float* i, const float* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
SMJ_DEV void cross_motion(float* r, const float* vel, const float* v) {
  float a[3], b[3], c[3];
  cross3(a, vel, v); cross3(b, vel, v + 3); cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
SMJ_DEV void cross_force(float* r, const float* vel, const float* f) {
  float a[3], b[3], c[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
