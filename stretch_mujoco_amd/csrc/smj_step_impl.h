// One environment per wavefront: the whole mj_step pipeline (reference call site
// stretch_mujoco/mujoco_server.py:378; stages per SURVEY.md Appendix B) for `nsteps` steps, state resident in
// LDS/registers between steps.  Lane mappings: lane = body (tree stages), lane = dof (joint-space stages),
// lane = constraint row (solver).  Written in the lane-region style of smj_wave.h.
//
// Stage map (MuJoCo names):  kinematics -> comPos -> comVel -> crb/factorM (sparse L'DL in LDS) ->
// passive/rne/actuation -> collision (plane pairs) -> makeConstraint/makeImpedance ->
// projectConstraint (Y = J L^-1, A = Y D^-1 Y' + R on MFMA) -> PGS (dual, elliptic cones, QCQP blocks) ->
// implicitfast integrate.
#pragma once
#ifndef SMJ_PROFILING
#define SMJ_PROFILING 1   // per-stage shader-cycle counters (DevState::prof) compiled in; the product build of the standard variant drops them
#endif
#include <cstddef>
#include "smj_model.h"
#include "smj_wave.h"
#include "smj_sat_mem.h"

// NVS: columns the dof-indexed matrices (J, M, H) hold -- the 64-lane variant is built for 40, 52 or 64 dofs (smj_model.h)
#define JS (NVS + 1)
#define MS (NVS + 1)
#define SMJ_MINVAL 1e-15f
#define SMJ_MINIMP 0.0001f
#define SMJ_MAXIMP 0.9999f

enum { JT_FREE = 0, JT_BALL = 1, JT_SLIDE = 2, JT_HINGE = 3 };
enum { GT_PLANE = 0, GT_SPHERE = 2, GT_CAPSULE = 3, GT_ELLIPSOID = 4, GT_CYLINDER = 5, GT_BOX = 6, GT_MESH = 7 };
enum { CT_EQUALITY = 0, CT_FRICTION = 1, CT_LIMIT = 3, CT_CONTACT_FRICTIONLESS = 5, CT_CONTACT_ELLIPTIC = 7, CT_NONE = -1 };

#if NSAT > 0
#define NCGS (NCG + 1)
#define NSURV 128
#else
#define NCGS NCG
#endif
// Two wavefronts per env under Newton (smj_kernels_sat2.hip): the collision stage runs on both -- the first wavefront takes the
// pairs with the static world (collision_static), the second one the moving-moving pairs (collide_helper) -- see collision_convex().
#if defined(SMJ_TWO_WAVES) && defined(SMJ_ONLY_NEWTON) && NSAT > 0 && !defined(SMJ_EMUL)
#define SMJ_SPLIT_COLLIDE 1
#else
#define SMJ_SPLIT_COLLIDE 0
#endif
// (the second wavefront's other jobs, helper(): switches for measurements)
#ifndef SMJ_W2_FORWARD
#define SMJ_W2_FORWARD SMJ_SPLIT_COLLIDE
#endif
#ifndef SMJ_W2_NEWTON
#define SMJ_W2_NEWTON SMJ_SPLIT_COLLIDE
#endif
#ifndef SMJ_W2_INTEGRATE
#define SMJ_W2_INTEGRATE SMJ_SPLIT_COLLIDE
#endif
struct TreeTmp {  // lives in the A region until A is built
  float cinert[NBP][10], crb[NBP][10], cvel[NBP][6], cfrc[NBP][6], buf[NVP][6], cdof[NVP][6], cdof_dot[NVP][6];
};

#define NEFP 64   // rows per register set of the PGS path (set p: rows 64 p .. 64 p + 63)
#if NSAT > 0
#define NEFC_P SMJ_SAT_PGS   // satellite builds: rows of the DENSE system of the PGS path (the rows that touch the main tree + the rows of coupled satellites, smj_sat_pgs.h)
#else
#define NEFC_P (NEFC > 160 ? 160 : NEFC)   // rows of the PGS path: every row of the variant up to 160 (the packed A of the 224-row build would not fit the CU's LDS)
#endif
#define NPS ((NEFC_P + 63) / 64)   // register sets of the PGS path
struct Smem {
  float MM[NVS][MS];    // strict upper: M; lower + diag: working copy -> L (unit, strictly lower), D on diag
  float Mdiag[NVP], Dinv[NVP];
  float xpos[NBT][3], xquat[NBT][4], xmat[NBT][9], com[NBT][3];   // main bodies, then the satellites' (NBT = NBP without them)
  float xaxis[NVP][3], xanchor[NVP][3];
  float qpos[NVP + 8], qvel[NVP], ctrl[16], g[NVP], uu[NVP], w[NVP], qacc[NVP], warm[NVP], tmp[NVP];
  float act_force[16], act_len[16], act_vel[16], act_free[16];
  float bctl[SMJ_BC_ROWS];   // relative base move in flight (BaseController): mode, start pose, increment, v, omega
  int etype[NEFC], eid[NEFC];
  int estate[NEFC];   // Newton: state of every row as of the last constraint update (cone blocks: the block's state)
  float epos[NEFC], emargin[NEFC], ediag[NEFC], efloss[NEFC], eR[NEFC], eK[NEFC], eBv[NEFC], eimp[NEFC], earef[NEFC],
      eb[NEFC], ef[NEFC];
  float cpos[NCON][3], cframe[NCON][9], cdist[NCON], cfric[NCON][5], csolref[NCON][2], csolimp[NCON][5], cmargin[NCON];
#if NSAT > 0
  int cdim[NCON], cefc[NCON];
  unsigned short cgeom1[NCON], cgeom2[NCON];
  int cpair[NCON];      // the contact's pair (index in the model's pair list): MuJoCo's contact order, which the PGS sweeps follow
#else
  int cdim[NCON], cgeom1[NCON], cgeom2[NCON], cefc[NCON];
#endif
  // The stage-local union comes LAST on purpose: the PGS path keeps its matrix A = J M^-1 J' + R there (packed lower triangle,
  // NEFC_P (NEFC_P + 1) / 2 floats, see A()).  In the standard variant the 80-row triangle (13 KB) ends inside the struct, so a PGS
  // launch needs no more LDS than a Newton launch (four workgroups per CU); the 160-row variants ask for the tail as dynamic
  // LDS (smj_lds_bytes).
#if NSAT > 0
  float J[NDR + 1][JS];   // dense Jacobian rows of the rows that touch the main tree (rows 0 .. nd-1 of a step); row NDR stays zero: the dense row of every other row
#else
  float J[NEFC][JS];    // constraint Jacobian, transformed in place to Y = J L^-1
#endif
  union {
    TreeTmp t;
    struct {                // collision: world frames of the geoms taking part in convex pairs (tree temporaries are dead)
      float pos[NCGS][3], mat[NCGS][9], cen[NCGS][3], half[NCGS][3];   // (NCGS = NCG, + 1 staging slot for a static geom in the satellite builds)
      float ccen[NCGS][3], size[NCGS][3];   // world centre used as MPR's interior point, geom size
      int meta[NCGS];                      // geom type (bits 0-3) | hull vertex count (4-15) | hull address (16-31)
      unsigned short list[1024];   // bounding-sphere survivors of the convex pair list, table order
      float mc[5][3];              // multiccd: contact points found so far for the pair in hand
#if NSAT > 0
      int sl_n, sl_sid[NSURV];     // static-geometry broadphase (collision_static): survivors = index into k_sprec ...
      unsigned char sl_c[NSURV], sl_ord[NSURV];   // ... the moving geom's cache slot; the survivors in pair-table order
      float mc_r[NCG];             // bounding radii of the cached geoms
#endif
#if SMJ_SPLIT_COLLIDE
      // two wavefronts in the collision stage (smj_kernels_sat2.hip, helper()): the second one's multiccd scratch, the contact
      // slots the two have claimed together
      float mc2[5][3];
      int contotal;
#endif
    } c;
#if NSAT > 0 && defined(SMJ_ONLY_NEWTON)
    float pa[4];                            // (a Newton-only build keeps no A: the 18 KB of the triangle are what lets smj_kernels_sat.hip / _sat2.hip carry 112 dense rows in half a CU's LDS)
#elif NSAT > 0
    float pa[NEFC_P * (NEFC_P + 1) / 2];   // PGS: A of the dense system, packed lower triangle
#endif
    struct {                // plane narrowphase staging: contacts of the pair owned by each lane, emitted in pair order
      int cnt[64], pair[64];
      float dist[64][4], pos[64][4][3], nrm[64][3];
    } p;
    struct {                // constraint assembly: per-contact body dof masks gathered lane-parallel before the row loop
      int m1lo[NCON], m1hi[NCON], m2lo[NCON], m2hi[NCON], b1[NCON], b2[NCON];
      float cd[NVP][6];     // cdof, so that both half-waves of the Jacobian fill can read any dof's motion axis
    } k;
    struct {                // Newton: the Hessian H = M + J' W J (or M - h*D of the integrator)
#if NSAT > 0
      float H[NXV][NXV + 1];   // main block + the dense extension of coupled satellites (smj_sat.h); the cone Hessians stay valid beside it (the satellites' blocks read them after the main block is stored)
      float cH[NCH][36];       // pool: block SatMem::chs[c] holds contact c's cone Hessian
#else
      union {               // the cone Hessians are consumed (as MFMA operands) before the Hessian of the same iteration is written
        float H[NVS][NVS + 1];
        float cH[NCON][36];   // cone Hessians of contacts in the middle zone
      };
#endif
      // per-row solver registers (NRow) of rows 64..NEFC-1: the second row pass loads them at the start of a stage and
      // stores them back at its end, so that the (rare) second pass holds no registers across the Newton loop
      int rxi[3][NEFC > 64 ? NEFC - 64 : 1];      // indexed by row - 64
      float rxf[15][NEFC > 64 ? NEFC - 64 : 1];
    } n;
  } u;
#if NSAT > 0
  SatMem sat;
#endif
#if SMJ_SPLIT_COLLIDE
  // mailbox of the env's two wavefronts (helper()): [0] command; [1] the env (first command of a launch) / the cone mask's low word;
  // [2] the second wavefront's contact count / the cone mask's high word; [3] its flags.  (The last 16 bytes of the 80 KB.)
  int mbox[4];
#endif
  SMJ_DEV float* A(int cap) { return &J[cap][0]; }   // PGS: A = Y D^-1 Y' + R as a packed lower triangle of cap (cap + 1) / 2 floats behind the rows a step may use (cap = NEFC_P: the union's start)
};
// Row passes of the Newton path: rows 0..63 on lanes 0..63 (rb = 0), then rows 64..NEFC-1 on lanes 0..NEFC-65 (rb = 64), the
// second pass only for an env that has that many rows (wave-uniform test).
#define ROWPASS(rb, ne) _Pragma("unroll") for (int rb = 0; rb < NEFC; rb += 64) if (rb == 0 || __builtin_expect((ne) > rb, 0))
#define ROWPASS_ALL(rb) _Pragma("unroll") for (int rb = 0; rb < NEFC; rb += 64)
// A solver stage over all rows: `nr` names the per-row registers of the pass -- nr0 (registers, live across the Newton loop)
// in the first pass, a stage-local copy of the LDS-resident state (rx_load / rx_store) in the second.
#if NEFC > 128
// Builds with more than two passes (160 .. 320 rows): the first pass (registers) and ONE copy of the later passes as a run-time
// loop -- unrolled, every solver stage existed four or five times over (the 320-row build: 2.4 KB of scratch per lane, most of it
// addresses of the LDS-resident row state kept alive across the copies).
#define ROWS_BEGIN_(rb, ne, LOAD) _Pragma("unroll") for (int rp_ = 0; rp_ < 2; rp_++) _Pragma("nounroll") for (int rb = rp_ ? 64 : 0; rp_ ? (rb < NEFC && (ne) > rb) : rb < 64; rb += 64) { NRow nrx_; if (rp_ && (LOAD)) rx_load(nrx_, rb); NRow& nr = rp_ ? nrx_ : nr0; (void)nr;
#define ROWS_END_RW(rb) if (rp_) rx_store(nrx_, rb); }
#else
#define ROWS_BEGIN_(rb, ne, LOAD) ROWPASS(rb, ne) { NRow nrx_; if (rb != 0 && (LOAD)) rx_load(nrx_, rb); NRow& nr = (rb != 0) ? nrx_ : nr0; (void)nr;
#define ROWS_END_RW(rb) if (rb != 0) rx_store(nrx_, rb); }
#endif
#define ROWS_BEGIN(rb, ne) ROWS_BEGIN_(rb, ne, true)
#define ROWS_END_RO() }
// PGS without dynamic LDS: the largest row count (a multiple of 16: the MFMA tiles of A's build must not read rows of J that A
// has taken) whose packed A, placed behind that many rows of J, ends inside the struct -- 80 rows for the standard variant (all
// it has), 96 for the 128-row tall build and big38, 112 for big50.  A launch whose steps are capped there (DevModel::pgs_cap)
// runs at the variant's Newton occupancy (3 / 2 envs per CU instead of 1); steps beyond it go to the escalation variant.
static inline int smj_pgs_rows_static() {
  const size_t a_sq = offsetof(Smem, J) + sizeof(float) * (NEFP * JS + NEFP * NEFP);
  if (a_sq > sizeof(Smem)) return 0;
  for (int n = NEFC_P & ~15; n > NEFP; n -= 16)
    if (offsetof(Smem, J) + sizeof(float) * ((size_t)n * JS + (size_t)n * (n + 1) / 2) <= sizeof(Smem)) return n;
  return 0;
}
static inline size_t smj_lds_bytes(bool pgs) {
#if NSAT > 0
  return sizeof(Smem);   // (the satellite builds run Newton only)
#endif
  const size_t a_wide = offsetof(Smem, u) + sizeof(float) * (NEFC_P * (NEFC_P + 1) / 2);      // packed triangle in the union
  const size_t a_sq = offsetof(Smem, J) + sizeof(float) * (NEFP * JS + NEFP * NEFP);           // 64 x 64 square from row 64 of J
  const size_t a_end = a_wide > a_sq ? a_wide : a_sq;
  return (pgs && a_end > sizeof(Smem)) ? a_end : sizeof(Smem);
}

// ---------------------------------------------------------------------------------------------- MFMA tile
// D(16x16) += A(16x4) * B(4x16), f32.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; holds D[(l>>4)*4+r][l&15].
struct F4v { float r[4]; };
#ifdef SMJ_EMUL
static inline void mfma16x16x4(PL<F4v>& acc, const PL<float>& a, const PL<float>& b) {
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      int row = (l >> 4) * 4 + r, col = l & 15;
      float s = acc.v[l].r[r];
      for (int k = 0; k < 4; k++) s = fmaf(a.v[row + 16 * k], b.v[col + 16 * k], s);
      acc.v[l].r[r] = s;
    }
}
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16x16x4(PL<F4v>& acc, const PL<float>& a, const PL<float>& b) {
  f32x4 c = {acc.v.r[0], acc.v.r[1], acc.v.r[2], acc.v.r[3]};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v, b.v, c, 0, 0, 0);
  acc.v.r[0] = c[0]; acc.v.r[1] = c[1]; acc.v.r[2] = c[2]; acc.v.r[3] = c[3];
}
#endif

// ---------------------------------------------------------------------------------------------- QCQP (uniform)
// min 0.5 x'Ax + x'b  s.t. sum (x_i/d_i)^2 <= r^2   [MJ] mju_QCQP2 / mju_QCQP3 / mju_QCQP: the multiplier la >= 0 with
// |x(la)| = r, x(la) = -(A + la I)^-1 b (scaled variables), by a Newton iteration with a Cholesky factor of (A + la I) per
// iterate (rank test 1e-10).  Returns 1 if the constraint is active.  Reciprocal-based factor / solves: no IEEE divides on the
// serial path; x'(A + la I)^-1 x = |L^-1 x|^2 takes one forward substitution, not a second full solve.
//  exact = true (option qcqp_exact): MuJoCo's iteration -- from la = 0 on val(la) = |x|^2 - r^2, stops at val < 1e-10,
//    delta < 1e-10 or after 20 iterates.  From la = 0 that iteration grows (A + la I) by at most 1.5x per step when the
//    unconstrained minimum lies far outside the cone (rolling / torsional friction of a driven wheel): 9.6 iterations per call on
//    the bench workload, 3 % of the active calls end at the cap -- 2/3 of the PGS path's time.
//  exact = false (default): the SAME root, found faster.  (a) The secular form 1 / |x(la)| - 1 / r (Hebden / More-Sorensen) is
//    concave and nearly linear in la: its Newton step is MuJoCo's times 2 |x|^2 / (r (|x| + r)) -- equal at the root, larger far
//    from it -- and lands left of the root from either side (20 000 random spectra: no overshoot), so a clamp at la = 0 is the
//    only safeguard.  (b) The iteration starts at the multiplier the contact's block had in the PREVIOUS SWEEP (la_io; 0 in the
//    first sweep of a step): 1.8 iterations per call.  (c) fp32 tolerances: val carries the rounding of the solve
//    (1e-7 r^2 times the condition number), a step of la matters relative to la + min A_ii; both tests get a relative part of
//    1e-5 (the caller rescales an active solution onto the cone anyway).  (d) A step below 3e-3 (la + min A_ii) is taken to
//    first order -- x(la + d) = x - d (A + la I)^-1 x, one more back substitution -- instead of a new factorisation (error
//    d^2 / (la + A)^2 <= 1e-5): from the second sweep on an active block costs ONE factorisation.  Where MuJoCo's iteration
//    converges the two agree to that tolerance; where it ends at its cap MuJoCo returns the direction of an unconverged x, this
//    returns the root's.
template <int N>
SMJ_DEV int qcqp(float* res, const float* Ain, const float* bin, const float* dd, float r, float& la_io, bool exact, float* iters = nullptr) {
  float A[N * N], b[N], L[N * N], Li[N], la = exact ? 0.f : la_io;
#pragma unroll
  for (int i = 0; i < N; i++) {
    b[i] = bin[i] * dd[i];
#pragma unroll
    for (int j = 0; j <= i; j++) A[i * N + j] = Ain[i * N + j] * dd[i] * dd[j];
  }
  float dmin = A[0];
#pragma unroll
  for (int i = 1; i < N; i++) dmin = fminf(dmin, A[i * N + i]);
  const float r2 = r * r, vtol = exact ? 1e-10f : 1e-10f + 1e-5f * r2, dtol = exact ? 1e-10f : 1e-10f + 1e-5f * dmin, drel = exact ? 0.f : 1e-5f;   // a step below 1e-5 (la + min A_ii) moves x by less than that, relatively
  for (int it = 0; it < 20; it++) {
    if (iters) *iters += 1.f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < N; j++) {
      float sd = A[j * N + j] + la;
#pragma unroll
      for (int k = 0; k < j; k++) sd -= L[j * N + k] * L[j * N + k];
      if (sd < 1e-10f) { bad = true; sd = 1e-10f; }
      const float li = fast_rsqrt(sd);
      Li[j] = li;
#pragma unroll
      for (int i = j + 1; i < N; i++) {
        float t = A[i * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) t -= L[i * N + k] * L[j * N + k];
        L[i * N + j] = t * li;
      }
    }
    if (bad) {
#pragma unroll
      for (int i = 0; i < N; i++) res[i] = 0;
      return 0;
    }
    // res = -(A + la)^-1 b
    float x[N], nn = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      float sv = -b[i];
#pragma unroll
      for (int k = 0; k < i; k++) sv -= L[i * N + k] * x[k];
      x[i] = sv * Li[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
      float sv = x[i];
#pragma unroll
      for (int k = i + 1; k < N; k++) sv -= L[k * N + i] * x[k];
      x[i] = sv * Li[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) { res[i] = x[i]; nn += x[i] * x[i]; }
    const float val = nn - r2;
    if (val < vtol && (exact || la == 0.f || val > -vtol)) break;  // converged, or (la = 0) the unconstrained minimum is inside the cone
    // q = res'(A + la)^-1 res = |L^-1 res|^2
    float q = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      float sv = res[i];
#pragma unroll
      for (int k = 0; k < i; k++) sv -= L[i * N + k] * x[k];
      x[i] = sv * Li[i];
      q += x[i] * x[i];
    }
    float delta;
    if (exact) {   // [MJ] delta = -val / deriv, deriv = -2 x'(A + la)^-1 x
      delta = val * fast_rcp(2.f * q);
      if (delta < dtol) break;
    } else {
      const float nrm = nn * fast_rsqrt(nn);
      delta = (nrm - r) * nn * fast_rcp(r * q);
      if (fabsf(delta) < dtol + drel * la) break;
      if (fabsf(delta) < 3e-3f * (la + dmin)) {
        // close (the usual case from the second sweep on): finish to first order instead of factoring again --
        // x(la + d) = x(la) - d (A + la I)^-1 x(la) + O(d^2 / (la + A)^2) <= 1e-5 relative; (A + la I)^-1 x = L^-T (L^-1 x), L^-1 x is in x[]
        const float lan = fmaxf(0.f, la + delta), d = lan - la;
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
          float sv = x[i];
#pragma unroll
          for (int k = i + 1; k < N; k++) sv -= L[k * N + i] * x[k];
          x[i] = sv * Li[i];
        }
#pragma unroll
        for (int i = 0; i < N; i++) res[i] -= d * x[i];
        la = lan;
        break;
      }
    }
    la = fmaxf(0.f, la + delta);
  }
#pragma unroll
  for (int i = 0; i < N; i++) res[i] *= dd[i];
  la_io = la;
  return la != 0;
}

SMJ_DEV float impedance(const float* solimp, float pos, float margin) {
  float dmin = fminf(SMJ_MAXIMP, fmaxf(SMJ_MINIMP, solimp[0])), dmax = fminf(SMJ_MAXIMP, fmaxf(SMJ_MINIMP, solimp[1]));
  float width = fmaxf(SMJ_MINVAL, solimp[2]), mid = fminf(SMJ_MAXIMP, fmaxf(SMJ_MINIMP, solimp[3])), power = fmaxf(1.0f, solimp[4]);
  if (dmin == dmax) return 0.5f * (dmin + dmax);
  float x = fabsf(pos - margin) / width, y;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  if (power == 1.0f) y = x;
  else if (power == 2.0f) y = (x <= mid) ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
  else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

#ifdef SMJ_EMUL
static long smj_emul_sep_skips = 0;
static long smj_emul_ext_steps = 0;
static long smj_emul_mc_hits = 0;
static long smj_emul_isl_total = 0, smj_emul_isl_swept = 0;   // PGS islands: satellite-lane sweeps with / without the island stop
#endif
// ---------------------------------------------------------------------------------------------- the step
struct StepKernel {
  const DevModel& M;
  const DevState& S;
  Smem& s;
  int env;   // (constant but in the second wavefront of a two-wavefront worker kernel, which follows the first one from env to env: helper())

  // Persistent lane-resident state is kept small on purpose (the kernel owns 512 registers per lane but also unrolls a
  // 32x32 elimination): model constants are (re)loaded at the top of the stage that needs them -- independent global loads
  // issued back to back cost one L2 round trip per stage -- instead of living in registers for the whole launch.
  PL<float[6]> cdof, cdof_dot;          // lane = dof
  PL<float> qvel_r, g_r, qacc_r;
  PL<float> f_r[NPS], r_r[NPS], ARinv_r[NPS];   // lane = row (PGS), one register set per 64 rows
  PL<float> qla_r;                              // lane = contact (PGS): the QCQP multiplier of the contact's last sweep, the next sweep's starting point
  int nefc, ncon, niter, flags;
  float* ppc = nullptr;  // profiling builds: the launch's counter array while the PGS sweeps run (pgs_block's QCQP time / iterations)
  int step_base = 0;     // steps of this launch that earlier chunks of the env already ran (pipelined chunks, DevState::pipe_len)
  int pipe_chunk = 0;    // the chunk this workgroup runs
  bool parked = false;   // run() handed the env to the escalation list
  // Collision stage on two wavefronts (SMJ_SPLIT_COLLIDE): `split_on` while the two run side by side (contact slots are claimed from
  // the shared counter Smem::u.c.contotal), `rev` in the second wavefront, whose contact k sits in slot NCON - 1 - k until the first
  // one appends them to its own
  bool split_on = false, rev = false;
  bool serial_redo = false;   // run(): the collision stage's second go at a step whose contact list overflowed

  SMJ_DEV StepKernel(const DevModel& M_, const DevState& S_, Smem& s_, int env_) : M(M_), S(S_), s(s_), env(env_) {}

  SMJ_DEV static uint64_t mk64(int lo, int hi) { return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32); }

  // ------------------------------------------------------------------ stage-local constant tables
  struct KinTab {   // lane = body: tree link + the body's (<= 2) joints
    PL<int> parent, level;
    PL<int[6]> jump;        // k_body_jump rows (pointer-jumping ancestors), fetched with the rest of the table
    PL<float[3]> pos;
    PL<float[4]> quat;
    PL<int[2]> jtype, jqadr, jdadr;
    PL<float[2]> jq0;
    PL<float[6]> jaxis, jpos;
  };
  // one lane's record of the per-lane stage tables (DevModel::k_lanerec): 16-byte aligned, contiguous
  SMJ_DEV const int* lanerec(int ol, int off) const {
    return static_cast<const int*>(__builtin_assume_aligned(M.k_lanerec + ol * SMJ_LR_STRIDE + off, 16));
  }
  SMJ_DEV static float asf(int b) { return __builtin_bit_cast(float, b); }
  SMJ_DEV void load(KinTab& t) {
    LANES {
      const int* r = lanerec(opaque(lane), SMJ_LR_KIN);
      int v[35];
      for (int k = 0; k < 35; k++) v[k] = r[k];
      t.parent[lane] = v[0]; t.level[lane] = v[1];
      for (int q = 0; q < 6; q++) t.jump[lane][q] = v[2 + q];
      for (int k = 0; k < 3; k++) t.pos[lane][k] = asf(v[8 + k]);
      for (int k = 0; k < 4; k++) t.quat[lane][k] = asf(v[11 + k]);
      for (int u = 0; u < 2; u++) {
        t.jtype[lane][u] = v[15 + u]; t.jqadr[lane][u] = v[17 + u]; t.jdadr[lane][u] = v[19 + u]; t.jq0[lane][u] = asf(v[21 + u]);
      }
      for (int k = 0; k < 6; k++) { t.jaxis[lane][k] = asf(v[23 + k]); t.jpos[lane][k] = asf(v[29 + k]); }
    }
  }
  struct BodyTab {  // lane = body: inertia and tree bookkeeping
    PL<int> root, subsize, parent, dofadr, dofnum;
    PL<int[6]> jump;
    PL<uint64_t> dofmask;
    PL<float[10]> inl;
  };
  SMJ_DEV void load(BodyTab& t) {
    LANES {
      const int* r = lanerec(opaque(lane), SMJ_LR_BODY);
      int v[23];
      for (int k = 0; k < 23; k++) v[k] = r[k];
      t.root[lane] = v[0]; t.subsize[lane] = v[1]; t.parent[lane] = v[2]; t.dofadr[lane] = v[3]; t.dofnum[lane] = v[4];
      for (int q = 0; q < 6; q++) t.jump[lane][q] = v[5 + q];
      t.dofmask[lane] = mk64(v[11], v[12]);
      for (int k = 0; k < 10; k++) t.inl[lane][k] = asf(v[13 + k]);
    }
  }
  struct DofTab {   // lane = dof
    PL<int> body, jtype, qadr, first, bsub;
    PL<uint64_t> velmask;
    PL<float> damp, stiff, spring;
    PL<int[2]> act;
    PL<float[2]> actmom;
  };
  SMJ_DEV void load(DofTab& t) {
    LANES {
      const int* r = lanerec(opaque(lane), SMJ_LR_DOF);
      int v[14];
      for (int k = 0; k < 14; k++) v[k] = r[k];
      t.body[lane] = v[0]; t.jtype[lane] = v[1]; t.qadr[lane] = v[2]; t.first[lane] = v[3]; t.bsub[lane] = v[4];
      t.velmask[lane] = mk64(v[5], v[6]);
      t.damp[lane] = asf(v[7]); t.stiff[lane] = asf(v[8]); t.spring[lane] = asf(v[9]);
      for (int u = 0; u < 2; u++) { t.act[lane][u] = v[10 + u]; t.actmom[lane][u] = asf(v[12 + u]); }
    }
  }
  struct EntryTab { // lane = mass-matrix pattern slots (NENT per lane)
    PL<int[NENT]> i, j, lact;
    PL<float[NENT]> arm, damp, dcoef, lcoef;
  };
  SMJ_DEV void load(EntryTab& t, bool implicit) {
    LANES {
      const int* r = lanerec(opaque(lane), SMJ_LR_ENT);
      for (int u = 0; u < NENT; u++) { t.i[lane][u] = r[u]; t.j[lane][u] = r[NENT + u]; t.arm[lane][u] = asf(r[2 * NENT + u]); }
      if (implicit)
        for (int u = 0; u < NENT; u++) {
          t.lact[lane][u] = r[3 * NENT + u]; t.damp[lane][u] = asf(r[4 * NENT + u]); t.dcoef[lane][u] = asf(r[5 * NENT + u]);
          t.lcoef[lane][u] = asf(r[6 * NENT + u]);
        }
    }
  }
  struct ActTab {   // lane = actuator, and lane = gravity-compensated body slot
    PL<int[4]> dof, qadr;
    PL<float[4]> mom;
    PL<float[8]> prm;   // gain, b0, b1, b2, ctrl lo/hi, force lo/hi
    PL<int> flags;      // bit0 ctrllimited, bit1 forcelimited, bit2 affine bias
    PL<int> gc_body, gc_mlo, gc_mhi;
    PL<float> gc_mass, gc_x, gc_y, gc_z;
  };
  SMJ_DEV void load(ActTab& t) {
    LANES {
      const int* r = lanerec(opaque(lane), SMJ_LR_ACT);
      int v[28];
      for (int k = 0; k < 28; k++) v[k] = r[k];
      for (int u = 0; u < 4; u++) { t.dof[lane][u] = v[u]; t.qadr[lane][u] = v[4 + u]; t.mom[lane][u] = asf(v[8 + u]); }
      for (int k = 0; k < 8; k++) t.prm[lane][k] = asf(v[12 + k]);
      t.flags[lane] = v[20]; t.gc_body[lane] = v[21]; t.gc_mlo[lane] = v[22]; t.gc_mhi[lane] = v[23];
      t.gc_mass[lane] = asf(v[24]); t.gc_x[lane] = asf(v[25]); t.gc_y[lane] = asf(v[26]); t.gc_z[lane] = asf(v[27]);
    }
  }

  // ------------------------------------------------------------------ setup (once per launch)
  SMJ_DEV void setup() {
    LANES {
      // zero the factor storage once: the sparsity pattern is static, non-pattern entries stay zero
      for (int k = lane; k < NVS * MS; k += 64) (&s.MM[0][0])[k] = 0.f;
      // world body
      if (lane == 0) {
        s.xpos[0][0] = s.xpos[0][1] = s.xpos[0][2] = 0;
        s.xquat[0][0] = 1; s.xquat[0][1] = s.xquat[0][2] = s.xquat[0][3] = 0;
        for (int k = 0; k < 9; k++) s.xmat[0][k] = (k % 4 == 0) ? 1.f : 0.f;
      }
    }
    SYNC();
#if NSAT > 0
    LANES {
      for (int k = lane; k < (NDR + 1) * JS; k += 64) (&s.J[0][0])[k] = 0.f;   // (row NDR: the dense row of every row without one)
      if (lane == 0) { s.sat.cand_ok = 0; s.sat.ncand = 0; }                    // the static-broadphase candidate list is rebuilt by the first step
      for (int k = lane; k < NCG * 3; k += 64) (&s.sat.refcen[0][0])[k] = 0.f;
    }
#endif
    // identity padding of the dof block nv..NVP-1, written once: the per-step mass-matrix entries never touch it, and the
    // Newton Hessian H = M + J'WJ / the dense M x products can then take MM as it is, with no `< nv` select per element
    LANES { if (lane >= M.nv && lane < NVS) s.MM[lane][lane] = 1.f; }
    SYNC();
  }

  // One env's row of the env-major staging copy (DevState::stage): contiguous words, lanes = consecutive words
  SMJ_DEV float* stage_row() const { return S.stage + (size_t)env * S.lay.stride; }
  // Capacity escalation: park the env at the START of step `st` (nothing of this step has been written to the state yet:
  // qpos / qvel / qacc_warmstart / ctrl / the base controller change in solve / integrate / base_controller only) and queue
  // it for the big kernel variant, which finishes the launch's remaining steps on it.
  SMJ_DEV void escalate(int st) {
#ifndef SMJ_EMUL
    float* row = stage_row();
    int* rowi = reinterpret_cast<int*>(row);
    LANES {
      for (int k = lane; k < M.nq; k += 64) st_coh(&row[S.lay.qpos + k], s.qpos[k]);
      if (lane < M.nv) { st_coh(&row[S.lay.qvel + lane], s.qvel[lane]); st_coh(&row[S.lay.warm + lane], s.warm[lane]); }
      if (lane < M.nu) st_coh(&row[S.lay.ctrl + lane], s.ctrl[lane]);
      if (lane < SMJ_BC_ROWS) st_coh(&row[S.lay.bctl + lane], s.bctl[lane]);
      if (lane == 0) {
        st_coh(&rowi[S.lay.nstep], ld_coh(&rowi[S.lay.nstep]) + st);
        st_coh(&S.done_steps[env], ld_coh(&S.done_steps[env]) + st);
      }
    }
#if NSAT > 0
    sat_store_state();
#endif
    coh_release();
    LANES {
      if (lane == 0) {
        // parked state first, then the list entry (a poller that takes the entry swaps exactly this value back)
        st_coh(&S.progress[env], -(pipe_chunk + 1));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int at = atomicAdd(&S.sched[SMJ_SCHED_COUNT], 1);
        st_coh(&S.redo[at], env);
      }
    }
#endif
    parked = true;
  }
  SMJ_DEV void load_state() {
    const long ld = S.ld;
    if (S.stage) {
      const float* st = stage_row();
      LANES {
        for (int k = lane; k < M.nq; k += 64) s.qpos[k] = ld_coh(&st[S.lay.qpos + k]);
        if (lane < M.nv) { s.qvel[lane] = ld_coh(&st[S.lay.qvel + lane]); s.warm[lane] = ld_coh(&st[S.lay.warm + lane]); }
        if (lane < M.nu) s.ctrl[lane] = ld_coh(&st[S.lay.ctrl + lane]);
        if (lane < SMJ_BC_ROWS) s.bctl[lane] = ld_coh(&st[S.lay.bctl + lane]);
      }
    } else {
      LANES {
        for (int k = lane; k < M.nq; k += 64) s.qpos[k] = S.qpos[k * ld + env];
        if (lane < M.nv) { s.qvel[lane] = S.qvel[lane * ld + env]; s.warm[lane] = S.warm[lane * ld + env]; }
        if (lane < M.nu) s.ctrl[lane] = S.ctrl[lane * ld + env];
        if (lane < SMJ_BC_ROWS) s.bctl[lane] = S.bctl ? S.bctl[lane * ld + env] : 0.f;
      }
    }
    SYNC();
#if NSAT > 0
    sat_load_state();
#endif
  }
  // State, counters and the post-step readout.  actuator_length / velocity and the base pose are those of the LAST FORWARD
  // PASS -- what MjData holds after mj_step, which integrates after computing them (pull_status reads exactly these fields,
  // mujoco_server.py:465-515, :124-129): they lag qpos by one step, as in the reference.
  SMJ_DEV void store_state(int nsteps) {
    const long ld = S.ld;
#if NSAT > 0
    sat_store_state();
#endif
    const int b = 1;  // base_link is the first body after the world
    const float bx = s.xpos[b][0], by = s.xpos[b][1], bth = atan2f(s.xmat[b][3], s.xmat[b][0]);
    if (S.stage) {
      float* st = stage_row();
      int* sti = reinterpret_cast<int*>(st);
      LANES {
        // the state words go out device-coherent (st_coh): the env's next chunk may run on another XCD (DevState::pipe_len)
        for (int k = lane; k < M.nq; k += 64) st_coh(&st[S.lay.qpos + k], s.qpos[k]);
        if (lane < M.nv) { st_coh(&st[S.lay.qvel + lane], s.qvel[lane]); st_coh(&st[S.lay.warm + lane], s.warm[lane]); }
        // (the readout words too: with pipelined chunks every chunk of an env stores them, possibly from different XCDs -- plain
        // stores would leave dirty copies of the same words in two L2s, written back in no defined order at kernel end)
        if (lane < M.nu) { st_coh(&st[S.lay.ctrl + lane], s.ctrl[lane]); st_coh(&st[S.lay.actlen + lane], s.act_len[lane]); st_coh(&st[S.lay.actvel + lane], s.act_vel[lane]); }
        if (lane < SMJ_BC_ROWS) st_coh(&st[S.lay.bctl + lane], s.bctl[lane]);
        if (lane == 0) {
          st_coh(&sti[S.lay.nstep], ld_coh(&sti[S.lay.nstep]) + nsteps);
          if (S.done_steps) st_coh(&S.done_steps[env], ld_coh(&S.done_steps[env]) + nsteps);
          st_coh(&sti[S.lay.info + SMJ_INFO_NEFC], nefc); st_coh(&sti[S.lay.info + SMJ_INFO_NCON], ncon);
          st_coh(&sti[S.lay.info + SMJ_INFO_NITER], niter); st_coh(&sti[S.lay.info + SMJ_INFO_FLAGS], ld_coh(&sti[S.lay.info + SMJ_INFO_FLAGS]) | flags);
          st_coh(&st[S.lay.base], bx); st_coh(&st[S.lay.base + 1], by); st_coh(&st[S.lay.base + 2], bth);
        }
      }
    } else {
      LANES {
        for (int k = lane; k < M.nq; k += 64) S.qpos[k * ld + env] = s.qpos[k];
        if (lane < M.nv) { S.qvel[lane * ld + env] = s.qvel[lane]; S.warm[lane * ld + env] = s.warm[lane]; }
        if (lane < M.nu) { S.ctrl[lane * ld + env] = s.ctrl[lane]; S.act_len[lane * ld + env] = s.act_len[lane]; S.act_vel[lane * ld + env] = s.act_vel[lane]; }
        if (lane < SMJ_BC_ROWS && S.bctl) S.bctl[lane * ld + env] = s.bctl[lane];
        if (lane == 0) {
          S.nstep[env] += nsteps;
          S.info[SMJ_INFO_NEFC * ld + env] = nefc; S.info[SMJ_INFO_NCON * ld + env] = ncon;
          S.info[SMJ_INFO_NITER * ld + env] = niter; S.info[SMJ_INFO_FLAGS * ld + env] |= flags;
          S.base[0 * ld + env] = bx; S.base[1 * ld + env] = by; S.base[2 * ld + env] = bth;
        }
      }
    }
  }

  // BaseController.update (mujoco_server.py:110-176) after every step, with the base pose of the step's forward pass (what
  // get_base_pose reads from MjData after mj_step): a translate-by / rotate-by in flight drives the wheels at the default
  // speed until the travelled distance / heading change exceeds the increment, then stops them and clears itself; velocity
  // mode keeps writing its wheel speeds.  No angle wrap in rotate-by (:160-163), as in the reference.  Wave-uniform.
  SMJ_DEV void base_controller() {
    const int mode = (int)uni(s.bctl[SMJ_BC_MODE]);
    if (mode == 0) return;
    const int b = 1;
    const float x = uni(s.xpos[b][0]), y = uni(s.xpos[b][1]), th = atan2f(uni(s.xmat[b][3]), uni(s.xmat[b][0]));
    const float inc = uni(s.bctl[SMJ_BC_INC]), sign = inc > 0.f ? 1.f : -1.f;
    float v = 0.f, w = 0.f;
    int next = mode;
    if (mode == 1) {
      const float dx = x - uni(s.bctl[SMJ_BC_X0]), dy = y - uni(s.bctl[SMJ_BC_Y0]);
      if (!(sqrtf(dx * dx + dy * dy) <= fabsf(inc))) next = 0; else v = SMJ_BASE_X_VEL * sign;
    } else if (mode == 2) {
      if (!(fabsf(uni(s.bctl[SMJ_BC_TH0]) - th) <= fabsf(inc))) next = 0; else w = SMJ_BASE_R_VEL * sign;
    } else { v = uni(s.bctl[SMJ_BC_V]); w = uni(s.bctl[SMJ_BC_W]); }
    const float wl = (v - (w * SMJ_WHEEL_SEPARATION / 2.f)) / SMJ_WHEEL_RADIUS, wr = (v + (w * SMJ_WHEEL_SEPARATION / 2.f)) / SMJ_WHEEL_RADIUS;
    LANES {
      if (lane == 0) { s.ctrl[0] = wl; s.ctrl[1] = wr; s.bctl[SMJ_BC_MODE] = (float)next; }
    }
    SYNC();
  }

  // ------------------------------------------------------------------ B.1 kinematics  [MJ] mj_kinematics
  // Body poses in three phases instead of one dependent pass per tree level (13 levels, each a chain of a few hundred
  // dependent instructions at one wave per SIMD):  A) every body's transform relative to its PARENT frame, joint motion
  // included -- all bodies at once, this is where the sincos / quaternion work is;  B) pointer jumping: round r composes a
  // body with its 2^r-th ancestor (k_body_jump), ceil(log2(depth)) = 4 rounds;  C) world matrices, joint anchors and axes
  // from the parent's world frame, all bodies at once.
  SMJ_DEV void kinematics() {
    const int nb = M.nbody;
    KinTab kt;
    load(kt);
    PL<float[3]> pl;          // position in the parent frame, later in the frame of the current jump ancestor
    PL<float[4]> ql;
    PL<float[6]> al, xl;      // joint anchors / axes in the parent frame
    LANES {
      float pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
      for (int k = 0; k < 6; k++) { al[lane][k] = 0.f; xl[lane][k] = 0.f; }
      if (lane > 0 && lane < nb) {
        if (kt.jtype[lane][0] == JT_FREE) {
          const int qa = kt.jqadr[lane][0];
          for (int k = 0; k < 3; k++) pos[k] = s.qpos[qa + k];
          for (int k = 0; k < 4; k++) quat[k] = s.qpos[qa + 3 + k];
        } else {
          for (int k = 0; k < 3; k++) pos[k] = kt.pos[lane][k];
          for (int k = 0; k < 4; k++) quat[k] = kt.quat[lane][k];
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int jt = kt.jtype[lane][t];
            if (jt < 0) continue;
            const float* ax = &kt.jaxis[lane][3 * t];
            const float* jp = &kt.jpos[lane][3 * t];
            float Rm[9], xa[3], a[3], anchor[3];
            quat2mat(Rm, quat);
            mulmat3vec(xa, Rm, ax);
            mulmat3vec(a, Rm, jp);
            for (int k = 0; k < 3; k++) { anchor[k] = pos[k] + a[k]; xl[lane][3 * t + k] = xa[k]; al[lane][3 * t + k] = anchor[k]; }
            const float q = s.qpos[kt.jqadr[lane][t]] - kt.jq0[lane][t];
            if (jt == JT_SLIDE) {
              for (int k = 0; k < 3; k++) pos[k] += xa[k] * q;
            } else {
              float sn, cs;
              sincosf(0.5f * q, &sn, &cs);
              const float dq[4] = {cs, ax[0] * sn, ax[1] * sn, ax[2] * sn};
              quat_mul(quat, quat, dq);
              quat2mat(Rm, quat);
              mulmat3vec(a, Rm, jp);
              for (int k = 0; k < 3; k++) pos[k] = anchor[k] - a[k];
            }
          }
        }
        quat_normalize(quat);
      }
      // (slots nb .. NBP-1 hold the satellites' bodies, written by sat_forward -- which runs beside this stage on the env's second
      // wavefront in the two-wavefront build: no padding writes there)
      const bool mine = lane < NBP && (!SMJ_W2_FORWARD || lane < nb);
      for (int k = 0; k < 3; k++) { pl[lane][k] = pos[k]; if (mine) s.xpos[lane][k] = pos[k]; }
      for (int k = 0; k < 4; k++) { ql[lane][k] = quat[k]; if (mine) s.xquat[lane][k] = quat[k]; }
    }
    SYNC();
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r >= M.njump) break;
      PL<float[3]> pa;
      PL<float[4]> qa;
      PL<int> anc;
      LANES {
        const int a = kt.jump[lane][r];
        anc[lane] = a;
        for (int k = 0; k < 3; k++) pa[lane][k] = s.xpos[a][k];
        for (int k = 0; k < 4; k++) qa[lane][k] = s.xquat[a][k];
      }
      SYNC();
      LANES {
        if (anc[lane] > 0) {   // ancestor 0 is the world: identity
          float Rm[9], v[3], q[4];
          quat2mat(Rm, qa[lane]);
          mulmat3vec(v, Rm, pl[lane]);
          quat_mul(q, qa[lane], ql[lane]);
          for (int k = 0; k < 3; k++) { pl[lane][k] = pa[lane][k] + v[k]; s.xpos[lane][k] = pl[lane][k]; }
          for (int k = 0; k < 4; k++) { ql[lane][k] = q[k]; s.xquat[lane][k] = q[k]; }
        }
      }
      SYNC();
    }
    LANES {
      if (lane < nb) {
        float q[4] = {ql[lane][0], ql[lane][1], ql[lane][2], ql[lane][3]}, Rm[9];
        quat_normalize(q);
        quat2mat(Rm, q);
        for (int k = 0; k < 4; k++) s.xquat[lane][k] = q[k];
        for (int k = 0; k < 9; k++) s.xmat[lane][k] = Rm[k];
      }
    }
    SYNC();
    LANES {
      if (lane > 0 && lane < nb) {
        if (kt.jtype[lane][0] == JT_FREE) {
          const int da = kt.jdadr[lane][0];
          for (int d = 0; d < 6; d++)
            for (int k = 0; k < 3; k++) s.xanchor[da + d][k] = s.xpos[lane][k];
          for (int d = 0; d < 3; d++)
            for (int k = 0; k < 3; k++) {
              s.xaxis[da + d][k] = (d == k) ? 1.f : 0.f;
              s.xaxis[da + 3 + d][k] = s.xmat[lane][3 * k + d];
            }
        } else {
          const int p = kt.parent[lane];
#pragma unroll
          for (int t = 0; t < 2; t++) {
            if (kt.jtype[lane][t] < 0) continue;
            const int da = kt.jdadr[lane][t];
            float xa[3], an[3];
            mulmat3vec(xa, s.xmat[p], &xl[lane][3 * t]);
            mulmat3vec(an, s.xmat[p], &al[lane][3 * t]);
            for (int k = 0; k < 3; k++) { s.xaxis[da][k] = xa[k]; s.xanchor[da][k] = s.xpos[p][k] + an[k]; }
          }
        }
      }
    }
    SYNC();
  }

  // ------------------------------------------------------------------ comPos + comVel + crb + M
  SMJ_DEV void com_crb() {
    const int nb = M.nbody, nv = M.nv;
    BodyTab bt;
    DofTab dt;
    EntryTab et;
    load(bt); load(dt); load(et, false);
    // subtree com of each tree root: masked wave reductions
    PL<float> mx, my, mz;
    PL<float[3]> xip;
    LANES {
      const int b = lane;
      float t[3] = {0, 0, 0};
      if (b > 0 && b < nb) {
        mulmat3vec(t, s.xmat[b], &bt.inl[lane][6]);
        for (int k = 0; k < 3; k++) t[k] += s.xpos[b][k];
      }
      for (int k = 0; k < 3; k++) xip[lane][k] = t[k];
    }
    for (int r = 0; r < M.nroot; r++) {
      const int root = uni(M.k_root_list[r]);
      LANES {
        const bool in = lane > 0 && lane < nb && bt.root[lane] == root;
        const float m = in ? bt.inl[lane][9] : 0.f;
        mx[lane] = m * xip[lane][0]; my[lane] = m * xip[lane][1]; mz[lane] = m * xip[lane][2];
      }
      const float inv = 1.0f / M.body_subtreemass[root];
      const float cx = wave_sum(mx) * inv, cy = wave_sum(my) * inv, cz = wave_sum(mz) * inv;
      LANES {
        if (lane < nb && bt.root[lane] == root) { s.com[lane][0] = cx; s.com[lane][1] = cy; s.com[lane][2] = cz; }
      }
    }
    SYNC();
    // cinert (body lanes) and cdof (dof lanes)
    LANES {
      const int b = lane;
      if (b > 0 && b < nb) {
        const float* R = s.xmat[b];
        const float* I = bt.inl[lane];
        // T = R * Iloc * R'   (Iloc symmetric: xx yy zz xy xz yz)
        float Il[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], T[9], Rt[9];
        mulmat3(RI, R, Il);
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
        mulmat3(T, RI, Rt);
        float dif[3] = {xip[lane][0] - s.com[b][0], xip[lane][1] - s.com[b][1], xip[lane][2] - s.com[b][2]};
        const float mass = I[9], dd = dot3(dif, dif);
        float* c = s.u.t.cinert[b];
        c[0] = T[0] + mass * (dd - dif[0] * dif[0]); c[1] = T[4] + mass * (dd - dif[1] * dif[1]);
        c[2] = T[8] + mass * (dd - dif[2] * dif[2]); c[3] = T[1] - mass * dif[0] * dif[1];
        c[4] = T[2] - mass * dif[0] * dif[2]; c[5] = T[5] - mass * dif[1] * dif[2];
        c[6] = mass * dif[0]; c[7] = mass * dif[1]; c[8] = mass * dif[2]; c[9] = mass;
      }
      if (lane < nv) {
        const int d = lane, b = dt.body[lane];
        float off[3] = {s.com[b][0] - s.xanchor[d][0], s.com[b][1] - s.xanchor[d][1], s.com[b][2] - s.xanchor[d][2]};
        float* c = cdof[lane];
        const int jt = dt.jtype[lane], k = d - dt.first[lane];
        if (jt == JT_SLIDE || (jt == JT_FREE && k < 3)) {
          c[0] = c[1] = c[2] = 0;
          for (int x = 0; x < 3; x++) c[3 + x] = s.xaxis[d][x];
        } else {
          for (int x = 0; x < 3; x++) c[x] = s.xaxis[d][x];
          cross3(c + 3, s.xaxis[d], off);
        }
        for (int x = 0; x < 6; x++) s.u.t.cdof[d][x] = c[x];
        qvel_r[lane] = s.qvel[d];
      } else {
        for (int x = 0; x < 6; x++) cdof[lane][x] = 0;
        qvel_r[lane] = 0;
      }
    }
    SYNC();
    // comVel.  cvel[b] = sum over the dofs on the path to b of cdof*qvel: own-body sums, then pointer jumping up the tree
    // (k_body_jump, as in kinematics) instead of a loop over up to 26 ancestor dofs per lane.
    LANES {
      if (lane < nv)
        for (int x = 0; x < 6; x++) s.u.t.buf[lane][x] = cdof[lane][x] * qvel_r[lane];
    }
    SYNC();
    PL<float[6]> cvb;
    LANES {
      float sv[6] = {0, 0, 0, 0, 0, 0};
      if (lane > 0 && lane < nb) {
        const int da = bt.dofadr[lane], dn = bt.dofnum[lane];
#pragma unroll
        for (int u = 0; u < 6; u++) {   // a body carries at most six dofs; fixed trip count, masked tail
          const float on = u < dn ? 1.f : 0.f;
          const int a = da + (u < dn ? u : 0);
          for (int x = 0; x < 6; x++) sv[x] += on * s.u.t.buf[dn > 0 ? a : 0][x];
        }
      }
      for (int x = 0; x < 6; x++) { cvb[lane][x] = sv[x]; if (lane < NBP) s.u.t.cvel[lane][x] = sv[x]; }
    }
    SYNC();
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r >= M.njump) break;
      PL<float[6]> up;
      PL<int> anc;
      LANES {
        const int a = bt.jump[lane][r];
        anc[lane] = a;
        for (int x = 0; x < 6; x++) up[lane][x] = s.u.t.cvel[a][x];
      }
      SYNC();
      LANES {
        if (anc[lane] > 0)
          for (int x = 0; x < 6; x++) { cvb[lane][x] += up[lane][x]; s.u.t.cvel[lane][x] = cvb[lane][x]; }
      }
      SYNC();
    }
    // cdof_dot (dof lanes): the velocity seen by dof d is the parent body's cvel plus the earlier dofs of its own body
    // that k_dof_velmask admits; crb = sum of cinert over the contiguous DFS subtree, four bodies per pass
    LANES {
      if (lane < nv) {
        const int b = dt.body[lane], pb = M.body_parentid[b], da = M.body_dofadr[b];
        float cv[6];
        for (int x = 0; x < 6; x++) cv[x] = s.u.t.cvel[pb][x];
        const uint64_t mk = dt.velmask[lane];
#pragma unroll
        for (int u = 0; u < 5; u++) {
          const int a = da + u;
          const float on = (a < lane && ((mk >> a) & 1)) ? 1.f : 0.f;
          for (int x = 0; x < 6; x++) cv[x] += on * s.u.t.buf[a < lane ? a : lane][x];
        }
        const int jt = dt.jtype[lane], k = lane - dt.first[lane];
        if (jt == JT_FREE && k < 3) { for (int x = 0; x < 6; x++) cdof_dot[lane][x] = 0; }
        else cross_motion(cdof_dot[lane], cv, cdof[lane]);
        for (int x = 0; x < 6; x++) s.u.t.cdof_dot[lane][x] = cdof_dot[lane][x];
      } else {
        for (int x = 0; x < 6; x++) cdof_dot[lane][x] = 0;
      }
    }
    {
      PL<float[10]> cr;
      LANES { for (int k = 0; k < 10; k++) cr[lane][k] = 0.f; }
      for (int x0 = 0; x0 < M.maxsubtree; x0 += 4) {
        LANES {
          if (lane > 0 && lane < nb) {
            const int sub = bt.subsize[lane];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const bool in = x0 + u < sub;
              const int x = lane + (in ? x0 + u : 0);
              const float on = in ? 1.f : 0.f;
              for (int k = 0; k < 10; k++) cr[lane][k] += on * s.u.t.cinert[x][k];
            }
          }
        }
      }
      LANES {
        if (lane < nb)
          for (int k = 0; k < 10; k++) s.u.t.crb[lane][k] = cr[lane][k];
      }
    }
    SYNC();
    LANES {
      if (lane < nv) {
        float bf[6];
        mul_inert_vec(bf, s.u.t.crb[dt.body[lane]], cdof[lane]);
        for (int x = 0; x < 6; x++) s.u.t.buf[lane][x] = bf[x];
      }
    }
    SYNC();
    // M entries on the static sparsity pattern: lower (working copy), upper (kept), Mdiag
    LANES {
      for (int t = 0; t < NENT; t++) {
        const int i = et.i[lane][t], j = et.j[lane][t];
        if (i < 0) continue;
        float v = 0;
        for (int x = 0; x < 6; x++) v += s.u.t.cdof[j][x] * s.u.t.buf[i][x];
        if (i == j) { v += et.arm[lane][t]; s.Mdiag[i] = v; }
        s.MM[i][j] = v;
        if (i != j) s.MM[j][i] = v;
      }
    }
    SYNC();
  }

  // ------------------------------------------------------------------ sparse L'DL in place (lower part of MM)
  // [MJ] mj_factorM: for k = nv-1..0, eliminate row k from its ancestors' block.  Entry (i,j), j<=i<k, both
  // ancestors of k:  M[i][j] -= M[k][i]*M[k][j]/M[k][k].  Non-ancestors hold zeros (static pattern, no fill-in).
  SMJ_DEV void factor() {
    const int nv = M.nv;
    EntryTab et;
    load(et, false);
    for (int k = nv - 1; k >= 0; k--) {
      const float dkk = s.MM[k][k], dinv = 1.0f / dkk;
      LANES {
        for (int t = 0; t < NENT; t++) {
          const int i = et.i[lane][t], j = et.j[lane][t];
          if (i < 0 || i >= k) continue;
          const float a = s.MM[k][i];
          if (a != 0.f) s.MM[i][j] -= a * s.MM[k][j] * dinv;
        }
      }
      SYNC();
      LANES {
        if (lane < k) s.MM[k][lane] *= dinv;
        if (lane == k) s.Dinv[k] = dinv;
      }
      SYNC();
    }
  }
  // x <- L^-T-phase: for i = nv-1..0: x[j] -= L[i][j]*x[i] for j<i   (x lane-resident, lane = dof)
  SMJ_DEV void solve_LT(PL<float>& x) {
    for (int i = M.nv - 1; i > 0; i--) {
      const float xi = wave_read(x, i);
      LANES { if (lane < i) x[lane] -= s.MM[i][lane] * xi; }
    }
  }
  // x <- L^-1-phase: for j = 0..nv-1: x[i] -= L[i][j]*x[j] for i>j
  SMJ_DEV void solve_L(PL<float>& x) {
    for (int j = 0; j < M.nv - 1; j++) {
      const float xj = wave_read(x, j);
      LANES { if (lane > j && lane < M.nv) x[lane] -= s.MM[lane][j] * xj; }
    }
  }

  // ------------------------------------------------------------------ B.5/B.6 smooth forces -> g
  SMJ_DEV void smooth_forces(bool dbg) {
    const int nb = M.nbody, nv = M.nv, nu = M.nu;
    PL<float> frc_passive, frc_bias, frc_act;
    DofTab dt;
    ActTab at;
    load(dt); load(at);
    // passive: damper + spring
    LANES {
      float f = 0;
      if (lane < nv) {
        f = -dt.damp[lane] * s.qvel[lane];
        if (dt.stiff[lane] != 0.f) f -= dt.stiff[lane] * (s.qpos[dt.qadr[lane]] - dt.spring[lane]);
      }
      frc_passive[lane] = f;
    }
    // gravity compensation  [MJ] mj_passive gravcomp: F = -g*m*gravcomp at the body's gravcomp point.  Lane = gravcomp slot
    // stages (offset from the subtree com, force, dof mask) in LDS; then every dof lane runs a fixed 16-slot loop.
    BodyTab bt;
    load(bt);
    LANES {
      if (lane < 16) {
        float off[3] = {0, 0, 0}, F[3] = {0, 0, 0};
        int mlo = 0, mhi = 0;
        if (lane < M.ngc) {
          const int b = at.gc_body[lane];
          const float gm = at.gc_mass[lane];
          const float lp[3] = {at.gc_x[lane], at.gc_y[lane], at.gc_z[lane]};
          float pt[3];
          mulmat3vec(pt, s.xmat[b], lp);
          for (int k = 0; k < 3; k++) { off[k] = pt[k] + s.xpos[b][k] - s.com[b][k]; F[k] = -M.gravity[k] * gm; }
          mlo = at.gc_mlo[lane]; mhi = at.gc_mhi[lane];
        }
        float* o = s.u.t.buf[lane];     // buf[slot] = off, F ; cfrc is free until the RNE pass below
        for (int k = 0; k < 3; k++) { o[k] = off[k]; o[3 + k] = F[k]; }
        s.u.t.cfrc[lane][0] = __builtin_bit_cast(float, mlo); s.u.t.cfrc[lane][1] = __builtin_bit_cast(float, mhi);
      }
    }
    SYNC();
    LANES {
      if (lane < nv) {
        float acc = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) {
          const uint64_t mk = mk64(__builtin_bit_cast(int, s.u.t.cfrc[t][0]), __builtin_bit_cast(int, s.u.t.cfrc[t][1]));
          const float* o = s.u.t.buf[t];
          float tv[3];
          cross3(tv, cdof[lane], o);
          const float v = (cdof[lane][3] + tv[0]) * o[3] + (cdof[lane][4] + tv[1]) * o[4] + (cdof[lane][5] + tv[2]) * o[5];
          acc += ((mk >> lane) & 1) ? v : 0.f;
        }
        frc_passive[lane] += acc;
      }
    }
    SYNC();
    // RNE bias: cacc (no qacc) per body = gravity + sum over the dofs on the path of cdof_dot*qvel (own-body sums, then
    // pointer jumping up the tree), body force, subtree sum projected on cdof  [MJ] mj_rne(flg_acc=0)
    LANES {
      if (lane < nv)
        for (int x = 0; x < 6; x++) s.u.t.buf[lane][x] = s.u.t.cdof_dot[lane][x] * s.qvel[lane];
    }
    SYNC();
    PL<float[6]> ca;
    LANES {
      float sv[6] = {0, 0, 0, 0, 0, 0};
      if (lane > 0 && lane < nb) {
        const int da = bt.dofadr[lane], dn = bt.dofnum[lane];
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const float on = u < dn ? 1.f : 0.f;
          const int a = dn > 0 ? da + (u < dn ? u : 0) : 0;
          for (int x = 0; x < 6; x++) sv[x] += on * s.u.t.buf[a][x];
        }
      }
      for (int x = 0; x < 6; x++) { ca[lane][x] = sv[x]; if (lane < NBP) s.u.t.cfrc[lane][x] = sv[x]; }
    }
    SYNC();
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r >= M.njump) break;
      PL<float[6]> up;
      PL<int> anc;
      LANES {
        const int a = bt.jump[lane][r];
        anc[lane] = a;
        for (int x = 0; x < 6; x++) up[lane][x] = s.u.t.cfrc[a][x];
      }
      SYNC();
      LANES {
        if (anc[lane] > 0)
          for (int x = 0; x < 6; x++) { ca[lane][x] += up[lane][x]; s.u.t.cfrc[lane][x] = ca[lane][x]; }
      }
      SYNC();
    }
    LANES {
      if (lane < nb) {
        const int b = lane;
        float t1[6] = {0, 0, 0, 0, 0, 0}, cf[6] = {0, 0, 0, 0, 0, 0};
        if (b > 0) {
          const float a[6] = {ca[lane][0], ca[lane][1], ca[lane][2], ca[lane][3] - M.gravity[0], ca[lane][4] - M.gravity[1], ca[lane][5] - M.gravity[2]};
          float t2[6];
          mul_inert_vec(t1, s.u.t.cinert[b], a);
          mul_inert_vec(t2, s.u.t.cinert[b], s.u.t.cvel[b]);
          cross_force(cf, s.u.t.cvel[b], t2);
        }
        for (int x = 0; x < 6; x++) s.u.t.cfrc[b][x] = t1[x] + cf[x];
      }
    }
    SYNC();
    {
      PL<float[6]> cfs;
      LANES { for (int k = 0; k < 6; k++) cfs[lane][k] = 0.f; }
      for (int x0 = 0; x0 < M.maxsubtree; x0 += 4) {   // subtree sums over the contiguous DFS range, four bodies per pass
        LANES {
          if (lane < nv) {
            const int b = dt.body[lane], sub = dt.bsub[lane];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const bool in = x0 + u < sub;
              const int x = b + (in ? x0 + u : 0);
              const float on = in ? 1.f : 0.f;
              for (int k = 0; k < 6; k++) cfs[lane][k] += on * s.u.t.cfrc[x][k];
            }
          }
        }
      }
      LANES {
        float v = 0;
        if (lane < nv)
          for (int k = 0; k < 6; k++) v += cdof[lane][k] * cfs[lane][k];
        frc_bias[lane] = v;
      }
    }
    // actuation  [MJ] mj_fwdActuation (static moments: joint / fixed-tendon transmissions), constants lane-resident
    LANES {
      if (lane < nu) {
        const int a = lane;
        float len = 0, vel = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int dd = at.dof[lane][t];
          if (dd >= 0) { vel += at.mom[lane][t] * s.qvel[dd]; len += at.mom[lane][t] * s.qpos[at.qadr[lane][t]]; }
        }
        const float* pr = at.prm[lane];
        const int fl = at.flags[lane];
        float ctrl = s.ctrl[a];
        if (fl & 1) ctrl = fminf(pr[5], fmaxf(pr[4], ctrl));
        float f = pr[0] * ctrl;
        if (fl & 4) f += pr[1] + pr[2] * len + pr[3] * vel;
        bool clamped = false;
        if (fl & 2) { clamped = (f <= pr[6] || f >= pr[7]); f = fminf(pr[7], fmaxf(pr[6], f)); }
        s.act_force[a] = f; s.act_len[a] = len; s.act_vel[a] = vel;
        s.act_free[a] = clamped ? 0.f : 1.f;   // the velocity derivative of a force-clamped actuator is dropped ([MJ] mjd_actuator_vel)
      }
    }
    SYNC();
    LANES {
      float v = 0;
      if (lane < nv) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int a = dt.act[lane][t];
          if (a >= 0) v += dt.actmom[lane][t] * s.act_force[a];
        }
      }
      frc_act[lane] = v;
      g_r[lane] = frc_passive[lane] - frc_bias[lane] + frc_act[lane];
      if (lane < nv) s.g[lane] = g_r[lane];
      if (dbg && S.debug && lane < nv) {
        S.debug[(SMJ_DBG_QFRC_BIAS + lane) * S.ld + env] = frc_bias[lane];
        S.debug[(SMJ_DBG_QFRC_PASSIVE + lane) * S.ld + env] = frc_passive[lane];
        S.debug[(SMJ_DBG_QFRC_ACT + lane) * S.ld + env] = frc_act[lane];
        S.debug[(SMJ_DBG_G + lane) * S.ld + env] = g_r[lane];
      }
    }
    SYNC();
  }

  // ------------------------------------------------------------------ B.3 collision (plane pairs)
  SMJ_DEV void geom_pose(int g, float* pos, float* mat) const {
    const int b = M.geom_bodyid[g];
    float lp[3] = {M.geom_pos[3 * g], M.geom_pos[3 * g + 1], M.geom_pos[3 * g + 2]};
    mulmat3vec(pos, s.xmat[b], lp);
    for (int k = 0; k < 3; k++) pos[k] += s.xpos[b][k];
    float lm[9];
    for (int k = 0; k < 9; k++) lm[k] = M.k_geom_mat[9 * g + k];
    mulmat3(mat, s.xmat[b], lm);
  }
  // the same from a record: body id, local position and local frame already in registers
  SMJ_DEV void pose_from(int b, const float* lp, const float* lm, float* pos, float* mat) const {
    mulmat3vec(pos, s.xmat[b], lp);
    for (int k = 0; k < 3; k++) pos[k] += s.xpos[b][k];
    mulmat3(mat, s.xmat[b], lm);
  }
  // [MJ] mju_makeFrame: fr[0:3] = normal given; tangents built around it
  SMJ_DEV static void make_frame(float* fr) {
    normalize3(fr);
    if (fr[1] > -0.5f && fr[1] < 0.5f) { fr[3] = 0; fr[4] = 1; fr[5] = 0; } else { fr[3] = 0; fr[4] = 0; fr[5] = 1; }
    const float t = dot3(fr, fr + 3);
    for (int k = 0; k < 3; k++) fr[3 + k] -= t * fr[k];
    normalize3(fr + 3);
    cross3(fr + 6, fr, fr + 3);
  }
  // per-lane: fill contact slot c from the convex pair's record (DevModel::k_cprec)
  SMJ_DEV int cslot(int c) const { return (SMJ_SPLIT_COLLIDE && rev) ? NCON - 1 - c : c; }
  SMJ_DEV float (*mcs())[3] {
#if SMJ_SPLIT_COLLIDE
    return rev ? s.u.c.mc2 : s.u.c.mc;
#else
    return s.u.c.mc;
#endif
  }
  // room for n more contacts?  One wavefront: its own count.  Two side by side: the slots are claimed from the shared counter (an
  // LDS atomic), so the two lists -- one growing from the bottom, one from the top -- cannot meet; a claim that fails leaves the
  // counter above NCON, and every later claim of either wavefront fails too (the step is flagged and redone by the larger build).
  SMJ_DEV bool con_claim(int n) {
#if SMJ_SPLIT_COLLIDE
    if (split_on) {
      if (ncon + n > NCON) return false;
      int old = 0;
      LANES { if (lane == 0) old = lds_atomic_add(&s.u.c.contotal, n); }
      return uni(old) + n <= NCON;
    }
#endif
    return ncon + n <= NCON;
  }
  SMJ_DEV void write_contact(int c_, const int* r, float dist, const float* pos, const float* n) {
    const int c = cslot(c_);
    s.cdist[c] = dist;
    float fr[9] = {n[0], n[1], n[2], 0, 0, 0, 0, 0, 0};
    make_frame(fr);
    for (int k = 0; k < 9; k++) s.cframe[c][k] = fr[k];
    for (int k = 0; k < 3; k++) s.cpos[c][k] = pos[k];
    for (int k = 0; k < 5; k++) { s.cfric[c][k] = asf(r[SMJ_CP_FRIC + k]); s.csolimp[c][k] = asf(r[SMJ_CP_SOLIMP + k]); }
    s.csolref[c][0] = asf(r[SMJ_CP_SOLREF]); s.csolref[c][1] = asf(r[SMJ_CP_SOLREF + 1]);
    s.cmargin[c] = asf(r[SMJ_CP_MG]);
    s.cdim[c] = r[SMJ_CP_CONDIM]; s.cgeom1[c] = r[SMJ_CP_G1]; s.cgeom2[c] = r[SMJ_CP_G2]; s.cefc[c] = -1;
#if NSAT > 0
    s.cpair[c] = r[SMJ_CP_PAIR];
#endif
  }
  SMJ_DEV void add_contact(const int* r, float dist, const float* pos, const float* n) {
    // uniform: every lane calls with identical arguments; lane 0 writes
    if (!con_claim(1)) { flags |= SMJ_FLAG_CON_OVERFLOW | 0x4000; return; }
    const int c = ncon++;
    LANES { if (lane == 0) write_contact(c, r, dist, pos, n); }
  }

  // Plane pairs.  Lane = pair: bounding sphere vs plane, then the primitive narrowphase of the hits runs lane-parallel
  // (types diverge, the work of all hit pairs overlaps); mesh hulls take the wave-serial vertex scan.  Results are staged
  // per lane and emitted by prefix sum, which keeps the contacts in pair-table order.
  SMJ_DEV void collision() {
    ncon = 0;
    for (int base = 0; base < M.nplanepair; base += 64) {
      PL<int> hull, cv;
      LANES {
        int cnt = 0, hl = 0;
        const int t = base + lane;
        if (t < M.nplanepair) {
          // the pair's record (DevModel::k_pprec): one level of wide loads instead of pair -> geoms -> bodies / frames / sizes
          const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_pprec + opaque(t) * SMJ_PP_STRIDE, 16));
          int v[SMJ_PP_CONDIM];
          for (int k = 0; k < SMJ_PP_CONDIM; k++) v[k] = r[k];
          const int b1 = v[SMJ_PP_B1], b2 = v[SMJ_PP_B2];
          float lp1[3], lm1[9], pp[3], pm[9];
          for (int k = 0; k < 3; k++) lp1[k] = asf(v[SMJ_PP_POS1 + k]);
          for (int k = 0; k < 9; k++) lm1[k] = asf(v[SMJ_PP_MAT1 + k]);
          pose_from(b1, lp1, lm1, pp, pm);
          float c2[3], lc[3] = {asf(v[SMJ_PP_BCEN2]), asf(v[SMJ_PP_BCEN2 + 1]), asf(v[SMJ_PP_BCEN2 + 2])};
          mulmat3vec(c2, s.xmat[b2], lc);
          const float n[3] = {pm[2], pm[5], pm[8]};   // plane normal = z axis of the plane geom frame
          const float dif[3] = {c2[0] + s.xpos[b2][0] - pp[0], c2[1] + s.xpos[b2][1] - pp[1], c2[2] + s.xpos[b2][2] - pp[2]};
          const float margin = asf(v[SMJ_PP_MARGIN]);
          if (dot3(dif, n) - asf(v[SMJ_PP_RBOUND2]) <= margin) {
            const int t2 = v[SMJ_PP_T2];
            if (t2 == GT_MESH) {
              // the hull's box (geom frame) against the plane before the wave-serial vertex scan: the lowest point of the box
              // along the normal bounds every vertex.  (The base hull's bounding sphere always reaches the floor, its box
              // stays a centimetre above it: one vertex scan per step saved.)
              float lp2[3], lm2[9], gp[3], gm[9], bc[3], wc[3];
              for (int k = 0; k < 3; k++) { lp2[k] = asf(v[SMJ_PP_POS2 + k]); bc[k] = asf(r[SMJ_PP_BOX2 + k]); }
              for (int k = 0; k < 9; k++) lm2[k] = asf(v[SMJ_PP_MAT2 + k]);
              pose_from(b2, lp2, lm2, gp, gm);
              mulmat3vec(wc, gm, bc);
              float low = (gp[0] + wc[0] - pp[0]) * n[0] + (gp[1] + wc[1] - pp[1]) * n[1] + (gp[2] + wc[2] - pp[2]) * n[2];
              for (int k = 0; k < 3; k++) low -= fabsf(n[0] * gm[k] + n[1] * gm[3 + k] + n[2] * gm[6 + k]) * (asf(r[SMJ_PP_BOX2 + 3 + k]) + 1e-5f);
              hl = low <= margin + 1e-6f;
            } else {
              float lp2[3], lm2[9], size[3];
              for (int k = 0; k < 3; k++) { lp2[k] = asf(v[SMJ_PP_POS2 + k]); size[k] = asf(v[SMJ_PP_SIZE2 + k]); }
              for (int k = 0; k < 9; k++) lm2[k] = asf(v[SMJ_PP_MAT2 + k]);
              cnt = plane_prim(b2, lp2, lm2, size, t2, pp, n, margin, s.u.p.dist[lane], s.u.p.pos[lane]);
            }
            for (int k = 0; k < 3; k++) s.u.p.nrm[lane][k] = n[k];
            s.u.p.pair[lane] = t;   // table slot; the emission below reads the same record
          }
        }
        s.u.p.cnt[lane] = cnt;
        hull[lane] = hl;
      }
      uint64_t hm = wave_ballot(hull);
      if (hm) {
        SYNC();
        while (hm) {
          const int l = ffs64(hm);
          hm &= hm - 1;
          narrow_plane_hull(uni(M.k_planepair[base + l]), l);   // (lane l staged the table slot; the hull scan wants the pair id)
        }
        SYNC();
      }
      // emit in pair order: exclusive prefix of the per-lane counts (0..4) from three ballots
      PL<int> bit;
      LANES { cv[lane] = s.u.p.cnt[lane]; bit[lane] = cv[lane] & 1; }
      const uint64_t m0 = wave_ballot(bit);
      LANES { bit[lane] = cv[lane] & 2; }
      const uint64_t m1 = wave_ballot(bit);
      LANES { bit[lane] = cv[lane] & 4; }
      const uint64_t m2 = wave_ballot(bit);
      const int total = popc64(m0) + 2 * popc64(m1) + 4 * popc64(m2);
      if (total) {
        LANES {
          const int c = cv[lane];
          if (c) {
            const uint64_t lt = (1ull << lane) - 1;
            const int off = ncon + popc64(m0 & lt) + 2 * popc64(m1 & lt) + 4 * popc64(m2 & lt);
            const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_pprec + s.u.p.pair[lane] * SMJ_PP_STRIDE, 16));
            const int g1 = r[SMJ_PP_G1], g2 = r[SMJ_PP_G2];
            // the contacts of one pair share normal, frame and parameters: fetch / build them once, then a fixed 4-slot loop
            float fr[9] = {s.u.p.nrm[lane][0], s.u.p.nrm[lane][1], s.u.p.nrm[lane][2], 0, 0, 0, 0, 0, 0};
            make_frame(fr);
            float fric[5], simp[5];
            for (int k = 0; k < 5; k++) { fric[k] = asf(r[SMJ_PP_FRIC + k]); simp[k] = asf(r[SMJ_PP_SOLIMP + k]); }
            const float sr0 = asf(r[SMJ_PP_SOLREF]), sr1 = asf(r[SMJ_PP_SOLREF + 1]), mg = asf(r[SMJ_PP_MG]);
            const int cd = r[SMJ_PP_CONDIM];
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int ci = off + k;
              if (k < c && ci < NCON) {
                s.cdist[ci] = s.u.p.dist[lane][k];
                for (int x = 0; x < 9; x++) s.cframe[ci][x] = fr[x];
                for (int x = 0; x < 3; x++) s.cpos[ci][x] = s.u.p.pos[lane][k][x];
                for (int x = 0; x < 5; x++) { s.cfric[ci][x] = fric[x]; s.csolimp[ci][x] = simp[x]; }
                s.csolref[ci][0] = sr0; s.csolref[ci][1] = sr1;
                s.cmargin[ci] = mg;
                s.cdim[ci] = cd; s.cgeom1[ci] = g1; s.cgeom2[ci] = g2; s.cefc[ci] = -1;
#if NSAT > 0
                s.cpair[ci] = r[SMJ_PP_PAIR];
#endif
              }
            }
          }
        }
        if (ncon + total > NCON) { flags |= SMJ_FLAG_CON_OVERFLOW; ncon = NCON; }
        else ncon += total;
      }
    }
    SYNC();
  }

  // per-lane narrowphase of a plane against a sphere / cylinder / box; returns the number of contacts written
  SMJ_DEV int plane_prim(int b2, const float* lp2, const float* lm2, const float* size, int t2, const float* pp, const float* n,
                         float margin, float* cdist, float (*cpos)[3]) const {
    float gp[3], gm[9];
    pose_from(b2, lp2, lm2, gp, gm);
    int cnt = 0;
#define put(d, q) do { cdist[cnt] = (d); cpos[cnt][0] = (q)[0]; cpos[cnt][1] = (q)[1]; cpos[cnt][2] = (q)[2]; cnt++; } while (0)
    if (t2 == GT_SPHERE) {
      const float dif[3] = {gp[0] - pp[0], gp[1] - pp[1], gp[2] - pp[2]};
      const float dist = dot3(dif, n) - size[0];
      if (dist <= margin) {
        float pos[3];
        for (int k = 0; k < 3; k++) pos[k] = gp[k] - n[k] * (size[0] + 0.5f * dist);
        put(dist, pos);
      }
    } else if (t2 == GT_CYLINDER) {  // [MJ] mjc_PlaneCylinder
      float axis[3] = {gm[2], gm[5], gm[8]};
      float prjaxis = dot3(n, axis);
      if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
      float vec[3] = {gp[0] - pp[0], gp[1] - pp[1], gp[2] - pp[2]};
      const float dist0 = dot3(vec, n);
      for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - n[k];
      const float len2 = dot3(vec, vec);
      // (vec = sin(tilt) x the downhill direction.  fp32: its components are differences of O(1) numbers, noise 1e-7 -- a cylinder
      // STANDING on the plane after a yaw gave vec = (0, 0, -1.2e-7), "downhill" straight into the plane, and lost its contacts or got
      // one 8 cm deep.  Below a tilt of 3e-5 rad the cylinder is upright: MuJoCo's branch for len2 < mjMINVAL^2.)
      if (len2 >= 1e-9f) { const float sc = size[0] / sqrtf(len2); for (int k = 0; k < 3; k++) vec[k] *= sc; }
      else { vec[0] = gm[0] * size[0]; vec[1] = gm[3] * size[0]; vec[2] = gm[6] * size[0]; }
      const float prjvec = dot3(vec, n);
      for (int k = 0; k < 3; k++) axis[k] *= size[1];
      prjaxis *= size[1];
      if (dist0 + prjaxis + prjvec <= margin) {
        float pos[3], dist = dist0 + prjaxis + prjvec;
        for (int k = 0; k < 3; k++) pos[k] = gp[k] + vec[k] + axis[k] - n[k] * dist * 0.5f;
        put(dist, pos);
        if (dist0 - prjaxis + prjvec <= margin) {
          dist = dist0 - prjaxis + prjvec;
          for (int k = 0; k < 3; k++) pos[k] = gp[k] + vec[k] - axis[k] - n[k] * dist * 0.5f;
          put(dist, pos);
        }
        const float prjvec1 = -prjvec * 0.5f;
        if (dist0 + prjaxis + prjvec1 <= margin) {
          float vec1[3];
          cross3(vec1, vec, axis);
          normalize3(vec1);
          for (int k = 0; k < 3; k++) vec1[k] *= size[0] * 0.8660254037844386f;
          dist = dist0 + prjaxis + prjvec1;
          for (int sg = 0; sg < 2 && cnt < 4; sg++) {
            const float sgn = sg ? -1.f : 1.f;
            for (int k = 0; k < 3; k++) pos[k] = gp[k] + sgn * vec1[k] + axis[k] - vec[k] * 0.5f - n[k] * dist * 0.5f;
            put(dist, pos);
          }
        }
      }
    } else if (t2 == GT_CAPSULE) {  // [MJ] mjc_PlaneCapsule: the two end spheres, +axis end first
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const float sg = e ? -1.f : 1.f;
        const float c[3] = {gp[0] + sg * size[1] * gm[2], gp[1] + sg * size[1] * gm[5], gp[2] + sg * size[1] * gm[8]};
        const float dif[3] = {c[0] - pp[0], c[1] - pp[1], c[2] - pp[2]};
        const float dist = dot3(dif, n) - size[0];
        if (dist <= margin) {
          float pos[3];
          for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (size[0] + 0.5f * dist);
          put(dist, pos);
        }
      }
    } else if (t2 == GT_ELLIPSOID) {  // [MJ] mjc_PlaneEllipsoid: the surface point whose outward normal is -n
      float nl[3], sv[3], loc[3], wv[3];
      mulmat3Tvec(nl, gm, n);
      for (int k = 0; k < 3; k++) sv[k] = -nl[k] * size[k];
      const float len = sqrtf(dot3(sv, sv));
      for (int k = 0; k < 3; k++) loc[k] = len > SMJ_MINVAL ? size[k] * sv[k] / len : 0.f;
      mulmat3vec(wv, gm, loc);
      const float dif[3] = {gp[0] + wv[0] - pp[0], gp[1] + wv[1] - pp[1], gp[2] + wv[2] - pp[2]};
      const float dist = dot3(dif, n);
      if (dist <= margin) {
        float pos[3];
        for (int k = 0; k < 3; k++) pos[k] = gp[k] + wv[k] - n[k] * dist * 0.5f;
        put(dist, pos);
      }
    } else if (t2 == GT_BOX) {  // [MJ] mjc_PlaneBox
      const float dif[3] = {gp[0] - pp[0], gp[1] - pp[1], gp[2] - pp[2]};
      const float dist = dot3(dif, n);
      for (int i = 0; i < 8 && cnt < 4; i++) {
        const float v[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]};
        float corner[3];
        mulmat3vec(corner, gm, v);
        const float ldist = dot3(n, corner);
        if (dist + ldist > margin || ldist > 0) continue;
        const float cd = dist + ldist;
        float pos[3];
        for (int k = 0; k < 3; k++) pos[k] = gp[k] + corner[k] - n[k] * cd * 0.5f;
        put(cd, pos);
      }
    }
#undef put
    return cnt;
  }

  // wave-serial narrowphase of plane pair p (a mesh hull), staged into row l
  SMJ_DEV void narrow_plane_hull(int p, int l) {
    const int g1 = uni(M.pair_geom1[p]), g2 = uni(M.pair_geom2[p]);
    const float margin = uni(M.pair_margin[p]);
    float pp[3], pm[9], gp[3], gm[9];
    geom_pose(g1, pp, pm);
    geom_pose(g2, gp, gm);
    const float n[3] = {pm[2], pm[5], pm[8]};
    plane_hull(l, g2, pp, n, gp, gm, margin);
  }

  // plane vs convex hull, vertices strided over lanes; same selection rule as oracle plane_hull()
  SMJ_DEV void plane_hull(int l, int g2, const float* pp, const float* n, const float* gp, const float* gm, float margin) {
    const float* verts = M.k_hull_vert4 + 4 * uni(M.geom_hulladr[g2]);
    const Vec4* v4 = reinterpret_cast<const Vec4*>(verts);
    const int nvert = uni(M.geom_hullnum[g2]);
    float nl[3];
    mulmat3Tvec(nl, gm, n);
    const float off = (gp[0] - pp[0]) * n[0] + (gp[1] - pp[1]) * n[1] + (gp[2] - pp[2]) * n[2];
    // Every pass scans the vertices strided over the lanes, four 16-byte loads in flight per lane (clamped tail: the last
    // vertex may be visited twice, which changes neither an extreme nor its lowest index).
    // pass 1: deepest vertex (ties -> lowest index)
    PL<float> best;
    PL<int> bidx;
    LANES {
      float bd = 3.0e38f;
      int bi = -1;
      for (int i0 = lane; i0 < nvert; i0 += 256) {
        Vec4 v[4];
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { id[u] = i0 + 64 * u < nvert ? i0 + 64 * u : nvert - 1; v[u] = v4[id[u]]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float dd = nl[0] * v[u].x + nl[1] * v[u].y + nl[2] * v[u].z + off;
          if (dd <= margin && (dd < bd || (dd == bd && id[u] < bi))) { bd = dd; bi = id[u]; }
        }
      }
      best[lane] = bd; bidx[lane] = bi;
    }
    const float dmin = wave_min(best);
    if (dmin > margin) return;
    int i1 = pick_index(best, bidx, dmin);
    int idx[4] = {i1, -1, -1, -1};
    const float v1[3] = {verts[4 * i1], verts[4 * i1 + 1], verts[4 * i1 + 2]};
    if (M.max_con_pair > 1) {
      LANES {
        float bd = 1e-12f;
        int bi = -1;
        for (int i0 = lane; i0 < nvert; i0 += 256) {
          Vec4 v[4];
          int id[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { id[u] = i0 + 64 * u < nvert ? i0 + 64 * u : nvert - 1; v[u] = v4[id[u]]; }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const float dd = nl[0] * v[u].x + nl[1] * v[u].y + nl[2] * v[u].z + off;
            const float e[3] = {v[u].x - v1[0], v[u].y - v1[1], v[u].z - v1[2]};
            const float r2 = dot3(e, e);
            if (dd <= margin && (r2 > bd || (r2 == bd && bi >= 0 && id[u] < bi))) { bd = r2; bi = id[u]; }
          }
        }
        best[lane] = bi >= 0 ? -bd : 3.0e38f; bidx[lane] = bi;
      }
      const float m2 = wave_min(best);
      if (m2 < 1.0e38f) idx[1] = pick_index(best, bidx, m2);
    }
    if (idx[1] >= 0 && M.max_con_pair > 2) {
      const int i2 = idx[1];
      float e12[3] = {verts[4 * i2] - v1[0], verts[4 * i2 + 1] - v1[1], verts[4 * i2 + 2] - v1[2]}, side[3];
      cross3(side, nl, e12);
      normalize3(side);
      PL<float> bmin;
      PL<int> imin;
      LANES {
        float smax = 1e-6f, smin = -1e-6f;
        int i3 = -1, i4 = -1;
        for (int i0 = lane; i0 < nvert; i0 += 256) {
          Vec4 v[4];
          int id[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { id[u] = i0 + 64 * u < nvert ? i0 + 64 * u : nvert - 1; v[u] = v4[id[u]]; }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const float dd = nl[0] * v[u].x + nl[1] * v[u].y + nl[2] * v[u].z + off;
            const float e[3] = {v[u].x - v1[0], v[u].y - v1[1], v[u].z - v1[2]};
            const float sv = dot3(e, side);
            if (dd <= margin && (sv > smax || (sv == smax && i3 >= 0 && id[u] < i3))) { smax = sv; i3 = id[u]; }
            if (dd <= margin && (sv < smin || (sv == smin && i4 >= 0 && id[u] < i4))) { smin = sv; i4 = id[u]; }
          }
        }
        best[lane] = i3 >= 0 ? -smax : 3.0e38f; bidx[lane] = i3;
        bmin[lane] = i4 >= 0 ? smin : 3.0e38f; imin[lane] = i4;
      }
      const float m3 = wave_min(best);
      if (m3 < 1.0e38f) idx[2] = pick_index(best, bidx, m3);
      if (M.max_con_pair > 3) {
        const float m4 = wave_min(bmin);
        if (m4 < 1.0e38f) idx[3] = pick_index(bmin, imin, m4);
      }
    }
    int cnt = 0;
    for (int k = 0; k < 4; k++) {
      if (idx[k] < 0) continue;
      const float v[3] = {verts[4 * idx[k]], verts[4 * idx[k] + 1], verts[4 * idx[k] + 2]};
      float wv[3];
      const float dd = dot3(nl, v) + off;
      mulmat3vec(wv, gm, v);
      LANES {
        if (lane == 0) {
          s.u.p.dist[l][cnt] = dd;
          for (int j = 0; j < 3; j++) s.u.p.pos[l][cnt][j] = gp[j] + wv[j] - n[j] * dd * 0.5f;
        }
      }
      cnt++;
    }
    LANES { if (lane == 0) s.u.p.cnt[l] = cnt; }
  }
  // among lanes whose key equals `val`, the lowest stored index (deterministic tie-break)
  SMJ_DEV int pick_index(const PL<float>& key, const PL<int>& idx, float val) {
    PL<float> cand;
    LANES { cand[lane] = (key[lane] == val && idx[lane] >= 0) ? (float)idx[lane] : 3.0e38f; }
    return uni((int)wave_min(cand));
  }


  // ------------------------------------------------------------------ convex-convex narrowphase (MPR)
  // Restates libccd's ccdMPRPenetration (the routine MuJoCo 3.2.6's mjc_Convex uses for mesh / cylinder / box pairs;
  // third-party, not in /root/reference) with the same control flow as oracle/smj_oracle.c mpr_penetration.  Wave-uniform
  // scalar logic; the hull support function is lane-parallel (vertices strided over lanes, arg-max by wave reduction).
  // vc: the first 256 hull vertices, strided over the lanes (vertex lane + 64 u in slot u), fetched once per pair by
  // load_shape and reused by every support query of the MPR run
  struct Shape { int type, nvert; const float* verts; float pos[3], mat[9], size[3]; PL<float[12]> vc; };
  struct MprPt { float v[3], a[3], b[3]; };

  // support point of a primitive in its own frame for the direction dl ([MJ] mjc_support)
  SMJ_DEV static void prim_support(int type, const float* size, const float* dl, float* pl) {
    pl[0] = pl[1] = pl[2] = 0.f;
    if (type == GT_SPHERE) {
      const float n = sqrtf(dot3(dl, dl));
      if (n > SMJ_MINVAL) for (int i = 0; i < 3; i++) pl[i] = size[0] * dl[i] / n;
    } else if (type == GT_BOX) {
      for (int i = 0; i < 3; i++) pl[i] = dl[i] >= 0 ? size[i] : -size[i];
    } else if (type == GT_CYLINDER) {
      const float n = sqrtf(dl[0] * dl[0] + dl[1] * dl[1]);
      if (n > SMJ_MINVAL) { pl[0] = size[0] * dl[0] / n; pl[1] = size[0] * dl[1] / n; }
      pl[2] = dl[2] >= 0 ? size[1] : -size[1];
    } else if (type == GT_CAPSULE) {     // a sphere swept along z: the sphere's support point, moved to the end the direction points to
      const float n = sqrtf(dot3(dl, dl));
      if (n > SMJ_MINVAL) for (int i = 0; i < 3; i++) pl[i] = size[0] * dl[i] / n;
      pl[2] += dl[2] > 0 ? size[1] : (dl[2] < 0 ? -size[1] : 0.f);
    } else if (type == GT_ELLIPSOID) {   // the unit sphere's support point for the scaled direction, scaled back
      const float t[3] = {dl[0] * size[0], dl[1] * size[1], dl[2] * size[2]};
      const float n = sqrtf(dot3(t, t));
      if (n > SMJ_MINVAL) for (int i = 0; i < 3; i++) pl[i] = size[i] * t[i] / n;
    }
  }
  SMJ_DEV void shape_support(const Shape& sh, const float* dir, float* out) {
    float dl[3], pl[3] = {0, 0, 0};
    mulmat3Tvec(dl, sh.mat, dir);
    if (sh.type != GT_MESH) {
      prim_support(sh.type, sh.size, dl, pl);
    } else {
      PL<float> best, bx, by, bz;
      PL<int> bidx;
      const Vec4* verts = reinterpret_cast<const Vec4*>(sh.verts);
      const int nvert = sh.nvert;
      LANES {
        float bd = -3.0e38f, b[3] = {0, 0, 0};
        int bi = -1;
#pragma unroll
        for (int u = 0; u < 4; u++) {   // register-resident vertices
          const int id = lane + 64 * u;
          const float x = sh.vc[lane][3 * u], y = sh.vc[lane][3 * u + 1], z = sh.vc[lane][3 * u + 2];
          const float d = x * dl[0] + y * dl[1] + z * dl[2];
          if (id < nvert && d > bd) { bd = d; bi = id; b[0] = x; b[1] = y; b[2] = z; }
        }
        for (int i0 = 256 + lane; i0 < nvert; i0 += 256) {   // larger hulls: the rest from memory, four 16-byte loads in flight
          Vec4 v[4];
          int id[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { id[u] = i0 + 64 * u < nvert ? i0 + 64 * u : nvert - 1; v[u] = verts[id[u]]; }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const float d = v[u].x * dl[0] + v[u].y * dl[1] + v[u].z * dl[2];
            if (d > bd) { bd = d; bi = id[u]; b[0] = v[u].x; b[1] = v[u].y; b[2] = v[u].z; }
          }
        }
        best[lane] = bd; bidx[lane] = bi; bx[lane] = b[0]; by[lane] = b[1]; bz[lane] = b[2];
      }
      const float mx = wave_max(best);
      // the lane(s) holding the maximum: almost always one -- then it is the owner; exact ties (symmetric hulls along an axis)
      // take the lowest vertex index, like the oracle's first-maximum scan
      PL<int> ism;
      LANES { ism[lane] = best[lane] == mx && bidx[lane] >= 0; }
      const uint64_t mm = wave_ballot(ism);
      // vertex i is scanned by lane i mod 64: the lane holding the maximum is the owner, its coordinates come by v_readlane
      const int owner = popc64(mm) == 1 ? ffs64(mm) : (pick_index(best, bidx, mx) & 63);
      pl[0] = wave_read(bx, owner); pl[1] = wave_read(by, owner); pl[2] = wave_read(bz, owner);
    }
    mulmat3vec(out, sh.mat, pl);
    for (int i = 0; i < 3; i++) out[i] += sh.pos[i];
  }
  // per-lane part of a hull support query: the lane's best vertex for direction dl (hull frame)
  SMJ_DEV void hull_scan(const Shape& sh, const float* dl, int lane, float& bd, int& bi, float* b) {
    const Vec4* verts = reinterpret_cast<const Vec4*>(sh.verts);
    const int nvert = sh.nvert;
    bd = -3.0e38f; bi = -1; b[0] = b[1] = b[2] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; u++) {   // register-resident vertices
      const int id = lane + 64 * u;
      const float x = sh.vc[lane][3 * u], y = sh.vc[lane][3 * u + 1], z = sh.vc[lane][3 * u + 2];
      const float d = x * dl[0] + y * dl[1] + z * dl[2];
      if (id < nvert && d > bd) { bd = d; bi = id; b[0] = x; b[1] = y; b[2] = z; }
    }
    for (int i0 = 256 + lane; i0 < nvert; i0 += 256) {   // larger hulls: the rest from memory, four 16-byte loads in flight
      Vec4 v[4];
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { id[u] = i0 + 64 * u < nvert ? i0 + 64 * u : nvert - 1; v[u] = verts[id[u]]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float d = v[u].x * dl[0] + v[u].y * dl[1] + v[u].z * dl[2];
        if (d > bd) { bd = d; bi = id[u]; b[0] = v[u].x; b[1] = v[u].y; b[2] = v[u].z; }
      }
    }
  }
  SMJ_DEV void mpr_support(const Shape& A, const Shape& Bs, const float* dir, MprPt& p) {
    const float nd[3] = {-dir[0], -dir[1], -dir[2]};
    if (A.type == GT_MESH && Bs.type == GT_MESH) {
      // hull against hull (most convex pairs of the robot): the two queries side by side -- one lane region scans both hulls and
      // the two arg-max reductions are independent chains the scheduler interleaves.  Same arithmetic as shape_support.
      float dla[3], dlb[3];
      mulmat3Tvec(dla, A.mat, dir);
      mulmat3Tvec(dlb, Bs.mat, nd);
      PL<float> besta, ax, ay, az, bestb, bx, by, bz;
      PL<int> ia, ib;
      LANES {
        float ba[3], bb[3];
        hull_scan(A, dla, lane, besta[lane], ia[lane], ba);
        hull_scan(Bs, dlb, lane, bestb[lane], ib[lane], bb);
        ax[lane] = ba[0]; ay[lane] = ba[1]; az[lane] = ba[2]; bx[lane] = bb[0]; by[lane] = bb[1]; bz[lane] = bb[2];
      }
      const float mxa = wave_max(besta), mxb = wave_max(bestb);
      PL<int> isa, isb;
      LANES { isa[lane] = besta[lane] == mxa && ia[lane] >= 0; isb[lane] = bestb[lane] == mxb && ib[lane] >= 0; }
      const uint64_t ma = wave_ballot(isa), mb = wave_ballot(isb);
      const int oa = popc64(ma) == 1 ? ffs64(ma) : (pick_index(besta, ia, mxa) & 63);
      const int ob = popc64(mb) == 1 ? ffs64(mb) : (pick_index(bestb, ib, mxb) & 63);
      const float pla[3] = {wave_read(ax, oa), wave_read(ay, oa), wave_read(az, oa)}, plb[3] = {wave_read(bx, ob), wave_read(by, ob), wave_read(bz, ob)};
      mulmat3vec(p.a, A.mat, pla);
      mulmat3vec(p.b, Bs.mat, plb);
      for (int i = 0; i < 3; i++) { p.a[i] += A.pos[i]; p.b[i] += Bs.pos[i]; p.v[i] = p.a[i] - p.b[i]; }
      return;
    }
    shape_support(A, dir, p.a);
    shape_support(Bs, nd, p.b);
    for (int i = 0; i < 3; i++) p.v[i] = p.a[i] - p.b[i];
  }
  // libccd's CCD_EPS as MuJoCo builds it (double precision: DBL_EPSILON).  These are ABSOLUTE tests on squared lengths and
  // triple products of centimetre-scale vectors: with FLT_EPSILON, |v0 x v1|^2 < eps holds for any two vectors shorter than
  // ~2 cm, and the "origin on the v0-v1 segment" exit of the portal discovery reported a 3 mm penetration for finger hulls
  // that are 0.5 mm apart.  fp32 represents 2.2e-16 without trouble, so the thresholds stay where the reference has them.
  static constexpr float CCD_EPS = 2.220446e-16f;
  SMJ_DEV static bool ccd_zero(float x) { return fabsf(x) < CCD_EPS; }
  SMJ_DEV static bool ccd_eq(float a, float b) {
    const float ab = fabsf(a - b);
    if (ab < CCD_EPS) return true;
    return ab < CCD_EPS * fmaxf(fabsf(a), fabsf(b));
  }
  SMJ_DEV static void portal_dir(const MprPt* P, float* dir) {
    float a[3], b[3];
    for (int i = 0; i < 3; i++) { a[i] = P[2].v[i] - P[1].v[i]; b[i] = P[3].v[i] - P[1].v[i]; }
    cross3(dir, a, b);
    const float n = sqrtf(dot3(dir, dir));
    if (n > 0) for (int i = 0; i < 3; i++) dir[i] /= n;
  }
  // P[t] <- q when `on`, component by component (selects): struct assignments under branches make the compiler address
  // the portal array dynamically, which would put it in scratch memory
  template <int T>
  SMJ_DEV static void portal_set(MprPt* P, const MprPt& q, bool on) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      P[T].v[i] = on ? q.v[i] : P[T].v[i];
      P[T].a[i] = on ? q.a[i] : P[T].a[i];
      P[T].b[i] = on ? q.b[i] : P[T].b[i];
    }
  }
  SMJ_DEV static void expand_portal(MprPt* P, const MprPt& v4) {
    float v4v0[3];
    cross3(v4v0, v4.v, P[0].v);
    int t;
    if (dot3(P[1].v, v4v0) > 0) t = dot3(P[2].v, v4v0) > 0 ? 1 : 3;
    else t = dot3(P[3].v, v4v0) > 0 ? 2 : 1;
    portal_set<1>(P, v4, t == 1);
    portal_set<2>(P, v4, t == 2);
    portal_set<3>(P, v4, t == 3);
  }
  SMJ_DEV static bool reach_tolerance(const MprPt* P, const MprPt& v4, const float* dir, float tol) {
    const float dv4 = dot3(v4.v, dir);
    const float d = fminf(dv4 - dot3(P[1].v, dir), fminf(dv4 - dot3(P[2].v, dir), dv4 - dot3(P[3].v, dir)));
    return ccd_eq(d, tol) || d < tol;
  }
#ifndef SMJ_MPR_LOCAL
#define SMJ_MPR_LOCAL 1     // (0: MPR on world coordinates, rounds 1-5; A/B builds of tools / tests only)
#endif
#ifndef SMJ_PORTAL_NORMAL
#define SMJ_PORTAL_NORMAL 1   // (0: the witness' direction as it comes, the fp32 behaviour of rounds 1-5; A/B builds of tools / tests only)
#endif
  // interior (optional): set when the witness is the projection of the origin INTO the triangle (not onto an edge / a corner)
  SMJ_DEV static float origin_tri_dist2(const float* a, const float* b, const float* c, float* w, bool* interior = nullptr) {
    float d1[3], d2[3];
    if (interior) *interior = false;
    for (int i = 0; i < 3; i++) { d1[i] = b[i] - a[i]; d2[i] = c[i] - a[i]; }
    const float u = dot3(a, a), v = dot3(d1, d1), ww = dot3(d2, d2), p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
    const float den = ww * v - r * r;
    float sc = -1, t = -1;
    if (!ccd_zero(den)) { sc = (q * r - ww * p) / den; t = (-sc * r - q) / ww; }
    if ((ccd_zero(sc) || sc > 0) && (ccd_eq(sc, 1) || sc < 1) && (ccd_zero(t) || t > 0) && (ccd_eq(t, 1) || t < 1) && (ccd_eq(t + sc, 1) || t + sc < 1)) {
      for (int i = 0; i < 3; i++) w[i] = a[i] + sc * d1[i] + t * d2[i];
      if (interior) *interior = true;
      // |w|^2 from the witness point itself: libccd's expanded form (u + s^2 v + t^2 w + 2sp + 2tq + 2str) cancels to noise in
      // fp32 when the penetration (1e-5 m) is small against the portal's distance from the origin (1e-2 m)
      return dot3(w, w);
    }
    float best = -1;
    for (int e = 0; e < 3; e++) {
      const float* s0 = e == 2 ? b : a;
      const float* s1 = e == 0 ? b : c;
      float dd[3], wp[3];
      for (int i = 0; i < 3; i++) dd[i] = s1[i] - s0[i];
      const float l2 = dot3(dd, dd);
      float tt = l2 > 0 ? -dot3(s0, dd) / l2 : 0.f;
      tt = tt < 0 ? 0.f : (tt > 1 ? 1.f : tt);
      for (int i = 0; i < 3; i++) wp[i] = s0[i] + tt * dd[i];
      const float dist = dot3(wp, wp);
      if (best < 0 || dist < best) { best = dist; for (int i = 0; i < 3; i++) w[i] = wp[i]; }
    }
    return best;
  }
  // sep (optional): on a `false` that came from a support point behind the origin, the direction that showed it (the shapes are
  // disjoint along it), else zero
  SMJ_DEV bool mpr_penetration(const Shape& A, const Shape& Bs, const float* c0, const float* c1, float& depth, float* pdir, float* pos, float* sep = nullptr) {
    const float tol = 1e-6f;
    const int maxit = 50;
    MprPt P[4], v4;
    float dir[3], va[3], vb[3], dot;
    if (sep) sep[0] = sep[1] = sep[2] = 0.f;
#define SMJ_SEP_OUT() { if (sep && dot < 0) { sep[0] = dir[0]; sep[1] = dir[1]; sep[2] = dir[2]; } }
    for (int i = 0; i < 3; i++) { P[0].a[i] = c0[i]; P[0].b[i] = c1[i]; P[0].v[i] = c0[i] - c1[i]; }
    if (ccd_zero(P[0].v[0]) && ccd_zero(P[0].v[1]) && ccd_zero(P[0].v[2])) P[0].v[0] += 10.f * CCD_EPS;
    for (int i = 0; i < 3; i++) dir[i] = -P[0].v[i];
    normalize3(dir);
    mpr_support(A, Bs, dir, P[1]);
    dot = dot3(P[1].v, dir);
    if (ccd_zero(dot) || dot < 0) { SMJ_SEP_OUT() return false; }
    cross3(dir, P[0].v, P[1].v);
    if (ccd_zero(dot3(dir, dir))) {
      if (ccd_zero(P[1].v[0]) && ccd_zero(P[1].v[1]) && ccd_zero(P[1].v[2])) { depth = 0; pdir[0] = pdir[1] = pdir[2] = 0; }
      else {
        depth = sqrtf(dot3(P[1].v, P[1].v));
        for (int i = 0; i < 3; i++) pdir[i] = P[1].v[i];
        normalize3(pdir);
      }
      for (int i = 0; i < 3; i++) pos[i] = 0.5f * (P[1].a[i] + P[1].b[i]);
      return true;
    }
    normalize3(dir);
    mpr_support(A, Bs, dir, P[2]);
    dot = dot3(P[2].v, dir);
    if (ccd_zero(dot) || dot < 0) { SMJ_SEP_OUT() return false; }
    for (int i = 0; i < 3; i++) { va[i] = P[1].v[i] - P[0].v[i]; vb[i] = P[2].v[i] - P[0].v[i]; }
    cross3(dir, va, vb);
    normalize3(dir);
    {   // swap P[1] and P[2] when the portal faces away from the origin -- by per-component selects: a struct swap under a
        // branch makes the compiler address the portal array dynamically, which puts it in scratch memory
      const bool sw = dot3(dir, P[0].v) > 0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const float v1 = P[1].v[i], v2 = P[2].v[i], a1 = P[1].a[i], a2 = P[2].a[i], b1 = P[1].b[i], b2 = P[2].b[i];
        P[1].v[i] = sw ? v2 : v1; P[2].v[i] = sw ? v1 : v2;
        P[1].a[i] = sw ? a2 : a1; P[2].a[i] = sw ? a1 : a2;
        P[1].b[i] = sw ? b2 : b1; P[2].b[i] = sw ? b1 : b2;
        dir[i] = sw ? -dir[i] : dir[i];
      }
    }
    for (int it = 0;; it++) {
      if (it > 100) return false;
      mpr_support(A, Bs, dir, P[3]);
      dot = dot3(P[3].v, dir);
      if (ccd_zero(dot) || dot < 0) { SMJ_SEP_OUT() return false; }
      bool cont = false;
      cross3(va, P[1].v, P[3].v);
      dot = dot3(va, P[0].v);
      if (dot < 0 && !ccd_zero(dot)) { const MprPt q = P[3]; portal_set<2>(P, q, true); cont = true; }
      if (!cont) {
        cross3(va, P[3].v, P[2].v);
        dot = dot3(va, P[0].v);
        if (dot < 0 && !ccd_zero(dot)) { const MprPt q = P[3]; portal_set<1>(P, q, true); cont = true; }
      }
      if (!cont) break;
      for (int i = 0; i < 3; i++) { va[i] = P[1].v[i] - P[0].v[i]; vb[i] = P[2].v[i] - P[0].v[i]; }
      cross3(dir, va, vb);
      normalize3(dir);
    }
    for (int it = 0;; it++) {
      portal_dir(P, dir);
      dot = dot3(dir, P[1].v);
      if (ccd_zero(dot) || dot > 0) break;
      mpr_support(A, Bs, dir, v4);
      dot = dot3(v4.v, dir);
      if (!(ccd_zero(dot) || dot > 0)) { SMJ_SEP_OUT() return false; }
      if (reach_tolerance(P, v4, dir, tol) || it > maxit) return false;
      expand_portal(P, v4);
    }
    for (int it = 0;; it++) {
      portal_dir(P, dir);
      mpr_support(A, Bs, dir, v4);
      if (reach_tolerance(P, v4, dir, tol) || it > maxit) {
        bool inside;
        depth = sqrtf(fmaxf(0.f, origin_tri_dist2(P[1].v, P[2].v, P[3].v, pdir, &inside)));
        if (ccd_zero(pdir[0]) && ccd_zero(pdir[1]) && ccd_zero(pdir[2])) { pdir[0] = dir[0]; pdir[1] = dir[1]; pdir[2] = dir[2]; }
        normalize3(pdir);
        // fp32: the witness is a vector of the penetration's length (1e-5 .. 1e-4 m for a resting body) put together from support
        // points that carry the rounding of world coordinates (1e-7 at 1.5 m): its direction is good to 1e-3 .. 1e-2 rad -- a bottle
        // standing on a counter got a contact normal 1.6e-3 rad off the counter's face, 5e-2 rad/s^2 on its tilt where the fp64
        // oracle has 0.  Where the witness is the origin's projection INTO the portal it is parallel to the portal's normal in exact
        // arithmetic (libccd's result, unchanged), and that normal comes from centimetre-long edges: good to 1e-5 rad.  Taken when it
        // is the better conditioned of the two: the witness' direction is uncertain by (rounding) / depth, the normal's by (rounding) /
        // (the portal's smallest height) -- a sliver portal's normal is no better than the witness.
        if (SMJ_PORTAL_NORMAL && inside && dot3(pdir, dir) > 0.9995f) {
          float ea[3], eb[3], ec[3], cr[3];
          for (int i = 0; i < 3; i++) { ea[i] = P[2].v[i] - P[1].v[i]; eb[i] = P[3].v[i] - P[1].v[i]; ec[i] = P[3].v[i] - P[2].v[i]; }
          cross3(cr, ea, eb);
          const float emax2 = fmaxf(dot3(ea, ea), fmaxf(dot3(eb, eb), dot3(ec, ec)));
          if (dot3(cr, cr) > 16.f * depth * depth * emax2) { pdir[0] = dir[0]; pdir[1] = dir[1]; pdir[2] = dir[2]; }   // height > 4 x depth
        }
        float b[4], vec[3], sum;
        cross3(vec, P[1].v, P[2].v); b[0] = dot3(vec, P[3].v);
        cross3(vec, P[3].v, P[2].v); b[1] = dot3(vec, P[0].v);
        cross3(vec, P[0].v, P[1].v); b[2] = dot3(vec, P[3].v);
        cross3(vec, P[2].v, P[1].v); b[3] = dot3(vec, P[0].v);
        sum = b[0] + b[1] + b[2] + b[3];
        if (ccd_zero(sum) || sum < 0) {
          b[0] = 0;
          cross3(vec, P[2].v, P[3].v); b[1] = dot3(vec, dir);
          cross3(vec, P[3].v, P[1].v); b[2] = dot3(vec, dir);
          cross3(vec, P[1].v, P[2].v); b[3] = dot3(vec, dir);
          sum = b[1] + b[2] + b[3];
        }
        if (!(fabsf(sum) > 1e-30f)) { b[0] = 0; b[1] = b[2] = b[3] = 1; sum = 3; }   // exactly touching faces
#pragma unroll
        for (int i = 0; i < 3; i++) {   // static indices only: a dynamically indexed portal array would live in scratch memory
          const float p1 = b[0] * P[0].a[i] + b[1] * P[1].a[i] + b[2] * P[2].a[i] + b[3] * P[3].a[i];
          const float p2 = b[0] * P[0].b[i] + b[1] * P[1].b[i] + b[2] * P[2].b[i] + b[3] * P[3].b[i];
          pos[i] = 0.5f * (p1 + p2) / sum;
        }
        return true;
      }
      expand_portal(P, v4);
    }
#undef SMJ_SEP_OUT
  }

  // [MJ] mjc_SphereBox / mjc_SphereSphere: primitive pairs have closed forms and do not go through MPR.  Wave-uniform.
  // dist < 0 = penetration; dir from the sphere (first sphere) to the other geom
  SMJ_DEV bool sphere_box(const float* spos, float r, const Shape& bx, float margin, float& dist_out, float* dir, float* pos) {
    const float tmp[3] = {spos[0] - bx.pos[0], spos[1] - bx.pos[1], spos[2] - bx.pos[2]};
    float cen[3], cl[3], dif[3], nl[3], pl[3];
    mulmat3Tvec(cen, bx.mat, tmp);
    for (int i = 0; i < 3; i++) { cl[i] = fminf(bx.size[i], fmaxf(-bx.size[i], cen[i])); dif[i] = cl[i] - cen[i]; }
    const float dist = sqrtf(dot3(dif, dif));
    if (dist - r > margin) return false;
    if (dist <= SMJ_MINVAL) {   // centre inside the box: nearest face
      float closest = 2.f * fmaxf(bx.size[0], fmaxf(bx.size[1], bx.size[2]));
      int k = 0;
      for (int i = 0; i < 6; i++) {
        const float cd = fabsf(((i & 1) ? 1.f : -1.f) * bx.size[i >> 1] - cen[i >> 1]);
        if (cd < closest) { closest = cd; k = i; }
      }
      for (int i = 0; i < 3; i++) nl[i] = (i == (k >> 1)) ? ((k & 1) ? -1.f : 1.f) : 0.f;
      for (int i = 0; i < 3; i++) pl[i] = cen[i] + nl[i] * (r - closest) * 0.5f;
      dist_out = -closest - r;
    } else {
      const float inv = 1.f / dist;
      for (int i = 0; i < 3; i++) { nl[i] = dif[i] * inv; pl[i] = 0.5f * (cl[i] + cen[i] + dif[i] * (r * inv)); }
      dist_out = dist - r;
    }
    mulmat3vec(dir, bx.mat, nl);
    mulmat3vec(pos, bx.mat, pl);
    for (int i = 0; i < 3; i++) pos[i] += bx.pos[i];
    return true;
  }
  SMJ_DEV bool sphere_sphere(const float* p1, float r1, const float* p2, float r2, float margin, float& dist_out, float* dir, float* pos) {
    float dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const float cdist = sqrtf(dot3(dif, dif));
    if (cdist - r1 - r2 > margin) return false;
    if (cdist < SMJ_MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; } else { const float inv = 1.f / cdist; for (int i = 0; i < 3; i++) dif[i] *= inv; }
    dist_out = cdist - r1 - r2;
    for (int i = 0; i < 3; i++) { dir[i] = dif[i]; pos[i] = p1[i] + dif[i] * (r1 + 0.5f * dist_out); }
    return true;
  }
  // [MJ] mjc_SphereCapsule: the sphere against the nearest point of the capsule's segment (size = radius, half length; axis = z)
  SMJ_DEV bool sphere_capsule(const float* p1, float r1, const Shape& cp, float margin, float& dist_out, float* dir, float* pos) {
    const float ax[3] = {cp.mat[2], cp.mat[5], cp.mat[8]}, vec[3] = {p1[0] - cp.pos[0], p1[1] - cp.pos[1], p1[2] - cp.pos[2]};
    const float x = fmaxf(-cp.size[1], fminf(cp.size[1], dot3(ax, vec)));
    const float q[3] = {cp.pos[0] + ax[0] * x, cp.pos[1] + ax[1] * x, cp.pos[2] + ax[2] * x};
    return sphere_sphere(p1, r1, q, cp.size[0], margin, dist_out, dir, pos);
  }
  // [MJ] mjc_CapsuleCapsule: nearest points of the two segments, then sphere-sphere; parallel axes: segment ends against the
  // other segment, up to two contacts.  fp32: MuJoCo's det = ma mc - mb^2 and the numerators mc u - mb v, ma v - mb u lose all
  // their digits below an angle of 3e-4 rad between the axes; by Lagrange's identity they are |c|^2, c . (a2 x d) and
  // c . (a1 x d) with c = a1 x a2 -- the same numbers without the cancellation, good down to 1e-6 rad (the "parallel" test
  // here; MuJoCo's is an absolute 1e-15 on det).  Emits its contacts; uniform.
  SMJ_DEV void capsule_capsule(const int* r, const Shape& A, const Shape& Bs, float margin) {
    const float a1[3] = {A.mat[2] * A.size[1], A.mat[5] * A.size[1], A.mat[8] * A.size[1]};
    const float a2[3] = {Bs.mat[2] * Bs.size[1], Bs.mat[5] * Bs.size[1], Bs.mat[8] * Bs.size[1]};
    const float dif[3] = {A.pos[0] - Bs.pos[0], A.pos[1] - Bs.pos[1], A.pos[2] - Bs.pos[2]};
    const float ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
    float c[3], c1[3], c2[3];
    cross3(c, a1, a2); cross3(c2, a2, dif); cross3(c1, a1, dif);
    const float det = dot3(c, c);
    float v1[3], v2[3], depth, dir[3], pos[3];
    if (det >= 1e-12f * ma * mc) {
      float x1 = dot3(c, c2) / det, x2 = dot3(c, c1) / det;
      if (x1 > 1) { x1 = 1; x2 = (v - mb) / mc; }
      else if (x1 < -1) { x1 = -1; x2 = (v + mb) / mc; }
      if (x2 > 1) { x2 = 1; x1 = fmaxf(-1.f, fminf(1.f, (u - mb) / ma)); }
      else if (x2 < -1) { x2 = -1; x1 = fmaxf(-1.f, fminf(1.f, (u + mb) / ma)); }
      for (int i = 0; i < 3; i++) { v1[i] = A.pos[i] + a1[i] * x1; v2[i] = Bs.pos[i] + a2[i] * x2; }
      if (sphere_sphere(v1, A.size[0], v2, Bs.size[0], margin, depth, dir, pos)) add_contact(r, depth, pos, dir);
      return;
    }
    int n = 0;
    for (int e = 0; e < 2; e++) {
      const float sg = e ? -1.f : 1.f, x2 = fmaxf(-1.f, fminf(1.f, (v - sg * mb) / mc));
      for (int i = 0; i < 3; i++) { v1[i] = A.pos[i] + sg * a1[i]; v2[i] = Bs.pos[i] + a2[i] * x2; }
      if (sphere_sphere(v1, A.size[0], v2, Bs.size[0], margin, depth, dir, pos)) { add_contact(r, depth, pos, dir); n++; }
    }
    for (int e = 0; e < 2 && n < 2; e++) {
      const float sg = e ? -1.f : 1.f, x1 = fmaxf(-1.f, fminf(1.f, (u - sg * mb) / ma));
      for (int i = 0; i < 3; i++) { v1[i] = A.pos[i] + a1[i] * x1; v2[i] = Bs.pos[i] + sg * a2[i]; }
      if (sphere_sphere(v1, A.size[0], v2, Bs.size[0], margin, depth, dir, pos)) { add_contact(r, depth, pos, dir); n++; }
    }
  }
  SMJ_DEV void load_shape(Shape& sh, int g, int slot, float* cen) {
    (void)g;
    const unsigned meta = (unsigned)uni(s.u.c.meta[slot]);
    sh.type = (int)(meta & 15u); sh.nvert = (int)((meta >> 4) & 4095u);
    sh.verts = M.k_hull_vert4 + 4 * (int)(meta >> 16);
    for (int k = 0; k < 3; k++) { sh.pos[k] = uni(s.u.c.pos[slot][k]); sh.size[k] = uni(s.u.c.size[slot][k]); cen[k] = uni(s.u.c.ccen[slot][k]); }
    for (int k = 0; k < 9; k++) sh.mat[k] = uni(s.u.c.mat[slot][k]);
    if (sh.type == GT_MESH) {
      const Vec4* verts = reinterpret_cast<const Vec4*>(sh.verts);
      const int nvert = sh.nvert;
      LANES {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int id = lane + 64 * u < nvert ? lane + 64 * u : (nvert > 0 ? nvert - 1 : 0);
          const Vec4 v = verts[id];
          sh.vc[lane][3 * u] = v.x; sh.vc[lane][3 * u + 1] = v.y; sh.vc[lane][3 * u + 2] = v.z;
        }
      }
    }
  }


  // ------------------------------------------------------------------ box-box, multi-point convex contacts
  // [MJ] mjc_BoxBox and the multiccd branch of mjc_Convex, in the formulation of oracle/smj_oracle.c box_box / convex_pair
  // (see there: behaviour restated, not MuJoCo's arithmetic).  The separating-axis test is wave-uniform; the polygon of a
  // face contact is enumerated lane-parallel -- lane = candidate: 4 incident corners inside the reference face, 4 reference
  // corners inside the incident face, 16 edge crossings -- and emitted in candidate order by ballot prefix.
  // Geom frames come from the collision stage's LDS cache (s.u.c, slot = cache index): dynamically chosen boxes / axes are
  // LDS addresses, not register selects.
  SMJ_DEV void ccol(float* o, int slot, int i) const {   // axis i of the cached frame of `slot`
    o[0] = uni(s.u.c.mat[slot][i]); o[1] = uni(s.u.c.mat[slot][3 + i]); o[2] = uni(s.u.c.mat[slot][6 + i]);
  }
  SMJ_DEV void box_box(const int* rec, int slot1, int slot2, float margin) {
    float p1[3], p2[3], A[3], B[3], R1[9], R2[9];
#pragma unroll
    for (int k = 0; k < 3; k++) { p1[k] = uni(s.u.c.pos[slot1][k]); p2[k] = uni(s.u.c.pos[slot2][k]); A[k] = uni(s.u.c.size[slot1][k]); B[k] = uni(s.u.c.size[slot2][k]); }
#pragma unroll
    for (int k = 0; k < 9; k++) { R1[k] = uni(s.u.c.mat[slot1][k]); R2[k] = uni(s.u.c.mat[slot2][k]); }
    const float p[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    float R[3][3], Q[3][3], pp[3], pq[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float ai[3] = {R1[i], R1[3 + i], R1[6 + i]}, bi[3] = {R2[i], R2[3 + i], R2[6 + i]};
      pp[i] = dot3(p, ai); pq[i] = dot3(p, bi);
#pragma unroll
      for (int j = 0; j < 3; j++) { const float bj[3] = {R2[j], R2[3 + j], R2[6 + j]}; R[i][j] = dot3(ai, bj); Q[i][j] = fabsf(R[i][j]); }
    }
    float best = -3.0e38f, en[3] = {0, 0, 0};
    int code = -1;
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float sv = fabsf(pp[i]) - (A[i] + B[0] * Q[i][0] + B[1] * Q[i][1] + B[2] * Q[i][2]);
      sep = sep || sv > margin;
      if (sv > best) { best = sv; code = i; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float sv = fabsf(pq[j]) - (B[j] + A[0] * Q[0][j] + A[1] * Q[1][j] + A[2] * Q[2][j]);
      sep = sep || sv > margin;
      if (sv > best) { best = sv; code = 3 + j; }
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const float l = sqrtf(fmaxf(0.f, 1.f - R[i][j] * R[i][j]));
        // (parallel edges are covered by the face axes.  fp32: 1 - R^2 resolves 6e-8, i.e. l = 2.4e-4 -- the z axes of two boxes
        // lying flat give R = 1 - 1 ulp as often as 1, and at l = 3.4e-4 e and the radii below are rounding: the axis "separated"
        // the boxes by ~0 and won over the face axis with ONE bogus point.  Edges within 0.11 degrees count as parallel.)
        if (l >= 2e-3f) {
          const float e = pp[i2] * R[i1][j] - pp[i1] * R[i2][j];
          const float sv = (fabsf(e) - (A[i1] * Q[i2][j] + A[i2] * Q[i1][j] + B[j1] * Q[i][j2] + B[j2] * Q[i][j1])) / l;
          sep = sep || sv > margin;
          if (sv * 1.05f > best && sv > best) {
            best = sv; code = 6 + 3 * i + j;
            const float ai[3] = {R1[i], R1[3 + i], R1[6 + i]}, bj[3] = {R2[j], R2[3 + j], R2[6 + j]};
            cross3(en, ai, bj);
            for (int k = 0; k < 3; k++) en[k] /= l;
          }
        }
      }
#ifdef SMJ_EMUL
    if (getenv("SMJ_BOXBOX_TRACE")) { fprintf(stderr, "box_box: code %d best %.9g sep %d R:", code, best, (int)sep); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) fprintf(stderr, " %.9g", R[i][j]); fprintf(stderr, " pp %.9g %.9g %.9g\n", pp[0], pp[1], pp[2]); }
#endif
    if (sep) return;
    if (code >= 6) {   // edge-edge: one point, midway between the closest points of the two edges
      float n[3] = {en[0], en[1], en[2]};
      if (dot3(n, p) < 0) for (int k = 0; k < 3; k++) n[k] = -n[k];
      const int i = (code - 6) / 3, j = (code - 6) % 3;
      float pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float ak[3] = {R1[k], R1[3 + k], R1[6 + k]}, bk[3] = {R2[k], R2[3 + k], R2[6 + k]};
        const float sa = dot3(n, ak) > 0 ? 1.f : -1.f, sb = dot3(n, bk) > 0 ? -1.f : 1.f;
        for (int x = 0; x < 3; x++) { pa[x] += sa * A[k] * ak[x]; pb[x] += sb * B[k] * bk[x]; }
      }
      float ua[3], ub[3];
      ccol(ua, slot1, i); ccol(ub, slot2, j);
      const float dd[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
      const float uaub = dot3(ua, ub), q1 = dot3(ua, dd), q2 = -dot3(ub, dd), den = 1.f - uaub * uaub;
      float al = 0, be = 0;
      if (den > 1e-12f) { al = (q1 + uaub * q2) / den; be = (uaub * q1 + q2) / den; }
      float pos[3];
      for (int x = 0; x < 3; x++) pos[x] = 0.5f * ((pa[x] + al * ua[x]) + (pb[x] + be * ub[x]));
      add_contact(rec, best, pos, n);
      return;
    }
    // face contact.  Reference box a (the one owning the axis), incident box b; n from a to b
    const bool swap = code >= 3;
    const int ia = swap ? code - 3 : code, sa = swap ? slot2 : slot1, sb = swap ? slot1 : slot2;
    float pa[3], pb[3], n[3], u[3], v[3];
    for (int k = 0; k < 3; k++) { pa[k] = uni(s.u.c.pos[sa][k]); pb[k] = uni(s.u.c.pos[sb][k]); }
    ccol(n, sa, ia);
    const float ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    if (dot3(n, ab) < 0) for (int k = 0; k < 3; k++) n[k] = -n[k];
    const int ja = (ia + 1) % 3, ka = (ia + 2) % 3;
    ccol(u, sa, ja); ccol(v, sa, ka);
    const float hu = uni(s.u.c.size[sa][ja]), hv = uni(s.u.c.size[sa][ka]), hn = uni(s.u.c.size[sa][ia]);
    float cA[3];
    for (int k = 0; k < 3; k++) cA[k] = pa[k] + n[k] * hn;
    int ib = 0;
    float bd = -1.f;
#pragma unroll
    for (int k = 0; k < 3; k++) { float bk[3]; ccol(bk, sb, k); const float t = fabsf(dot3(bk, n)); if (t > bd) { bd = t; ib = k; } }
    float nb[3], pv[3], qv[3];
    ccol(nb, sb, ib);
    if (dot3(nb, n) > 0) for (int k = 0; k < 3; k++) nb[k] = -nb[k];
    const int jb = (ib + 1) % 3, kb = (ib + 2) % 3;
    ccol(pv, sb, jb); ccol(qv, sb, kb);
    const float hp = uni(s.u.c.size[sb][jb]), hq = uni(s.u.c.size[sb][kb]), hbn = uni(s.u.c.size[sb][ib]);
    float cB[3];
    for (int k = 0; k < 3; k++) cB[k] = pb[k] + nb[k] * hbn;
    const float nnb = dot3(n, nb);
    PL<int> okv;
    PL<float> dv, px, py, pz;
    LANES {
      const int cand = lane;
      int ok = 0;
      float x[3] = {0, 0, 0};
      // corner c of a face, counter-clockwise: (-,-) (+,-) (+,+) (-,+)
#define SG0(c) (((c) == 1 || (c) == 2) ? 1.f : -1.f)
#define SG1(c) ((c) >= 2 ? 1.f : -1.f)
      if (cand < 4) {
        float d[3];
        for (int k = 0; k < 3; k++) { x[k] = cB[k] + SG0(cand) * hp * pv[k] + SG1(cand) * hq * qv[k]; d[k] = x[k] - cA[k]; }
        ok = fabsf(dot3(d, u)) <= hu * (1.f + 2e-6f) && fabsf(dot3(d, v)) <= hv * (1.f + 2e-6f);   // (a corner ON the reference face's edge -- boxes of equal size stacked in line -- is inside)
      } else if (cand < 8) {
        const int c = cand - 4;
        float r[3], d[3];
        for (int k = 0; k < 3; k++) { r[k] = cA[k] + SG0(c) * hu * u[k] + SG1(c) * hv * v[k]; d[k] = cB[k] - r[k]; }
        const float t = dot3(d, nb) / nnb;
        for (int k = 0; k < 3; k++) { x[k] = r[k] + t * n[k]; d[k] = x[k] - cB[k]; }
        ok = fabsf(dot3(d, pv)) <= hp * (1.f + 2e-6f) && fabsf(dot3(d, qv)) <= hq * (1.f + 2e-6f);   // (inclusive, as the incident corners: coincident edges)
      } else if (cand < 24) {
        const int e = (cand - 8) >> 2, r = (cand - 8) & 3, c0 = e, c1 = (e + 1) & 3;
        float w0[3], w1[3], d0[3], d1[3];
        for (int k = 0; k < 3; k++) {
          w0[k] = cB[k] + SG0(c0) * hp * pv[k] + SG1(c0) * hq * qv[k]; w1[k] = cB[k] + SG0(c1) * hp * pv[k] + SG1(c1) * hq * qv[k];
          d0[k] = w0[k] - cA[k]; d1[k] = w1[k] - cA[k];
        }
        const float u0 = dot3(d0, u), v0 = dot3(d0, v), u1 = dot3(d1, u), v1 = dot3(d1, v);
        const bool along_u = r < 2;
        const float lim = ((r & 1) ? 1.f : -1.f) * (along_u ? hu : hv);
        const float a0 = along_u ? u0 : v0, a1 = along_u ? u1 : v1, b0 = along_u ? v0 : u0, b1 = along_u ? v1 : u1;
        const float den = a1 - a0;
        if (fabsf(den) > 1e-12f) {
          const float t = (lim - a0) / den, o = b0 + t * (b1 - b0);
          ok = t > 0 && t < 1 && fabsf(o) < (along_u ? hv : hu);
          for (int k = 0; k < 3; k++) x[k] = w0[k] + t * (w1[k] - w0[k]);
        }
      }
#undef SG0
#undef SG1
      const float d[3] = {x[0] - cA[0], x[1] - cA[1], x[2] - cA[2]};
      const float depth = -dot3(d, n);
      if (-depth > margin) ok = 0;
      okv[lane] = ok; dv[lane] = -depth;
      px[lane] = x[0] + 0.5f * depth * n[0]; py[lane] = x[1] + 0.5f * depth * n[1]; pz[lane] = x[2] + 0.5f * depth * n[2];
    }
    uint64_t mask = wave_ballot(okv);
    if (mask == 0) return;
    // candidates that coincide (a corner of one face ON an edge or a corner of the other: boxes of equal size stacked in line) are one
    // point: the later one goes, before the count is taken (within 1e-5 m)
    {
      PL<int> dup;
      LANES {
        int dd = 0;
        if (okv[lane])
          for (int c2 = 0; c2 < 24; c2++) {
            const float ox = wave_read(px, c2), oy = wave_read(py, c2), oz = wave_read(pz, c2);
            if (c2 < lane && ((mask >> c2) & 1) && fabsf(ox - px[lane]) < 1e-5f && fabsf(oy - py[lane]) < 1e-5f && fabsf(oz - pz[lane]) < 1e-5f) dd = 1;
          }
        dup[lane] = dd;
      }
      mask &= ~wave_ballot(dup);
      LANES { okv[lane] = (int)((mask >> lane) & 1); }
    }
    const int maxcon = M.max_con_pair < 4 ? 4 : M.max_con_pair;
    if (popc64(mask) > maxcon) {
      // more points than max_contacts_per_pair: keep the extreme ones along the two axes of the reference face (ties: lowest
      // candidate), a support polygon as wide as the full one
      PL<float> ku, kv, kk;
      PL<int> li;
      LANES {
        const float pos[3] = {px[lane], py[lane], pz[lane]};
        const float d[3] = {pos[0] - cA[0], pos[1] - cA[1], pos[2] - cA[2]};   // (the midpoint shift is along n: same u, v)
        ku[lane] = okv[lane] ? dot3(d, u) : 3.0e38f; kv[lane] = okv[lane] ? dot3(d, v) : 3.0e38f; li[lane] = okv[lane] ? lane : -1;
      }
      uint64_t keep = 0;
      keep |= 1ull << pick_index(ku, li, wave_min(ku));
      keep |= 1ull << pick_index(kv, li, wave_min(kv));
      LANES { kk[lane] = okv[lane] ? -ku[lane] : 3.0e38f; }
      keep |= 1ull << pick_index(kk, li, wave_min(kk));
      LANES { kk[lane] = okv[lane] ? -kv[lane] : 3.0e38f; }
      keep |= 1ull << pick_index(kk, li, wave_min(kk));
      // (one point can be extreme in two directions: the places left go to the remaining candidates, lowest first)
      uint64_t rest = mask & ~keep;
      while (popc64(keep) < maxcon && rest) { keep |= rest & (~rest + 1); rest &= rest - 1; }
      mask &= keep;
      LANES { okv[lane] = (int)((mask >> lane) & 1); }
    }
    const int total = popc64(mask) < 8 ? popc64(mask) : 8;
    const float cn[3] = {swap ? -n[0] : n[0], swap ? -n[1] : n[1], swap ? -n[2] : n[2]};
    // (side by side with the other wavefront the slots are claimed first, and a batch that does not fit is not written at all)
    const bool fits = (SMJ_SPLIT_COLLIDE && split_on) ? con_claim(total) : ncon + total <= NCON;
    LANES {
      if (okv[lane] && (fits || !(SMJ_SPLIT_COLLIDE && split_on))) {
        const int k = popc64(mask & ((1ull << lane) - 1));
        if (k < 8 && ncon + k < NCON) {
          const float pos[3] = {px[lane], py[lane], pz[lane]};
          write_contact(ncon + k, rec, dv[lane], pos, cn);
        }
      }
    }
    if (!fits) { flags |= SMJ_FLAG_CON_OVERFLOW; if (!(SMJ_SPLIT_COLLIDE && split_on)) ncon = NCON; }   // (side by side nothing was written: the list stays the valid prefix it is)
    else ncon += total;
  }

  // rotation by +-1e-3 rad about a unit axis.  The angle is multiccd's constant, so cos / sin / 1 - cos are literals (fp64 values
  // rounded once: 1 - cosf(1e-3f) evaluated in fp32 would be off by 5 %), not two libm calls per matrix
  SMJ_DEV static void axis_angle_mat(float* Rm, const float* ax, bool negative) {
    const float c = 0.9999995000000417f, sn = negative ? -9.999998333333417e-4f : 9.999998333333417e-4f, t = 4.999999583333347e-7f, x = ax[0], y = ax[1], z = ax[2];
    Rm[0] = t * x * x + c; Rm[1] = t * x * y - sn * z; Rm[2] = t * x * z + sn * y;
    Rm[3] = t * x * y + sn * z; Rm[4] = t * y * y + c; Rm[5] = t * y * z - sn * x;
    Rm[6] = t * x * z - sn * y; Rm[7] = t * y * z + sn * x; Rm[8] = t * z * z + c;
  }
  SMJ_DEV static void rotate_point(float* q, const float* c, const float* Rm) {
    const float d[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
    float r[3];
    mulmat3vec(r, Rm, d);
    for (int k = 0; k < 3; k++) q[k] = c[k] + r[k];
  }
  // multiccd: the two geoms counter-rotated by +-1e-3 rad about the two tangent axes through the first contact point, the
  // penetration query repeated; new points farther than 1e-3 x min(rbound) from the earlier ones join the manifold, which
  // shares the first normal.  A, Bs are modified in place (the pair is done afterwards).
  SMJ_DEV void convex_multi(const int* rec, Shape& A, Shape& Bs, int slotA, int slotB, const float* pos0, const float* dir0,
                            float margin, float tol, const float* org) {
    const float p0l[3] = {pos0[0] - org[0], pos0[1] - org[1], pos0[2] - org[2]};   // (the first contact in the query's local frame, narrow_pair_run)
    float fr[9] = {dir0[0], dir0[1], dir0[2], 0, 0, 0, 0, 0, 0};
    make_frame(fr);
    LANES { if (lane == 0) for (int k = 0; k < 3; k++) mcs()[0][k] = pos0[k]; }
    SYNC();
    int n = 1;
#pragma nounroll
    for (int q = 0; q < 4; q++) {
      const float* ax = fr + 3 * (1 + (q >> 1));
      const float axv[3] = {uni(ax[0]), uni(ax[1]), uni(ax[2])};
      const bool neg = (q & 1) != 0;
      // the unrotated poses and MPR interior points come from the LDS geom cache every round (no register copies kept)
      float Rp[9], Rn[9], ca[3], cb[3], mA[9], mB[9];
      axis_angle_mat(Rp, axv, neg); axis_angle_mat(Rn, axv, !neg);
      for (int k = 0; k < 3; k++) {
        A.pos[k] = uni(s.u.c.pos[slotA][k]) - org[k]; Bs.pos[k] = uni(s.u.c.pos[slotB][k]) - org[k]; ca[k] = uni(s.u.c.ccen[slotA][k]) - org[k]; cb[k] = uni(s.u.c.ccen[slotB][k]) - org[k];
      }
      for (int k = 0; k < 9; k++) { mA[k] = uni(s.u.c.mat[slotA][k]); mB[k] = uni(s.u.c.mat[slotB][k]); }
      rotate_point(A.pos, p0l, Rp); rotate_point(Bs.pos, p0l, Rn); rotate_point(ca, p0l, Rp); rotate_point(cb, p0l, Rn);
      mulmat3(A.mat, Rp, mA); mulmat3(Bs.mat, Rn, mB);
      float dp, dr[3], ps[3];
      if (!mpr_penetration(A, Bs, ca, cb, dp, dr, ps)) continue;
      if (-dp > margin || dot3(dr, dr) < 0.5f) continue;
      for (int k = 0; k < 3; k++) ps[k] += org[k];
      bool dup = false;
#pragma nounroll
      for (int k = 0; k < n; k++) {
        const float e[3] = {ps[0] - uni(mcs()[k][0]), ps[1] - uni(mcs()[k][1]), ps[2] - uni(mcs()[k][2])};
        dup = dup || dot3(e, e) < tol * tol;
      }
      if (dup) continue;
      LANES { if (lane == 0) for (int k = 0; k < 3; k++) mcs()[n][k] = ps[k]; }
      SYNC();
      n++;
      add_contact(rec, -dp, ps, dir0);
    }
  }

  // ---- multiccd, four queries at once.  The four counter-rotated penetration queries of a pair are independent MPR runs on
  // the same two shapes: run serially they are four times ~30 k cycles of wave-uniform scalar code with the vector lanes idle
  // (measured: 0.38 hits per env-step under random actions, 5 MPR runs per hit = most of the collision stage).  Here each
  // 16-lane row of the wave owns one query: the MPR state is per lane (identical within a row), the control flow of
  // mpr_penetration becomes a small state machine that advances one support point per round, and the hull support scan
  // evaluates every lane's four register-resident vertices against the four rows' directions at once (four arg-max
  // reductions per shape and round instead of sixteen dependent ones per query).  Same arithmetic per query as
  // mpr_penetration / shape_support, same tie-breaks: the contacts are those of the serial code.
  struct MprLane {
    MprPt P[4], v4;
    float dir[3], apos[3], amat[9], bpos[3], bmat[9];
    int phase, it, ok;
    float depth, pdir[3], pos[3];
  };
  enum { MQ_W1 = 0, MQ_W2, MQ_W3, MQ_R1, MQ_R2, MQ_DONE };
  // support points of `sh` for every lane's own frame (pos, mat: per lane, equal within a row) and direction
  SMJ_DEV void shape_support4(const Shape& sh, PL<MprLane>& st, bool second, PL<float[3]>& out) {
    if (sh.type != GT_MESH) {
      LANES {
        MprLane& m = st[lane];
        const float* mat = second ? m.bmat : m.amat;
        const float* ps = second ? m.bpos : m.apos;
        const float dir[3] = {second ? -m.dir[0] : m.dir[0], second ? -m.dir[1] : m.dir[1], second ? -m.dir[2] : m.dir[2]};
        float dl[3], pl[3];
        mulmat3Tvec(dl, mat, dir);
        prim_support(sh.type, sh.size, dl, pl);
        float o[3];
        mulmat3vec(o, mat, pl);
        for (int i = 0; i < 3; i++) out[lane][i] = o[i] + ps[i];
      }
      return;
    }
    // the four rows' directions in the hull frame, broadcast to the wave
    PL<float> dlx, dly, dlz;
    LANES {
      const MprLane& m = st[lane];
      const float dir[3] = {second ? -m.dir[0] : m.dir[0], second ? -m.dir[1] : m.dir[1], second ? -m.dir[2] : m.dir[2]};
      float dl[3];
      mulmat3Tvec(dl, second ? m.bmat : m.amat, dir);
      dlx[lane] = dl[0]; dly[lane] = dl[1]; dlz[lane] = dl[2];
    }
    float dq[4][3];
#pragma unroll
    for (int q = 0; q < 4; q++) { dq[q][0] = wave_read(dlx, 16 * q); dq[q][1] = wave_read(dly, 16 * q); dq[q][2] = wave_read(dlz, 16 * q); }
    PL<float> best[4], bx[4], by[4], bz[4];
    PL<int> bidx[4];
    const Vec4* verts = reinterpret_cast<const Vec4*>(sh.verts);
    const int nvert = sh.nvert;
    LANES {
      float bd[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, b[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      int bi[4] = {-1, -1, -1, -1};
#pragma unroll
      for (int u = 0; u < 4; u++) {   // register-resident vertices
        const int id = lane + 64 * u;
        const float x = sh.vc[lane][3 * u], y = sh.vc[lane][3 * u + 1], z = sh.vc[lane][3 * u + 2];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float d = x * dq[q][0] + y * dq[q][1] + z * dq[q][2];
          if (id < nvert && d > bd[q]) { bd[q] = d; bi[q] = id; b[q][0] = x; b[q][1] = y; b[q][2] = z; }
        }
      }
      for (int i0 = 256 + lane; i0 < nvert; i0 += 256) {   // larger hulls: the rest from memory
        Vec4 v[4];
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { id[u] = i0 + 64 * u < nvert ? i0 + 64 * u : nvert - 1; v[u] = verts[id[u]]; }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float d = v[u].x * dq[q][0] + v[u].y * dq[q][1] + v[u].z * dq[q][2];
            if (d > bd[q]) { bd[q] = d; bi[q] = id[u]; b[q][0] = v[u].x; b[q][1] = v[u].y; b[q][2] = v[u].z; }
          }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { best[q][lane] = bd[q]; bidx[q][lane] = bi[q]; bx[q][lane] = b[q][0]; by[q][lane] = b[q][1]; bz[q][lane] = b[q][2]; }
    }
    float plq[4][3];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float mx = wave_max(best[q]);
      PL<int> ism;
      LANES { ism[lane] = best[q][lane] == mx && bidx[q][lane] >= 0; }
      const uint64_t mm = wave_ballot(ism);
      const int owner = popc64(mm) == 1 ? ffs64(mm) : (pick_index(best[q], bidx[q], mx) & 63);
      plq[q][0] = wave_read(bx[q], owner); plq[q][1] = wave_read(by[q], owner); plq[q][2] = wave_read(bz[q], owner);
    }
    LANES {
      const MprLane& m = st[lane];
      const int q = lane >> 4;
      const float pl[3] = {q == 0 ? plq[0][0] : q == 1 ? plq[1][0] : q == 2 ? plq[2][0] : plq[3][0],
                           q == 0 ? plq[0][1] : q == 1 ? plq[1][1] : q == 2 ? plq[2][1] : plq[3][1],
                           q == 0 ? plq[0][2] : q == 1 ? plq[1][2] : q == 2 ? plq[2][2] : plq[3][2]};
      float o[3];
      mulmat3vec(o, second ? m.bmat : m.amat, pl);
      const float* ps = second ? m.bpos : m.apos;
      for (int i = 0; i < 3; i++) out[lane][i] = o[i] + ps[i];
    }
  }
  // the end of mpr_penetration: depth, direction and position from the final portal (per lane)
  SMJ_DEV static void mpr_finish(MprLane& m) {
    m.depth = sqrtf(fmaxf(0.f, origin_tri_dist2(m.P[1].v, m.P[2].v, m.P[3].v, m.pdir)));
    if (ccd_zero(m.pdir[0]) && ccd_zero(m.pdir[1]) && ccd_zero(m.pdir[2])) { m.pdir[0] = m.dir[0]; m.pdir[1] = m.dir[1]; m.pdir[2] = m.dir[2]; }
    normalize3(m.pdir);
    float b[4], vec[3], sum;
    const MprPt* P = m.P;
    cross3(vec, P[1].v, P[2].v); b[0] = dot3(vec, P[3].v);
    cross3(vec, P[3].v, P[2].v); b[1] = dot3(vec, P[0].v);
    cross3(vec, P[0].v, P[1].v); b[2] = dot3(vec, P[3].v);
    cross3(vec, P[2].v, P[1].v); b[3] = dot3(vec, P[0].v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (ccd_zero(sum) || sum < 0) {
      b[0] = 0;
      cross3(vec, P[2].v, P[3].v); b[1] = dot3(vec, m.dir);
      cross3(vec, P[3].v, P[1].v); b[2] = dot3(vec, m.dir);
      cross3(vec, P[1].v, P[2].v); b[3] = dot3(vec, m.dir);
      sum = b[1] + b[2] + b[3];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float p1 = b[0] * P[0].a[i] + b[1] * P[1].a[i] + b[2] * P[2].a[i] + b[3] * P[3].a[i];
      const float p2 = b[0] * P[0].b[i] + b[1] * P[1].b[i] + b[2] * P[2].b[i] + b[3] * P[3].b[i];
      m.pos[i] = 0.5f * (p1 + p2) / sum;
    }
  }
  SMJ_DEV int convex_multi4(const int* rec, const Shape& A, const Shape& Bs, int slotA, int slotB, const float* pos0, const float* dir0,
                             float margin, float tol, const float* org) {
    const float p0l[3] = {pos0[0] - org[0], pos0[1] - org[1], pos0[2] - org[2]};   // (the first contact in the query's local frame, narrow_pair_run)
    float fr[9] = {dir0[0], dir0[1], dir0[2], 0, 0, 0, 0, 0, 0};
    make_frame(fr);
    const float tolm = 1e-6f;
    const int maxit = 50;
    PL<MprLane> st;
    LANES {
      MprLane& m = st[lane];
      const int q = lane >> 4;
      const float* ax = fr + 3 * (1 + (q >> 1));
      const float axv[3] = {ax[0], ax[1], ax[2]};
      const bool neg = (q & 1) != 0;
      float Rp[9], Rn[9], ca[3], cb[3], mA[9], mB[9];
      axis_angle_mat(Rp, axv, neg); axis_angle_mat(Rn, axv, !neg);
      for (int k = 0; k < 3; k++) { m.apos[k] = s.u.c.pos[slotA][k] - org[k]; m.bpos[k] = s.u.c.pos[slotB][k] - org[k]; ca[k] = s.u.c.ccen[slotA][k] - org[k]; cb[k] = s.u.c.ccen[slotB][k] - org[k]; }
      for (int k = 0; k < 9; k++) { mA[k] = s.u.c.mat[slotA][k]; mB[k] = s.u.c.mat[slotB][k]; }
      rotate_point(m.apos, p0l, Rp); rotate_point(m.bpos, p0l, Rn); rotate_point(ca, p0l, Rp); rotate_point(cb, p0l, Rn);
      mulmat3(m.amat, Rp, mA); mulmat3(m.bmat, Rn, mB);
      // mpr_penetration up to its first support query
      for (int i = 0; i < 3; i++) { m.P[0].a[i] = ca[i]; m.P[0].b[i] = cb[i]; m.P[0].v[i] = ca[i] - cb[i]; }
      if (ccd_zero(m.P[0].v[0]) && ccd_zero(m.P[0].v[1]) && ccd_zero(m.P[0].v[2])) m.P[0].v[0] += 10.f * CCD_EPS;
      for (int i = 0; i < 3; i++) m.dir[i] = -m.P[0].v[i];
      normalize3(m.dir);
      m.phase = MQ_W1; m.it = 0; m.ok = 0; m.depth = 0;
      for (int i = 0; i < 3; i++) { m.pdir[i] = 0; m.pos[i] = 0; }
    }
    int round = 0;
    for (; round < 256; round++) {
      // what each query needs before its next support point (loop heads of mpr_penetration)
      PL<int> need;
      LANES {
        MprLane& m = st[lane];
        if (m.phase == MQ_W3 && m.it > 100) m.phase = MQ_DONE;
        if (m.phase == MQ_R1 || m.phase == MQ_R2) {   // both refinement loops start from the portal's normal
          portal_dir(m.P, m.dir);
          const float dot = dot3(m.dir, m.P[1].v);
          if (m.phase == MQ_R1 && (ccd_zero(dot) || dot > 0)) { m.phase = MQ_R2; m.it = 0; }
        }
        need[lane] = m.phase != MQ_DONE;
      }
      if (wave_ballot(need) == 0) break;
      PL<float[3]> sa, sb;
      shape_support4(A, st, false, sa);
      shape_support4(Bs, st, true, sb);
      LANES {
        MprLane& m = st[lane];
        MprPt np;
        for (int i = 0; i < 3; i++) { np.a[i] = sa[lane][i]; np.b[i] = sb[lane][i]; np.v[i] = np.a[i] - np.b[i]; }
        float va[3], vb[3];
        if (m.phase == MQ_W1) {
          m.P[1] = np;
          float dot = dot3(m.P[1].v, m.dir);
          if (ccd_zero(dot) || dot < 0) m.phase = MQ_DONE;
          else {
            cross3(m.dir, m.P[0].v, m.P[1].v);
            if (ccd_zero(dot3(m.dir, m.dir))) {
              if (ccd_zero(m.P[1].v[0]) && ccd_zero(m.P[1].v[1]) && ccd_zero(m.P[1].v[2])) { m.depth = 0; m.pdir[0] = m.pdir[1] = m.pdir[2] = 0; }
              else {
                m.depth = sqrtf(dot3(m.P[1].v, m.P[1].v));
                for (int i = 0; i < 3; i++) m.pdir[i] = m.P[1].v[i];
                normalize3(m.pdir);
              }
              for (int i = 0; i < 3; i++) m.pos[i] = 0.5f * (m.P[1].a[i] + m.P[1].b[i]);
              m.ok = 1; m.phase = MQ_DONE;
            } else { normalize3(m.dir); m.phase = MQ_W2; }
          }
        } else if (m.phase == MQ_W2) {
          m.P[2] = np;
          const float dot = dot3(m.P[2].v, m.dir);
          if (ccd_zero(dot) || dot < 0) m.phase = MQ_DONE;
          else {
            for (int i = 0; i < 3; i++) { va[i] = m.P[1].v[i] - m.P[0].v[i]; vb[i] = m.P[2].v[i] - m.P[0].v[i]; }
            cross3(m.dir, va, vb);
            normalize3(m.dir);
            const bool sw = dot3(m.dir, m.P[0].v) > 0;
#pragma unroll
            for (int i = 0; i < 3; i++) {
              const float v1 = m.P[1].v[i], v2 = m.P[2].v[i], a1 = m.P[1].a[i], a2 = m.P[2].a[i], b1 = m.P[1].b[i], b2 = m.P[2].b[i];
              m.P[1].v[i] = sw ? v2 : v1; m.P[2].v[i] = sw ? v1 : v2;
              m.P[1].a[i] = sw ? a2 : a1; m.P[2].a[i] = sw ? a1 : a2;
              m.P[1].b[i] = sw ? b2 : b1; m.P[2].b[i] = sw ? b1 : b2;
              m.dir[i] = sw ? -m.dir[i] : m.dir[i];
            }
            m.phase = MQ_W3; m.it = 0;
          }
        } else if (m.phase == MQ_W3) {
          m.P[3] = np;
          float dot = dot3(m.P[3].v, m.dir);
          if (ccd_zero(dot) || dot < 0) m.phase = MQ_DONE;
          else {
            bool cont = false;
            cross3(va, m.P[1].v, m.P[3].v);
            dot = dot3(va, m.P[0].v);
            if (dot < 0 && !ccd_zero(dot)) { const MprPt q = m.P[3]; portal_set<2>(m.P, q, true); cont = true; }
            if (!cont) {
              cross3(va, m.P[3].v, m.P[2].v);
              dot = dot3(va, m.P[0].v);
              if (dot < 0 && !ccd_zero(dot)) { const MprPt q = m.P[3]; portal_set<1>(m.P, q, true); cont = true; }
            }
            if (cont) {
              for (int i = 0; i < 3; i++) { va[i] = m.P[1].v[i] - m.P[0].v[i]; vb[i] = m.P[2].v[i] - m.P[0].v[i]; }
              cross3(m.dir, va, vb);
              normalize3(m.dir);
              m.it++;
            } else { m.phase = MQ_R1; m.it = 0; }
          }
        } else if (m.phase == MQ_R1 || m.phase == MQ_R2) {
          // one body for the two refinement loops: the first also leaves when the new point is behind the origin (no
          // penetration), the second leaves with a result -- computed after the rounds, once for all four queries
          m.v4 = np;
          const float dot = dot3(m.v4.v, m.dir);
          const bool behind = m.phase == MQ_R1 && !(ccd_zero(dot) || dot > 0);
          if (behind || reach_tolerance(m.P, m.v4, m.dir, tolm) || m.it > maxit) { m.ok = m.phase == MQ_R2 ? 2 : 0; m.phase = MQ_DONE; }
          else { expand_portal(m.P, m.v4); m.it++; }
        }
      }
    }
    LANES {
      MprLane& m = st[lane];
      if (m.ok == 2) { mpr_finish(m); m.ok = 1; }
    }
    // the results in query order: the manifold's duplicate test is sequential (convex_multi)
    PL<int> okp;
    PL<float> dpp, drx, dry, drz, psx, psy, psz;
    LANES {
      const MprLane& m = st[lane];
      okp[lane] = m.ok; dpp[lane] = m.depth; drx[lane] = m.pdir[0]; dry[lane] = m.pdir[1]; drz[lane] = m.pdir[2];
      psx[lane] = m.pos[0]; psy[lane] = m.pos[1]; psz[lane] = m.pos[2];
    }
    LANES { if (lane == 0) for (int k = 0; k < 3; k++) mcs()[0][k] = pos0[k]; }
    SYNC();
    int n = 1;
#pragma nounroll
    for (int q = 0; q < 4; q++) {
      const int l = 16 * q;
      if (!wave_read(okp, l)) continue;
      const float dp = wave_read(dpp, l);
      const float dr[3] = {wave_read(drx, l), wave_read(dry, l), wave_read(drz, l)}, ps[3] = {wave_read(psx, l) + org[0], wave_read(psy, l) + org[1], wave_read(psz, l) + org[2]};
      if (-dp > margin || dot3(dr, dr) < 0.5f) continue;
      bool dup = false;
#pragma nounroll
      for (int k = 0; k < n; k++) {
        const float e[3] = {ps[0] - uni(mcs()[k][0]), ps[1] - uni(mcs()[k][1]), ps[2] - uni(mcs()[k][2])};
        dup = dup || dot3(e, e) < tol * tol;
      }
      if (dup) continue;
      LANES { if (lane == 0) for (int k = 0; k < 3; k++) mcs()[n][k] = ps[k]; }
      SYNC();
      n++;
      add_contact(rec, -dp, ps, dir0);
    }
    return round;
  }

  // narrowphase of one convex pair: record r (DevModel::k_cprec / k_sprec), s1 / s2 = the geoms' slots in the collision stage's
  // LDS cache.  sepslot / septag: the pair's entry of the separating-direction cache (or null), sep_hit: it holds this pair's
  // direction sd.
  // motion of a body since a manifold was stored: translation dx and small rotation vector dth (= 2 vec(q1 conj(q0)), q = w x y z)
  SMJ_DEV static void smj_mc_motion(const float* x0, const float* q0, const float* x1, const float* q1, float* dx, float* dth) {
    for (int k = 0; k < 3; k++) dx[k] = x1[k] - x0[k];
    const float sw = q1[0] * q0[0] + q1[1] * q0[1] + q1[2] * q0[2] + q1[3] * q0[3];
    const float sg = sw < 0.f ? -2.f : 2.f;
    dth[0] = sg * (q0[0] * q1[1] - q1[0] * q0[1] - (q1[2] * q0[3] - q1[3] * q0[2]));
    dth[1] = sg * (q0[0] * q1[2] - q1[0] * q0[2] - (q1[3] * q0[1] - q1[1] * q0[3]));
    dth[2] = sg * (q0[0] * q1[3] - q1[0] * q0[3] - (q1[1] * q0[2] - q1[2] * q0[1]));
  }
  // a stored contact (dist, p, normal n from geom 1 to geom 2) carried along by the two bodies: u_b(p) = dx_b + dth_b x (p - x_b)
  SMJ_DEV static void smj_mc_carry(const float* xa, const float* dxa, const float* dta, const float* xb, const float* dxb, const float* dtb, const float* n,
                                   float& dist, float* p) {
    const float ra[3] = {p[0] - xa[0], p[1] - xa[1], p[2] - xa[2]}, rb[3] = {p[0] - xb[0], p[1] - xb[1], p[2] - xb[2]};
    const float ua[3] = {dxa[0] + dta[1] * ra[2] - dta[2] * ra[1], dxa[1] + dta[2] * ra[0] - dta[0] * ra[2], dxa[2] + dta[0] * ra[1] - dta[1] * ra[0]};
    const float ub[3] = {dxb[0] + dtb[1] * rb[2] - dtb[2] * rb[1], dxb[1] + dtb[2] * rb[0] - dtb[0] * rb[2], dxb[2] + dtb[0] * rb[1] - dtb[1] * rb[0]};
    dist += n[0] * (ub[0] - ua[0]) + n[1] * (ub[1] - ua[1]) + n[2] * (ub[2] - ua[2]);
    for (int k = 0; k < 3; k++) p[k] += 0.5f * (ua[k] + ub[k]);
  }
  // (moving-moving pairs in the lower half of the env's slots, pairs with the static world -- tag bit 30 -- in the upper half: the two
  // groups never evict each other, so the cache's contents do not depend on the order the two groups are worked in)
  SMJ_DEV float* mc_entry(int tag) const {
    const unsigned slot = (((unsigned)tag >> 30) & 1u) << SMJ_MC_LOG2 | (((unsigned)tag * 2654435761u) >> (32 - SMJ_MC_LOG2));
    return S.mcache + ((size_t)env * SMJ_MC_SLOTS + slot) * SMJ_MC_WORDS;
  }
  SMJ_DEV void narrow_pair(const int* r, int s1, int s2, float* sepslot, int septag, bool sep_hit, const float* sd, float* pc, bool prof, bool lookup = true) {
    const int g1 = uni(r[SMJ_CP_G1]), g2 = uni(r[SMJ_CP_G2]);
    // the pair's stored manifold (DevState::mcache): valid while neither body has moved
    const int ta0 = uni(s.u.c.meta[s1]) & 15, tb0 = uni(s.u.c.meta[s2]) & 15;
    float* mc = nullptr;
    const int ncon0 = ncon;
    // (not for capsule against capsule: a closed form, and its two contacts of the parallel case have normals of their own -- an entry keeps one)
    if (S.mcache && M.manifold_cache && ta0 != GT_SPHERE && tb0 != GT_SPHERE && !(ta0 == GT_CAPSULE && tb0 == GT_CAPSULE)) {
      mc = mc_entry(septag);
      const int b1 = uni(r[SMJ_CP_B1]), b2 = uni(r[SMJ_CP_B2]);
      PL<float> w;
      PL<int> moved;
      if (lookup) LANES {
        w[lane] = lane < SMJ_MC_WORDS ? mc[lane] : 0.f;
        float cur = w[lane];
        if (lane >= 1 && lane < 15) {
          const int b = lane < 8 ? b1 : b2, k = lane < 8 ? lane - 1 : lane - 8;
          cur = k < 3 ? s.xpos[b][k] : s.xquat[b][k - 3];
        }
        moved[lane] = !(fabsf(cur - w[lane]) <= SMJ_MC_EPS);
      }
      if (lookup && __builtin_bit_cast(int, wave_read(w, 0)) == septag && wave_ballot(moved) == 0) {
        const int n = (int)wave_read(w, 15);
        const float nrm[3] = {wave_read(w, 16), wave_read(w, 17), wave_read(w, 18)};
        float x0a[3], q0a[4], x0b[3], q0b[4], x1a[3], q1a[4], x1b[3], q1b[4], dxa[3], dta[3], dxb[3], dtb[3];
        for (int k = 0; k < 3; k++) { x0a[k] = wave_read(w, 1 + k); x0b[k] = wave_read(w, 8 + k); x1a[k] = uni(s.xpos[b1][k]); x1b[k] = uni(s.xpos[b2][k]); }
        for (int k = 0; k < 4; k++) { q0a[k] = wave_read(w, 4 + k); q0b[k] = wave_read(w, 11 + k); q1a[k] = uni(s.xquat[b1][k]); q1b[k] = uni(s.xquat[b2][k]); }
        smj_mc_motion(x0a, q0a, x1a, q1a, dxa, dta);
        smj_mc_motion(x0b, q0b, x1b, q1b, dxb, dtb);
        for (int k = 0; k < n && k < 5; k++) {
          float p3[3] = {wave_read(w, 20 + 4 * k), wave_read(w, 21 + 4 * k), wave_read(w, 22 + 4 * k)};
          float dist = wave_read(w, 19 + 4 * k);
          smj_mc_carry(x0a, dxa, dta, x0b, dxb, dtb, nrm, dist, p3);
          add_contact(r, dist, p3, nrm);
        }
#ifdef SMJ_EMUL
        smj_emul_mc_hits++;
#endif
        return;
      }
    }
    narrow_pair_run(r, g1, g2, s1, s2, sepslot, septag, sep_hit, sd, pc, prof);
    if (mc && ncon > ncon0 && ncon - ncon0 <= 5 && !(flags & SMJ_FLAG_CON_OVERFLOW)) {   // keep what the narrowphase found, with the poses it was found at -- unless the contact list has overflowed: the manifold may be cut short, and the worker that redoes the step (same poses) must not replay it
      const int b1 = uni(r[SMJ_CP_B1]), b2 = uni(r[SMJ_CP_B2]), n = ncon - ncon0;
      SYNC();
      LANES {
        if (lane < SMJ_MC_WORDS) {
          float v = 0.f;
          if (lane == 0) v = asf(septag);
          else if (lane < 15) { const int b = lane < 8 ? b1 : b2, k = lane < 8 ? lane - 1 : lane - 8; v = k < 3 ? s.xpos[b][k] : s.xquat[b][k - 3]; }
          else if (lane == 15) v = (float)n;
          else if (lane < 19) v = s.cframe[cslot(ncon0)][lane - 16];
          else if (lane < 19 + 4 * n) { const int k = (lane - 19) >> 2, q = (lane - 19) & 3; v = q == 0 ? s.cdist[cslot(ncon0 + k)] : s.cpos[cslot(ncon0 + k)][q - 1]; }
          mc[lane] = v;
        }
      }
    }
  }
  SMJ_DEV void narrow_pair_run(const int* r, int g1, int g2, int s1, int s2, float* sepslot, int septag, bool sep_hit, const float* sd, float* pc, bool prof) {
    Shape A, Bs;
    float c0[3], c1[3], depth, dir[3], pos[3];
    load_shape(A, g1, s1, c0);
    load_shape(Bs, g2, s2, c1);
    const float margin = asf(uni(r[SMJ_CP_MARGIN]));
    if (A.type == GT_SPHERE && Bs.type == GT_SPHERE) {
      if (sphere_sphere(A.pos, A.size[0], Bs.pos, Bs.size[0], margin, depth, dir, pos)) add_contact(r, depth, pos, dir);
      return;
    }
    if (A.type == GT_SPHERE && Bs.type == GT_BOX) {
      if (sphere_box(A.pos, A.size[0], Bs, margin, depth, dir, pos)) add_contact(r, depth, pos, dir);
      return;
    }
    if (A.type == GT_BOX && Bs.type == GT_SPHERE) {
      if (sphere_box(Bs.pos, Bs.size[0], A, margin, depth, dir, pos)) {
        for (int k = 0; k < 3; k++) dir[k] = -dir[k];   // the contact keeps the pair's geom order
        add_contact(r, depth, pos, dir);
      }
      return;
    }
    if (A.type == GT_SPHERE && Bs.type == GT_CAPSULE) {
      if (sphere_capsule(A.pos, A.size[0], Bs, margin, depth, dir, pos)) add_contact(r, depth, pos, dir);
      return;
    }
    if (A.type == GT_CAPSULE && Bs.type == GT_SPHERE) {
      if (sphere_capsule(Bs.pos, Bs.size[0], A, margin, depth, dir, pos)) {
        for (int k = 0; k < 3; k++) dir[k] = -dir[k];
        add_contact(r, depth, pos, dir);
      }
      return;
    }
    if (A.type == GT_CAPSULE && Bs.type == GT_CAPSULE) { capsule_capsule(r, A, Bs, margin); return; }
    if (A.type == GT_BOX && Bs.type == GT_BOX && M.multiccd) {
      const long long tb = prof ? smj_clock() : 0;
      box_box(r, s1, s2, margin);
      if (prof) pc[SMJ_PROF_C_TBOXBOX] += (float)(smj_clock() - tb);
      return;
    }
    const long long tm = prof ? smj_clock() : 0;
    // MPR in a LOCAL frame (round 6): the interior point of the smaller of the two geoms is the origin.  A support point of the Minkowski
    // difference is a difference of two world points; at 1.5 m from the world origin each carries 1.2e-7 m of fp32 rounding -- the scale
    // of MPR's own 1e-6 m tolerance and of a resting body's penetration (3e-5 m) -- while the same points a few centimetres from a local
    // origin carry 4e-9.  (The poses themselves keep the rounding kinematics gave them: an offset common to all support points of a
    // query, not noise between them.)  The contact position goes back to world coordinates at the end; directions do not change.
    float org[3];
    {
      const float ha = uni(s.u.c.half[s1][0]) + uni(s.u.c.half[s1][1]) + uni(s.u.c.half[s1][2]), hb = uni(s.u.c.half[s2][0]) + uni(s.u.c.half[s2][1]) + uni(s.u.c.half[s2][2]);
      for (int k = 0; k < 3; k++) org[k] = SMJ_MPR_LOCAL ? (ha <= hb ? c0[k] : c1[k]) : 0.f;
      for (int k = 0; k < 3; k++) { A.pos[k] -= org[k]; Bs.pos[k] -= org[k]; c0[k] -= org[k]; c1[k] -= org[k]; }
    }
    float sep[3];
    bool pen = false, skipped = false;
    if (sep_hit) {   // (tag = pair + 1: a zeroed cache holds no entry)
      // one support query along the direction that separated the pair last time: still behind the origin (by more than the
      // rounding of the test) -> disjoint, what the full query would find
      MprPt q;
      mpr_support(A, Bs, sd, q);
      skipped = dot3(q.v, sd) < -1e-6f;
#ifdef SMJ_EMUL
      if (skipped) smj_emul_sep_skips++;   // (tests: the cache is exercised)
#endif
    }
    if (!skipped) {
      pen = mpr_penetration(A, Bs, c0, c1, depth, dir, pos, sep);
      if (sepslot && !pen && (sep[0] != 0.f || sep[1] != 0.f || sep[2] != 0.f)) {
        LANES { if (lane == 0) *reinterpret_cast<Vec4*>(sepslot) = Vec4{sep[0], sep[1], sep[2], asf(septag)}; }
      }
    }
    if (prof) pc[SMJ_PROF_C_TMPR1] += (float)(smj_clock() - tm);
    if (!pen) return;
    if (-depth > margin || dot3(dir, dir) < 0.5f) return;
    for (int k = 0; k < 3; k++) pos[k] += org[k];
    add_contact(r, -depth, pos, dir);
    if (prof) pc[SMJ_PROF_C_NHIT] += 1.f;
    if (prof && M.multiccd && A.type != GT_SPHERE && Bs.type != GT_SPHERE) pc[SMJ_PROF_C_NMULTI] += 1.f;
    if (M.multiccd && A.type != GT_SPHERE && Bs.type != GT_SPHERE) {
#ifdef SMJ_EMUL   // the serial formulation stays in the lane emulator as the comparator of the four-wide one (tests/test_emul_parity.py)
      if (M.multi_serial) { convex_multi(r, A, Bs, s1, s2, pos, dir, margin, 1e-3f * asf(uni(r[SMJ_CP_RBMIN])), org); return; }
#endif
      const long long t4 = prof ? smj_clock() : 0;
      const int rounds = convex_multi4(r, A, Bs, s1, s2, pos, dir, margin, 1e-3f * asf(uni(r[SMJ_CP_RBMIN])), org);
      if (prof) { pc[SMJ_PROF_C_TMULTI] += (float)(smj_clock() - t4); pc[SMJ_PROF_C_ROUNDS] += (float)rounds; }
    }
  }

  // non-plane pairs: cache world frames of the participating geoms, sphere + oriented-box broadphase with lane = pair,
  // MPR on the survivors in pair-table order
  SMJ_DEV void collision_convex(float* pc, bool prof) {
    if (!M.convex_pairs || (M.nconvpair == 0 && (NSAT == 0 || M.nsgeom == 0))) return;
    long long tc = prof ? smj_clock() : 0;
#define CTICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - tc); tc = t1; }
    for (int c0 = 0; c0 < M.ncgeom; c0 += 64) {
      LANES {
        const int c = c0 + lane;
        if (c < M.ncgeom) {
          const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_cgrec + opaque(c) * SMJ_CG_STRIDE, 16));
          int v[SMJ_CG_STRIDE];
          for (int k = 0; k < SMJ_CG_STRIDE; k++) v[k] = r[k];
          float lp[3], lm[9], pos[3], mat[9], cw[3];
          for (int k = 0; k < 3; k++) lp[k] = asf(v[SMJ_CG_POS + k]);
          for (int k = 0; k < 9; k++) lm[k] = asf(v[SMJ_CG_MAT + k]);
          pose_from(v[SMJ_CG_BODY], lp, lm, pos, mat);
          const float lc[3] = {asf(v[SMJ_CG_LCEN]), asf(v[SMJ_CG_LCEN + 1]), asf(v[SMJ_CG_LCEN + 2])};
          mulmat3vec(cw, mat, lc);
          for (int k = 0; k < 3; k++) { s.u.c.pos[c][k] = pos[k]; s.u.c.cen[c][k] = pos[k] + cw[k]; s.u.c.half[c][k] = asf(v[SMJ_CG_HALF + k]); }
          for (int k = 0; k < 9; k++) s.u.c.mat[c][k] = mat[k];
          // everything load_shape needs, gathered here lane-parallel so that the (wave-serial) MPR set-up reads LDS only
          const float lcc[3] = {asf(v[SMJ_CG_CCEN]), asf(v[SMJ_CG_CCEN + 1]), asf(v[SMJ_CG_CCEN + 2])};
          float wc[3];
          mulmat3vec(wc, mat, lcc);
          for (int k = 0; k < 3; k++) { s.u.c.ccen[c][k] = pos[k] + wc[k]; s.u.c.size[c][k] = asf(v[SMJ_CG_SIZE + k]); }
          s.u.c.meta[c] = v[SMJ_CG_META];
        }
      }
    }
    SYNC();
    CTICK(SMJ_PROF_C_POSE)
#if SMJ_SPLIT_COLLIDE
    // The env's second wavefront works the moving-moving pairs (collide_helper -> collision_moving) while this one works the pairs with
    // the static world: two barriers per step.  Its contacts come back in the slots NCON - 1, NCON - 2, ... and are appended here,
    // behind the static ones: the one-wavefront build's list, contact for contact.
    // (serial_redo: the second go at a step whose contact list overflowed, run() -- this wavefront alone, pairs in table order)
    if (!serial_redo) {
      LANES {
        if (lane == 0) { s.mbox[0] = W2_COLLIDE; s.mbox[1] = env; s.u.c.contotal = ncon; }
      }
      WG_BARRIER();
      split_on = true;
    }
    collision_static(pc, prof);
    split_on = false;
    CTICK(SMJ_PROF_C_NARROW)
    if (serial_redo) {
      collision_moving(pc, prof);
      SYNC();
      return;
    }
    WG_BARRIER();
    {
      const int nh = uni(s.mbox[2]);
      flags |= uni(s.mbox[3]);
      // (also on a flagged step: every contact either wavefront wrote sits in a slot it had claimed, and the claims that succeeded add up
      // to NCON at most -- the list keeps everything that fitted, as the one-wavefront kernel's does)
      if (nh > 0 && ncon + nh <= NCON) {
        PL<float[31]> w;
        LANES {
          if (lane < nh) {
            const int c = NCON - 1 - lane;
            float* v = w[lane];
            v[0] = s.cdist[c]; v[1] = s.cmargin[c];
            for (int k = 0; k < 9; k++) v[2 + k] = s.cframe[c][k];
            for (int k = 0; k < 3; k++) v[11 + k] = s.cpos[c][k];
            for (int k = 0; k < 5; k++) { v[14 + k] = s.cfric[c][k]; v[19 + k] = s.csolimp[c][k]; }
            v[24] = s.csolref[c][0]; v[25] = s.csolref[c][1];
            v[26] = __builtin_bit_cast(float, (int)s.cdim[c]); v[27] = __builtin_bit_cast(float, (int)s.cgeom1[c]);
            v[28] = __builtin_bit_cast(float, (int)s.cgeom2[c]); v[29] = __builtin_bit_cast(float, (int)s.cefc[c]);
            v[30] = __builtin_bit_cast(float, (int)s.cpair[c]);
          }
        }
        SYNC();
        LANES {
          if (lane < nh) {
            const int c = ncon + lane;
            const float* v = w[lane];
            s.cdist[c] = v[0]; s.cmargin[c] = v[1];
            for (int k = 0; k < 9; k++) s.cframe[c][k] = v[2 + k];
            for (int k = 0; k < 3; k++) s.cpos[c][k] = v[11 + k];
            for (int k = 0; k < 5; k++) { s.cfric[c][k] = v[14 + k]; s.csolimp[c][k] = v[19 + k]; }
            s.csolref[c][0] = v[24]; s.csolref[c][1] = v[25];
            s.cdim[c] = __builtin_bit_cast(int, v[26]); s.cgeom1[c] = (unsigned short)__builtin_bit_cast(int, v[27]);
            s.cgeom2[c] = (unsigned short)__builtin_bit_cast(int, v[28]); s.cefc[c] = __builtin_bit_cast(int, v[29]);
            s.cpair[c] = __builtin_bit_cast(int, v[30]);
          }
        }
        ncon += nh;
      } else if (nh > 0) flags |= SMJ_FLAG_CON_OVERFLOW | 0x4000;   // (cannot happen: see above; the list would stay this wavefront's own contacts)
    }
    SYNC();
    CTICK(SMJ_PROF_C_SPHERE)
#undef CTICK
  }
  enum { W2_COLLIDE = 1, W2_EXIT = 2, W2_SAT_NEWTON = 3, W2_SAT_FORWARD = 4, W2_SAT_INTEGRATE = 5 };
  // The second wavefront of the env (smj_kernels_sat2.hip).  It waits at a workgroup barrier until the first one has a job for it
  // (the command in the mailbox), does it, meets the first one at a second barrier where the results change hands, and leaves when
  // the first one is through with the launch.  Jobs -- all of them work the first wavefront does itself in the one-wavefront build,
  // in the same arithmetic order, so the two builds agree bit for bit:
  //   W2_SAT_FORWARD  sat_forward() (pose, mass block, smooth forces of every satellite) beside the main tree's kinematics;
  //   W2_COLLIDE      the moving-moving pairs (collision_moving) beside the pairs with the static world;
  //   W2_SAT_NEWTON   the satellites' Newton blocks and the search direction of the uncoupled ones (sat_hessian, sat_solve_own)
  //                   beside the main block's H = M + J' W J on the matrix cores;
  //   W2_SAT_INTEGRATE sat_integrate() (implicit velocity update and position integration of every satellite) beside the main
  //                   tree's.
  // (Entered behind the first of those barriers -- the kernel reads the env from the mailbox there, smj_step_tu.h.)
  SMJ_DEV void helper() {
    rev = true;
    split_on = true;
    for (int cmd = uni(s.mbox[0]);;) {
      if (cmd == W2_COLLIDE) {
        ncon = 0;
        flags = 0;
        collision_moving(nullptr, false);
        LANES { if (lane == 0) { s.mbox[2] = ncon; s.mbox[3] = flags; } }
      } else if (cmd == W2_SAT_NEWTON) {
        const uint64_t conemask = mk64(uni(s.mbox[1]), uni(s.mbox[2]));
        sat_hessian(conemask);
        SYNC();
        sat_solve_own();
      } else if (cmd == W2_SAT_FORWARD) {
        env = uni(s.mbox[1]);   // the first job of every step names the env (a worker kernel goes from env to env)
        sat_forward();
      } else if (cmd == W2_SAT_INTEGRATE) {
        PL<int> bad;
        LANES { bad[lane] = 0; }
        sat_integrate(bad);
        const int anybad = wave_ballot(bad) != 0;
        LANES { if (lane == 0) s.mbox[2] = anybad; }
      }
      WG_BARRIER();   // the first wavefront takes the results over
      WG_BARRIER();   // its next job, or the end of the launch
      cmd = uni(s.mbox[0]);
      if (cmd == W2_EXIT) return;
    }
  }
  SMJ_DEV void fork2(int cmd, int w1 = 0, int w2 = 0) {
    LANES { if (lane == 0) { s.mbox[0] = cmd; s.mbox[1] = w1; s.mbox[2] = w2; } }
    WG_BARRIER();
  }
  SMJ_DEV void collision_moving(float* pc, bool prof) {
    long long tc = prof ? smj_clock() : 0;
#define CTICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - tc); tc = t1; }
#else
#if NSAT > 0
    collision_static(pc, prof);   // moving geoms against the world body's geoms, through the uniform grid (before the moving-moving pairs: pair-table order)
    CTICK(SMJ_PROF_C_NARROW)
#endif
#endif
    // pass 1: bounding spheres of all pairs (lane = pair), survivors compacted in table order.  The pair words of eight
    // chunks are fetched up front so that their load latency is paid once per group, not once per chunk.
    int nsurv = 0;
    const int ncp = M.nconvpair;
    const int* const pair_ss = M.k_convpair_ss;
    const float* const pair_rr = M.k_convpair_rsum;
    for (int g0 = 0; g0 < ncp; g0 += 512) {
      PL<int> ssv[8];
      PL<float> rrv[8];
      LANES {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          int t = g0 + 64 * j + lane;
          t = t < ncp ? t : ncp;   // entry ncp is padding with radius -1
          ssv[j][lane] = pair_ss[t];
          rrv[j][lane] = pair_rr[t];
        }
      }
      // the eight chunks' tests first (independent: their LDS gathers overlap), then the eight compactions (a chain through nsurv)
      PL<int> hit[8];
      LANES {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int ss = ssv[j][lane], s1 = ss & 255, s2 = ss >> 8;
          const float rr = rrv[j][lane];
          const float dv[3] = {s.u.c.cen[s2][0] - s.u.c.cen[s1][0], s.u.c.cen[s2][1] - s.u.c.cen[s1][1], s.u.c.cen[s2][2] - s.u.c.cen[s1][2]};
          hit[j][lane] = rr >= 0.f && dot3(dv, dv) <= rr * rr;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int base = g0 + 64 * j;
        if (base < ncp) {
          const uint64_t mask = wave_ballot(hit[j]);
          LANES {
            if (hit[j][lane]) {
              const int at = nsurv + popc64(mask & ((1ull << lane) - 1));
              if (at < 1024) s.u.c.list[at] = (unsigned short)(base + lane);
            }
          }
          nsurv += popc64(mask);
        }
      }
    }
    if (nsurv > 1024) { nsurv = 1024; flags |= SMJ_FLAG_CON_OVERFLOW; }   // pairs beyond the survivor list are lost: flagged
    SYNC();
    CTICK(SMJ_PROF_C_SPHERE)
    if (prof) pc[SMJ_PROF_C_NSPHERE] += (float)nsurv;
    // pass 2: oriented boxes on the survivors, then MPR in table order
    for (int base = 0; base < nsurv; base += 64) {
      PL<int> hit;
      PL<int> tt;
      LANES {
        int h = 0, t = 0;
        if (base + lane < nsurv) {
          t = s.u.c.list[base + lane];
          const int ss = M.k_convpair_ss[t], s1 = ss & 255, s2 = ss >> 8;
          const float dv[3] = {s.u.c.cen[s2][0] - s.u.c.cen[s1][0], s.u.c.cen[s2][1] - s.u.c.cen[s1][1], s.u.c.cen[s2][2] - s.u.c.cen[s1][2]};
          // the six face axes of the two boxes, straight-line: frames and half extents are fetched once (their LDS reads issue
          // back to back), no early exit (a chain of dependent LDS gathers per axis cost more than the arithmetic it spared)
          float Ra[9], Rb[9], ha[3], hb[3], Rm[3][3], ta[3], tb[3];
#pragma unroll
          for (int k = 0; k < 9; k++) { Ra[k] = s.u.c.mat[s1][k]; Rb[k] = s.u.c.mat[s2][k]; }
#pragma unroll
          for (int k = 0; k < 3; k++) { ha[k] = s.u.c.half[s1][k]; hb[k] = s.u.c.half[s2][k]; }
#pragma unroll
          for (int i = 0; i < 3; i++) {
            ta[i] = Ra[i] * dv[0] + Ra[3 + i] * dv[1] + Ra[6 + i] * dv[2];
            tb[i] = Rb[i] * dv[0] + Rb[3 + i] * dv[1] + Rb[6 + i] * dv[2];
#pragma unroll
            for (int j = 0; j < 3; j++) Rm[i][j] = fabsf(Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]);
          }
          h = 1;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            if (fabsf(ta[k]) > ha[k] + (Rm[k][0] * hb[0] + Rm[k][1] * hb[1] + Rm[k][2] * hb[2])) h = 0;   // axis k of box 1
            if (fabsf(tb[k]) > hb[k] + (Rm[0][k] * ha[0] + Rm[1][k] * ha[1] + Rm[2][k] * ha[2])) h = 0;   // axis k of box 2
          }
        }
        hit[lane] = h;
        tt[lane] = t;
      }
      uint64_t mask = wave_ballot(hit);
      // the survivors' cached separating directions, fetched lane-parallel (lane = pair) ahead of the serial loop
      PL<float> sdx, sdy, sdz;
      PL<int> stag;
      float* const sepbase = (S.sepcache && M.sep_cache) ? S.sepcache + (size_t)env * (SMJ_SEP_SLOTS * 4) : nullptr;
      LANES {
        Vec4 e = {0.f, 0.f, 0.f, 0.f};
        if (sepbase && hit[lane]) e = *reinterpret_cast<const Vec4*>(sepbase + 4 * (tt[lane] & (SMJ_SEP_SLOTS / 2 - 1)));
        sdx[lane] = e.x; sdy[lane] = e.y; sdz[lane] = e.z; stag[lane] = __builtin_bit_cast(int, e.w);
      }
      CTICK(SMJ_PROF_C_OBB)
      if (prof) pc[SMJ_PROF_C_NOBB] += (float)popc64(mask);
      while (mask) {
        const int l = ffs64(mask);
        mask &= mask - 1;
        const int t = wave_read(tt, l);
        const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_cprec + t * SMJ_CP_STRIDE, 16));
        const float sd[3] = {wave_read(sdx, l), wave_read(sdy, l), wave_read(sdz, l)};
        narrow_pair(r, uni(r[SMJ_CP_S1]), uni(r[SMJ_CP_S2]), sepbase ? sepbase + 4 * (t & (SMJ_SEP_SLOTS / 2 - 1)) : nullptr, t + 1,
                    sepbase && wave_read(stag, l) == t + 1, sd, pc, prof);
      }
      CTICK(SMJ_PROF_C_NARROW)
    }
    SYNC();
#undef CTICK
  }

  // ------------------------------------------------------------------ B.4 constraint rows
  SMJ_DEV void make_constraint() {
    const int nv = M.nv, neq = M.neq, nfric = M.nfric, nlimit = M.nlimit;
    // rows 0..63 on lanes 0..63, rows 64..NEFC-1 on lanes 0..NEFC-65 (second pass of every row stage, ROWPASS)
    // (rows 64.. are written only by a step with more than 64 rows: cleared on the first step of a launch -- run() starts
    // with nefc = NEFC -- and after such a step; nefc still holds the previous step's row count here)
    ROWPASS(rb, nefc) LANES {
      const int row = lane + rb;
      // J rows rb .. rb+63 (or up to NEFC-1) cleared as one 16-byte-aligned block: 9 wide stores per lane instead of 33
      static_assert((64 * JS) % 4 == 0 && ((NEFC % 64) * JS) % 4 == 0, "row blocks are whole float4s");
      const int nrow = NEFC - rb < 64 ? NEFC - rb : 64, nf4 = nrow * JS / 4;
      Vec4* jz = reinterpret_cast<Vec4*>(&s.J[rb][0]);
      for (int k = lane; k < nf4; k += 64) jz[k] = Vec4{0.f, 0.f, 0.f, 0.f};
      if (row < NEFC) {
        s.etype[row] = CT_NONE; s.efloss[row] = 0; s.eid[row] = 0; s.epos[row] = 0; s.emargin[row] = 0; s.ediag[row] = 0;
      }
    }
    SYNC();
    int cap = newton() ? NEFC : M.pgs_cap > 0 ? M.pgs_cap : NEFC_P;
    if (M.row_limit > 0 && M.row_limit < cap) cap = M.row_limit;
    // static rows (equalities -- all active, an inactive one gets an empty row with R large -> force 0 -- then friction-loss
    // dofs) and the limit slots, each from its row record (DevModel::k_rowrec): one level of loads
    const int nstat = neq + nfric;
    PL<int> act, lrow[5];   // limit slot: joint dof, side, and (bit patterns) range bound, margin, diag
    PL<float> lq;
    LANES {
      const int ol = opaque(lane);
      if (lane < nstat) {
        const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_rowrec + ol * SMJ_RR_STRIDE, 16));
        int v[SMJ_RR_SOLREF];
        for (int k = 0; k < SMJ_RR_SOLREF; k++) v[k] = r[k];
        const int e = lane, d1 = v[SMJ_RR_D1];
        if (v[SMJ_RR_TYPE] == CT_EQUALITY) {
          const int d2 = v[SMJ_RR_D2];
          float pos = s.qpos[v[SMJ_RR_Q1]] - asf(v[SMJ_RR_V1]), deriv = 0;
          const float a[5] = {asf(v[SMJ_RR_DATA]), asf(v[SMJ_RR_DATA + 1]), asf(v[SMJ_RR_DATA + 2]), asf(v[SMJ_RR_DATA + 3]), asf(v[SMJ_RR_DATA + 4])};
          if (d2 >= 0) {
            const float dif = s.qpos[v[SMJ_RR_Q2]] - asf(v[SMJ_RR_V2]);
            pos -= a[0] + dif * (a[1] + dif * (a[2] + dif * (a[3] + dif * a[4])));
            deriv = a[1] + dif * (2 * a[2] + dif * (3 * a[3] + dif * 4 * a[4]));
            s.J[e][d2] = -deriv;
          } else pos -= a[0];
          s.J[e][d1] = 1.f;
          s.etype[e] = CT_EQUALITY; s.eid[e] = v[SMJ_RR_ID]; s.epos[e] = pos; s.emargin[e] = 0; s.ediag[e] = asf(v[SMJ_RR_DIAG]);
        } else {
          s.J[e][d1] = 1.f;
          s.etype[e] = CT_FRICTION; s.eid[e] = v[SMJ_RR_ID]; s.efloss[e] = asf(v[SMJ_RR_FLOSS]); s.ediag[e] = asf(v[SMJ_RR_DIAG]);
        }
      }
      // limits: lane -> (joint, side), lower side first
      int a = 0;
      float q = 0;
      for (int k = 0; k < 5; k++) lrow[k][lane] = 0;
      if (lane < 2 * nlimit) {
        const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_rowrec + (nstat + ol) * SMJ_RR_STRIDE, 16));
        int v[SMJ_RR_DIAG + 1];
        for (int k = 0; k < SMJ_RR_DIAG + 1; k++) v[k] = r[k];
        const int side = v[SMJ_RR_D2];
        q = s.qpos[v[SMJ_RR_Q1]];
        const float dist = side * (asf(v[SMJ_RR_V1]) - q);
        a = dist < asf(v[SMJ_RR_V2]);
        lrow[0][lane] = v[SMJ_RR_D1]; lrow[1][lane] = side; lrow[2][lane] = v[SMJ_RR_V1]; lrow[3][lane] = v[SMJ_RR_V2]; lrow[4][lane] = v[SMJ_RR_DIAG];
      }
      act[lane] = a; lq[lane] = q;
    }
    const uint64_t lm = wave_ballot(act);
    int row0 = nstat;
    LANES {
      if (act[lane]) {
        const int r = row0 + popc64(lm & ((1ull << lane) - 1));
        if (r < cap) {
          const int side = lrow[1][lane];
          s.J[r][lrow[0][lane]] = (float)(-side);
          s.etype[r] = CT_LIMIT; s.eid[r] = lane;   // the limit SLOT (its row record is k_rowrec[nstat + slot])
          s.epos[r] = side * (asf(lrow[2][lane]) - lq[lane]);
          s.emargin[r] = asf(lrow[3][lane]); s.ediag[r] = asf(lrow[4][lane]);
        }
      }
    }
    row0 += popc64(lm);
    if (row0 > cap) { row0 = cap; flags |= SMJ_FLAG_EFC_OVERFLOW; }
    SYNC();
    // contact rows.  Phase 1, lane = contact: everything that needs model tables (body ids, dof masks, diagonal
    // approximations) is gathered at once instead of per contact in the serial loop below; rows are assigned in contact
    // order (a contact that does not fit is skipped and flagged, later smaller ones may still fit -- as before).
    PL<int> cact, cdimv, crow;
    PL<float> ctran, crot;
    LANES {
      int act = 0, dim = 0;
      float tran = 0, rot = 0;
      if (lane < ncon) {
        const int c = lane;
        dim = s.cdim[c];
        act = s.cdist[c] < s.cmargin[c];
        const int g1 = s.cgeom1[c], g2 = s.cgeom2[c], b1 = M.geom_bodyid[g1], b2 = M.geom_bodyid[g2];
        s.u.k.b1[c] = b1; s.u.k.b2[c] = b2;
        s.u.k.m1lo[c] = M.k_body_dofmask_lo[b1]; s.u.k.m1hi[c] = M.k_body_dofmask_hi[b1];
        s.u.k.m2lo[c] = M.k_body_dofmask_lo[b2]; s.u.k.m2hi[c] = M.k_body_dofmask_hi[b2];
        tran = M.geom_invweight0[2 * g1] + M.geom_invweight0[2 * g2];
        rot = M.geom_invweight0[2 * g1 + 1] + M.geom_invweight0[2 * g2 + 1];
      }
      cact[lane] = act; cdimv[lane] = dim; ctran[lane] = tran; crot[lane] = rot; crow[lane] = -1;
      if (lane < NVP)
        for (int x = 0; x < 6; x++) s.u.k.cd[lane][x] = lane < nv ? cdof[lane][x] : 0.f;
    }
    for (int c = 0; c < ncon; c++) {
      int d = wave_read(cdimv, c);
      if (!wave_read(cact, c)) continue;
      if (row0 + d > cap) {
        // Out of rows (flagged: the result is outside parity from here on).  Degrade gracefully instead of dropping the
        // contact: give up its rolling / torsional rows first, then its friction, and only then the contact itself -- a
        // contact that keeps its normal row still prevents penetration.
        flags |= SMJ_FLAG_EFC_OVERFLOW;
        if (d > 3 && row0 + 3 <= cap) d = 3;
        else if (row0 + 1 <= cap) d = 1;
        else continue;
        LANES { if (lane == c) { cdimv[lane] = d; s.cdim[c] = d; } }
      }
      LANES { if (lane == c) crow[lane] = row0; }
      row0 += d;
    }
    LANES {
      if (lane < ncon) {
        const int c = lane, r0 = crow[lane], dim = cdimv[lane];
        s.cefc[c] = r0;
        if (r0 >= 0) {
#pragma unroll
          for (int r = 0; r < 6; r++)
            if (r < dim) {
              s.etype[r0 + r] = dim == 1 ? CT_CONTACT_FRICTIONLESS : CT_CONTACT_ELLIPTIC;
              s.eid[r0 + r] = c; s.epos[r0 + r] = s.cdist[c]; s.emargin[r0 + r] = s.cmargin[c];
              s.ediag[r0 + r] = r < 3 ? ctran[lane] : crot[lane];
            }
        }
      }
    }
    SYNC();
    // Phase 2, lanes = dofs fill the Jacobian columns; per contact only LDS is read.  64 / NVP contacts per pass: with 32 dof
    // lanes, lanes 0-31 take contact c0 and lanes 32-63 contact c0 + 1.
    constexpr int CPP = 64 / NVP;
    for (int c0 = 0; c0 < ncon; c0 += CPP) {
      LANES {
        const int c = c0 + lane / NVP, d = lane % NVP;
        if (c < ncon && d < nv) {
          const int r0 = s.cefc[c];
          if (r0 >= 0) {
            const int dim = s.cdim[c], b1 = s.u.k.b1[c], b2 = s.u.k.b2[c];
            const uint64_t m1 = mk64(s.u.k.m1lo[c], s.u.k.m1hi[c]), m2 = mk64(s.u.k.m2lo[c], s.u.k.m2hi[c]);
            const int in1 = (int)((m1 >> d) & 1), in2 = (int)((m2 >> d) & 1);
            const float sg = (float)(in2 - in1);
            if (sg != 0.f) {
              // a dof moves at most one of the two bodies differently (same tree: the common ancestors cancel; different
              // trees: disjoint dof sets): offsets relative to the subtree com of the body it moves
              const int bb = in2 ? b2 : b1;
              const float off[3] = {s.cpos[c][0] - s.com[bb][0], s.cpos[c][1] - s.com[bb][1], s.cpos[c][2] - s.com[bb][2]};
              float cd[6], tv[3];
              for (int x = 0; x < 6; x++) cd[x] = s.u.k.cd[d][x];
              cross3(tv, cd, off);
              const float jp[3] = {cd[3] + tv[0], cd[4] + tv[1], cd[5] + tv[2]};
#pragma unroll
              for (int r = 0; r < 6; r++) {
                if (r < dim) {
                  const float* ax = s.cframe[c] + 3 * (r < 3 ? r : r - 3);
                  s.J[r0 + r][d] = sg * (r < 3 ? dot3(ax, jp) : dot3(ax, cd));
                }
              }
            }
          }
        }
      }
    }
    nefc = row0;
    SYNC();
    // impedance, R, K, B  [MJ] mj_makeImpedance
    ROWPASS(rb, nefc) LANES {
      const int i = lane + rb;
      if (i < nefc) {
        const int t = s.etype[i], id = s.eid[i];
        float solref[2], solimp[5];
        if (t == CT_CONTACT_FRICTIONLESS || t == CT_CONTACT_ELLIPTIC) {
          solref[0] = s.csolref[id][0]; solref[1] = s.csolref[id][1];
          for (int k = 0; k < 5; k++) solimp[k] = s.csolimp[id][k];
        } else {   // static rows sit at their own index, limit rows carry their slot
          const int* r = M.k_rowrec + (t == CT_LIMIT ? M.neq + M.nfric + id : i) * SMJ_RR_STRIDE;
          solref[0] = asf(r[SMJ_RR_SOLREF]); solref[1] = asf(r[SMJ_RR_SOLREF + 1]);
          for (int k = 0; k < 5; k++) solimp[k] = asf(r[SMJ_RR_SOLIMP + k]);
        }
        const float imp = impedance(solimp, s.epos[i], s.emargin[i]);
        s.eR[i] = fmaxf(SMJ_MINVAL, (1 - imp) * s.ediag[i] / imp);
        const float dmax = fminf(SMJ_MAXIMP, fmaxf(SMJ_MINIMP, solimp[1]));
        float K, B;
        if (solref[0] > 0) {
          const float tc = fmaxf(solref[0], 2 * M.timestep), dr = solref[1];
          K = 1.0f / fmaxf(SMJ_MINVAL, dmax * dmax * tc * tc * dr * dr);
          B = 2.0f / fmaxf(SMJ_MINVAL, dmax * tc);
        } else { K = -solref[0] / fmaxf(SMJ_MINVAL, dmax * dmax); B = -solref[1] / fmaxf(SMJ_MINVAL, dmax); }
        const bool fr = (t == CT_FRICTION) || (t == CT_CONTACT_ELLIPTIC && i != s.cefc[id]);
        if (fr) K = 0;
        s.eK[i] = K; s.eBv[i] = B; s.eimp[i] = imp;
      }
    }
    SYNC();
    LANES {
      if (lane < ncon) {
        const int c = lane, i = s.cefc[c], dim = s.cdim[c];
        if (i >= 0 && dim >= 3) {
          const float r1 = s.eR[i] / fmaxf(SMJ_MINVAL, M.impratio);
          s.eR[i + 1] = r1;
          const float f0 = s.cfric[c][0];
          for (int j = 1; j < dim - 1; j++) s.eR[i + 1 + j] = r1 * f0 * f0 / (s.cfric[c][j] * s.cfric[c][j]);
        }
      }
    }
    SYNC();
  }

  // the dense Jacobian row of constraint row `row` (satellite builds: rows beyond the dense ones share the zero row NDR)
  SMJ_DEV const float* jrow(int row) const {
#if NSAT > 0
    return s.J[row < NDR ? row : NDR];
#else
    return s.J[row];
#endif
  }
  SMJ_DEV int ndense() const {
#if NSAT > 0
    return nd;
#else
    return nefc;
#endif
  }
  // contact regularised-cone mu  [MJ] con->mu = friction[0]*sqrt(R[1]/R[0])
  SMJ_DEV float contact_mu(int c) const {
    const int i = s.cefc[c];
    return s.cfric[c][0] * sqrtf(s.eR[i + 1] / s.eR[i]);
  }

  // ------------------------------------------------------------------ projectConstraint + PGS
  // The sweeps are lane = row.  Two builds of the same code (template WIDE), chosen per step by the env's row count:
  //   WIDE = false -- at most 64 rows (99 % of the steps): one register set, A = Y D^-1 Y' + R a 64 x 64 square that starts at
  //     row 64 of J (free when there are no more rows) and runs on through the union; row i = column i is a conflict-free read;
  //   WIDE = true -- every row the variant holds: the per-row state (force, residual, type ...) lives in NPS register sets --
  //     set p holds rows 64 p .. 64 p + 63 -- and A is a packed lower triangle in the union (row i doubles as column i), which
  //     is what lets the standard variant's 80 x 80 matrix fit with nothing added to the launch's LDS.
  SMJ_DEV static int tri(int r, int c) { return r >= c ? ((r * (r + 1)) >> 1) + c : ((c * (c + 1)) >> 1) + r; }
#define PSETS(p, ne) _Pragma("unroll") for (int p = 0; p < NP; p++) if (p == 0 || (ne) > 64 * p)
#define PSETS_ALL(p) _Pragma("unroll") for (int p = 0; p < NP; p++)
  // value of row i (wave-uniform) from the register sets
  template <int NP, class T>
  SMJ_DEV T prow(const PL<T>* a, int i) const {
    T v = wave_read(a[0], i & 63);
#pragma unroll
    for (int p = 1; p < NP; p++)
      if (i >= 64 * p) v = wave_read(a[p], i & 63);
    return v;
  }
#if NSAT > 0
  template <bool WIDE>
  SMJ_DEV float* Amat() { return s.u.pa; }
#else
  template <bool WIDE>
  SMJ_DEV float* Amat() { return WIDE ? s.A(M.pgs_cap > 0 ? M.pgs_cap : NEFC_P) : &s.J[NEFP][0]; }
#endif
  template <bool WIDE>
  SMJ_DEV static int ai(int r, int c) { return WIDE ? tri(r, c) : r * NEFP + c; }
#if NSAT == 0
  // Identity of a constraint row from one step to the next (option pgs_dual_ws; the satellite builds' twin is in smj_sat_pgs.h): type
  // and equality / friction-loss dof / limit slot, or, for a contact row, (collision pair, ordinal of the contact within the pair's
  // manifold -- a pair's contacts are contiguous in the list --, row within the contact).
  SMJ_DEV int pgs_row_key(int row) const {
    const int t = s.etype[row];
    if (t == CT_CONTACT_ELLIPTIC || t == CT_CONTACT_FRICTIONLESS) {
      const int c = s.eid[row], pr = ((s.cgeom1[c] & 0x3ff) << 10) | (s.cgeom2[c] & 0x3ff);   // (the dense builds hold <= 1023 geoms)
      int ord = 0;
      for (int k = c - 1; k >= 0 && (((s.cgeom1[k] & 0x3ff) << 10) | (s.cgeom2[k] & 0x3ff)) == pr; k--) ord++;
      return (int)(0x80000000u | ((unsigned)pr << 6) | ((unsigned)(ord & 7) << 3) | (unsigned)((row - s.cefc[c]) & 7));
    }
    return 0x40000000 | (t << 20) | (s.eid[row] & 0xfffff);
  }
#endif
  template <bool WIDE>
  SMJ_DEV void solve(bool dbg, float* pc, long long& t0, bool prof) {
#define TICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - t0); t0 = t1; }
    constexpr int NP = WIDE ? NPS : 1;
    const int nv = M.nv, ne = nefc;
    float* const A = Amat<WIDE>();
    // efc_vel, aref, warm-start residual jar (rows = lanes), all from J before it is transformed
    PL<float> aref[NP], jar[NP], Rr[NP], bb[NP];
    PSETS(p, ne) LANES {
      const int row = lane + 64 * p;
      float vel = 0, jw = 0;
      if (row < ne)
        for (int k = 0; k < nv; k++) { const float jv = s.J[row][k]; vel += jv * s.qvel[k]; jw += jv * s.warm[k]; }
      const float ar = row < ne ? -s.eBv[row] * vel - s.eK[row] * s.eimp[row] * (s.epos[row] - s.emargin[row]) : 0.f;
      aref[p][lane] = ar; jar[p][lane] = jw - ar; Rr[p][lane] = row < ne ? s.eR[row] : 1.f;
      if (row < NEFC) s.earef[row] = ar;
    }
    // u = L^-T phase of g (dof lanes)
    PL<float> u;
    LANES { u[lane] = lane < nv ? g_r[lane] : 0.f; }
    solve_LT(u);
    LANES { if (lane < NVP) s.uu[lane] = lane < nv ? u[lane] : 0.f; }
    // Y = J L^-1 : every lane runs the L^-T phase on its own row (private LDS row)
    for (int i = nv - 1; i > 0; i--) {
      const int na = uni(M.k_dof_anc_num[i]), adr = uni(M.k_dof_anc_adr[i]);
      if (na == 0) continue;
      PSETS(p, ne) LANES {
        const int row = lane + 64 * p < NEFC ? lane + 64 * p : NEFC - 1;
        const float xi = lane + 64 * p < NEFC ? s.J[row][i] : 0.f;
        if (xi != 0.f)
          for (int a = 0; a < na; a++) {
            const int j = uni(M.k_dof_anc[adr + a]);
            s.J[row][j] -= s.MM[i][j] * xi;
          }
      }
    }
    SYNC();
    // b = Y Dinv u - aref
    PSETS(p, ne) LANES {
      const int row = lane + 64 * p;
      float v = 0;
      if (row < ne)
        for (int k = 0; k < nv; k++) v += s.J[row][k] * s.Dinv[k] * s.uu[k];
      bb[p][lane] = v - aref[p][lane];
      if (row < NEFC) s.eb[row] = bb[p][lane];
    }
    SYNC();  // all reads of the tree temporaries that alias A are done (cdof etc. live in registers from here on)
    // A = Y Dinv Y' (+R on the diagonal) on the matrix cores, 16x16 tiles of the lower triangle, K = nv padded to 4
    {
      const int ntile = (ne + 15) >> 4, ksteps = (nv + 3) >> 2;
      for (int tr = 0; tr < ntile; tr++)
        for (int tc = 0; tc <= tr; tc++) {
          PL<F4v> acc;
          LANES { for (int r = 0; r < 4; r++) acc[lane].r[r] = 0.f; }
          for (int ks = 0; ks < ksteps; ks++) {
            PL<float> a, b;
            LANES {
              const int k = 4 * ks + (lane >> 4);
              const float dk = k < nv ? s.Dinv[k] : 0.f;
              const int ra = 16 * tr + (lane & 15), rb = 16 * tc + (lane & 15);
              a[lane] = (k < nv && ra < NEFC) ? s.J[ra][k] * dk : 0.f;
              b[lane] = (k < nv && rb < NEFC) ? s.J[rb][k] : 0.f;
            }
            mfma16x16x4(acc, a, b);
          }
          LANES {
            for (int r = 0; r < 4; r++) {
              const int row = 16 * tr + (lane >> 4) * 4 + r, col = 16 * tc + (lane & 15);
              float v = acc[lane].r[r];
              if (row == col) v += row < ne ? s.eR[row] : 1.f;
              if (WIDE) { if (col <= row && row < (M.pgs_cap > 0 ? M.pgs_cap : NEFC_P)) A[tri(row, col)] = v; }
              else { A[row * NEFP + col] = v; if (tr != tc) A[col * NEFP + row] = v; }
            }
          }
        }
    }
    SYNC();
    TICK(SMJ_PROF_PROJECT)
    // warm start  [MJ] mj_warmstart (PGS branch): forces from the primal residual at qacc_warmstart
    PSETS(p, ne) LANES { if (lane + 64 * p < NEFC) s.earef[lane + 64 * p] = jar[p][lane]; }  // stash jar in LDS so a contact's first row can see its block
    SYNC();
    PSETS(p, ne) LANES {
      float f = 0;
      const int i = lane + 64 * p;
      if (i < ne && M.warmstart) {
        const int t = s.etype[i];
        const float D = 1.0f / Rr[p][lane], jr = jar[p][lane];
        if (t == CT_EQUALITY) f = -D * jr;
        else if (t == CT_FRICTION) {
          const float fl = s.efloss[i];
          f = (jr <= -Rr[p][lane] * fl) ? fl : (jr >= Rr[p][lane] * fl) ? -fl : -D * jr;
        } else if (t == CT_LIMIT || t == CT_CONTACT_FRICTIONLESS) f = jr < 0 ? -D * jr : 0.f;
      }
      if (i < NEFC) s.ef[i] = f;  // elliptic rows: overwritten below by the contact's lane
    }
    SYNC();
    LANES {
      if (lane < ncon && M.warmstart) {
        const int c = lane, i = s.cefc[c], dim = s.cdim[c];
        if (i >= 0 && dim >= 3) {
          const float mu = contact_mu(c);
          float U[6], T = 0;
          U[0] = s.earef[i] * mu;
          for (int j = 1; j < dim; j++) { U[j] = s.earef[i + j] * s.cfric[c][j - 1]; T += U[j] * U[j]; }
          const float N = U[0];
          T = sqrtf(T);
          if ((T <= 0 && N >= 0) || (T > 0 && N >= mu * T)) { for (int j = 0; j < dim; j++) s.ef[i + j] = 0; }
          else if ((T <= 0 && N < 0) || (T > 0 && mu * N + T <= 0)) { for (int j = 0; j < dim; j++) s.ef[i + j] = -s.earef[i + j] / s.eR[i + j]; }
          else {
            const float Dm = (1.0f / s.eR[i]) / fmaxf(mu * mu * (1 + mu * mu), SMJ_MINVAL), NmT = N - mu * T;
            const float fn = -Dm * NmT * mu;
            s.ef[i] = fn;
            for (int j = 1; j < dim; j++) s.ef[i + j] = -fn / T * U[j] * s.cfric[c][j - 1];
          }
        }
      }
    }
    SYNC();
    PSETS_ALL(p) LANES { const int row = lane + 64 * p; if (row < NEFC) s.earef[row] = row < ne ? aref[p][lane] : 0.f; }
    // ---- NOT MuJoCo (option pgs_dual_ws, default on): a second start -- the forces the rows had at the end of the PREVIOUS step's solve
    // (DevState::pgsprev), matched by row identity, projected onto this step's bounds and cones -- taken when its dual cost is below that
    // of MuJoCo's start.  Same fixed point (the dual is strictly convex, R > 0), fewer sweeps to it; smj_sat_pgs.h has the measurements.
    bool have_prev = false;
#if NSAT == 0
    if (M.pgs_dual_ws && M.warmstart && S.pgsprev) {
      const int* const pk = reinterpret_cast<const int*>(S.pgsprev + (size_t)env * SMJ_PGSPREV_STRIDE);
      const float* const pf = S.pgsprev + (size_t)env * SMJ_PGSPREV_STRIDE + 1 + SMJ_PGSPREV_ROWS;
      int np = uni(pk[0]);
      np = np < 0 ? 0 : np > SMJ_PGSPREV_ROWS ? SMJ_PGSPREV_ROWS : np;
      if (np > 0) {
        have_prev = true;
        // the stored rows into LDS, lane-parallel (keys -> emargin, forces -> eimp: both dead once aref is known); a row usually sits
        // where it sat a step ago: its own index first, the scan of the whole list only for the lanes that miss
        int* const lk = reinterpret_cast<int*>(s.emargin);
        float* const lf = s.eimp;
#pragma nounroll
        for (int rb = 0; rb < np; rb += 64) LANES { const int j = lane + rb; if (j < np && j < NEFC) { lk[j] = pk[1 + j]; lf[j] = pf[j]; } }
        if (np > NEFC) np = NEFC;
        SYNC();
#pragma nounroll
        for (int rb = 0; rb < ne; rb += 64) LANES {
          const int row = lane + rb;
          if (row < ne) {
            const int key = pgs_row_key(row), t = s.etype[row];
            float f = 0.f;
            if (row < np && lk[row] == key) f = lf[row];
            else
              for (int j = 0; j < np; j++) f = lk[j] == key ? lf[j] : f;
            if (t == CT_FRICTION) { const float fl = s.efloss[row]; f = fminf(fl, fmaxf(-fl, f)); }
            else if (t == CT_LIMIT || t == CT_CONTACT_FRICTIONLESS) f = fmaxf(0.f, f);
            s.ediag[row] = f;   // (free: R was its last reader)
          }
        }
        SYNC();
        LANES {
          if (lane < ncon) {
            const int c = lane, i = s.cefc[c], dim = s.cdim[c];
            if (i >= 0 && dim >= 3) {
              const float fn = s.ediag[i];
              if (fn < SMJ_MINVAL) { for (int j = 0; j < dim; j++) s.ediag[i + j] = 0.f; }
              else {
                float s2 = 0.f;
                for (int j = 1; j < dim; j++) { const float tq = s.ediag[i + j] * fast_rcp(s.cfric[c][j - 1]); s2 += tq * tq; }
                if (s2 > fn * fn) { const float sc = fn * fast_rsqrt(s2); for (int j = 1; j < dim; j++) s.ediag[i + j] *= sc; }
              }
            }
          }
        }
        SYNC();
      }
    }
#endif
    // residual r = A f + b (lanes = rows; column reads via symmetry) and the dual cost of a start.  Pass 0: MuJoCo's (s.ef).  Pass 1: the
    // previous step's forces, swapped into s.ef; kept when cheaper.  Pass 2: MuJoCo's once more when it was the better one.
    PL<float> cost;
    float wcost = 0.f;
    for (int pass = 0;; pass++) {
      LANES { cost[lane] = 0.f; }
      PSETS_ALL(p) LANES {
        const int row = lane + 64 * p;
        f_r[p][lane] = row < ne ? s.ef[row] : 0.f;
        r_r[p][lane] = 0.f;
      }
      residual_refresh<WIDE>(bb);
      PSETS(p, ne) LANES { cost[lane] += lane + 64 * p < ne ? f_r[p][lane] * 0.5f * (r_r[p][lane] + bb[p][lane]) : 0.f; }
      const float cst = wave_sum(cost);
      bool swap = false;
      if (pass == 0) { wcost = cst; swap = have_prev; }
      else if (pass == 1) { if (cst < wcost) wcost = cst; else swap = true; }
      if (!swap) break;
#pragma nounroll
      for (int rb = 0; rb < ne; rb += 64) LANES {
        const int row = lane + rb;
        if (row < ne) { const float tq = s.ef[row]; s.ef[row] = s.ediag[row]; s.ediag[row] = tq; }
      }
      SYNC();
    }
    if (wcost > 0) { PSETS(p, ne) LANES { f_r[p][lane] = 0.f; r_r[p][lane] = bb[p][lane]; } }

    TICK(SMJ_PROF_WARM)
    // ---- PGS sweeps  [MJ] mj_solPGS
    // Row metadata lives in the registers of the row's lane and is fetched with v_readlane (no LDS round trip on the
    // serial path); the A row needed for the residual update is loaded first so its latency overlaps the scalar math.
    PL<int> type_r[NP], dimc_r[NP];   // row type; for the first row of an elliptic block: dim | contact << 8
    PL<float> aii_r[NP], lo_r[NP], hi_r[NP];   // diagonal of A; the row's bounds (equality: none, friction loss: -floss..floss, limit / frictionless contact: 0..)
    PSETS_ALL(p) LANES {
      const int row = lane + 64 * p;
      const int t = row < ne ? s.etype[row] : CT_NONE;
      type_r[p][lane] = t;
      int dc = 0;
      if (t == CT_CONTACT_ELLIPTIC) { const int c = s.eid[row]; dc = s.cdim[c] | (c << 8); }
      dimc_r[p][lane] = dc;
      const float aii = row < ne ? A[ai<WIDE>(row, row)] : 1.f;
      aii_r[p][lane] = aii; ARinv_r[p][lane] = 1.0f / aii;
      const float fl = row < ne ? s.efloss[row] : 0.f;
      lo_r[p][lane] = t == CT_EQUALITY ? -INFINITY : t == CT_FRICTION ? -fl : 0.f;
      hi_r[p][lane] = t == CT_FRICTION ? fl : INFINITY;
    }
    LANES { qla_r[lane] = 0.f; }
    const float scale = 1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1));
    int iter = 0;
    for (; iter < M.iterations; iter++) {
      float improvement = 0;
      if (iter > 0 && (iter & 7) == 0) residual_refresh<WIDE>(bb);
      ppc = prof ? pc : nullptr;
      long long tp = prof ? smj_clock() : 0;   // profiling builds: cycles in scalar rows / elliptic blocks (the Newton slots are free under PGS)
      for (int i = 0; i < ne;) {
        const int t = prow<NP>(type_r, i);
        if (t != CT_CONTACT_ELLIPTIC) {
          // A run of scalar rows (equality, friction loss, limit, frictionless contact): f <- clamp(f - r / A_ii) between the row's
          // bounds (-inf..inf, -floss..floss, 0..inf: one code path), straight-line; the A row of the NEXT row is fetched while
          // this one is worked on (its LDS latency was a third of a row's time on the serial path).
          PL<float> arow[NP];
          PSETS(p, ne) LANES { const int col = lane + 64 * p; arow[p][lane] = (!WIDE || col < ne) ? A[ai<WIDE>(i, col)] : 0.f; }
          int nrun = 0;
          for (;;) {
            const bool more = i + 1 < ne;
            const int inext = more ? i + 1 : i;   // (past the end: row i again, not used)
            PL<float> anext[NP];
            PSETS(p, ne) LANES { const int col = lane + 64 * p; anext[p][lane] = (!WIDE || col < ne) ? A[ai<WIDE>(inext, col)] : 0.f; }
            const int tnext = more ? prow<NP>(type_r, inext) : CT_CONTACT_ELLIPTIC;
            const float res = prow<NP>(r_r, i), old = prow<NP>(f_r, i), ainv = prow<NP>(ARinv_r, i);
            const float aii = prow<NP>(aii_r, i), lo = prow<NP>(lo_r, i), hi = prow<NP>(hi_r, i);
            const float fn = fminf(hi, fmaxf(lo, old - res * ainv));
            float delta = fn - old;
            float change = delta * (0.5f * aii * delta + res);
            if (change > 1e-10f) { delta = 0; change = 0; }
            improvement -= change;
            PSETS(p, ne) LANES {
              r_r[p][lane] += arow[p][lane] * delta;
              if (lane + 64 * p == i) f_r[p][lane] += delta;
            }
            i += 1; nrun++;
            if (tnext == CT_CONTACT_ELLIPTIC) break;   // the next row starts a contact block, or there is none
            PSETS(p, ne) LANES { arow[p][lane] = anext[p][lane]; }
          }
          if (prof) { const long long t1 = smj_clock(); pc[SMJ_PROF_N_UPDATE] += (float)(t1 - tp); tp = t1; pc[SMJ_PROF_N_FACTSOLVE] += (float)nrun; }
        } else {
          const int dc = prow<NP>(dimc_r, i), dim = dc & 255, c = dc >> 8;
          if (dim == 3) improvement += pgs_block<3, WIDE>(i, c);
          else if (dim == 4) improvement += pgs_block<4, WIDE>(i, c);
          else improvement += pgs_block<6, WIDE>(i, c);
          i += dim;
          if (prof) { const long long t1 = smj_clock(); pc[SMJ_PROF_N_GRAD] += (float)(t1 - tp); tp = t1; pc[SMJ_PROF_N_SOLVE] += 1.f; }
        }
      }
      improvement *= scale;
      if (!M.pgs_fixed_iter && improvement < M.tolerance) { iter++; break; }
    }
    niter = iter;
    TICK(SMJ_PROF_PGS)
#undef TICK
    PSETS_ALL(p) LANES { if (lane + 64 * p < NEFC) s.ef[lane + 64 * p] = lane + 64 * p < ne ? f_r[p][lane] : 0.f; }
    SYNC();
#if NSAT == 0
    if (M.pgs_dual_ws && M.warmstart && S.pgsprev) {   // the rows of this step and where their forces ended: the next step's second start
      int* const pk = reinterpret_cast<int*>(S.pgsprev + (size_t)env * SMJ_PGSPREV_STRIDE);
      float* const pf = S.pgsprev + (size_t)env * SMJ_PGSPREV_STRIDE + 1 + SMJ_PGSPREV_ROWS;
      const int nst = ne < SMJ_PGSPREV_ROWS ? ne : SMJ_PGSPREV_ROWS;
#pragma nounroll
      for (int rb = 0; rb < nst; rb += 64) LANES {
        const int row = lane + rb;
        if (row < nst) { pk[1 + row] = pgs_row_key(row); pf[row] = s.ef[row]; }
      }
      LANES { if (lane == 0) pk[0] = nst; }
    }
#endif
    // w = Y' f (dof lanes); qfrc_constraint = L' w ; qacc = L^-1 ( Dinv (u + w) )
    PL<float> w, qc;
    LANES {
      float v = 0;
      if (lane < nv)
        for (int r = 0; r < ne; r++) v += s.J[r][lane] * s.ef[r];
      w[lane] = v;
      if (lane < NVP) s.w[lane] = v;
    }
    SYNC();
    LANES {
      float v = w[lane];
      if (lane < nv)
        for (int i = lane + 1; i < nv; i++) v += s.MM[i][lane] * s.w[i];
      qc[lane] = v;  // J' f
      qacc_r[lane] = lane < nv ? s.Dinv[lane] * (u[lane] + w[lane]) : 0.f;
    }
    solve_L(qacc_r);
    LANES {
      if (lane < nv) { s.qacc[lane] = qacc_r[lane]; s.warm[lane] = qacc_r[lane]; s.tmp[lane] = g_r[lane] + qc[lane]; }
    }
    SYNC();
    if (dbg && S.debug) {   // the debug layout holds the first 64 rows
      LANES {
        if (lane < nv) S.debug[(SMJ_DBG_QACC + lane) * S.ld + env] = qacc_r[lane];
        S.debug[(SMJ_DBG_EFC_FORCE + lane) * S.ld + env] = lane < ne ? f_r[0][lane] : 0.f;
        S.debug[(SMJ_DBG_EFC_B + lane) * S.ld + env] = lane < ne ? bb[0][lane] : 0.f;
        S.debug[(SMJ_DBG_EFC_R + lane) * S.ld + env] = lane < ne ? s.eR[lane] : 0.f;
        S.debug[(SMJ_DBG_EFC_AREF + lane) * S.ld + env] = lane < ne ? aref[0][lane] : 0.f;
        S.debug[(SMJ_DBG_AR_DIAG + lane) * S.ld + env] = lane < ne ? A[ai<WIDE>(lane, lane)] : 0.f;
        for (int k = 0; k < 64; k++) S.debug[(SMJ_DBG_AR + k * 64 + lane) * S.ld + env] = (k < ne && lane < ne) ? A[ai<WIDE>(k, lane)] : 0.f;
      }
    }
  }

  template <bool WIDE>
  SMJ_DEV void residual_refresh(const PL<float>* bb) {
    constexpr int NP = WIDE ? NPS : 1;
#if NSAT > 0
    const int ne = ndp;          // rows of the dense system; their forces are staged in s.epos (free once aref is known): s.ef is indexed by row
    float* const fv = s.epos;
#else
    const int ne = nefc;
    float* const fv = s.ef;
#endif
    const float* const A = Amat<WIDE>();
    PSETS_ALL(p) LANES { if (lane + 64 * p < NEFC) fv[lane + 64 * p] = f_r[p][lane]; }
    SYNC();
    PSETS(p, ne) LANES {
      const int row = lane + 64 * p;
      float v = bb[p][lane];
      if (row < ne)
        for (int k = 0; k < ne; k++) v += A[ai<WIDE>(k, row)] * fv[k];
      r_r[p][lane] = v;
    }
    SYNC();
  }

  // one elliptic contact block of the PGS sweep  [MJ] mj_solPGS elliptic branch (ray update + QCQP); uniform math.
  // The block's own DIM x DIM part of A is never read as such: rows i..i+DIM of A are in the lanes anyway (arow, for the
  // residual update; A is symmetric, so lane i + r holds row r of the block), hence At old and At delta are DIM vector FMAs
  // each, read back from the block's lanes, instead of DIM^2 uniform ones on DIM^2 uniform loads; the QCQP takes the lower
  // triangle of the friction part.
  template <int DIM, bool WIDE>
  SMJ_DEV float pgs_block(int i, int c) {
    constexpr int NP = WIDE ? NPS : 1;
    constexpr int NF = DIM - 1;
#if NSAT > 0
    const int ne = ndp;
#else
    const int ne = nefc;
#endif
    const float* const A = Amat<WIDE>();
    float res[DIM], old[DIM], f[DIM], v1[DIM], mu[NF], Ac[NF * NF], a0[NF];
    PL<float[DIM]> arow[NP];  // rows i..i+DIM of A for the residual update, issued up front
    PSETS(p, ne) LANES {
      const int col = lane + 64 * p;
#pragma unroll
      for (int r = 0; r < DIM; r++) arow[p][lane][r] = (!WIDE || col < ne) ? A[ai<WIDE>(i + r, col)] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NF; j++) {
      mu[j] = s.cfric[c][j];
      a0[j] = A[ai<WIDE>(i + 1 + j, i)];
#pragma unroll
      for (int q = 0; q <= j; q++) Ac[j * NF + q] = A[ai<WIDE>(i + 1 + j, i + 1 + q)];   // (the QCQP reads the lower triangle)
    }
#pragma unroll
    for (int r = 0; r < DIM; r++) { res[r] = prow<NP>(r_r, i + r); old[r] = prow<NP>(f_r, i + r); f[r] = old[r]; }
    // v1 = At * old (used by the ray update and by the friction right-hand side)
    PL<float> vv[NP];
    PSETS(p, ne) LANES {
      float a = arow[p][lane][0] * old[0];
#pragma unroll
      for (int q = 1; q < DIM; q++) a += arow[p][lane][q] * old[q];
      vv[p][lane] = a;
    }
    float denom = 0, num = 0;
#pragma unroll
    for (int r = 0; r < DIM; r++) { v1[r] = prow<NP>(vv, i + r); denom += old[r] * v1[r]; num += old[r] * res[r]; }
    if (f[0] < SMJ_MINVAL) {  // normal update
      f[0] -= res[0] * prow<NP>(ARinv_r, i);
      if (f[0] < 0) f[0] = 0;
#pragma unroll
      for (int j = 1; j < DIM; j++) f[j] = 0;
    } else if (denom >= SMJ_MINVAL) {  // ray update
      const float x = -num * fast_rcp(denom);
      // [MJ] a step that would take the normal force below zero ends AT the cone's apex: x = -f0 / old0 = -1 (f == old here), every
      // component exactly 0.  Computed as f + x old with an fp32 reciprocal it left 2^-31 .. 1e-9 of the old force behind -- above
      // mjMINVAL, so the next sweeps took neither the normal update (f0 >= MINVAL) nor the ray update (old' A old < MINVAL): the contact
      // stayed switched off for the rest of the solve and a resting mug tilted at 4 rad/s^2 under PGS (round 6, found on the objects'
      // own scale; the fp64 oracle's residue is 1e-19, below MINVAL).
      const bool apex = f[0] + x * old[0] < 0;
#pragma unroll
      for (int r = 0; r < DIM; r++) f[r] = apex ? 0.f : f[r] + x * old[r];
    }
    // friction update with the normal fixed
    float bc[NF], v[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) bc[j] = res[j + 1] - v1[j + 1] + a0[j] * f[0];
    if (f[0] < SMJ_MINVAL) {
#pragma unroll
      for (int j = 1; j < DIM; j++) f[j] = 0;
    } else {
      const long long tq = (SMJ_PROFILING && ppc) ? smj_clock() : 0;
      float la = wave_read(qla_r, c);
      const int active = qcqp<NF>(v, Ac, bc, mu, f[0], la, M.qcqp_exact != 0, (SMJ_PROFILING && ppc) ? &ppc[SMJ_PROF_N_HMFMA] : nullptr);
      LANES { if (lane == c) qla_r[lane] = la; }
      if (SMJ_PROFILING && ppc) ppc[SMJ_PROF_N_XA] += (float)(smj_clock() - tq);
      if (active) {
        float sc = 0;
#pragma unroll
        for (int j = 0; j < NF; j++) { const float t = v[j] * fast_rcp(mu[j]); sc += t * t; }
        sc = f[0] * fast_rsqrt(fmaxf(SMJ_MINVAL, sc));
#pragma unroll
        for (int j = 0; j < NF; j++) v[j] *= sc;
      }
#pragma unroll
      for (int j = 0; j < NF; j++) f[j + 1] = v[j];
    }
    float change = 0, delta[DIM];
#pragma unroll
    for (int r = 0; r < DIM; r++) delta[r] = f[r] - old[r];
    // sv = At * delta is the block's part of the residual update A[:, i..i+DIM] delta
    PL<float> dr[NP];
    PSETS(p, ne) LANES {
      float a = arow[p][lane][0] * delta[0];
#pragma unroll
      for (int r = 1; r < DIM; r++) a += arow[p][lane][r] * delta[r];
      dr[p][lane] = a;
    }
#pragma unroll
    for (int r = 0; r < DIM; r++) change += delta[r] * (0.5f * prow<NP>(dr, i + r) + res[r]);
    if (change > 1e-10f) return 0.f;
    PSETS(p, ne) LANES {
      r_r[p][lane] += dr[p][lane];
#pragma unroll
      for (int r = 0; r < DIM; r++)
        if (lane + 64 * p == i + r) f_r[p][lane] += delta[r];
    }
    return -change;
  }
#undef PSETS
#undef PSETS_ALL
#if NSAT > 0
#include "smj_sat_pgs.h"
#endif


  // ------------------------------------------------------------------ B.7' Newton solver (primal)
  // [MJ] mj_solNewton: exact Newton steps on  0.5 (a-a_s)' M (a-a_s) + s(J a - aref)  with H = M + J' W J on the
  // matrix cores, an in-register Gauss-Jordan solve (lane i owns row i of [H | g]) and an exact line search.  This is the solver
  // the reference model runs (stretch.xml names none -> MuJoCo default Newton); same restatement as the oracle's.
  struct NRow {            // lane = constraint row
    PL<int> type, state, c0;          // c0: contact index if this lane is the first row of an elliptic contact, else -1
    PL<float> aref, D, R, fl, jar, jv, force, q0, q1, q2;
    PL<float[7]> cq;                  // u0 v0 uu uv vv Dm mu   (first row of an elliptic contact)
  };

  enum { RX_AREF = 0, RX_D, RX_JAR, RX_JV, RX_FORCE, RX_Q0, RX_Q1, RX_Q2, RX_CQ };   // R and the friction loss stay where make_constraint left them (s.eR, s.efloss)
  SMJ_DEV void rx_load(NRow& t, int rb) {
    LANES {
      const bool on = lane + rb < NEFC;
      const int l = on ? lane + rb - 64 : 0;
      t.type[lane] = on ? s.u.n.rxi[0][l] : CT_NONE; t.state[lane] = on ? s.u.n.rxi[1][l] : 0; t.c0[lane] = on ? s.u.n.rxi[2][l] : -1;
      t.aref[lane] = on ? s.u.n.rxf[RX_AREF][l] : 0.f; t.D[lane] = on ? s.u.n.rxf[RX_D][l] : 0.f; t.R[lane] = on ? s.eR[l + 64] : 1.f;
      t.fl[lane] = on ? s.efloss[l + 64] : 0.f; t.jar[lane] = on ? s.u.n.rxf[RX_JAR][l] : 0.f; t.jv[lane] = on ? s.u.n.rxf[RX_JV][l] : 0.f;
      t.force[lane] = on ? s.u.n.rxf[RX_FORCE][l] : 0.f;
      t.q0[lane] = on ? s.u.n.rxf[RX_Q0][l] : 0.f; t.q1[lane] = on ? s.u.n.rxf[RX_Q1][l] : 0.f; t.q2[lane] = on ? s.u.n.rxf[RX_Q2][l] : 0.f;
      for (int k = 0; k < 7; k++) t.cq[lane][k] = on ? s.u.n.rxf[RX_CQ + k][l] : 0.f;
    }
  }
  SMJ_DEV void rx_store(const NRow& t, int rb) {
    LANES {
      if (lane + rb < NEFC) {
        const int l = lane + rb - 64;
        s.u.n.rxi[0][l] = t.type[lane]; s.u.n.rxi[1][l] = t.state[lane]; s.u.n.rxi[2][l] = t.c0[lane];
        s.u.n.rxf[RX_AREF][l] = t.aref[lane]; s.u.n.rxf[RX_D][l] = t.D[lane];
        s.u.n.rxf[RX_JAR][l] = t.jar[lane]; s.u.n.rxf[RX_JV][l] = t.jv[lane];
        s.u.n.rxf[RX_FORCE][l] = t.force[lane];
        s.u.n.rxf[RX_Q0][l] = t.q0[lane]; s.u.n.rxf[RX_Q1][l] = t.q1[lane]; s.u.n.rxf[RX_Q2][l] = t.q2[lane];
        for (int k = 0; k < 7; k++) s.u.n.rxf[RX_CQ + k][l] = t.cq[lane][k];
      }
    }
  }

  // constraint forces / states / cost at the residual nr.jar  ([MJ] mj_constraintUpdate).  Returns the cost.
  SMJ_DEV float newton_update(NRow& nr0, bool want_hess) {
    const int ne = nefc;
    PL<float> cost;
    LANES { cost[lane] = 0.f; }
    ROWS_BEGIN(rb, ne) LANES { if (lane + rb < NEFC) s.eb[lane + rb] = nr.jar[lane]; } ROWS_END_RO()
#if NSAT > 0
    LANES { if (lane == 0) s.sat.nch = 0; }
#endif
    SYNC();
    ROWS_BEGIN(rb, ne) LANES {
      float c = 0, f = 0;
      int st = 0;
      const int i = lane + rb, t = nr.type[lane];
      const float jar = nr.jar[lane], D = nr.D[lane], R = nr.R[lane];
      if (i < ne) {
        if (t == CT_EQUALITY) { f = -D * jar; st = 1; c = 0.5f * D * jar * jar; }
        else if (t == CT_FRICTION) {
          const float fl = nr.fl[lane];
          if (jar <= -R * fl) { f = fl; st = 2; c = -fl * (0.5f * R * fl + jar); }
          else if (jar >= R * fl) { f = -fl; st = 3; c = -fl * (0.5f * R * fl - jar); }
          else { f = -D * jar; st = 1; c = 0.5f * D * jar * jar; }
        } else if (t == CT_LIMIT || t == CT_CONTACT_FRICTIONLESS) {
          if (jar < 0) { f = -D * jar; st = 1; c = 0.5f * D * jar * jar; }
        }
      }
      nr.force[lane] = f; nr.state[lane] = st; cost[lane] += c;
      if (i < NEFC) s.ef[i] = f;
    } ROWS_END_RW(rb)
    SYNC();
    // elliptic contacts: the lane of the contact's first row handles the block.  All loops run to the maximum block size 6
    // with the tail masked off (zero friction coefficient, clamped row index): fixed trip counts let the LDS reads issue
    // back to back instead of one latency per row.
    ROWS_BEGIN(rb, ne) LANES {
      const int c = nr.c0[lane];
      if (c >= 0) {
        const int i = lane + rb, dim = s.cdim[c];
        const float mu = nr.cq[lane][6];
        float U[6], jr[6], S[6], Rj[6], T = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
          const int jj = j < dim ? j : 0;
          jr[j] = s.eb[i + jj];
          Rj[j] = s.eR[i + jj];
          S[j] = j == 0 ? mu : (j < dim ? s.cfric[c][j - 1] : 0.f);
          U[j] = jr[j] * S[j];
          if (j > 0) T += U[j] * U[j];
        }
        const float N = U[0];
        const float Tinv = T > 0 ? fast_rsqrt(T) : 0.f;   // v_rsq_f32, see ls_eval
        T = T * Tinv;
        int st;
        float cc = 0, ef[6] = {0, 0, 0, 0, 0, 0};
        if (N >= mu * T || (T <= 0 && N >= 0)) { st = 0; }
        else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
          st = 1;
#pragma unroll
          for (int j = 0; j < 6; j++)
            if (j < dim) { const float Dj = fast_rcp(Rj[j]); ef[j] = -Dj * jr[j]; cc += 0.5f * Dj * jr[j] * jr[j]; }
        } else {
          st = 4;
          const float Dm = nr.cq[lane][5], NT = N - mu * T, Ti = Tinv;
          cc = 0.5f * Dm * NT * NT;
          const float fn = -Dm * NT * mu;
          ef[0] = fn;
#pragma unroll
          for (int j = 1; j < 6; j++) ef[j] = -fn * Ti * U[j] * S[j];
#if NSAT > 0
          const int hslot = want_hess ? lds_atomic_inc(&s.sat.nch) : NCH;   // a block of the cone-Hessian pool (beyond it: none, see NCH)
          s.sat.chs[c] = hslot < NCH ? (signed char)hslot : (signed char)-1;
          if (want_hess && hslot < NCH) {
            float* H = s.u.n.cH[hslot];
#else
          if (want_hess) {
            float* H = s.u.n.cH[c];
#endif
            const float a = mu * N * Ti * Ti * Ti, b = mu * NT * Ti;
            // stored 6x6 with a fixed stride: S[j] = 0 beyond the contact's condim zero-pads the block by itself, and the
            // readers (XA stage) get immediate offsets instead of condim-dependent addresses
            H[0] = Dm * S[0] * S[0];
#pragma unroll
            for (int j = 1; j < 6; j++) H[j] = H[6 * j] = -mu * U[j] * Ti * Dm * S[0] * S[j];
#pragma unroll
            for (int j = 1; j < 6; j++)
#pragma unroll
              for (int k = 1; k < 6; k++) H[6 * j + k] = (a * U[j] * U[k] - (j == k ? b : 0.f)) * Dm * S[j] * S[k];
          }
        }
        cost[lane] += cc;
#pragma unroll
        for (int j = 0; j < 6; j++)
          if (j < dim) { s.ef[i + j] = ef[j]; s.estate[i + j] = st; }   // block state broadcast through LDS
      }
    } ROWS_END_RO()
    SYNC();
    ROWS_BEGIN(rb, ne) LANES {
      const int i = lane + rb;
      if (i < NEFC) {
        if (nr.type[lane] == CT_CONTACT_ELLIPTIC && i < ne) { nr.force[lane] = s.ef[i]; nr.state[lane] = s.estate[i]; }
        s.estate[i] = (i < ne) ? nr.state[lane] : 0;   // every row's state, for stages that are not mapped lane = row
      }
    } ROWS_END_RW(rb)
    return wave_sum(cost);
  }

  // acc += a * b on two adjacent elements at once (v_pk_fma_f32); the emulator keeps the scalar form
#ifdef SMJ_EMUL
  struct F2 { float x, y; };
  static inline void pk_fma(F2& acc, float a0, float a1, float b0, float b1) { acc.x += a0 * b0; acc.y += a1 * b1; }
#else
  typedef float F2 __attribute__((ext_vector_type(2)));
  SMJ_DEV static void pk_fma(F2& acc, float a0, float a1, float b0, float b1) {
    const F2 a = {a0, a1}, b = {b0, b1};
    acc = acc + a * b;
  }
#endif
  // y = M x for lane-resident x (lane = dof, x zero beyond nv).  Newton path only: factor() is not run there, so MM holds the
  // full symmetric M (both triangles + diagonal, identity beyond nv) and lane i reads its row as it is -- no select per element.
  // Lanes 32..63 mirror lanes 0..31; their y is never used.
  SMJ_DEV void mat_M(PL<float>& y, const PL<float>& x) {
    PL<float[NVS]> m;   // row `lane` of M: all LDS reads are issued before the first use (fixed trip count)
    LANES {
      const int i = NVS == 32 ? (lane & 31) : (lane < NVS ? lane : NVS - 1);   // lanes beyond the matrix: any row, y unused
#pragma unroll
      for (int j = 0; j < NVS; j++) m[lane][j] = s.MM[i][j];
    }
    PL<F2> acc;
    LANES { acc[lane] = F2{0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < NVS; j += 2) {
      const float x0 = wave_read(x, j), x1 = wave_read(x, j + 1);
      LANES { pk_fma(acc[lane], m[lane][j], m[lane][j + 1], x0, x1); }
    }
    LANES { y[lane] = acc[lane].x + acc[lane].y; }
  }
  // out[row] = J[row] . x - sub[row] evaluated with error-free transformations (TwoProduct / TwoSum: ~fp64
  // accuracy from fp32 operations).  The primal residual jar = J qacc - aref cancels to |R f| << |aref|, and the
  // constraint force is D * jar with D = 1/R up to 1e4: a plain fp32 dot product would put 1e-3 noise on the
  // forces.  Only the starting residual needs this; the Newton loop then updates jar incrementally.
  SMJ_DEV void mat_J_exact(PL<float>& out, const PL<float>& x, const PL<float>& sub, int rb) {
    // the row is fetched up front (fixed trip count: columns >= nv of J and entries >= nv of x are zero) and the compensated
    // sum runs as two independent chains (even / odd columns), merged by a last TwoSum
    PL<float[NVS]> a;
    PL<float> hi0, lo0, hi1, lo1;
    LANES {
      const int row = lane + rb < NEFC ? lane + rb : 0;
      const float* jr = jrow(row);
#pragma unroll
      for (int k = 0; k < NVS; k++) a[lane][k] = jr[k];
      hi0[lane] = -sub[lane]; lo0[lane] = 0.f; hi1[lane] = 0.f; lo1[lane] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NVS; k += 2) {
      const float x0 = wave_read(x, k), x1 = wave_read(x, k + 1);
      LANES {
        {
          const float av = a[lane][k];
          const float p = av * x0, pe = fmaf(av, x0, -p);          // p + pe = a*x exactly
          const float t = hi0[lane] + p, z = t - hi0[lane];
          const float se = (hi0[lane] - (t - z)) + (p - z);        // hi + p = t + se exactly
          hi0[lane] = t; lo0[lane] += se + pe;
        }
        {
          const float av = a[lane][k + 1];
          const float p = av * x1, pe = fmaf(av, x1, -p);
          const float t = hi1[lane] + p, z = t - hi1[lane];
          const float se = (hi1[lane] - (t - z)) + (p - z);
          hi1[lane] = t; lo1[lane] += se + pe;
        }
      }
    }
#if NSAT > 0
    // the satellite columns of the row, in the same compensated sum (x = the satellites' qacc)
    LANES {
      const int row = lane + rb < NEFC ? lane + rb : 0;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int si = s.sat.esat[row][u];
        if (si >= 0 && lane + rb < NEFC)
          for (int k = 0; k < 6; k++) {
            const float av = jsp(row, u)[k], xv = s.sat.x[SX_QA][si][k];
            const float p = av * xv, pe = fmaf(av, xv, -p);
            const float t = hi0[lane] + p, z = t - hi0[lane];
            const float se = (hi0[lane] - (t - z)) + (p - z);
            hi0[lane] = t; lo0[lane] += se + pe;
          }
      }
    }
#endif
    LANES {
      const float t = hi0[lane] + hi1[lane], z = t - hi0[lane];
      const float se = (hi0[lane] - (t - z)) + (hi1[lane] - z);
      out[lane] = t + (se + lo0[lane] + lo1[lane]);
    }
  }
  // out[dof] = sum_rows J[row][dof] * f[row]   (lane = dof, f lane-resident over rows), sixteen rows per pass
  SMJ_DEV void matT_J(PL<float>& out, const NRow& nr0) {
    const int ne = ndense();
    LANES { out[lane] = 0.f; }
#pragma unroll
    for (int r0 = 0; r0 < NDR; r0 += 16) {   // (NDR = NEFC unless the build has satellites: only rows 0 .. nd-1 have dense columns)
      if (r0 < 64 ? r0 < ne : __builtin_expect(r0 < ne, 0)) {
        PL<float[16]> a;
        LANES {
#pragma unroll
          // rows >= ne of J are zero.  Lanes 32..63 mirror lanes 0..31 (their `out` is never used): no `lane < NVP` select --
          // that mask, hoisted and spilled, was reloaded with two v_readlane per use
          for (int u = 0; u < 16; u++) a[lane][u] = s.J[r0 + u][NVS == 32 ? (lane & 31) : (lane < NVS ? lane : 0)];
        }
        PL<F2> acc;
        LANES { acc[lane] = F2{0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          const float f0 = r0 < 64 ? wave_read(nr0.force, (r0 & 63) + u) : s.u.n.rxf[RX_FORCE][(r0 >= 64 ? r0 - 64 : 0) + u];   // rows >= 64: LDS
          const float f1 = r0 < 64 ? wave_read(nr0.force, (r0 & 63) + u + 1) : s.u.n.rxf[RX_FORCE][(r0 >= 64 ? r0 - 64 : 0) + u + 1];
          LANES { pk_fma(acc[lane], a[lane][u], a[lane][u + 1], f0, f1); }
        }
        LANES { out[lane] += acc[lane].x + acc[lane].y; }
      }
    }
  }
  // out[row] = J[row] . x   (lane = row - rb, x lane-resident over dofs; columns nv..NVP of J are zero)
  SMJ_DEV void mat_J(PL<float>& out, const PL<float>& x, int rb) {
    PL<float[NVS]> a;
    LANES {
      const int row = lane + rb < NEFC ? lane + rb : 0;
      const float* jr = jrow(row);
#pragma unroll
      for (int k = 0; k < NVS; k++) a[lane][k] = jr[k];
    }
    PL<F2> acc;
    LANES { acc[lane] = F2{0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NVS; k += 2) {
      const float x0 = wave_read(x, k), x1 = wave_read(x, k + 1);
      LANES { pk_fma(acc[lane], a[lane][k], a[lane][k + 1], x0, x1); }
    }
    LANES { out[lane] = acc[lane].x + acc[lane].y; }
  }

  // x <- H^-1 x for the symmetric positive definite H in s.u.n.H (Newton Hessian, or M - h*D of the implicit integrator).
  // Gauss-Jordan elimination with lane i owning row i of [H | x] in registers, fully unrolled.  Why not Cholesky plus two
  // triangular solves: at one wavefront per SIMD a substitution step is a chain of dependent instructions (update ->
  // v_readlane -> scale -> update), measured at ~300 cycles per row and 19k cycles per solve, more than the factorisation
  // itself.  Eliminating above AND below the pivot costs the same n^2/2 (v_readlane, v_fma) pairs as the Cholesky update,
  // all of them independent within a column step, and leaves the solution in x with no substitution and no LDS traffic.
  // Pivots of an SPD matrix stay positive; no pivoting (same as the Cholesky it replaces).
  // The update of pivot K, four columns per step: the four pivot-row entries are fetched first (v_readlane), then two packed fp32
  // FMAs (v_pk_fma_f32) apply them -- two columns per FMA, and enough distance between a v_readlane and the FMA that consumes
  // its SGPR that no hazard s_nop is needed (they were 15 % of the solve's instructions with one column pair per step).
  template <int K, int J, int N>
  SMJ_DEV void gj_pair(PL<float[NVS]>& hrow, const PL<float>& mult) {
    if constexpr (J + 3 < N) {
      PL<float> c0, c1, c2, c3;
      LANES { c0[lane] = hrow[lane][J]; c1[lane] = hrow[lane][J + 1]; c2[lane] = hrow[lane][J + 2]; c3[lane] = hrow[lane][J + 3]; }
      const float h0 = wave_read(c0, K), h1 = wave_read(c1, K), h2 = wave_read(c2, K), h3 = wave_read(c3, K);
#ifdef SMJ_EMUL
      LANES {
        hrow[lane][J] -= mult[lane] * h0; hrow[lane][J + 1] -= mult[lane] * h1;
        hrow[lane][J + 2] -= mult[lane] * h2; hrow[lane][J + 3] -= mult[lane] * h3;
      }
#else
      LANES {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 m = {mult[lane], mult[lane]}, p = {h0, h1}, q = {h2, h3};
        f2 a = {hrow[lane][J], hrow[lane][J + 1]}, b = {hrow[lane][J + 2], hrow[lane][J + 3]};
        a = a - m * p; b = b - m * q;
        hrow[lane][J] = a.x; hrow[lane][J + 1] = a.y; hrow[lane][J + 2] = b.x; hrow[lane][J + 3] = b.y;
      }
#endif
      gj_pair<K, J + 4, N>(hrow, mult);
    } else if constexpr (J + 1 < N) {
      PL<float> c0, c1;
      LANES { c0[lane] = hrow[lane][J]; c1[lane] = hrow[lane][J + 1]; }
      const float h0 = wave_read(c0, K), h1 = wave_read(c1, K);   // pivot-row entries H[K][J], H[K][J+1]
#ifdef SMJ_EMUL
      LANES { hrow[lane][J] -= mult[lane] * h0; hrow[lane][J + 1] -= mult[lane] * h1; }
#else
      LANES {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 m = {mult[lane], mult[lane]}, p = {h0, h1};
        f2 h = {hrow[lane][J], hrow[lane][J + 1]};
        h = h - m * p;
        hrow[lane][J] = h.x; hrow[lane][J + 1] = h.y;
      }
#endif
      gj_pair<K, J + 2, N>(hrow, mult);
    } else if constexpr (J < N) {
      PL<float> cj;
      LANES { cj[lane] = hrow[lane][J]; }
      const float hkj = wave_read(cj, K);              // pivot-row entry H[K][J]
      LANES { hrow[lane][J] -= mult[lane] * hkj; }
    }
  }
  template <int K, int N>
  SMJ_DEV void gj_cols(PL<float[NVS]>& hrow, PL<float>& x, PL<float>& pinv, const PL<int>& ol) {
    if constexpr (K < N) {
      PL<float> col, mult;
      LANES { col[lane] = hrow[lane][K]; }
      const float rp = fast_rcp(fmaxf(wave_read(col, K), 1e-30f));
      LANES {
        // (ol = the lane id made opaque once per solve: the 28-32 `lane == K` masks are then compared where they are used
        // instead of being hoisted out of the step loop, where each would pin -- and spill -- an SGPR pair for the whole kernel)
        mult[lane] = ol[lane] == K ? 0.f : col[lane] * rp;   // the pivot row itself is left alone; its scale is applied at the end
        if (ol[lane] == K) pinv[lane] = rp;
      }
      gj_pair<K, K + 1, N>(hrow, mult);
      const float xk = wave_read(x, K);
      LANES { x[lane] -= mult[lane] * xk; }
      gj_cols<K + 1, N>(hrow, x, pinv, ol);
    }
  }
  // N = matrix order actually eliminated (rows / columns >= nv are identity padding and never touched)
  template <int N>
  SMJ_DEV void gj_solve(PL<float>& x) {
    PL<float[NVS]> hrow;
    PL<float> pinv;
    PL<int> ol;
    LANES { ol[lane] = opaque(lane); }
    LANES {
      // lanes >= N: rows N..31 are identity padding (never touched by the elimination), lanes 32..63 mirror lanes 0..31;
      // their x is zeroed below and scaled by pinv = 0 at the end, so they need no `lane < N` select per element
      const int i = NVS == 32 ? (lane & 31) : (lane < NVS ? lane : NVS - 1);
#pragma unroll
      for (int k = 0; k < N; k++) hrow[lane][k] = s.u.n.H[i][k];
      pinv[lane] = 0.f;
      if (lane >= N) x[lane] = 0.f;
    }
    gj_cols<0, N>(hrow, x, pinv, ol);
    LANES { x[lane] *= pinv[lane]; }
  }
  // The same Gauss-Jordan elimination with the matrix left in LDS (s.u.n.H, destroyed): lane i updates row i in place, the
  // pivot row is read as a broadcast, x rides along as column NVP.  For the 64-dof variant: 50 x 50 rows in registers plus the
  // solver's per-row state do not fit the register file (the unrolled version spilled to scratch and took 30x longer).
  SMJ_DEV void gj_solve_lds(PL<float>& x) {
    const int n = M.nv;
    LANES { if (lane < NVS) s.u.n.H[lane][NVS] = lane < n ? x[lane] : 0.f; }
    SYNC();
    for (int k = 0; k < n; k++) {
      const float rp = fast_rcp(fmaxf(uni(s.u.n.H[k][k]), 1e-30f));
      LANES {
        if (lane < n && lane != k) {
          float* row = s.u.n.H[lane];
          const float* piv = s.u.n.H[k];
          const float mult = row[k] * rp;
          // eight columns per round: the sixteen LDS reads issue back to back (one latency per round, not per column)
          int j = k + 1;
          for (; j + 8 <= n; j += 8) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = row[j + u]; b[u] = piv[j + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) row[j + u] = a[u] - mult * b[u];
          }
          for (; j < n; j++) row[j] -= mult * piv[j];
          row[NVS] -= mult * piv[NVS];
        }
      }
      SYNC();
    }
    LANES { x[lane] = lane < n ? s.u.n.H[lane][NVS] * fast_rcp(fmaxf(s.u.n.H[lane][lane], 1e-30f)) : 0.f; }
    SYNC();
  }
  SMJ_DEV void solve_H(PL<float>& x, bool rare = false) {
#if NVP == 32
    if (M.nv <= 26) gj_solve<26>(x);   // Stretch: 26 dofs; a third fewer column pairs than the full 32
    else gj_solve<NVP>(x);
#else
    // the 64-lane variants: the robot with two free objects (the reference's scene.xml: 38 dofs) or four (50 dofs) in registers,
    // anything larger -- and the call sites that hardly ever run -- with the matrix left in LDS
    if (rare || M.nv > 50) gj_solve_lds(x);
#if NVS >= 50
    else if (M.nv > 38) gj_solve<50>(x);
#endif
    else gj_solve<38>(x);
#endif
  }

  // cost and derivatives along the search line  ([MJ] CGeval); lanes = rows, three wave reductions
  SMJ_DEV float ls_eval(NRow& nr0, const float* qg, float a, float& d1, float& d2) {
    const int ne = nefc;
    PL<float> p0, p1, p2;
    LANES { p0[lane] = 0.f; p1[lane] = 0.f; p2[lane] = 0.f; }
    ROWS_BEGIN(rb, ne) LANES {
      float c0 = 0, c1 = 0, c2 = 0;
      if (lane + rb < ne) {
        const int t = nr.type[lane];
        const float x = nr.jar[lane] + a * nr.jv[lane];
        if (t == CT_EQUALITY) { c0 = nr.q0[lane]; c1 = nr.q1[lane]; c2 = nr.q2[lane]; }
        else if (t == CT_FRICTION) {
          const float fl = nr.fl[lane], rf = nr.R[lane] * fl;
          if (x <= -rf) { c0 = fl * (-0.5f * rf - nr.jar[lane]); c1 = -fl * nr.jv[lane]; }
          else if (x >= rf) { c0 = fl * (-0.5f * rf + nr.jar[lane]); c1 = fl * nr.jv[lane]; }
          else { c0 = nr.q0[lane]; c1 = nr.q1[lane]; c2 = nr.q2[lane]; }
        } else if (t == CT_LIMIT || t == CT_CONTACT_FRICTIONLESS) {
          if (x < 0) { c0 = nr.q0[lane]; c1 = nr.q1[lane]; c2 = nr.q2[lane]; }
        } else if (nr.c0[lane] >= 0) {
          const float* k = nr.cq[lane];
          const float mu = k[6], Dm = k[5], N = k[0] + a * k[1], Tsq = k[2] + a * (2 * k[3] + a * k[4]);
          if (Tsq <= 0) { if (N < 0) { c0 = nr.q0[lane]; c1 = nr.q1[lane]; c2 = nr.q2[lane]; } }
          else {
            // v_rsq_f32 (1 ulp) instead of the IEEE sqrt + divide expansions (~25 instructions per evaluation): the cost is
            // continuous across the zone tests, an ulp of T moves nothing that the 1e-8 solver tolerance can see
            const float Ti = fast_rsqrt(Tsq), T = Tsq * Ti;
            if (N >= mu * T) {}
            else if (mu * N + T <= 0) { c0 = nr.q0[lane]; c1 = nr.q1[lane]; c2 = nr.q2[lane]; }
            else {
              const float w = k[3] + a * k[4], N1 = k[1], T1 = w * Ti, T2 = k[4] * Ti - w * w * Ti * Ti * Ti;
              const float NT = N - mu * T, NT1 = N1 - mu * T1;
              // encode the non-quadratic cone piece so that the caller's quadratic formula reproduces it at this alpha:
              // value v, slope g, curvature h  ->  c2 = h/2, c1 = g - h a, c0 = v - a g + h a a / 2
              const float v = 0.5f * Dm * NT * NT, g = Dm * NT * NT1, h = Dm * (NT1 * NT1 - NT * mu * T2);
              c2 = 0.5f * h; c1 = g - h * a; c0 = v - a * g + 0.5f * h * a * a;
            }
          }
        }
      }
      p0[lane] += c0; p1[lane] += c1; p2[lane] += c2;
    } ROWS_END_RO()
    const float q0 = qg[0] + wave_sum(p0), q1 = qg[1] + wave_sum(p1), q2 = qg[2] + wave_sum(p2);
    d1 = 2 * a * q2 + q1;
    d2 = 2 * q2;
    return a * a * q2 + a * q1 + q0;
  }

  SMJ_DEV void solve_newton(bool dbg, float* pc, long long& t0, bool prof) {
#define TICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - t0); t0 = t1; }
    const int nv = M.nv, ne = nefc;
    NRow nr0;      // per-row registers of rows 0..63; rows 64.. keep theirs in LDS (ROWS_BEGIN)
    PL<float> qacc, Ma, Mv, grad, search, tmpv;
    // The Gauss term is used up to its constant: 0.5 (a-a_s)'M(a-a_s) = 0.5 a'Ma - a'g + const, so neither
    // qacc_smooth nor a factorisation of M is needed on the Newton path (the constant never enters a derivative).
    if (ne == 0) {   // unconstrained: qacc = M^-1 g
      build_dense(false);
      LANES { qacc[lane] = lane < nv ? g_r[lane] : 0.f; }
      solve_H(qacc, true);
      LANES {
        qacc_r[lane] = lane < nv ? qacc[lane] : 0.f;
        if (lane < nv) { s.qacc[lane] = qacc[lane]; s.warm[lane] = qacc[lane]; s.tmp[lane] = g_r[lane]; }
#if NSAT > 0
        if (lane >= 32 && lane - 32 < M.nsat) {   // the satellites: Mb^-1 g on their own lanes
          const int si = lane - 32;
          float A[21], x[6];
          for (int k = 0; k < 21; k++) A[k] = s.sat.Mb[si][k];
          for (int k = 0; k < 6; k++) { x[k] = s.sat.x[SX_G][si][k]; s.sat.x[SX_TMP][si][k] = x[k]; }
          sat_solve6(A, x);
          for (int k = 0; k < 6; k++) s.sat.x[SX_QA][si][k] = k < s.sat.ndof[si] ? x[k] : 0.f;
        }
#endif
      }
      niter = 0;
      SYNC();
      return;
    }
    // per-row constants, efc_vel, aref
    ROWS_BEGIN_(rb, ne, false) LANES {   // the first stage: nothing to load yet
      const int row = lane + rb;
      float vel = 0;
      const bool on = row < ne;
      if (on) {
        const float* jr = jrow(row);
        for (int k = 0; k < nv; k++) vel += jr[k] * s.qvel[k];
#if NSAT > 0
        vel += sat_jdot(row, SX_V);
#endif
      }
      nr.type[lane] = on ? s.etype[row] : CT_NONE;
      nr.R[lane] = on ? s.eR[row] : 1.f;
      nr.D[lane] = on ? 1.0f / s.eR[row] : 0.f;
      nr.fl[lane] = on ? s.efloss[row] : 0.f;
      nr.aref[lane] = on ? -s.eBv[row] * vel - s.eK[row] * s.eimp[row] * (s.epos[row] - s.emargin[row]) : 0.f;
      int c0 = -1;
      if (on && s.etype[row] == CT_CONTACT_ELLIPTIC && s.cefc[s.eid[row]] == row) c0 = s.eid[row];
      nr.c0[lane] = c0;
      nr.state[lane] = 0; nr.jar[lane] = 0.f; nr.jv[lane] = 0.f; nr.force[lane] = 0.f; nr.q0[lane] = 0.f; nr.q1[lane] = 0.f; nr.q2[lane] = 0.f;
      for (int k = 0; k < 7; k++) nr.cq[lane][k] = 0.f;
      if (c0 >= 0) {
        const float mu = contact_mu(c0);
        nr.cq[lane][6] = mu;
        nr.cq[lane][5] = (1.0f / s.eR[row]) / fmaxf(mu * mu * (1 + mu * mu), SMJ_MINVAL);
      }
      if (dbg && row < NEFC) s.earef[row] = nr.aref[lane];
    } ROWS_END_RW(rb)
#if NSAT > 0
    const int nvt = M.nv_all;   // ([MJ] the solver's scale and tolerances use the whole model's dof count)
#else
    const int nvt = nv;
#endif
    const float scale = 1.0f / (M.meaninertia * (float)(nvt > 1 ? nvt : 1));
    // start from qacc_warmstart (MuJoCo also tries qacc_smooth and keeps the cheaper; Newton reaches the same unique
    // optimum from either, so the extra M^-1 g solve is skipped)
    LANES { qacc[lane] = (lane < nv && M.warmstart) ? s.warm[lane] : 0.f; }
    mat_M(Ma, qacc);
#if NSAT > 0
    if (!M.warmstart) { LANES { if (lane >= 32 && lane - 32 < M.nsat) for (int k = 0; k < 6; k++) s.sat.x[SX_QA][lane - 32][k] = 0.f; } }
    sat_matM(SX_MA, SX_QA);
    SYNC();
#endif
    ROWS_BEGIN(rb, ne) mat_J_exact(nr.jar, qacc, nr.aref, rb); ROWS_END_RW(rb)
    float cost = 0;
    TICK(SMJ_PROF_WARM)
    int iter = 0;
    bool at_update = false;   // the loop was left right after a constraint update + gradient: forces and J'f are those of the accepted point
    for (; iter < M.iterations;) {
      TICK(SMJ_PROF_PGS)
      cost = newton_update(nr0, true);
      float gauss;
      {
        PL<float> gs;
        LANES {
          gs[lane] = lane < nv ? qacc[lane] * (0.5f * Ma[lane] - g_r[lane]) : 0.f;
#if NSAT > 0
          if (lane >= 32 && lane - 32 < M.nsat)
            for (int k = 0; k < 6; k++) gs[lane] += s.sat.x[SX_QA][lane - 32][k] * (0.5f * s.sat.x[SX_MA][lane - 32][k] - s.sat.x[SX_G][lane - 32][k]);
#endif
        }
        gauss = wave_sum(gs);
      }
      cost += gauss;
      (void)cost;   // the total cost is not needed: acceptance and termination use derivatives (see the fp32 note below)
      TICK(SMJ_PROF_N_UPDATE)
      // gradient = Ma - g - J'f   (lanes = dofs; force broadcast by readlane)
      matT_J(tmpv, nr0);
#if NSAT > 0
      sat_JTf();   // (handed to the second wavefront of the two-wavefront build as well, round 5: no gain -- the job is shorter than its two barriers)
#endif
      PL<float> g2;
      PL<int> gsig;
      LANES {
        grad[lane] = lane < nv ? Ma[lane] - g_r[lane] - tmpv[lane] : 0.f; g2[lane] = grad[lane] * grad[lane];
        gsig[lane] = fabsf(grad[lane]) > M.grad_noise * (fabsf(Ma[lane]) + fabsf(g_r[lane]) + fabsf(tmpv[lane]));
#if NSAT > 0
        if (lane >= 32 && lane - 32 < M.nsat) {
          const int si = lane - 32;
          for (int k = 0; k < 6; k++) {
            const float ma = s.sat.x[SX_MA][si][k], gg = s.sat.x[SX_G][si][k], jf = s.sat.x[SX_TMP][si][k], gr = k < s.sat.ndof[si] ? ma - gg - jf : 0.f;
            s.sat.x[SX_GRAD][si][k] = gr;
            g2[lane] += gr * gr;
            gsig[lane] |= fabsf(gr) > M.grad_noise * (fabsf(ma) + fabsf(gg) + fabsf(jf));
          }
        }
#endif
      }
      const float gnorm = sqrtf(wave_sum(g2));
      // [MJ] "gradient < tolerance" ends the iteration before the Hessian is built.  fp32: 1e-8 is below the rounding of the
      // gradient's own terms (Ma, g and J'f are ~1e2..1e3 and cancel), so that test never fired and every step paid a last
      // iteration -- Hessian, solve, line search -- whose gain came out as 1e-13 (tolerance 1e-8).  A gradient whose every
      // component is below 64 ulp of the three terms it is the difference of carries no signal a Newton step could use: stop
      // there.  (Measured on the emulator, bench workload, state-synchronised with the oracle: 3.65 -> 3.05 iterations per step,
      // oracle 3.25, one-step velocity error unchanged to three digits; 599 states of the stiff self-collision scenario, same
      // protocol: identical error quantiles.  The scale is the size of the SUMS Ma, g, J'f; the size of the terms of J'f --
      // sum over rows of |J_ri f_r| -- would be the sharper rounding bound where contact forces cancel, but on the device the
      // extra pass over J cost the scenes with free objects more than the iterations it saved: measured, dropped.)
      if (iter > 0 && (scale * gnorm < M.tolerance || wave_ballot(gsig) == 0)) { at_update = true; break; }
      TICK(SMJ_PROF_N_GRAD)
      // H = M + J' W J on the matrix cores, without a weighted copy of J.  W is diagonal (D for rows in the quadratic zone, 0 for
      // satisfied / linear rows) except for the contacts whose block sits in the cone (middle) zone, which carry a dense
      // dim x dim Hessian Hc.  Diagonal part: the operand pair of a k-step is (w_k J[k][:], J[k][:]) -- one LDS read and one
      // multiply per element, the row weights staged in s.ediag (dead since the per-row constants were taken).  Cone part:
      // further k-steps per contact, operands (Hc Jc, Jc) built in registers from the contact's rows.
      ROWS_BEGIN(rb, ne) LANES {
        const int row = lane + rb;
        if (row < NEFC) s.ediag[row] = (row < ne && nr.state[lane] == 1) ? nr.D[lane] : 0.f;
      } ROWS_END_RO()
      uint64_t conemask;
      {
        PL<int> cz;
#if NSAT > 0
        SYNC();
#endif
        LANES {
          int z = 0;
          if (lane < ncon) {
            const int r0 = s.cefc[lane];
            z = r0 >= 0 && s.cdim[lane] >= 3 && s.estate[r0 >= 0 ? r0 : 0] == 4;   // row states as left by newton_update
#if NSAT > 0
            // A sliding contact beyond the cone-Hessian pool (more than NCH of them at once: a toppled robot ploughing through
            // the kitchen) enters H with the diagonal weights of its rows instead of its cone Hessian -- a positive definite
            // stand-in of the same scale.  Left out altogether (the first version), H lacked those contacts' stiffness, the
            // iteration crept and ran into its cap of 100 with a wrong acceleration: a robot lying on its side rose at 2 m/s
            // (found by the long soak of round 4; the search direction stays a descent direction, the line search is exact).
            if (z && s.sat.chs[lane] < 0) {
              for (int j = 0; j < s.cdim[lane]; j++) s.ediag[r0 + j] = 1.0f / s.eR[r0 + j];
              z = 0;
            }
#endif
          }
          cz[lane] = z;
        }
        conemask = wave_ballot(cz);
      }
      SYNC();
      TICK(SMJ_PROF_N_XA)
#if SMJ_W2_NEWTON
      fork2(W2_SAT_NEWTON, (int)(uint32_t)conemask, (int)(uint32_t)(conemask >> 32));   // the satellites' blocks: the second wavefront, beside the main block (helper())
#endif
      // The lower 16x16 tiles over dofs (3 for 32 dofs, 10 for 64), K = constraint rows.  The operands of all tiles are fetched
      // first (J column blocks, shared between tiles) and the accumulation chains are interleaved, so that neither the LDS
      // latency nor the MFMA latency of one tile serialises the others.
      {
        constexpr int NT = (NVS + 15) / 16, NTRI = NT * (NT + 1) / 2, KB = NVP == 32 ? 16 : 8;   // (the last tile may read past column NVS: those products land in rows / columns >= NVS of H, which are not stored)
        const int ksteps = (ndense() + 3) >> 2;   // (rows beyond the dense ones have no main columns)
        PL<F4v> acc[NTRI];
        LANES {
#pragma unroll
          for (int t = 0; t < NTRI; t++)
            for (int r = 0; r < 4; r++) acc[t][lane].r[r] = 0.f;
        }
        // rows 0..63 in batches of KB k-steps, rows 64..NEFC-1 (rare) in further ones, entered only by an env that has them
#pragma unroll
        for (int kb = 0; kb < NDR / 4; kb += KB) {
          if (kb < 16 ? kb < ksteps : __builtin_expect(kb < ksteps, 0)) {
            PL<float[KB]> b[NT], wk;
            LANES {
#pragma unroll
              for (int ks = 0; ks < KB; ks++) {
                const int k = 4 * (kb + ks) + (lane >> 4), c = lane & 15;
                if (kb + ks < NDR / 4) {                       // compile-time: k stays inside J
                  // unconditional: rows >= ne of J are zero, and the k-steps beyond ne are skipped below anyway
                  wk[lane][ks] = s.ediag[k];
#pragma unroll
                  for (int t = 0; t < NT; t++) b[t][lane][ks] = s.J[k][16 * t + c];
                }
              }
            }
#pragma unroll
            for (int ks = 0; ks < KB; ks++) {
              if (kb + ks < ksteps && kb + ks < NDR / 4) {
                PL<float> pa[NT], pb[NT];
                LANES {
#pragma unroll
                  for (int t = 0; t < NT; t++) { pb[t][lane] = b[t][lane][ks]; pa[t][lane] = wk[lane][ks] * pb[t][lane]; }
                }
#pragma unroll
                for (int ta = 0; ta < NT; ta++)
#pragma unroll
                  for (int tb = 0; tb <= ta; tb++) mfma16x16x4(acc[ta * (ta + 1) / 2 + tb], pa[ta], pb[tb]);
              }
            }
          }
        }
        long long th = 0;
        if (prof) { th = smj_clock(); pc[SMJ_PROF_H_K] += (float)(th - t0); }
        // contacts in the cone zone: k = the contact's rows q = 0..dim-1 (zero-padded to 4 or 8), A = (Hc Jc)[q][:], B = Jc[q][:].
        // cH is stored 6 x 6 with zero padding beyond the contact's condim (newton_update), rows past the block meet those zeros.
        int nks = 0;
        for (uint64_t cm = conemask; cm;) {
          const int c = ffs64(cm);
          cm &= cm - 1;
          const int r0 = uni(s.cefc[c]), dim = uni(s.cdim[c]);
          if (r0 >= ndense()) continue;   // (satellite builds: a contact that touches no main body has no main columns)
          for (int q0 = 0; q0 < dim; q0 += 4) {
            nks++;
            PL<float> pa[NT], pb[NT];
            LANES {
              const int q = q0 + (lane >> 4), col = lane & 15, qc = q < 6 ? q : 5;
              const float on = q < dim ? 1.f : 0.f;
              float hq[6];
#if NSAT > 0
              const int hc = s.sat.chs[c];
#else
              const int hc = c;
#endif
#pragma unroll
              for (int p = 0; p < 6; p++) hq[p] = on * s.u.n.cH[hc][6 * qc + p];
#pragma unroll
              for (int t = 0; t < NT; t++) {
                float jv[6], av = 0.f;
#pragma unroll
                for (int p = 0; p < 6; p++) jv[p] = s.J[r0 + p < NDR ? r0 + p : NDR - 1][16 * t + col];
#pragma unroll
                for (int p = 0; p < 6; p++) av += hq[p] * jv[p];
                pa[t][lane] = av;
                pb[t][lane] = on * s.J[r0 + qc < NDR ? r0 + qc : NDR - 1][16 * t + col];
              }
            }
#pragma unroll
            for (int ta = 0; ta < NT; ta++)
#pragma unroll
              for (int tb = 0; tb <= ta; tb++) mfma16x16x4(acc[ta * (ta + 1) / 2 + tb], pa[ta], pb[tb]);
          }
        }
        if (prof) { const long long t1 = smj_clock(); pc[SMJ_PROF_H_CONE] += (float)(t1 - th); th = t1; pc[SMJ_PROF_H_NKS] += (float)nks; }
        LANES {
#pragma unroll
          for (int ta = 0; ta < NT; ta++)
#pragma unroll
            for (int tb = 0; tb <= ta; tb++) {
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const int row = 16 * ta + (lane >> 4) * 4 + r, col = 16 * tb + (lane & 15);
                if (NVS % 16 == 0 || (row < NVS && col < NVS)) {
                  float v = acc[ta * (ta + 1) / 2 + tb][lane].r[r];
                  v += s.MM[row][col];   // the full symmetric M, identity beyond nv (setup); acc is zero there (J columns >= nv are zero)
                  s.u.n.H[row][col] = v;
                  if (ta != tb) s.u.n.H[col][row] = v;
                }
              }
            }
        }
      }
      SYNC();
      if (prof) pc[SMJ_PROF_H_STORE] += (float)(smj_clock() - t0);   // (whole stage: k-steps + cone + store)
      TICK(SMJ_PROF_N_HMFMA)
      LANES { search[lane] = lane < nv ? grad[lane] : 0.f; }
#if NSAT > 0
      // the satellites' blocks; the coupled ones (contacts with the main tree / with each other) extend the dense system of this step
      const long long tsh = prof ? smj_clock() : 0;
#if SMJ_W2_NEWTON
      WG_BARRIER();   // Hb of every satellite and the search direction of the uncoupled ones are there
#else
      sat_hessian(conemask);
      SYNC();
#endif
      if (prof) pc[SMJ_PROF_SAT_H] += (float)(smj_clock() - tsh);
      if (next_sat > 0) {
        sat_extend_hessian(next_sat, conemask);
        LANES {
          if (lane >= NVS && lane < NVS + 6 * next_sat) { const int e = (lane - NVS) / 6; search[lane] = s.sat.x[SX_GRAD][s.sat.xs[e]][lane - NVS - 6 * e]; }
        }
        solve_ext_schur(search, next_sat);
        LANES {
          if (lane >= NVS && lane < NVS + 6 * next_sat) { const int e = (lane - NVS) / 6; s.sat.x[SX_SRCH][s.sat.xs[e]][lane - NVS - 6 * e] = -search[lane]; }
        }
      } else solve_H(search);
#if !SMJ_W2_NEWTON
      sat_solve_own();
#endif
      SYNC();
#else
      solve_H(search);
#endif
      TICK(SMJ_PROF_N_FACTSOLVE)
      PL<float> sq;
      LANES {
        search[lane] = lane < nv ? -search[lane] : 0.f; sq[lane] = search[lane] * search[lane];
#if NSAT > 0
        if (lane >= 32 && lane - 32 < M.nsat)
          for (int k = 0; k < 6; k++) sq[lane] += s.sat.x[SX_SRCH][lane - 32][k] * s.sat.x[SX_SRCH][lane - 32][k];
#endif
      }
      const float snorm = sqrtf(wave_sum(sq));
      TICK(SMJ_PROF_N_SOLVE)
      // line-search preparation  ([MJ] CGprepare)
      mat_M(Mv, search);
#if NSAT > 0
      sat_matM(SX_MV, SX_SRCH);
      SYNC();
      ROWS_BEGIN(rb, ne) {
        mat_J(nr.jv, search, rb);
        LANES { if (lane + rb < ne) nr.jv[lane] += sat_jdot(lane + rb, SX_SRCH); }
      } ROWS_END_RW(rb)
#else
      ROWS_BEGIN(rb, ne) mat_J(nr.jv, search, rb); ROWS_END_RW(rb)
#endif
      float qg[3];
      {
        PL<float> a1, a2;
        LANES {
          a1[lane] = lane < nv ? search[lane] * (Ma[lane] - g_r[lane]) : 0.f;
          a2[lane] = lane < nv ? 0.5f * search[lane] * Mv[lane] : 0.f;
#if NSAT > 0
          if (lane >= 32 && lane - 32 < M.nsat) {
            const int si = lane - 32;
            for (int k = 0; k < 6; k++) {
              a1[lane] += s.sat.x[SX_SRCH][si][k] * (s.sat.x[SX_MA][si][k] - s.sat.x[SX_G][si][k]);
              a2[lane] += 0.5f * s.sat.x[SX_SRCH][si][k] * s.sat.x[SX_MV][si][k];
            }
          }
#endif
        }
        qg[0] = gauss; qg[1] = wave_sum(a1); qg[2] = wave_sum(a2);
      }
      ROWS_BEGIN(rb, ne) LANES {
        const int row = lane + rb;
        const float D = nr.D[lane], ja = nr.jar[lane], jv = nr.jv[lane];
        nr.q0[lane] = 0.5f * D * ja * ja; nr.q1[lane] = D * ja * jv; nr.q2[lane] = 0.5f * D * jv * jv;
        if (row < NEFC) { s.eb[row] = ja; s.ef[row] = jv; s.earef[row] = nr.q0[lane]; s.eK[row] = nr.q1[lane]; s.eBv[row] = nr.q2[lane]; }
      } ROWS_END_RW(rb)
      SYNC();
      ROWS_BEGIN(rb, ne) LANES {
        const int c = nr.c0[lane];
        if (c >= 0) {
          const int i = lane + rb, dim = s.cdim[c];
          float a0 = 0, a1 = 0, a2 = 0, uu = 0, uv = 0, vv = 0;
#pragma unroll
          for (int j = 1; j < 6; j++) {   // fixed trip count, tail masked by a zero weight
            const int jj = j < dim ? j : 1;
            const float on = j < dim ? 1.f : 0.f;
            a0 += on * s.earef[i + jj]; a1 += on * s.eK[i + jj]; a2 += on * s.eBv[i + jj];
            const float fr = on * s.cfric[c][jj - 1], u = s.eb[i + jj] * fr, v = s.ef[i + jj] * fr;
            uu += u * u; uv += u * v; vv += v * v;
          }
          nr.q0[lane] += a0; nr.q1[lane] += a1; nr.q2[lane] += a2;
          const float mu = nr.cq[lane][6];
          nr.cq[lane][0] = nr.jar[lane] * mu; nr.cq[lane][1] = nr.jv[lane] * mu;
          nr.cq[lane][2] = uu; nr.cq[lane][3] = uv; nr.cq[lane][4] = vv;
        }
      } ROWS_END_RW(rb)
      TICK(SMJ_PROF_N_PREP)
      // exact line search: safeguarded Newton on the directional derivative (same scheme as the oracle's ls_search).
      // fp32 note: acceptance and the improvement estimate use derivatives, not cost differences -- near the optimum
      // the decrease is far below one ulp of the cost.
      float alpha = 0, d10 = 0;
      {
        const float gtol = M.tolerance * M.ls_tolerance * snorm * M.meaninertia * (float)(nvt > 1 ? nvt : 1);
        // the derivatives at 0 need no evaluation along the line: d1(0) = grad . search, and with the exact Hessian H search =
        // -grad gives d2(0) = search' H search = -d1(0) -- the first trial point is the full Newton step (MuJoCo evaluates at 0
        // because its CG directions share the code; one evaluation of ~2.3 per iteration saved)
        float d1, d2, lo = 0, hi = -1;
        {
          PL<float> gs;
          LANES {
            gs[lane] = lane < nv ? grad[lane] * search[lane] : 0.f;
#if NSAT > 0
            if (lane >= 32 && lane - 32 < M.nsat)
              for (int k = 0; k < 6; k++) gs[lane] += s.sat.x[SX_GRAD][lane - 32][k] * s.sat.x[SX_SRCH][lane - 32][k];
#endif
          }
          d1 = wave_sum(gs); d2 = -d1;
        }
        d10 = d1;
        float bestd = fabsf(d1), a = 0;
        if (d1 < 0 && d2 > 0) {
          a = -d1 / d2;
          for (int it = 0; it < M.ls_iterations; it++) {
            ls_eval(nr0, qg, a, d1, d2);
            if (prof) pc[SMJ_PROF_N_LSEVALS] += 1.f;
            if (fabsf(d1) < bestd) { bestd = fabsf(d1); alpha = a; }
            // fp32: besides the absolute tolerance, stop once the slope has dropped to 1e-4 of its value at 0 -- along a
            // (piecewise) quadratic the cost still to be gained is then 1e-8 of what this step gained, below the rounding of
            // the next gradient; chasing gtol = 1e-10 * |search| through fp32 noise cost ~7 evaluations per Newton iteration
            if (fabsf(d1) < gtol || fabsf(d1) < 1e-4f * fabsf(d10)) break;
            if (d1 < 0) lo = a; else hi = a;
            float an = (d2 > 0) ? a - d1 / d2 : -1.f;
            if (hi < 0) { if (an <= lo) an = 2 * a + 1e-12f; }
            else if (!(an > lo && an < hi)) an = 0.5f * (lo + hi);
            if (hi >= 0 && hi - lo < 1e-6f * fmaxf(1.f, hi)) break;
            if (fabsf(an - a) <= 1e-7f * fabsf(a)) break;
            a = an;
          }
        }
      }
      TICK(SMJ_PROF_N_LS)
      iter++;
      if (alpha == 0.f) break;
      LANES {
        qacc[lane] += alpha * search[lane]; Ma[lane] += alpha * Mv[lane];
#if NSAT > 0
        if (lane >= 32 && lane - 32 < M.nsat)
          for (int k = 0; k < 6; k++) { s.sat.x[SX_QA][lane - 32][k] += alpha * s.sat.x[SX_SRCH][lane - 32][k]; s.sat.x[SX_MA][lane - 32][k] += alpha * s.sat.x[SX_MV][lane - 32][k]; }
#endif
      }
#if NSAT > 0
      SYNC();
#endif
      if (NSAT > 0 || M.nroot > 1) {
        // scenes with free objects: the residual is re-evaluated from the new qacc with error-free transformations instead
        // of being advanced by alpha * jv.  jv = J search is a plain fp32 product (error ~1e-5 of terms that cancel to 1e-2),
        // and with D = 1/R up to 1e4 that is 0.1 N of force noise per iteration -- invisible on the 20 kg robot, 5 % of the
        // acceleration of a 0.5 kg object (inertia 3e-4).  The line search keeps using jv (it only sets the step length).
        PL<float> qn;
        LANES { qn[lane] = qacc[lane]; }
        ROWS_BEGIN(rb, ne) mat_J_exact(nr.jar, qn, nr.aref, rb); ROWS_END_RW(rb)
      } else {
        ROWS_BEGIN(rb, ne) LANES { nr.jar[lane] += alpha * nr.jv[lane]; } ROWS_END_RW(rb)
      }
      // decrease of the cost along the accepted step, from the slope at 0 (exact for the quadratic pieces)
      if (scale * (-0.5f * alpha * d10) < M.tolerance) break;
    }
    niter = iter;
    TICK(SMJ_PROF_PGS)
    // final forces at the accepted point, qfrc_constraint = J' f
    if (!at_update) {
      newton_update(nr0, false);
      matT_J(tmpv, nr0);
#if NSAT > 0
      SYNC();
      sat_JTf();
#endif
    }
    LANES {
      qacc_r[lane] = lane < nv ? qacc[lane] : 0.f;
      if (lane < nv) { s.qacc[lane] = qacc[lane]; s.warm[lane] = qacc[lane]; s.tmp[lane] = g_r[lane] + tmpv[lane]; }
#if NSAT > 0
      if (lane >= 32 && lane - 32 < M.nsat)   // qfrc_smooth + qfrc_constraint of the satellite: the integrator's right-hand side
        for (int k = 0; k < 6; k++) s.sat.x[SX_TMP][lane - 32][k] += s.sat.x[SX_G][lane - 32][k];
#endif
    }
    SYNC();
    if (dbg && S.debug) {
      LANES {
        if (lane < nv) S.debug[(SMJ_DBG_QACC + lane) * S.ld + env] = qacc[lane];
        const NRow& nr = nr0;   // the debug layout holds the first 64 rows
        S.debug[(SMJ_DBG_EFC_FORCE + lane) * S.ld + env] = lane < ne ? nr.force[lane] : 0.f;
        S.debug[(SMJ_DBG_EFC_R + lane) * S.ld + env] = lane < ne ? nr.R[lane] : 0.f;
        S.debug[(SMJ_DBG_EFC_AREF + lane) * S.ld + env] = lane < ne ? nr.aref[lane] : 0.f;
      }
    }
#undef TICK
  }

  // dense symmetric NVP x NVP copy of M (implicit = false) or of M - h*D (implicit = true) into s.u.n.H, identity padded
  SMJ_DEV void build_dense(bool implicit) {
    const float h = M.timestep;
    EntryTab et;
    load(et, implicit);
    LANES {
      for (int k = lane; k < NVS * (NVS + 1); k += 64) {
        const int r = k / (NVS + 1), c = k - r * (NVS + 1);
#if NSAT > 0
        s.u.n.H[r][c] = (r == c && r >= M.nv) ? 1.f : 0.f;   // (the row stride of H is that of the extended system here)
#else
        (&s.u.n.H[0][0])[k] = (r == c && r >= M.nv) ? 1.f : 0.f;
#endif
      }
    }
    SYNC();
    LANES {
      for (int t = 0; t < NENT; t++) {
        const int i = et.i[lane][t], j = et.j[lane][t];
        if (i < 0) continue;
        float v = (i == j) ? s.Mdiag[i] : s.MM[j][i];
        if (implicit) {
          float d = et.damp[lane][t] - et.dcoef[lane][t];
          const int la = et.lact[lane][t];
          if (la >= 0) d -= et.lcoef[lane][t] * s.act_free[la];
          v += h * d;
        }
        s.u.n.H[i][j] = v;
        s.u.n.H[j][i] = v;
      }
    }
    SYNC();
  }

  // ------------------------------------------------------------------ B.8 implicitfast integrate
  // (M - h*D) qacc' = qfrc_smooth + qfrc_constraint, D = d(passive + actuator force)/d(qvel)  ([MJ] mj_implicit, fast)
  SMJ_DEV void integrate() {
    const int nv = M.nv;
    const float h = M.timestep;
#if SMJ_W2_INTEGRATE
    fork2(W2_SAT_INTEGRATE);   // the satellites' own integration: the second wavefront, beside the main tree's (helper())
#endif
    build_dense(true);
    PL<float> x;
    LANES { x[lane] = lane < nv ? s.tmp[lane] : 0.f; }
    solve_H(x);
    LANES {
      if (lane < nv) s.qvel[lane] += h * x[lane];
    }
    SYNC();
    LANES {
      if (lane < M.njnt) {
        const int j = lane, qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
        if (M.jnt_type[j] == JT_FREE) {
          for (int k = 0; k < 3; k++) s.qpos[qa + k] += h * s.qvel[da + k];
          float wv[3] = {s.qvel[da + 3], s.qvel[da + 4], s.qvel[da + 5]};
          const float nrm = sqrtf(dot3(wv, wv)), ang = nrm * h;
          float q[4] = {s.qpos[qa + 3], s.qpos[qa + 4], s.qpos[qa + 5], s.qpos[qa + 6]};
          if (ang > 0) {
            const float sn = sinf(0.5f * ang) / nrm, dq[4] = {cosf(0.5f * ang), wv[0] * sn, wv[1] * sn, wv[2] * sn};
            quat_mul(q, q, dq);
          }
          quat_normalize(q);
          for (int k = 0; k < 4; k++) s.qpos[qa + 3 + k] = q[k];
        } else s.qpos[qa] += h * s.qvel[da];
      }
    }
    SYNC();
    // [MJ] mj_checkPos / mj_checkVel: a non-finite or absurd state resets the environment (and is flagged)
    PL<int> bad;
    LANES {
      int b = 0;
      for (int k = lane; k < M.nq; k += 64) { const float q = s.qpos[k]; b |= !(fabsf(q) < 1e10f); }
      if (lane < nv) { const float v = s.qvel[lane]; b |= !(fabsf(v) < 1e10f); }
      bad[lane] = b;
    }
#if SMJ_W2_INTEGRATE
    WG_BARRIER();
    LANES { bad[lane] |= s.mbox[2]; }
#elif NSAT > 0
    sat_integrate(bad);
#endif
    if (wave_ballot(bad) != 0) {
      flags |= SMJ_FLAG_BAD_STATE;
      LANES {
        for (int k = lane; k < M.nq; k += 64) s.qpos[k] = M.qpos0[k];
        if (lane < nv) { s.qvel[lane] = 0.f; s.warm[lane] = 0.f; }
      }
#if NSAT > 0
      sat_reset_state();
#endif
      SYNC();
    }
  }

  // IMU: gyro + accelerometer at the IMU site from the last forward pass  [MJ] mj_sensorVel / mj_sensorAcc
  SMJ_DEV void imu() {
    if (M.imu_site < 0 || (!S.gyro && !S.stage)) return;
    const int sid = M.imu_site, b = M.site_bodyid[sid];
    const uint64_t mk = mk64(M.k_body_dofmask_lo[b], M.k_body_dofmask_hi[b]);
    float cv[6], ca[6];
    for (int x = 0; x < 6; x++) {
      PL<float> tv, ta;
      LANES {
        const bool in = (mk >> lane) & 1;
        tv[lane] = in ? cdof[lane][x] * qvel_r[lane] : 0.f;
        ta[lane] = in ? cdof_dot[lane][x] * qvel_r[lane] + cdof[lane][x] * qacc_r[lane] : 0.f;
      }
      cv[x] = wave_sum(tv); ca[x] = wave_sum(ta);
    }
    for (int k = 0; k < 3; k++) ca[3 + k] -= M.gravity[k];
    float lp[3] = {M.site_pos[3 * sid], M.site_pos[3 * sid + 1], M.site_pos[3 * sid + 2]}, sp[3], lm[9], R[9];
    mulmat3vec(sp, s.xmat[b], lp);
    for (int k = 0; k < 9; k++) lm[k] = M.k_site_mat[9 * sid + k];
    mulmat3(R, s.xmat[b], lm);
    const float off[3] = {sp[0] + s.xpos[b][0] - s.com[b][0], sp[1] + s.xpos[b][1] - s.com[b][1], sp[2] + s.xpos[b][2] - s.com[b][2]};
    float t[3], vlin[3], alin[3], c2[3], gy[3], ac[3];
    cross3(t, cv, off);
    for (int k = 0; k < 3; k++) vlin[k] = cv[3 + k] + t[k];
    cross3(t, ca, off);
    for (int k = 0; k < 3; k++) alin[k] = ca[3 + k] + t[k];
    cross3(c2, cv, vlin);
    for (int k = 0; k < 3; k++) alin[k] += c2[k];
    mulmat3Tvec(gy, R, cv);
    mulmat3Tvec(ac, R, alin);
    if (S.stage) {
      float* st = stage_row();
      LANES { if (lane < 3) { st[S.lay.gyro + lane] = gy[lane]; st[S.lay.accel + lane] = ac[lane]; } }
    } else {
      LANES { if (lane < 3) { S.gyro[lane * S.ld + env] = gy[lane]; S.accel[lane * S.ld + env] = ac[lane]; } }
    }
  }

  // body poses of the last step for the ray-casting kernels (smj_render.hip: lidar, depth cameras): what mj_sensorPos /
  // mjv_updateScene read from mjData
  SMJ_DEV void dump_poses() {
    if (S.stage) {   // contiguous: 12 words per body
      float* st = stage_row() + S.lay.xpose;
      LANES {
        for (int k = lane; k < 12 * (M.nbody + M.nsat); k += 64) {
          const int b = k / 12, c = k - 12 * b;
          st[k] = c < 3 ? s.xpos[b][c] : s.xmat[b][c - 3];
        }
      }
      return;
    }
    if (!S.xpose) return;
    LANES {
      if (lane < M.nbody + M.nsat) {
        for (int k = 0; k < 3; k++) S.xpose[(long)(12 * lane + k) * S.ld + env] = s.xpos[lane][k];
        for (int k = 0; k < 9; k++) S.xpose[(long)(12 * lane + 3 + k) * S.ld + env] = s.xmat[lane][k];
      }
    }
  }
  SMJ_DEV void dump_debug() {
    if (!S.debug) return;
    LANES {
      for (int i = 0; i < NVP; i++)
        if (lane < NVP) {
          float v = 0;
          if (i < M.nv && lane < M.nv) v = (i == lane) ? s.Mdiag[i] : (i < lane ? s.MM[i][lane] : s.MM[lane][i]);
          S.debug[(SMJ_DBG_QM + i * NVP + lane) * S.ld + env] = v;
        }
      if (lane < NBP)
        for (int k = 0; k < 3; k++) S.debug[(SMJ_DBG_XPOS + 3 * lane + k) * S.ld + env] = lane < M.nbody ? s.xpos[lane][k] : 0.f;
    }
  }
  SMJ_DEV void dump_contacts() {
    if (!S.debug) return;
    LANES {
      if (lane < NCON) {
        const int c = lane;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < ncon) {
          v[0] = s.cdist[c]; v[1] = s.cpos[c][0]; v[2] = s.cpos[c][1]; v[3] = s.cpos[c][2];
          v[4] = s.cframe[c][0]; v[5] = s.cframe[c][1]; v[6] = s.cframe[c][2];
          v[7] = (float)(s.cdim[c] + 16 * s.cgeom1[c] + 16 * 1024 * s.cgeom2[c]);   // condim | geom1 << 4 | geom2 << 14 (exact in fp32)
        }
        for (int k = 0; k < 8; k++) S.debug[(SMJ_DBG_CON + 8 * c + k) * S.ld + env] = v[k];
      }
    }
  }

#if NSAT > 0
#include "smj_sat.h"
#endif

  // ------------------------------------------------------------------ driver
  // Which solver a build carries: both (chosen per launch by DevModel::solver) or, where a translation unit says so, ONE -- the
  // standard variant is built twice (smj_kernels.hip: Newton only, smj_kernels_pgs.hip: PGS only) because the other solver's code, dead
  // in a launch, still costs the live one registers (the allocation is per kernel).
#if defined(SMJ_ONLY_NEWTON)
  SMJ_DEV bool newton() const { return true; }
#elif defined(SMJ_ONLY_PGS)
  SMJ_DEV bool newton() const { return false; }
#else
  SMJ_DEV bool newton() const { return M.solver == 2; }
#endif
  SMJ_DEV void run(int nsteps, unsigned read_flags) {
    const int want_imu = read_flags & 1;
    flags = 0; nefc = NEFC; ncon = 0; niter = 0;   // nefc = NEFC: the first make_constraint clears every row
    float pc[SMJ_PROF_SLOTS];
    for (int k = 0; k < SMJ_PROF_SLOTS; k++) pc[k] = 0.f;
    const bool prof = SMJ_PROFILING && S.prof != nullptr;
    const long long tlaunch = S.cost ? smj_clock() : 0;
    long long t0 = prof ? smj_clock() : 0, tstart = t0;
#define TICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - t0); t0 = t1; }
    setup();
    load_state();
    TICK(SMJ_PROF_SETUP)
    for (int st = 0; st < nsteps; st++) {
      const bool last = st == nsteps - 1;
#if SMJ_W2_FORWARD
      fork2(W2_SAT_FORWARD, env);   // the satellites' forward pass: the second wavefront, beside the main tree's kinematics (helper())
      kinematics();
      WG_BARRIER();
#else
      kinematics();
#if NSAT > 0
      sat_forward();   // pose, mass block, smooth forces of every satellite (one lane each)
#endif
#endif
      if (last && (read_flags & 4)) dump_poses();
      TICK(SMJ_PROF_KIN)
      com_crb();
      if (last) dump_debug();
      TICK(SMJ_PROF_COMCRB)
      smooth_forces(last);
      TICK(SMJ_PROF_SMOOTH)
      if (!newton()) factor();   // the sparse L'DL of M is only needed by the PGS path (Y = J L^-1)
      TICK(SMJ_PROF_FACTOR)

      collision();
      {
        // More contacts than slots?  With two wavefronts side by side WHICH claims failed depends on their timing (that the list
        // overflowed does not: the total is what it is), and which manifolds were kept (mcache) up to the overflow depends on the same.
        // So the stage is run again by this wavefront alone, pairs in table order -- planes, static world, moving pairs: the truncated
        // list is the same on every run and in the one- and two-wavefront builds -- after the env's kept manifolds have been dropped;
        // the one-wavefront builds do the same, so that the two stay bit for bit equal through such a step (which the larger build may
        // still finish unflagged: 57 .. 64 contacts).  Rare: <= 0.03 % of the kitchen's env-steps.  (One call site in a loop: two would
        // inline the stage twice.)
        const int ncon_planes = ncon;
        const unsigned flags_planes = flags;
#pragma nounroll
        for (int pass = 0; pass < 2; pass++) {
          collision_convex(pc, prof && pass == 0);
          if (pass || !((flags & ~flags_planes) & SMJ_FLAG_CON_OVERFLOW) || !(SMJ_SPLIT_COLLIDE || (S.mcache && M.manifold_cache))) break;
          if (S.mcache) {
            LANES { if (lane < SMJ_MC_SLOTS) S.mcache[((size_t)env * SMJ_MC_SLOTS + lane) * SMJ_MC_WORDS] = 0.f; }
          }
          SYNC();
          ncon = ncon_planes;
          flags = flags_planes;
          serial_redo = true;
        }
        serial_redo = false;
      }
      if (last) dump_contacts();
      TICK(SMJ_PROF_COLLISION)
#if NSAT > 0
      make_constraint_sat(pc, prof);
#else
      if (!newton()) nefc = NEFC;   // PGS: the A of a step with at most 64 rows sits in rows 64.. of J (solve<false>) -- have them cleared
      make_constraint();
#endif
      TICK(SMJ_PROF_MAKECON)
      if (S.redo && !S.redo_worker && (flags & (SMJ_FLAG_EFC_OVERFLOW | SMJ_FLAG_CON_OVERFLOW))) {
        escalate(st);
        if (S.cost) { const int cst = (int)((smj_clock() - tlaunch) >> 6); LANES { if (lane == 0) st_coh(&S.cost[env], step_base ? ld_coh(&S.cost[env]) + cst : cst); } }
        return;
      }
#if NSAT > 0
      if (newton()) solve_newton(last, pc, t0, prof);
      else solve_pgs_sat(last, pc, t0, prof);   // islands: the dense system + one lane per uncoupled satellite (smj_sat_pgs.h)
      if (last && S.debug) {
        LANES {
          if (lane >= 32 && lane - 32 < M.nsat)
            for (int k = 0; k < 6; k++) S.debug[(SMJ_DBG_SATQACC + 6 * (lane - 32) + k) * S.ld + env] = s.sat.x[SX_QA][lane - 32][k];
        }
      }
#else
      if (newton()) solve_newton(last, pc, t0, prof);
      else if (nefc > NEFP) solve<true>(last, pc, t0, prof);
      else solve<false>(last, pc, t0, prof);
#endif
      if (last && want_imu) imu();
      TICK(SMJ_PROF_POST)
      integrate();
      base_controller();   // _ctrl_callback order: after mj_step, with the pose of its forward pass; sets the next step's wheel ctrl
      TICK(SMJ_PROF_INTEGRATE)
      pc[SMJ_PROF_PGS_SWEEPS] += (float)niter;
    }
    store_state(nsteps);
    if (S.cost) {   // shader time of this env's launch (units of 64 clocks): the key of the next launch's order (DevState::order)
      const int cst = (int)((smj_clock() - tlaunch) >> 6);
      LANES { if (lane == 0) st_coh(&S.cost[env], (S.redo_worker || step_base) ? ld_coh(&S.cost[env]) + cst : cst); }
    }
    if (prof) {
      pc[SMJ_PROF_TOTAL] = (float)(smj_clock() - tstart);
      LANES { if (lane < SMJ_PROF_SLOTS) S.prof[lane * S.ld + env] = pc[lane]; }
    }
#undef TICK
  }
};
