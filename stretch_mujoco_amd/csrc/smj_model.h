// Device-side view of the compiled (fused) model: plain pointers to fp32 / int32 arrays, named exactly as in
// the SMJB blob (model_blob.py, model_fuse.py).  Filled by smj_create() (smj_capi.hip) on the GPU and by the
// test-only lane emulator on the CPU.
#pragma once
#include <stdint.h>

// Three capacity variants of the step kernel are compiled from the same source (smj_kernels.hip, smj_kernels_tall.hip,
// smj_kernels_big.hip):
//   standard -- the robot alone or with ONE free object: 32 dofs, 80 constraint rows, 16 contacts; 40 KB of LDS per env,
//               four envs per CU;
//   tall (SMJ_TALL) -- the same 32 dofs with 160 rows and 48 contacts (~80 KB of LDS, two envs per CU): contact-rich scenes
//               (kitchen fixtures all around the robot), and the escalation target of the standard variant -- an env whose
//               step needs more rows / contacts than the standard kernel holds is finished by this one (DevState::redo);
//   big (SMJ_BIG) -- scenes with several free objects (the reference's own scene.xml: table + 2 objects; kitchens):
//               64 dofs, 160 rows, 48 contacts; ~130 KB of LDS per env, one env per CU.
// smj_create picks the variant from the model's dimensions (and the model compiler's capacity hint).
// Satellite builds (SMJ_SAT = satellite capacity, csrc/smj_sat.h, model_fuse.find_satellites): the MAIN tree (the robot: 32 dof lanes /
// columns, as the standard variant) plus up to SMJ_SAT single-body mechanisms -- free objects, doors, drawers -- that meet it
// only through contacts.  Each satellite is one LANE (32 + s) with its own 6 x 6 mass block; constraint rows that touch the main
// tree come first and are the only ones with a dense Jacobian row (NDR of them), every row has two 6-column satellite slots.
#if defined(SMJ_SAT)
#define NSAT SMJ_SAT
#define NVP 32
#ifndef SMJ_SAT_ROWS
#define SMJ_SAT_ROWS 320
#define SMJ_SAT_CONTACTS 64
#define SMJ_SAT_DENSE 256
#endif
#define NEFC SMJ_SAT_ROWS
#define NCON SMJ_SAT_CONTACTS
#define NDR SMJ_SAT_DENSE     // rows with a dense Jacobian row (rows 0 .. nd-1 of a step: the rows that touch the main tree)
#ifndef SMJ_SAT_EXT
#define SMJ_SAT_EXT 4
#endif
#ifndef SMJ_SAT_PGS
#define SMJ_SAT_PGS (SMJ_SAT_DENSE < 160 ? SMJ_SAT_DENSE : 160)   // rows of the PGS path's dense system (packed A in LDS)
#endif
#define NXS SMJ_SAT_EXT       // satellites that may be coupled to the main tree / to each other in one step (dense extension of the Newton system)
static_assert(NDR % 16 == 0 && NDR <= NEFC, "dense rows: whole 16-row blocks (matT_J, MFMA k-steps)");
#define NENT 5
#elif defined(SMJ_BIG)
#define NVP 64    // dof LANES (lane = dof stages run on 64 lanes)
#ifndef SMJ_NVS
#define SMJ_NVS 64
#endif
#define NVS SMJ_NVS   // dof COLUMNS of the LDS-resident matrices J / M / H (model nv <= NVS): the big variant is built three times,
                      // for 38 (the reference's scene.xml: robot + 2 free objects), 50 (robot + 4) and 64 dofs -- with 38 / 50
                      // columns an env's working set stays under 80 KB of LDS and two envs share a CU
#if SMJ_NVS == 64   // one env per CU anyway: the build that takes over when a 38- / 50-column env runs out of rows / contacts
#define NEFC 224
#define NCON 64
#elif defined(SMJ_BIG_ROWS)   // (capacities chosen per build so that the env's LDS is a whole number of 1280-byte granules a CU holds three of)
#define NEFC SMJ_BIG_ROWS
#define NCON SMJ_BIG_CONTACTS
#else
#define NEFC 160  // constraint-row capacity: rows 64.. take further passes on lanes 0..63
#define NCON 48   // contact capacity (<= 64: contact stages are lane = contact)
#endif
#define NENT 8    // mass-matrix pattern entries per lane (64 lanes)
#elif defined(SMJ_TALL)
#define NVP 32
#ifdef SMJ_TALL_ROWS   // the three-envs-per-CU build of the tall variant (smj_kernels_mid.hip)
#define NEFC SMJ_TALL_ROWS
#define NCON SMJ_TALL_CONTACTS
#else
#define NEFC 160
#define NCON 48
#endif
#define NENT 5
#else
#define NVP 32
#define NEFC 80
#define NCON 16
#define NENT 5
#endif
#ifndef NVS
#define NVS NVP
#endif
#ifndef NSAT
#define NSAT 0
#endif
#ifndef NDR
#define NDR NEFC
#endif
static_assert(NVS % 2 == 0 && NVS <= NVP && (NVP == 64 ? NVS >= 38 : NVS == NVP), "column capacity: even (packed pairs), odd row stride NVS + 1");
#define NBP 32   // fused-body capacity of the main tree (lane = body stages)
#define NBT (NBP + NSAT)   // body slots: main tree + satellites (body nbody + s = satellite s)
#define NCG 128  // geoms that take part in non-plane collision pairs (world-frame cache of the broadphase)

#define SMJ_MODEL_I32(X)                                                                                          \
  X(body_parentid) X(k_body_jump) X(body_rootid) X(body_jntadr) X(body_jntnum) X(body_dofadr) X(body_dofnum) X(k_body_level)     \
  X(k_body_subtreesize) X(k_body_dofmask_lo) X(k_body_dofmask_hi) X(k_root_list) X(k_gc_body)                     \
  X(jnt_type) X(jnt_qposadr) X(jnt_dofadr) X(jnt_bodyid) X(jnt_limited)                                           \
  X(dof_bodyid) X(dof_jntid) X(dof_parentid) X(k_dof_anc_adr) X(k_dof_anc_num) X(k_dof_anc) X(k_dof_velmask_lo)   \
  X(k_dof_velmask_hi) X(k_dof_qposadr) X(k_ldl_i) X(k_ldl_j) X(k_ldl_lact) X(k_fric_dof) X(k_limit_jnt) X(k_act_dof) X(k_dof_act)                         \
  X(geom_type) X(geom_bodyid) X(geom_hulladr) X(geom_hullnum)                                                     \
  X(site_bodyid)                                                                           \
  X(eq_obj1id) X(eq_obj2id) X(eq_active)                                                                          \
  X(actuator_trntype) X(actuator_trnid) X(actuator_ctrllimited) X(actuator_forcelimited) X(actuator_biastype)     \
  X(pair_geom1) X(pair_geom2) X(pair_condim) X(k_planepair) X(k_cgeom) X(k_convpair) X(k_convpair_s1) X(k_convpair_s2) X(k_convpair_ss)

#define SMJ_MODEL_F32(X)                                                                                          \
  X(body_pos) X(body_quat) X(k_body_inertia_local) X(body_gcmass) X(body_gcipos) X(body_subtreemass)              \
  X(jnt_pos) X(jnt_axis) X(jnt_stiffness) X(jnt_range) X(jnt_margin) X(jnt_solref) X(jnt_solimp) X(qpos0)         \
  X(qpos_spring)                                                                                                  \
  X(dof_armature) X(dof_damping) X(dof_frictionloss) X(dof_invweight0) X(dof_solref) X(dof_solimp)                \
  X(geom_pos) X(k_geom_mat) X(geom_size) X(geom_rbound) X(k_geom_bcenter) X(geom_rgba) X(geom_invweight0)         \
  X(k_hull_vert4) X(geom_aabb) X(geom_ccenter) X(k_convpair_rsum) X(k_cgeom_half) X(k_cgeom_lcen)                                                                             \
  X(site_pos) X(k_site_mat)                                                                                       \
  X(eq_data) X(eq_solref) X(eq_solimp)                                                                            \
  X(actuator_gear) X(actuator_gainprm) X(actuator_biasprm) X(actuator_ctrlrange) X(actuator_forcerange)           \
  X(k_act_moment) X(key_ctrl) X(k_ldl_damp) X(k_ldl_dcoef) X(k_ldl_lcoef) X(k_act_mom) X(k_dof_actmom)                                                                                     \
  X(pair_friction) X(pair_solref) X(pair_solimp) X(pair_margin) X(pair_gap)

struct DevModel {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, neq, npair, nlevel, nfric, nlimit, nplanepair, nldl, nlidar, imu_site,
      ngc, nroot, nkey, ncgeom, nconvpair, njump, maxsubtree;
  int iterations, warmstart, pgs_fixed_iter, max_con_pair, solver /* 0 PGS, 2 Newton */, ls_iterations, convex_pairs;
  float grad_noise;       // Newton: a gradient component below grad_noise * (|Ma| + |g| + |J'f|) is rounding (default 4e-6 = 64 ulp; 0 = MuJoCo's scale * |grad| < tolerance test only)
  int pgs_island_stop;    // PGS, satellite builds: 1 (default) = a satellite island whose own scaled improvement fell below tolerance / 64 stops sweeping (smj_sat_pgs.h); 0 = every island sweeps until the whole system stops
  int qcqp_exact;   // PGS: 1 = the friction QCQP iterates exactly as mju_QCQP does (from la = 0 on |x|^2 - r^2, cap 20); 0 (default) = same root, secular form, started at the last sweep's multiplier
  int pgs_dual_ws;  // PGS: 1 (default) = the sweeps may also start from the forces the rows had at the end of the previous step's solve (DevState::pgsprev; NOT MuJoCo's rule, same fixed point: smj_sat_pgs.h), 0 = MuJoCo's warm start only
  int sep_cache;      // 1 = use DevState::sepcache (separating directions of convex pairs kept between steps)
  int manifold_cache; // 1 = use DevState::mcache (contact manifolds of convex pairs whose two bodies have not moved)
  int multi_serial;   // lane emulator only: 1 = the four multiccd queries of a pair one after the other (convex_multi), the comparator of convex_multi4
  int pgs_cap;    // > 0 (set per launch, smj_step_tu.h): a PGS step of this launch holds at most this many rows -- its packed A then starts at row pgs_cap of J and ends inside the struct (no dynamic LDS); a step with more rows goes to the escalation variant
  int row_limit;  // > 0: the primary variant hands an env over to the escalation variant beyond this many constraint rows (tests; option "primary_rows")
  int multiccd;   // stretch.xml:8 <flag multiccd="enable"/>: multi-point contacts for convex pairs (box-box polygon, counter-rotated queries)
  float ls_tolerance;
  float timestep, gravity[3], impratio, tolerance, meaninertia, lidar_cutoff;
  // satellites: nq / nv / nbody / njnt above are those of the MAIN part (a prefix of the model's coordinates); the *_all counts
  // cover the whole model (equal to the main ones when nsat = 0)
  int nsat, nq_all, nv_all, nbody_all, nfric_main, nlimit_main;
  const int* k_satrec;    // [nsat][SMJ_SR_STRIDE], floats as bit patterns
  // static collision geometry (the world body's geoms: a kitchen's fixtures) behind a uniform grid (model_fuse._static_grid_tables)
  int nsgeom, nstatpair, grid_dim[3];
  float grid_org[3], grid_h, grid_margin;
  const int* k_sgrec;     // [nsgeom][SMJ_CG_STRIDE]: as k_cgrec, body 0 -- local frame = world frame
  const int* k_sprec;     // [nstatpair][SMJ_CP_STRIDE]: as k_cprec; the static geom's slot is -1 - (its index)
  const int* k_spair;     // [ncgeom][nsgeom] -> index into k_sprec, or -1 (pair filtered out)
  const int* k_grid_adr;  // [cells + 1]
  const int* k_grid_list; // static-geom indices, cell after cell
  const int* k_sg_cell;   // [nsgeom][6] cell range of the geom (lo xyz, hi xyz)
  const float* k_sg_bound;   // [nsgeom][8] world AABB of the geom's oriented box: lo xyz, pad, hi xyz, pad
  const float* k_sgw;        // [nsgeom][32] the geom in the world frame, in the order of the collision cache's arrays: pos 3, mat 9, box centre 3, half sizes 3, MPR centre 3, size 3, meta (bits)
#define X(n) const int* n;
  SMJ_MODEL_I32(X)
#undef X
#define X(n) const float* n;
  SMJ_MODEL_F32(X)
#undef X
  // Per-lane stage records, built at load time from the tables above (smj_build_lanerec): everything the kernel's stage-table
  // loaders need for lane L, already gathered (no dependent loads) and contiguous -- one base pointer and a few wide loads
  // instead of ~60 table pointers and three levels of pointer chasing.  Floats are stored as their bit patterns.
  const int* k_lanerec;   // [64][SMJ_LR_STRIDE]
  // Same idea for the collision stage: one record per plane pair (table order) and per geom of the convex-pair cache.
  const int* k_pprec;     // [nplanepair][SMJ_PP_STRIDE]
  const int* k_cgrec;     // [ncgeom][SMJ_CG_STRIDE]
  // Constraint assembly: one record per static row (equalities, then friction-loss dofs -- rows 0..neq+nfric-1 of every step)
  // followed by one per limit slot (2 per limited joint: lower, upper side)
  const int* k_rowrec;    // [neq + nfric + 2 nlimit][SMJ_RR_STRIDE]
  // Convex pairs that reach the narrowphase: pair, geoms, cache slots, margin and the contact parameters in one record
  const int* k_cprec;     // [nconvpair][SMJ_CP_STRIDE]
};
enum { SMJ_CP_PAIR = 0, SMJ_CP_G1, SMJ_CP_G2, SMJ_CP_S1, SMJ_CP_S2, SMJ_CP_MARGIN, SMJ_CP_MG, SMJ_CP_CONDIM, SMJ_CP_FRIC = 8, SMJ_CP_SOLIMP = 13,
       SMJ_CP_SOLREF = 18, SMJ_CP_RBMIN = 20 /* min bounding radius of the two geoms (multiccd duplicate tolerance) */,
       SMJ_CP_B1 = 21, SMJ_CP_B2 = 22 /* bodies of the two geoms */, SMJ_CP_STRIDE = 24 };
// row record: type, id (equality / dof / joint), dofs (limit: dof, side), qpos addresses, reference values (equality: qpos0 of
// both joints; limit: range bound of the side, margin), equality polynomial, diagonal approximation, friction loss, solref, solimp
enum { SMJ_RR_TYPE = 0, SMJ_RR_ID, SMJ_RR_D1, SMJ_RR_D2, SMJ_RR_Q1, SMJ_RR_Q2, SMJ_RR_V1, SMJ_RR_V2, SMJ_RR_DATA = 8, SMJ_RR_DIAG = 13,
       SMJ_RR_FLOSS = 14, SMJ_RR_SOLREF = 15, SMJ_RR_SOLIMP = 17, SMJ_RR_SAT = 22 /* satellite owning the row's dof, or -1 */, SMJ_RR_SDOF = 23 /* its dof inside the satellite */, SMJ_RR_STRIDE = 24 };
// per-lane stage record: KinTab 36 words, BodyTab 24, DofTab 16, EntryTab 7 arrays of `nent` slots (padded to 4 words), ActTab 28
constexpr int smj_lr_act(int nent) { return 76 + ((7 * nent + 3) & ~3); }
constexpr int smj_lr_stride(int nent) { return smj_lr_act(nent) + 28; }
enum { SMJ_LR_KIN = 0, SMJ_LR_BODY = 36, SMJ_LR_DOF = 60, SMJ_LR_ENT = 76, SMJ_LR_ACT = smj_lr_act(NENT), SMJ_LR_STRIDE = smj_lr_stride(NENT) };
// plane-pair record: pair, geom1 (the plane), geom2, their bodies, geom2 type, margin, geom2 bounding radius / centre,
// local frames of both geoms, geom2 size, then the contact parameters of the pair
enum { SMJ_PP_PAIR = 0, SMJ_PP_G1, SMJ_PP_G2, SMJ_PP_B1, SMJ_PP_B2, SMJ_PP_T2, SMJ_PP_MARGIN, SMJ_PP_RBOUND2, SMJ_PP_BCEN2 = 8,
       SMJ_PP_POS1 = 11, SMJ_PP_MAT1 = 14, SMJ_PP_POS2 = 23, SMJ_PP_MAT2 = 26, SMJ_PP_SIZE2 = 35, SMJ_PP_CONDIM = 38, SMJ_PP_MG = 39,
       SMJ_PP_FRIC = 40, SMJ_PP_SOLIMP = 45, SMJ_PP_SOLREF = 50, SMJ_PP_BOX2 = 52 /* geom2's box in its frame: centre, half sizes */,
       SMJ_PP_STRIDE = 60 };
// convex-cache geom record: geom, body, packed type|hull count|hull address, local frame, bounding-box centre and half
// sizes, MPR interior point, geom size
enum { SMJ_CG_GEOM = 0, SMJ_CG_BODY, SMJ_CG_META, SMJ_CG_POS = 3, SMJ_CG_MAT = 6, SMJ_CG_LCEN = 15, SMJ_CG_HALF = 18, SMJ_CG_CCEN = 21,
       SMJ_CG_SIZE = 24, SMJ_CG_RBOUND = 27, SMJ_CG_STRIDE = 28 };

// satellite record: ints, then floats (bit patterns); model_fuse.SAT_I / SAT_F
enum { SMJ_SR_BODY = 0, SMJ_SR_JTYPE, SMJ_SR_QADR, SMJ_SR_DADR, SMJ_SR_NDOF, SMJ_SR_JNT, SMJ_SR_F = 8,
       SMJ_SR_POS = SMJ_SR_F + 0, SMJ_SR_QUAT = SMJ_SR_F + 3, SMJ_SR_JPOS = SMJ_SR_F + 7, SMJ_SR_JAXIS = SMJ_SR_F + 10, SMJ_SR_Q0 = SMJ_SR_F + 13,
       SMJ_SR_INL = SMJ_SR_F + 14, SMJ_SR_ARM = SMJ_SR_F + 24, SMJ_SR_DAMP = SMJ_SR_F + 30, SMJ_SR_STIFF = SMJ_SR_F + 36, SMJ_SR_SPRING = SMJ_SR_F + 37,
       SMJ_SR_GCMASS = SMJ_SR_F + 38, SMJ_SR_GCIPOS = SMJ_SR_F + 39, SMJ_SR_STRIDE = 52 };
// layout of one env's staging row (4-byte words); a function of the variant's capacities, so that host code compiled once
// (smj_capi.hip) can address the rows of either kernel variant
struct SmjStageLayout {
  int qpos, qvel, warm, ctrl, bctl, nstep, info, actlen, actvel, base, gyro, accel, xpose, stride;
};
constexpr SmjStageLayout smj_stage_layout(int nvp, int nbp, int nsat = 0) {   // nbp: ALL body slots (main + satellites)
  SmjStageLayout L{};
  L.qpos = 0; L.qvel = L.qpos + nvp + 8 + 7 * nsat; L.warm = L.qvel + nvp + 6 * nsat; L.ctrl = L.warm + nvp + 6 * nsat; L.bctl = L.ctrl + 16; L.nstep = L.bctl + 8;
  L.info = L.nstep + 4; L.actlen = L.info + 4; L.actvel = L.actlen + 16; L.base = L.actvel + 16; L.gyro = L.base + 4;
  L.accel = L.gyro + 4; L.xpose = L.accel + 4; L.stride = L.xpose + 12 * nbp;
  return L;
}
// Batch-major simulator state bound through smj_bind() (include/smj.h).  ld = row stride in elements (>= B).
struct DevState {
  int B;           // environments in this context
  long ld;         // leading dimension of every [dim][B] array
  float* qpos;     // [nq][B]
  float* qvel;     // [nv][B]
  float* ctrl;     // [nu][B]
  float* warm;     // [nv][B]  qacc_warmstart
  int* nstep;      // [B]      steps taken since reset (time = nstep * timestep)
  float* act_len;  // [nu][B]  actuator_length   (readout, pull_status)
  float* act_vel;  // [nu][B]  actuator_velocity
  float* base;     // [3][B]   x, y, theta of base_link
  float* gyro;     // [3][B]
  float* accel;    // [3][B]
  float* lidar;    // [nlidar][B]
  int* info;       // [4][B]   nefc, ncon, solver iterations, flags (bit0: efc overflow, bit1: contact overflow, bit2: nan reset)
  float* debug;    // [SMJ_DEBUG_FLOATS][B] or null: stage dumps for parity tests
  float* prof;     // [SMJ_PROF_SLOTS][B] or null: shader cycles per stage, summed over the launch
  float* xpose;    // [nbody*12][B] or null: world pose (xpos 3, xmat 9) of every fused body at the last step (depth cameras)
  float* bctl;     // [8][B] or null: per-env state of the relative base moves (BaseController, mujoco_server.py:93-176), SMJ_BC_*
  // Env-major staging copy of the state, [B][SMJ_ST_STRIDE] words, owned by the library (null: the step kernel reads / writes
  // the batch-major arrays directly -- the lane emulator does).  One env per wavefront means lane = dim: on the batch-major
  // arrays every lane of a load / store touches its own 64-byte sector.  smj_step therefore runs import (batch-major ->
  // env-major, tiles transposed through LDS, both sides coalesced), the step kernel on contiguous 512-byte rows, export.
  float* stage;
  SmjStageLayout lay;   // word offsets inside a staging row (those of the context's primary kernel variant)
  // Capacity escalation (standard variant only; null = off): an env whose step needs more than the variant's constraint rows /
  // contacts stops BEFORE that step, leaves its state as of the start of the step in the staging row and appends
  // (env, steps done) to this list -- [0] = count, then pairs; the big variant then finishes the env's steps (smj_step).
  // Launch order (null = env id order): workgroup w takes env order[w].  smj_step sorts the envs by the shader time their last
  // launch took, longest first (`cost`, written by the kernel): the hardware dispatches workgroups in index order, so the
  // expensive envs (many convex contacts, Newton iterations) start in the first round and the cheap ones fill the tail.
  const int* order;
  int* cost;
  int* redo;
  int redo_worker;      // the launch is the tall variant working the list (redo[i] = env, -1 until published) off: 1 sweep, 2 poller
  // Pipelined chunks (standard variant; pipe_len 0 = off): the launch's n steps are cut into chunks of pipe_len steps and the
  // grid holds one workgroup per (chunk, env), chunk-major -- workgroup w = chunk * B + slot.  Chunk c of an env starts when
  // progress[env] >= c (published by the workgroup that ran chunk c - 1, which has a lower index and was therefore dispatched
  // earlier).  No barrier between chunks: the tail of a launch is the longest CHUNK of one env instead of the longest env
  // (smj_step_tu.h).
  // progress[env]: chunks finished (0..C) | -(c + 1): parked during chunk c, on the escalation list | SMJ_PIPE_ABANDONED: a
  // later chunk found the env parked with no poller alive and gave it up to the sweep | SMJ_PIPE_SWEPT: claimed by the sweep.
  // done_steps[env]: steps of this launch the env has completed (whoever ran them).
  // sched: [0] escalation entries claimed by pollers, [1] workgroups of the standard kernel that have exited, [2] pollers
  // alive, [3] escalation entries published (the list's count).
  // Escalation with pollers (redo_worker == 2): a few workgroups of the tall variant run CONCURRENTLY with the standard kernel
  // (second stream), take parked envs off the list as they appear, finish the env's current chunk and hand it back
  // (progress: -(c+1) -> c+1); they exit when every standard workgroup has exited and the list is drained.  The sweep
  // (redo_worker == 1, after both) finishes whatever is left of every env on the list -- all of it when there are no pollers.
  int* progress;
  int* done_steps;
  int* sched;
  int pipe_len;
  int pipe_total;   // workgroups of the standard launch
  int pollers;      // workgroups of the poller launch; negative: they stay even if the previous launch had no escalation
  int* hot;         // > 0: one of the last SMJ_HOT_LAUNCHES launches had escalations (kept by the sweep)
  // Separating directions of convex pairs (null = off): [B][SMJ_SEP_SLOTS] x {dir[3], pair id as int bits}, direct-mapped by
  // pair id.  A pair whose penetration query ended with "the origin lies outside the Minkowski difference" leaves the
  // direction that showed it; the next steps test that direction first (one support query) and skip the query while it
  // still separates.  Any direction with a negative support proves the shapes disjoint, so a stale or foreign entry can only
  // fail the test, never change a result.
  float* sepcache;
  // Contact manifolds of convex pairs (null = off): [B][SMJ_MC_SLOTS][SMJ_MC_WORDS], direct-mapped by pair key.  An entry holds the
  // poses (xpos, xquat) the pair's two BODIES had when its narrowphase last ran and the contacts it found; while both poses are
  // within SMJ_MC_EPS of those, the pair keeps that manifold and MPR / multiccd / the box-box polygon are skipped -- an object
  // resting on a counter costs one 160-byte load per step instead of five penetration queries.  Round 5: the kept manifold MOVES
  // WITH THE BODIES to first order (smj_mc_motion): a contact point p rides on both bodies, its depth changes by the normal component
  // of their relative displacement at p, dist' = dist + n . (u2(p) - u1(p)), u_b(p) = dx_b + dtheta_b x (p - x_b).  What is dropped is
  // second order in the motion since the manifold was built (<= SMJ_MC_EPS^2 / feature size: 1e-9 m), far below MPR's own 1e-6 m
  // tolerance -- which is what made resting bodies jitter by 1e-6 per step and miss the round-4 cache (3e-7, no update) every step.
  float* mcache;
  // PGS (null = off): the constraint rows of the env's previous step, [B][SMJ_PGSPREV_STRIDE] words: count, then
  // SMJ_PGSPREV_ROWS identity keys (type | equality / dof / limit record, or contact: pair | ordinal in the pair's manifold | row in
  // the contact), then as many forces.  Read by the next step's warm start (option pgs_dual_ws), whichever variant runs it.
  float* pgsprev;
};
enum { SMJ_PGSPREV_ROWS = 320, SMJ_PGSPREV_STRIDE = 2 * SMJ_PGSPREV_ROWS + 4 };
#ifndef SMJ_PGS_GUARD
#define SMJ_PGS_GUARD 1e-10f   // [MJ] mj_solPGS: an update whose cost change comes out above this is undone (A/B builds of tools only)
#endif
#ifndef SMJ_MC_LOG2
#define SMJ_MC_LOG2 5
#endif
enum { SMJ_SEP_SLOTS = 128 /* 64 for the moving-moving pairs (slot = pair & 63), 64 for the pairs with the static world */, SMJ_MC_SLOTS = 2 << SMJ_MC_LOG2, SMJ_MC_WORDS = 40 };   // (manifold slots: 1 << SMJ_MC_LOG2 for the moving-moving pairs, as many for the pairs with the static world)
#ifndef SMJ_MC_EPS
#define SMJ_MC_EPS 2e-5f
#endif
enum { SMJ_PIPE_ABANDONED = 0x7fffffff, SMJ_PIPE_SWEPT = 0x7ffffffe, SMJ_HOT_LAUNCHES = 8 };
enum { SMJ_SCHED_CLAIMED = 0, SMJ_SCHED_EXITED = 1, SMJ_SCHED_POLLERS = 2, SMJ_SCHED_COUNT = 3, SMJ_SCHED_WORDS = 4 };
// BaseController state rows (floats; mode: 0 none, 1 translate-by, 2 rotate-by, 3 velocity)
enum { SMJ_BC_MODE = 0, SMJ_BC_X0, SMJ_BC_Y0, SMJ_BC_TH0, SMJ_BC_INC, SMJ_BC_V, SMJ_BC_W, SMJ_BC_ROWS = 8 };
// wheel geometry and default speeds of the relative base moves (stretch_mujoco/config.py:2-3,11)
#define SMJ_WHEEL_RADIUS 0.0508f
#define SMJ_WHEEL_SEPARATION 0.3153f
#define SMJ_BASE_X_VEL 0.3f
#define SMJ_BASE_R_VEL 1.0f

enum { SMJ_PROF_KIN = 0, SMJ_PROF_COMCRB, SMJ_PROF_SMOOTH, SMJ_PROF_FACTOR, SMJ_PROF_COLLISION, SMJ_PROF_MAKECON,
       SMJ_PROF_PROJECT, SMJ_PROF_WARM, SMJ_PROF_PGS, SMJ_PROF_POST, SMJ_PROF_INTEGRATE, SMJ_PROF_TOTAL, SMJ_PROF_PGS_SWEEPS,
       SMJ_PROF_SETUP, SMJ_PROF_N_UPDATE, SMJ_PROF_N_GRAD, SMJ_PROF_N_XA, SMJ_PROF_N_HMFMA, SMJ_PROF_N_FACTSOLVE, SMJ_PROF_N_SOLVE,
       SMJ_PROF_N_PREP, SMJ_PROF_N_LS, SMJ_PROF_N_LSEVALS,
       // convex collision: cycles of the pose / bounding-sphere / oriented-box / narrowphase parts, and counts per step
       SMJ_PROF_C_POSE, SMJ_PROF_C_SPHERE, SMJ_PROF_C_OBB, SMJ_PROF_C_NARROW, SMJ_PROF_C_NSPHERE, SMJ_PROF_C_NOBB, SMJ_PROF_C_NHIT,
       SMJ_PROF_C_NMULTI, SMJ_PROF_C_TBOXBOX, SMJ_PROF_C_TMPR1, SMJ_PROF_C_TMULTI, SMJ_PROF_C_ROUNDS, SMJ_PROF_H_K, SMJ_PROF_H_CONE, SMJ_PROF_H_STORE, SMJ_PROF_H_NKS,
       // satellite builds: static broadphase / its narrowphase loop; make_constraint_sat: clear + static rows, contact rows, Jacobian fill, row items, impedance; satellites' Hessian blocks + own solves
       SMJ_PROF_S_BROAD = 40, SMJ_PROF_S_NARROW, SMJ_PROF_MC_ROWS, SMJ_PROF_MC_CON, SMJ_PROF_MC_JAC, SMJ_PROF_MC_ITEMS, SMJ_PROF_MC_IMP, SMJ_PROF_SAT_H, SMJ_PROF_SLOTS = 48 };
enum { SMJ_INFO_NEFC = 0, SMJ_INFO_NCON = 1, SMJ_INFO_NITER = 2, SMJ_INFO_FLAGS = 3 };
enum { SMJ_FLAG_EFC_OVERFLOW = 1, SMJ_FLAG_CON_OVERFLOW = 2, SMJ_FLAG_BAD_STATE = 4, SMJ_FLAG_PIPE_TIMEOUT = 8 };

// layout of the optional debug dump (floats), one column per env; a function of the variant's capacities (smj_dims reports
// the offsets of the running variant, lib.py: debug_layout)
struct SmjDebugLayout { int qm, g, qacc, efc_force, efc_b, efc_r, efc_aref, ar_diag, xpos, qfrc_bias, qfrc_passive, qfrc_act, con, ar, satqacc, floats; };
constexpr SmjDebugLayout smj_debug_layout(int nvp, int ncon, int nsat = 0) {
  SmjDebugLayout L{};
  L.qm = 0; L.g = nvp * nvp; L.qacc = L.g + nvp; L.efc_force = L.qacc + nvp; L.efc_b = L.efc_force + 64; L.efc_r = L.efc_b + 64;
  L.efc_aref = L.efc_r + 64; L.ar_diag = L.efc_aref + 64; L.xpos = L.ar_diag + 64; L.qfrc_bias = L.xpos + 96; L.qfrc_passive = L.qfrc_bias + nvp;
  L.qfrc_act = L.qfrc_passive + nvp; L.con = L.qfrc_act + nvp; L.ar = L.con + 8 * ncon; L.satqacc = L.ar + 64 * 64; L.floats = L.satqacc + 6 * nsat;
  return L;
}
constexpr SmjDebugLayout SMJ_DBG = smj_debug_layout(NVP, NCON, NSAT);
enum {
  SMJ_DBG_QM = SMJ_DBG.qm,                     // NVP*NVP dense mass matrix (row-major, stride NVP)
  SMJ_DBG_G = SMJ_DBG.g,                       // NVP qfrc_smooth
  SMJ_DBG_QACC = SMJ_DBG.qacc,                 // NVP qacc (forward dynamics)
  SMJ_DBG_EFC_FORCE = SMJ_DBG.efc_force,       // the first 64 rows of each efc array
  SMJ_DBG_EFC_B = SMJ_DBG.efc_b, SMJ_DBG_EFC_R = SMJ_DBG.efc_r, SMJ_DBG_EFC_AREF = SMJ_DBG.efc_aref, SMJ_DBG_AR_DIAG = SMJ_DBG.ar_diag,
  SMJ_DBG_XPOS = SMJ_DBG.xpos,                 // 32*3
  SMJ_DBG_QFRC_BIAS = SMJ_DBG.qfrc_bias, SMJ_DBG_QFRC_PASSIVE = SMJ_DBG.qfrc_passive, SMJ_DBG_QFRC_ACT = SMJ_DBG.qfrc_act,   // NVP each
  SMJ_DBG_CON = SMJ_DBG.con,                   // NCON contacts x (dist, pos3, normal3, condim | geom1 << 4 | geom2 << 14) = 8 floats
  SMJ_DBG_AR = SMJ_DBG.ar,                     // 64*64 AR (PGS)
  SMJ_DBG_SATQACC = SMJ_DBG.satqacc,           // 6 per satellite: qacc of the satellite's dofs (satellite builds)
  SMJ_DEBUG_FLOATS = SMJ_DBG.floats
};
static_assert(NVP != 32 || NCON != 16 || (SMJ_DBG_QACC == 1056 && SMJ_DBG_CON == 1600 && SMJ_DBG_AR == 1728), "standard variant: the layout the tests index");
