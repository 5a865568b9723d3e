/* TEST INFRASTRUCTURE ONLY -- see smj_oracle.h.  PARITY UNPINNED (no MuJoCo in this image).
 *
 * fp64, single environment, body-for-body, stage-by-stage restatement of what
 * `mj_step` (reference call site stretch_mujoco/mujoco_server.py:378) computes for the
 * Stretch model (stretch_mujoco/models/stretch.xml) under its options
 * (stretch.xml:7: integrator=implicitfast, cone=elliptic, impratio=20) with the PGS solver
 * named by BASELINE.json north_star.  Stage names are MuJoCo's.
 */
#include "smj_oracle.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MAXCON 256
#define MAXEFC 1024
enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { C_EQUALITY = 0, C_FRICTION_DOF = 1, C_LIMIT_JOINT = 3, C_CONTACT_FRICTIONLESS = 5, C_CONTACT_ELLIPTIC = 7 };

/* ------------------------------------------------------------------ blob */
typedef struct {
  char name[48];
  uint32_t dtype, ndim, shape[4];
  uint64_t offset, nbytes;
} blob_entry;

static const blob_entry* blob_find(const uint8_t* b, const char* name) {
  uint32_t n;
  memcpy(&n, b + 8, 4);
  const blob_entry* e = (const blob_entry*)(b + 16);
  for (uint32_t i = 0; i < n; i++)
    if (strncmp(e[i].name, name, 48) == 0) return &e[i];
  return NULL;
}
static double* blob_f64(const uint8_t* b, const char* name, int* count) {
  const blob_entry* e = blob_find(b, name);
  if (!e || e->dtype != 0) { fprintf(stderr, "smj_oracle: missing f64 array %s\n", name); abort(); }
  double* p = (double*)malloc(e->nbytes ? e->nbytes : 8);
  memcpy(p, b + e->offset, e->nbytes);
  if (count) *count = (int)(e->nbytes / 8);
  return p;
}
static int* blob_i32(const uint8_t* b, const char* name, int* count) {
  const blob_entry* e = blob_find(b, name);
  if (!e || e->dtype != 1) { fprintf(stderr, "smj_oracle: missing i32 array %s\n", name); abort(); }
  int* p = (int*)malloc(e->nbytes ? e->nbytes : 4);
  memcpy(p, b + e->offset, e->nbytes);
  if (count) *count = (int)(e->nbytes / 4);
  return p;
}

/* ------------------------------------------------------------------ model / data */
struct smjo_model {
  void** allocs; int nalloc, calloc_cap;   /* every array the model owns (smjo_free_model) */
  int nq, nv, nu, nbody, njnt, ngeom, nsite, ncam, neq, ntendon, nwrap, nkey, npair, nhullvert, nlidar;
  double timestep, gravity[3], impratio, tolerance, meaninertia, lidar_cutoff;
  int iterations, cone, warmstart, pgs_fixed_iter, max_con_pair, solver, ls_iterations;
  int qcqp_cap;      /* iterates of mju_QCQP (20 = MuJoCo); option "qcqp_cap" */
  int pgs_dual_warmstart; /* NOT MuJoCo (default 0): PGS may also start from the previous step's constraint forces, row by row (option "pgs_dual_warmstart"; the kernels' option of the same name) */
  int multiccd;      /* stretch.xml:8 <flag multiccd="enable"/>: multi-point contacts for convex pairs (default on) */
  /* NOT MuJoCo (default 0; option "manifold_keep"): the TWIN of the kernels' contact-manifold cache (option manifold_cache there,
   * smj_step_impl.h narrow_pair / smj_sat.h): a convex pair whose two bodies stand within manifold_keep_eps of the poses its
   * manifold was built at keeps that manifold, carried to first order by the bodies' motion since -- same slots, same keep rule,
   * same carry, in fp64.  Tests bound "kernel vs this twin" (the implementation) and "twin vs unmodified" (the rule) separately. */
  int manifold_keep;
  double manifold_keep_eps;
  int* pair_tag;     /* per pair: the kernels' cache tag (index in k_convpair + 1, or 0x40000000 | index in k_statpair), 0 = none */
  double ls_tolerance;
  int *body_parentid, *body_weldid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_gravcomp, *body_invweight0,
      *body_subtreemass, *body_gcmass, *body_gcipos, *geom_invweight0;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  double *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
  double *qpos0, *qpos_spring;
  int *geom_type, *geom_bodyid, *geom_hulladr, *geom_hullnum;
  double *geom_pos, *geom_quat, *geom_size, *geom_rbound, *geom_center, *geom_rgba, *hull_vert, *geom_aabb, *geom_ccenter;
  int convex_pairs; /* 1: evaluate non-plane pairs with MPR */
  int *site_bodyid, *cam_bodyid;
  double *site_pos, *site_quat, *cam_pos, *cam_quat, *cam_fovy;
  int *tendon_adr, *tendon_num, *wrap_objid;
  double* wrap_prm;
  int *eq_obj1id, *eq_obj2id, *eq_active;
  double *eq_data, *eq_solref, *eq_solimp;
  int *actuator_trntype, *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited, *actuator_biastype;
  double *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  double* key_ctrl;
  int *pair_geom1, *pair_geom2, *pair_condim;
  double *pair_friction, *pair_solref, *pair_solimp, *pair_margin, *pair_gap;
  int imu_site, *lidar_site;
  double* lidar_static;
  /* depth cameras: camera-visible mesh geoms and their triangles (render tables of the product blob; may be absent) */
  int *geom_group, *geom_rmeshid, *rmesh_vertadr, *rmesh_faceadr, *rmesh_facenum, *rmesh_face, nrmesh;
  double* rmesh_vert;
  double znear, zfar; /* absolute: vis.map.znear / zfar times stat.extent */
  struct rbvh* rbvh;  /* one per render mesh, built on first use */
};

typedef struct {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], includemargin, mu;
  int dim, geom1, geom2, efc_address;
} contact_t;

struct smjo_data {
  void** allocs; int nalloc, calloc_cap;   /* every array the data owns (smjo_free_data) */
  int nv, ncon, nefc, ne, nf, solver_niter, ncon_dropped;
  /* tests: contact list handed in from outside (smjo_set_contacts) replaces the collision stage of the next forward pass,
   * so that the DYNAMICS can be compared on identical contacts even where MPR's portal facet differs (curved rims, vertices) */
  int override_ncon;
  double* override_con;   /* per contact: dist, pos[3], normal[3], geom1, geom2 */
  double time;
  double *qpos, *qvel, *ctrl, *qacc_warmstart, *qacc, *qacc_smooth, *qfrc_bias, *qfrc_passive, *qfrc_actuator,
      *qfrc_smooth, *qfrc_constraint, *qfrc_applied;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat,
      *cam_xpos, *cam_xmat;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *qM, *qL, *qH; /* dense nv*nv: mass matrix, its Cholesky factor, implicit matrix factor */
  double *ten_length, *ten_J, *actuator_length, *actuator_velocity, *actuator_moment, *actuator_force;
  contact_t* contact;
  double *efc_J, *efc_pos, *efc_margin, *efc_D, *efc_R, *efc_aref, *efc_b, *efc_force, *efc_vel, *efc_KBIP,
      *efc_diagApprox, *efc_frictionloss, *efc_AR;
  int *efc_type, *efc_id;
  /* option pgs_dual_warmstart: the rows of the previous step (identity key, force) */
  int prev_n;
  long long* prev_key;
  double* prev_force;
  double *gyro, *accel, *lidar;
  double* scratch;
  /* option manifold_keep: MC_SLOTS entries of {tag, n, pose of body 1 (pos 3, quat 4), pose of body 2, normal 3, n x (dist, pos 3)} */
  int mc_tag[64], mc_n[64], mc_hits;
  double mc_pose[64][14], mc_nrm[64][3], mc_con[64][5][4];
};

static void* track_ptr(void*** list, int* n, int* cap, void* p) {
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 128; *list = (void**)realloc(*list, sizeof(void*) * (size_t)*cap); }
  (*list)[(*n)++] = p;
  return p;
}
#define MTRACK(m, p) track_ptr(&(m)->allocs, &(m)->nalloc, &(m)->calloc_cap, (p))
#define DTRACK(d, p) track_ptr(&(d)->allocs, &(d)->nalloc, &(d)->calloc_cap, (p))
#define LOADF(name) m->name = (double*)MTRACK(m, blob_f64(b, #name, NULL))
#define LOADI(name) m->name = (int*)MTRACK(m, blob_i32(b, #name, NULL))

smjo_model* smjo_load(const void* blob, size_t nbytes) {
  const uint8_t* b = (const uint8_t*)blob;
  if (nbytes < 16 || memcmp(b, "SMJB0001", 8) != 0) return NULL;
  smjo_model* m = (smjo_model*)calloc(1, sizeof(smjo_model));
  int* dims = blob_i32(b, "dims", NULL);
  m->nq = dims[0]; m->nv = dims[1]; m->nu = dims[2]; m->nbody = dims[3]; m->njnt = dims[4]; m->ngeom = dims[5];
  m->nsite = dims[6]; m->ncam = dims[7]; m->neq = dims[8]; m->ntendon = dims[9]; m->nwrap = dims[10];
  m->nkey = dims[11]; m->npair = dims[12]; m->nhullvert = dims[13];
  free(dims);
  double* t;
  t = blob_f64(b, "opt_timestep", NULL); m->timestep = t[0]; free(t);
  t = blob_f64(b, "opt_gravity", NULL); memcpy(m->gravity, t, 24); free(t);
  t = blob_f64(b, "opt_impratio", NULL); m->impratio = t[0]; free(t);
  t = blob_f64(b, "opt_tolerance", NULL); m->tolerance = t[0]; free(t);
  t = blob_f64(b, "stat_meaninertia", NULL); m->meaninertia = t[0]; free(t);
  t = blob_f64(b, "sensor_lidar_cutoff", NULL); m->lidar_cutoff = t[0]; free(t);
  int* ti;
  ti = blob_i32(b, "opt_iterations", NULL); m->iterations = ti[0]; free(ti);
  ti = blob_i32(b, "opt_cone", NULL); m->cone = ti[0]; free(ti);
  ti = blob_i32(b, "sensor_imu_site", NULL); m->imu_site = ti[0]; free(ti);
  m->lidar_site = (int*)MTRACK(m, blob_i32(b, "sensor_lidar_site", &m->nlidar));
  m->lidar_static = (double*)MTRACK(m, blob_f64(b, "sensor_lidar_static", NULL));
  m->warmstart = 1; m->pgs_fixed_iter = 0; m->max_con_pair = 4; m->qcqp_cap = 20; m->solver = 0; m->ls_iterations = 50; m->ls_tolerance = 0.01; m->multiccd = 1;
  LOADI(body_parentid); LOADI(body_weldid); LOADI(body_rootid); LOADI(body_jntadr); LOADI(body_jntnum);
  LOADI(body_dofadr); LOADI(body_dofnum);
  LOADF(body_pos); LOADF(body_quat); LOADF(body_ipos); LOADF(body_iquat); LOADF(body_mass); LOADF(body_inertia);
  LOADF(body_gravcomp); LOADF(body_invweight0); LOADF(body_subtreemass); LOADF(body_gcmass); LOADF(body_gcipos);
  LOADF(geom_invweight0);
  LOADI(jnt_type); LOADI(jnt_qposadr); LOADI(jnt_dofadr); LOADI(jnt_bodyid); LOADI(jnt_limited);
  LOADF(jnt_pos); LOADF(jnt_axis); LOADF(jnt_stiffness); LOADF(jnt_range); LOADF(jnt_margin); LOADF(jnt_solref);
  LOADF(jnt_solimp);
  LOADI(dof_bodyid); LOADI(dof_jntid); LOADI(dof_parentid);
  LOADF(dof_armature); LOADF(dof_damping); LOADF(dof_frictionloss); LOADF(dof_invweight0); LOADF(dof_solref);
  LOADF(dof_solimp);
  LOADF(qpos0); LOADF(qpos_spring);
  LOADI(geom_type); LOADI(geom_bodyid); LOADI(geom_hulladr); LOADI(geom_hullnum);
  LOADF(geom_pos); LOADF(geom_quat); LOADF(geom_size); LOADF(geom_rbound); LOADF(geom_center); LOADF(geom_rgba);
  LOADF(hull_vert); LOADF(geom_aabb); LOADF(geom_ccenter);
  m->convex_pairs = 1;
  LOADI(site_bodyid); LOADI(cam_bodyid); LOADF(site_pos); LOADF(site_quat); LOADF(cam_pos); LOADF(cam_quat);
  LOADF(cam_fovy);
  LOADI(tendon_adr); LOADI(tendon_num); LOADI(wrap_objid); LOADF(wrap_prm);
  LOADI(eq_obj1id); LOADI(eq_obj2id); LOADI(eq_active); LOADF(eq_data); LOADF(eq_solref); LOADF(eq_solimp);
  LOADI(actuator_trntype); LOADI(actuator_trnid); LOADI(actuator_ctrllimited); LOADI(actuator_forcelimited);
  LOADI(actuator_biastype);
  LOADF(actuator_gear); LOADF(actuator_gainprm); LOADF(actuator_biasprm); LOADF(actuator_ctrlrange);
  LOADF(actuator_forcerange);
  LOADF(key_ctrl);
  LOADI(pair_geom1); LOADI(pair_geom2); LOADI(pair_condim);
  LOADF(pair_friction); LOADF(pair_solref); LOADF(pair_solimp); LOADF(pair_margin); LOADF(pair_gap);
  LOADI(geom_group);
  m->manifold_keep = 0; m->manifold_keep_eps = 2e-5;   /* SMJ_MC_EPS, smj_model.h */
  m->pair_tag = (int*)MTRACK(m, calloc(m->npair > 0 ? m->npair : 1, sizeof(int)));
  for (int st = 0; st < 2; st++) {   /* the kernels' pair tables: moving-moving pairs (all convex pairs in the builds without satellites), pairs with the static world */
    const char *tab = st ? "k_statpair" : "k_convpair", *cnt = st ? "k_nstatpair" : "k_nconvpair";
    if (!blob_find(b, tab) || !blob_find(b, cnt)) continue;
    int nt = 0, *tp = blob_i32(b, tab, &nt), *np_ = blob_i32(b, cnt, NULL);
    for (int k = 0; k < np_[0] && k < nt; k++)
      if (tp[k] >= 0 && tp[k] < m->npair) m->pair_tag[tp[k]] = st ? (0x40000000 | k) : k + 1;
    free(tp); free(np_);
  }
  if (blob_find(b, "geom_rmeshid") && blob_find(b, "rmesh_vert")) {
    LOADI(geom_rmeshid); LOADI(rmesh_vertadr); LOADI(rmesh_faceadr); LOADI(rmesh_face);
    m->rmesh_facenum = (int*)MTRACK(m, blob_i32(b, "rmesh_facenum", &m->nrmesh));
    const blob_entry* e = blob_find(b, "rmesh_vert");
    if (e->dtype != 3) { fprintf(stderr, "smj_oracle: rmesh_vert must be f32\n"); abort(); }
    size_t cnt = e->nbytes / 4;
    m->rmesh_vert = (double*)MTRACK(m, malloc((cnt ? cnt : 1) * sizeof(double)));
    for (size_t i = 0; i < cnt; i++) { float v; memcpy(&v, b + e->offset + 4 * i, 4); m->rmesh_vert[i] = v; }
    t = blob_f64(b, "vis_znear_zfar_extent", NULL); m->znear = t[0] * t[2]; m->zfar = t[1] * t[2]; free(t);
  }
  return m;
}

void smjo_free_model(smjo_model* m) {
  if (!m) return;
  for (int i = 0; i < m->nalloc; i++) free(m->allocs[i]);
  free(m->allocs);
  free(m);
}


int smjo_set_option(smjo_model* m, const char* name, double v) {
  if (!strcmp(name, "iterations")) m->iterations = (int)v;
  else if (!strcmp(name, "tolerance")) m->tolerance = v;
  else if (!strcmp(name, "warmstart")) m->warmstart = (int)v;
  else if (!strcmp(name, "pgs_fixed_iter")) m->pgs_fixed_iter = (int)v;
  else if (!strcmp(name, "qcqp_cap")) m->qcqp_cap = (int)v;   /* per model; 20 = MuJoCo */
  else if (!strcmp(name, "pgs_dual_warmstart")) m->pgs_dual_warmstart = (int)v;
  else if (!strcmp(name, "max_contacts_per_pair")) m->max_con_pair = (int)v;
  else if (!strcmp(name, "solver")) m->solver = (int)v; /* 0 = PGS (north_star), 2 = Newton (the reference model's default) */
  else if (!strcmp(name, "convex_pairs")) m->convex_pairs = (int)v;
  else if (!strcmp(name, "multiccd")) m->multiccd = (int)v;
  else if (!strcmp(name, "manifold_keep")) m->manifold_keep = (int)v;
  else if (!strcmp(name, "manifold_keep_eps")) m->manifold_keep_eps = v;
  else if (!strcmp(name, "timestep")) m->timestep = v;
  else if (!strcmp(name, "gravity_z")) m->gravity[2] = v;
  else if (!strcmp(name, "impratio")) m->impratio = v;
  else return -1;
  return 0;
}

int smjo_dim(const smjo_model* m, const char* n) {
  if (!strcmp(n, "nq")) return m->nq;
  if (!strcmp(n, "nv")) return m->nv;
  if (!strcmp(n, "nu")) return m->nu;
  if (!strcmp(n, "nbody")) return m->nbody;
  if (!strcmp(n, "njnt")) return m->njnt;
  if (!strcmp(n, "ngeom")) return m->ngeom;
  if (!strcmp(n, "nsite")) return m->nsite;
  if (!strcmp(n, "npair")) return m->npair;
  if (!strcmp(n, "nlidar")) return m->nlidar;
  if (!strcmp(n, "ncam")) return m->ncam;
  return -1;
}

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

smjo_data* smjo_make_data(const smjo_model* m) {
  smjo_data* d = (smjo_data*)calloc(1, sizeof(smjo_data));
  int nv = m->nv, nb = m->nbody;
  d->nv = nv;
  d->qpos = (double*)DTRACK(d, dalloc(m->nq)); d->qvel = (double*)DTRACK(d, dalloc(nv)); d->ctrl = (double*)DTRACK(d, dalloc(m->nu)); d->qacc_warmstart = (double*)DTRACK(d, dalloc(nv));
  d->qacc = (double*)DTRACK(d, dalloc(nv)); d->qacc_smooth = (double*)DTRACK(d, dalloc(nv)); d->qfrc_bias = (double*)DTRACK(d, dalloc(nv)); d->qfrc_passive = (double*)DTRACK(d, dalloc(nv));
  d->qfrc_actuator = (double*)DTRACK(d, dalloc(nv)); d->qfrc_smooth = (double*)DTRACK(d, dalloc(nv)); d->qfrc_constraint = (double*)DTRACK(d, dalloc(nv));
  d->qfrc_applied = (double*)DTRACK(d, dalloc(nv));
  d->xpos = (double*)DTRACK(d, dalloc(3 * nb)); d->xquat = (double*)DTRACK(d, dalloc(4 * nb)); d->xmat = (double*)DTRACK(d, dalloc(9 * nb)); d->xipos = (double*)DTRACK(d, dalloc(3 * nb));
  d->ximat = (double*)DTRACK(d, dalloc(9 * nb)); d->xanchor = (double*)DTRACK(d, dalloc(3 * m->njnt)); d->xaxis = (double*)DTRACK(d, dalloc(3 * m->njnt));
  d->geom_xpos = (double*)DTRACK(d, dalloc(3 * m->ngeom)); d->geom_xmat = (double*)DTRACK(d, dalloc(9 * m->ngeom));
  d->site_xpos = (double*)DTRACK(d, dalloc(3 * m->nsite)); d->site_xmat = (double*)DTRACK(d, dalloc(9 * m->nsite));
  d->cam_xpos = (double*)DTRACK(d, dalloc(3 * m->ncam)); d->cam_xmat = (double*)DTRACK(d, dalloc(9 * m->ncam));
  d->subtree_com = (double*)DTRACK(d, dalloc(3 * nb)); d->cinert = (double*)DTRACK(d, dalloc(10 * nb)); d->crb = (double*)DTRACK(d, dalloc(10 * nb)); d->cdof = (double*)DTRACK(d, dalloc(6 * nv));
  d->cdof_dot = (double*)DTRACK(d, dalloc(6 * nv)); d->cvel = (double*)DTRACK(d, dalloc(6 * nb)); d->cacc = (double*)DTRACK(d, dalloc(6 * nb)); d->cfrc = (double*)DTRACK(d, dalloc(6 * nb));
  d->qM = (double*)DTRACK(d, dalloc(nv * nv)); d->qL = (double*)DTRACK(d, dalloc(nv * nv)); d->qH = (double*)DTRACK(d, dalloc(nv * nv));
  d->ten_length = (double*)DTRACK(d, dalloc(m->ntendon)); d->ten_J = (double*)DTRACK(d, dalloc(m->ntendon * nv));
  d->actuator_length = (double*)DTRACK(d, dalloc(m->nu)); d->actuator_velocity = (double*)DTRACK(d, dalloc(m->nu)); d->actuator_moment = (double*)DTRACK(d, dalloc(m->nu * nv));
  d->actuator_force = (double*)DTRACK(d, dalloc(m->nu));
  d->contact = (contact_t*)DTRACK(d, calloc(MAXCON, sizeof(contact_t)));
  d->override_ncon = -1; d->override_con = NULL;
  d->efc_J = (double*)DTRACK(d, dalloc((size_t)MAXEFC * nv)); d->efc_pos = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_margin = (double*)DTRACK(d, dalloc(MAXEFC));
  d->efc_D = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_R = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_aref = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_b = (double*)DTRACK(d, dalloc(MAXEFC));
  d->efc_force = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_vel = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_KBIP = (double*)DTRACK(d, dalloc(4 * MAXEFC));
  d->efc_diagApprox = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_frictionloss = (double*)DTRACK(d, dalloc(MAXEFC)); d->efc_AR = (double*)DTRACK(d, dalloc((size_t)MAXEFC * MAXEFC));
  d->efc_type = (int*)DTRACK(d, calloc(MAXEFC, sizeof(int))); d->efc_id = (int*)DTRACK(d, calloc(MAXEFC, sizeof(int)));
  d->gyro = (double*)DTRACK(d, dalloc(3)); d->accel = (double*)DTRACK(d, dalloc(3)); d->lidar = (double*)DTRACK(d, dalloc(m->nlidar));
  d->scratch = (double*)DTRACK(d, dalloc((size_t)MAXEFC * nv + 16 * nv));
  smjo_reset(m, d);
  return d;
}
void smjo_free_data(smjo_data* d) {
  if (!d) return;
  for (int i = 0; i < d->nalloc; i++) free(d->allocs[i]);
  free(d->allocs); free(d->prev_key); free(d->prev_force); free(d->override_con);
  free(d);
}

void smjo_reset(const smjo_model* m, smjo_data* d) {
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->qfrc_applied, 0, sizeof(double) * m->nv);
  d->time = 0;
  memset(d->mc_tag, 0, sizeof(d->mc_tag)); d->mc_hits = 0;
}

#define GETD(nm, cnt) if (!strcmp(name, #nm)) { *n = (cnt); return d->nm; }
double* smjo_get(smjo_data* d, const char* name, int* n) {
  int nv = d->nv, dummy;
  if (!n) n = &dummy;
  if (!strcmp(name, "time")) { *n = 1; return &d->time; }
  GETD(qpos, -1) GETD(qvel, nv) GETD(ctrl, -1) GETD(qacc_warmstart, nv) GETD(qacc, nv) GETD(qacc_smooth, nv)
  GETD(qfrc_bias, nv) GETD(qfrc_passive, nv) GETD(qfrc_actuator, nv) GETD(qfrc_smooth, nv) GETD(qfrc_constraint, nv)
  GETD(qfrc_applied, nv)
  GETD(xpos, -1) GETD(xquat, -1) GETD(xmat, -1) GETD(xipos, -1) GETD(ximat, -1) GETD(xanchor, -1) GETD(xaxis, -1)
  GETD(geom_xpos, -1) GETD(geom_xmat, -1) GETD(site_xpos, -1) GETD(site_xmat, -1) GETD(cam_xpos, -1) GETD(cam_xmat, -1)
  GETD(subtree_com, -1) GETD(cinert, -1) GETD(crb, -1) GETD(cdof, 6 * nv) GETD(cdof_dot, 6 * nv) GETD(cvel, -1)
  GETD(cacc, -1) GETD(qM, nv * nv) GETD(qL, nv * nv)
  GETD(ten_length, -1) GETD(actuator_length, -1) GETD(actuator_velocity, -1) GETD(actuator_force, -1)
  GETD(actuator_moment, -1)
  GETD(efc_J, d->nefc * nv) GETD(efc_pos, d->nefc) GETD(efc_D, d->nefc) GETD(efc_R, d->nefc) GETD(efc_aref, d->nefc)
  GETD(efc_b, d->nefc) GETD(efc_force, d->nefc) GETD(efc_vel, d->nefc) GETD(efc_AR, d->nefc * d->nefc)
  GETD(efc_diagApprox, d->nefc) GETD(efc_KBIP, 4 * d->nefc)
  GETD(gyro, 3) GETD(accel, 3) GETD(lidar, -1)
  if (!strcmp(name, "contact")) { *n = d->ncon * (int)(sizeof(contact_t) / 8); return (double*)d->contact; }
  return NULL;
}
int* smjo_get_int(smjo_data* d, const char* name, int* n) {
  int dummy;
  if (!n) n = &dummy;
  if (!strcmp(name, "efc_type")) { *n = d->nefc; return d->efc_type; }
  if (!strcmp(name, "efc_id")) { *n = d->nefc; return d->efc_id; }
  if (!strcmp(name, "ncon")) { *n = 1; return &d->ncon; }
  if (!strcmp(name, "nefc")) { *n = 1; return &d->nefc; }
  if (!strcmp(name, "solver_niter")) { *n = 1; return &d->solver_niter; }
  if (!strcmp(name, "ncon_dropped")) { *n = 1; return &d->ncon_dropped; }
  if (!strcmp(name, "mc_hits")) { *n = 1; return &d->mc_hits; }
  return NULL;
}

/* ------------------------------------------------------------------ small math */
/* Floating-point operation counter (multiplies, adds, divides, square roots each count 1; an FMA counts 2): incremented in the
 * vector primitives and at the dense loops of every stage, so that a rollout yields the flops MuJoCo's algorithm spends per
 * step on this model (tools/flop_count.py -> profiles/flops_per_env_step.json, the constant bench.py's fp32 roofline uses). */
static long long g_flop = 0;
#define FL(n) (g_flop += (long long)(n))
long long smjo_flops(int reset) { long long v = g_flop; if (reset) g_flop = 0; return v; }
static inline double dot3(const double* a, const double* b) { FL(5); return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) { FL(9);
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline double normalize3(double* a) { FL(4);
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
static void quat_mul(double* r, const double* a, const double* b) { FL(28);
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, 32);
}
static void quat_normalize(double* q) { FL(12);
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quat2mat(double* R, const double* q) { FL(27);
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
static void axisangle2quat(double* q, const double* axis, double ang) {
  double s = sin(0.5 * ang);
  q[0] = cos(0.5 * ang); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static inline void mulmat3vec(double* r, const double* R, const double* v) { FL(15);
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
         z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmat3Tvec(double* r, const double* R, const double* v) { FL(15);
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
         z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulmat3(double* r, const double* A, const double* B) { FL(45);
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(r, t, 72);
}

/* spatial algebra on [angular(3); linear(3)] about the tree-root subtree COM.  [MJ] mju_mulInertVec etc. */
static void mul_inert_vec(double* r, const double* i, const double* v) { FL(33);
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(double* r, const double* vel, const double* v) { FL(6);
  double a[3], b[3], c[3];
  cross3(a, vel, v); cross3(b, vel, v + 3); cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* vel, const double* f) { FL(6);
  double a[3], b[3], c[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

/* dense Cholesky A = L L^T (lower, row-major n x n); returns rank */
static int chol_factor(double* A, int n, double mindiag) { FL((long long)n * n * n / 3 + 2LL * n * n);
  int rank = n;
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (s < mindiag) { s = mindiag; rank--; }
    double l = sqrt(s);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / l;
    }
  }
  return rank;
}
static void chol_solve(double* x, const double* L, const double* b, int n) { FL(2LL * n * n + 2 * n);
  if (x != b) memcpy(x, b, sizeof(double) * n);
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* ------------------------------------------------------------------ B.1 position stage */
/* [MJ] mj_kinematics */
static void kinematics(const smjo_model* m, smjo_data* d) {
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  memset(d->xipos, 0, 24);
  quat2mat(d->ximat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b], ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    double pos[3], quat[4], R[9];
    if (jn == 1 && m->jnt_type[ja] == JNT_FREE) {
      const double* q = d->qpos + m->jnt_qposadr[ja];
      memcpy(pos, q, 24); memcpy(quat, q + 3, 32);
      quat_normalize(quat);
      memcpy(d->xanchor + 3 * ja, pos, 24);
      d->xaxis[3 * ja] = 0; d->xaxis[3 * ja + 1] = 0; d->xaxis[3 * ja + 2] = 1;
    } else {
      mulmat3vec(pos, d->xmat + 9 * p, m->body_pos + 3 * b);
      for (int k = 0; k < 3; k++) pos[k] += d->xpos[3 * p + k];
      quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
      for (int j = ja; j < ja + jn; j++) {
        quat2mat(R, quat);
        mulmat3vec(d->xaxis + 3 * j, R, m->jnt_axis + 3 * j);
        double a[3];
        mulmat3vec(a, R, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) d->xanchor[3 * j + k] = pos[k] + a[k];
        double q = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == JNT_SLIDE) {
          for (int k = 0; k < 3; k++) pos[k] += d->xaxis[3 * j + k] * q;
        } else {
          double dq[4];
          axisangle2quat(dq, m->jnt_axis + 3 * j, q);
          quat_mul(quat, quat, dq);
          quat2mat(R, quat);
          mulmat3vec(a, R, m->jnt_pos + 3 * j);
          for (int k = 0; k < 3; k++) pos[k] = d->xanchor[3 * j + k] - a[k];
        }
      }
    }
    quat_normalize(quat);
    memcpy(d->xpos + 3 * b, pos, 24); memcpy(d->xquat + 4 * b, quat, 32);
    quat2mat(d->xmat + 9 * b, quat);
    double t[3], Ri[9];
    mulmat3vec(t, d->xmat + 9 * b, m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = pos[k] + t[k];
    quat2mat(Ri, m->body_iquat + 4 * b);
    mulmat3(d->ximat + 9 * b, d->xmat + 9 * b, Ri);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], R[9];
    mulmat3vec(t, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + t[k];
    quat2mat(R, m->geom_quat + 4 * g);
    mulmat3(d->geom_xmat + 9 * g, d->xmat + 9 * b, R);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3], R[9];
    mulmat3vec(t, d->xmat + 9 * b, m->site_pos + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] = d->xpos[3 * b + k] + t[k];
    quat2mat(R, m->site_quat + 4 * s);
    mulmat3(d->site_xmat + 9 * s, d->xmat + 9 * b, R);
  }
  for (int c = 0; c < m->ncam; c++) {
    int b = m->cam_bodyid[c];
    double t[3], R[9];
    mulmat3vec(t, d->xmat + 9 * b, m->cam_pos + 3 * c);
    for (int k = 0; k < 3; k++) d->cam_xpos[3 * c + k] = d->xpos[3 * b + k] + t[k];
    quat2mat(R, m->cam_quat + 4 * c);
    mulmat3(d->cam_xmat + 9 * c, d->xmat + 9 * b, R);
  }
}

/* [MJ] mj_comPos: subtree COMs, cinert, cdof */
static void com_pos(const smjo_model* m, smjo_data* d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = m->body_mass[b] * d->xipos[3 * b + k];
  for (int b = nb - 1; b > 0; b--)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * m->body_parentid[b] + k] += d->subtree_com[3 * b + k];
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < MINVAL) memcpy(d->subtree_com + 3 * b, d->xipos + 3 * b, 24);
    else for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] /= m->body_subtreemass[b];
  }
  memset(d->cinert, 0, 80);
  for (int b = 1; b < nb; b++) {
    const double* R = d->ximat + 9 * b;
    const double* I = m->body_inertia + 3 * b;
    double mass = m->body_mass[b], dif[3], *c = d->cinert + 10 * b;
    for (int k = 0; k < 3; k++) dif[k] = d->xipos[3 * b + k] - d->subtree_com[3 * m->body_rootid[b] + k];
    double T[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        T[3 * i + j] = R[3 * i] * I[0] * R[3 * j] + R[3 * i + 1] * I[1] * R[3 * j + 1] + R[3 * i + 2] * I[2] * R[3 * j + 2];
    double dd = dot3(dif, dif);
    c[0] = T[0] + mass * (dd - dif[0] * dif[0]);
    c[1] = T[4] + mass * (dd - dif[1] * dif[1]);
    c[2] = T[8] + mass * (dd - dif[2] * dif[2]);
    c[3] = T[1] - mass * dif[0] * dif[1];
    c[4] = T[2] - mass * dif[0] * dif[2];
    c[5] = T[5] - mass * dif[1] * dif[2];
    c[6] = mass * dif[0]; c[7] = mass * dif[1]; c[8] = mass * dif[2]; c[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double* com = d->subtree_com + 3 * m->body_rootid[b];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = com[k] - d->xanchor[3 * j + k];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        double* c = d->cdof + 6 * (da + k);
        memset(c, 0, 48); c[3 + k] = 1;
        double ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        c = d->cdof + 6 * (da + 3 + k);
        memcpy(c, ax, 24); cross3(c + 3, ax, off);
      }
    } else if (m->jnt_type[j] == JNT_SLIDE) {
      double* c = d->cdof + 6 * da;
      c[0] = c[1] = c[2] = 0; memcpy(c + 3, d->xaxis + 3 * j, 24);
    } else {
      double* c = d->cdof + 6 * da;
      memcpy(c, d->xaxis + 3 * j, 24); cross3(c + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* [MJ] mj_tendon (fixed) + mj_transmission */
static void tendon_transmission(const smjo_model* m, smjo_data* d) {
  int nv = m->nv;
  memset(d->ten_J, 0, sizeof(double) * m->ntendon * nv);
  for (int t = 0; t < m->ntendon; t++) {
    double L = 0;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
      int j = m->wrap_objid[w];
      L += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]];
      d->ten_J[t * nv + m->jnt_dofadr[j]] = m->wrap_prm[w];
    }
    d->ten_length[t] = L;
  }
  memset(d->actuator_moment, 0, sizeof(double) * m->nu * nv);
  for (int a = 0; a < m->nu; a++) {
    double gear = m->actuator_gear[a];
    if (m->actuator_trntype[a] == 0) {
      int j = m->actuator_trnid[a];
      d->actuator_length[a] = gear * d->qpos[m->jnt_qposadr[j]];
      d->actuator_moment[a * nv + m->jnt_dofadr[j]] = gear;
    } else {
      int t = m->actuator_trnid[a];
      d->actuator_length[a] = gear * d->ten_length[t];
      for (int k = 0; k < nv; k++) d->actuator_moment[a * nv + k] = gear * d->ten_J[t * nv + k];
    }
  }
}

/* ------------------------------------------------------------------ B.2 inertia */
/* [MJ] mj_crb (dense storage) + factorisation (dense Cholesky instead of sparse L'DL: same M^-1) */
static void crb_factor(const smjo_model* m, smjo_data* d) {
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * nb);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) { FL(10); for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k]; }
  }
  memset(d->qM, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = 0;
      FL(12);
      for (int k = 0; k < 6; k++) v += d->cdof[6 * j + k] * buf[k];
      d->qM[i * nv + j] = d->qM[j * nv + i] = v;
    }
    d->qM[i * nv + i] += m->dof_armature[i];
  }
  memcpy(d->qL, d->qM, sizeof(double) * nv * nv);
  chol_factor(d->qL, nv, MINVAL);
}

/* Jacobian of a world point attached to body: jacp (3 x nv), jacr (3 x nv).  [MJ] mj_jac */
static void jac_point(const smjo_model* m, const smjo_data* d, double* jacp, double* jacr, const double* point, int body) {
  int nv = m->nv;
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body == 0) return;
  const double* com = d->subtree_com + 3 * m->body_rootid[body];
  double off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]};
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* c = d->cdof + 6 * i;
    if (jacr) { jacr[i] = c[0]; jacr[nv + i] = c[1]; jacr[2 * nv + i] = c[2]; }
    if (jacp) {
      double t[3];
      cross3(t, c, off);
      FL(3);
      jacp[i] = c[3] + t[0]; jacp[nv + i] = c[4] + t[1]; jacp[2 * nv + i] = c[5] + t[2];
    }
  }
}

/* ------------------------------------------------------------------ B.3 collision */
static void make_frame(double* f) { /* [MJ] mju_makeFrame: f[0:3] = normal given */
  normalize3(f);
  double* y = f + 3;
  if (f[1] > -0.5 && f[1] < 0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
  double t = dot3(f, y);
  for (int k = 0; k < 3; k++) y[k] -= t * f[k];
  normalize3(y);
  cross3(f + 6, f, y);
}

typedef struct { double dist, pos[3], normal[3]; } rawcon;

static int plane_sphere(const double* ppos, const double* pmat, const double* spos, double r, double margin, rawcon* c) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  double dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  c->dist = dist;
  memcpy(c->normal, n, 24);
  for (int k = 0; k < 3; k++) c->pos[k] = spos[k] - n[k] * (r + 0.5 * dist);
  return 1;
}

/* [MJ] mjc_PlaneCylinder */
static int plane_cylinder(const double* ppos, const double* pmat, const double* cpos, const double* cmat, const double* size,
                          double margin, rawcon* con) {
  double normal[3] = {pmat[2], pmat[5], pmat[8]}, axis[3] = {cmat[2], cmat[5], cmat[8]};
  double prjaxis = dot3(normal, axis);
  if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
  double vec[3] = {cpos[0] - ppos[0], cpos[1] - ppos[1], cpos[2] - ppos[2]};
  double dist0 = dot3(vec, normal);
  for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - normal[k];
  double len2 = dot3(vec, vec);
  if (len2 >= MINVAL * MINVAL) { double s = size[0] / sqrt(len2); for (int k = 0; k < 3; k++) vec[k] *= s; }
  else { vec[0] = cmat[0] * size[0]; vec[1] = cmat[3] * size[0]; vec[2] = cmat[6] * size[0]; }
  double prjvec = dot3(vec, normal);
  for (int k = 0; k < 3; k++) axis[k] *= size[1];
  prjaxis *= size[1];
  int cnt = 0;
  if (dist0 + prjaxis + prjvec <= margin) {
    double dist = dist0 + prjaxis + prjvec;
    con[cnt].dist = dist; memcpy(con[cnt].normal, normal, 24);
    for (int k = 0; k < 3; k++) con[cnt].pos[k] = cpos[k] + vec[k] + axis[k] - normal[k] * dist * 0.5;
    cnt++;
  } else return 0;
  if (dist0 - prjaxis + prjvec <= margin) {
    double dist = dist0 - prjaxis + prjvec;
    con[cnt].dist = dist; memcpy(con[cnt].normal, normal, 24);
    for (int k = 0; k < 3; k++) con[cnt].pos[k] = cpos[k] + vec[k] - axis[k] - normal[k] * dist * 0.5;
    cnt++;
  }
  double prjvec1 = -prjvec * 0.5;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    double vec1[3];
    cross3(vec1, vec, axis);
    normalize3(vec1);
    for (int k = 0; k < 3; k++) vec1[k] *= size[0] * sqrt(3.0) * 0.5;
    double dist = dist0 + prjaxis + prjvec1;
    for (int s = 0; s < 2; s++) {
      double sg = s ? -1.0 : 1.0;
      con[cnt].dist = dist; memcpy(con[cnt].normal, normal, 24);
      for (int k = 0; k < 3; k++) con[cnt].pos[k] = cpos[k] + sg * vec1[k] + axis[k] - vec[k] * 0.5 - normal[k] * dist * 0.5;
      cnt++;
    }
  }
  return cnt;
}

/* [MJ] mjc_PlaneBox */
static int plane_box(const double* ppos, const double* pmat, const double* bpos, const double* bmat, const double* size,
                     double margin, rawcon* con) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3] = {bpos[0] - ppos[0], bpos[1] - ppos[1], bpos[2] - ppos[2]};
  double dist = dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3] = {(i & 1 ? size[0] : -size[0]), (i & 2 ? size[1] : -size[1]), (i & 4 ? size[2] : -size[2])}, corner[3];
    mulmat3vec(corner, bmat, vec);
    double ldist = dot3(n, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    double cd = dist + ldist;
    con[cnt].dist = cd; memcpy(con[cnt].normal, n, 24);
    for (int k = 0; k < 3; k++) con[cnt].pos[k] = bpos[k] + corner[k] - n[k] * cd * 0.5;
    if (++cnt >= 4) return 4;
  }
  return cnt;
}

/* plane vs convex hull: the build's own manifold rule (MuJoCo's mjc_PlaneConvex picks support vertex +
 * up to 3 neighbours; not restatable bit-for-bit): among hull vertices with dist <= margin take
 * (1) the deepest, (2) the one farthest from (1), (3,4) the extremes of signed distance to the line (1)-(2). */
static int plane_hull(const double* ppos, const double* pmat, const double* gpos, const double* gmat, const double* verts,
                      int nvert, double margin, int maxc, rawcon* con) {
  double n[3] = {pmat[2], pmat[5], pmat[8]};
  double nl[3];
  mulmat3Tvec(nl, gmat, n); /* plane normal in geom frame */
  double off = (gpos[0] - ppos[0]) * n[0] + (gpos[1] - ppos[1]) * n[1] + (gpos[2] - ppos[2]) * n[2];
  int i1 = -1, i2 = -1, i3 = -1, i4 = -1;
  double best = margin;
  for (int i = 0; i < nvert; i++) {
    double dd = dot3(nl, verts + 3 * i) + off;
    if (dd <= margin && (i1 < 0 || dd < best)) { best = dd; i1 = i; }
  }
  if (i1 < 0) return 0;
  const double* v1 = verts + 3 * i1;
  double far2 = 1e-12;
  for (int i = 0; i < nvert && maxc > 1; i++) {
    double dd = dot3(nl, verts + 3 * i) + off;
    if (dd > margin) continue;
    double e[3] = {verts[3 * i] - v1[0], verts[3 * i + 1] - v1[1], verts[3 * i + 2] - v1[2]};
    double r2 = dot3(e, e);
    if (r2 > far2) { far2 = r2; i2 = i; }
  }
  if (i2 >= 0 && maxc > 2) {
    const double* v2 = verts + 3 * i2;
    double e12[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]}, side[3];
    cross3(side, nl, e12);
    normalize3(side);
    double smax = 1e-6, smin = -1e-6;
    for (int i = 0; i < nvert; i++) {
      double dd = dot3(nl, verts + 3 * i) + off;
      if (dd > margin) continue;
      double e[3] = {verts[3 * i] - v1[0], verts[3 * i + 1] - v1[1], verts[3 * i + 2] - v1[2]};
      double s = dot3(e, side);
      if (s > smax) { smax = s; i3 = i; }
      if (s < smin) { smin = s; i4 = i; }
    }
    if (maxc < 4) i4 = -1;
  }
  int idx[4] = {i1, i2, i3, i4}, cnt = 0;
  for (int k = 0; k < 4; k++) {
    if (idx[k] < 0) continue;
    const double* v = verts + 3 * idx[k];
    double w[3], dd = dot3(nl, v) + off;
    mulmat3vec(w, gmat, v);
    con[cnt].dist = dd; memcpy(con[cnt].normal, n, 24);
    for (int j = 0; j < 3; j++) con[cnt].pos[j] = gpos[j] + w[j] - n[j] * dd * 0.5;
    cnt++;
  }
  return cnt;
}


/* ---- convex-convex narrowphase: Minkowski Portal Refinement, restating libccd's ccdMPRPenetration (mpr.c), the
 * routine MuJoCo 3.2.6's mjc_Convex calls for mesh / cylinder / box pairs without an analytic function
 * ([MJ]; libccd is a third-party dependency of mujoco, not in /root/reference).  One contact per pair (MuJoCo's
 * multiccd adds up to three more by re-running with a perturbed normal: not restated).  mpr_tolerance 1e-6,
 * mpr_iterations 50 are MuJoCo's option defaults. */
typedef struct { double v[3], a[3], b[3]; } mprpt;
typedef struct {
  const smjo_model* m;
  int g[2], type[2], nvert[2];
  const double *pos[2], *mat[2], *size[2], *verts[2];
} mprctx;

static void shape_support(const mprctx* c, int k, const double* dir, double* out) {
  double dl[3], pl[3] = {0, 0, 0};
  mulmat3Tvec(dl, c->mat[k], dir);
  const double* sz = c->size[k];
  int t = c->type[k];
  if (t == G_SPHERE) { double n = norm3(dl); if (n > MINVAL) for (int i = 0; i < 3; i++) pl[i] = sz[0] * dl[i] / n; }
  else if (t == G_BOX) { for (int i = 0; i < 3; i++) pl[i] = dl[i] >= 0 ? sz[i] : -sz[i]; }
  else if (t == G_CYLINDER) {
    double n = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
    if (n > MINVAL) { pl[0] = sz[0] * dl[0] / n; pl[1] = sz[0] * dl[1] / n; }
    pl[2] = dl[2] >= 0 ? sz[1] : -sz[1];
  } else if (t == G_CAPSULE) {      /* [MJ] mjc_support: the sphere's support point, moved to the end the direction points to */
    double n = norm3(dl);
    if (n > MINVAL) for (int i = 0; i < 3; i++) pl[i] = sz[0] * dl[i] / n;
    pl[2] += dl[2] > 0 ? sz[1] : (dl[2] < 0 ? -sz[1] : 0);
  } else if (t == G_ELLIPSOID) {    /* [MJ] mjc_support: unit-sphere support of the scaled direction, scaled back */
    double tv[3] = {dl[0] * sz[0], dl[1] * sz[1], dl[2] * sz[2]}, n = norm3(tv);
    if (n > MINVAL) for (int i = 0; i < 3; i++) pl[i] = sz[i] * tv[i] / n;
  } else if (t == G_MESH) {
    int best = 0;
    double bd = -1e300;
    for (int i = 0; i < c->nvert[k]; i++) {
      double d = dot3(c->verts[k] + 3 * i, dl);
      if (d > bd) { bd = d; best = i; }
    }
    memcpy(pl, c->verts[k] + 3 * best, 24);
  }
  mulmat3vec(out, c->mat[k], pl);
  for (int i = 0; i < 3; i++) out[i] += c->pos[k][i];
}
static void mpr_support(const mprctx* c, const double* dir, mprpt* p) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  shape_support(c, 0, dir, p->a);
  shape_support(c, 1, nd, p->b);
  for (int i = 0; i < 3; i++) p->v[i] = p->a[i] - p->b[i];
}
#define CCD_EPS 2.220446049250313e-16
static int ccd_zero(double x) { return fabs(x) < CCD_EPS; }
static int ccd_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < CCD_EPS) return 1;
  double aa = fabs(a), bb = fabs(b);
  return ab < CCD_EPS * (bb > aa ? bb : aa);
}
static void portal_dir(const mprpt* P, double* dir) {
  double a[3], b[3];
  for (int i = 0; i < 3; i++) { a[i] = P[2].v[i] - P[1].v[i]; b[i] = P[3].v[i] - P[1].v[i]; }
  cross3(dir, a, b);
  double n = norm3(dir);
  if (n > 0) for (int i = 0; i < 3; i++) dir[i] /= n;
}
static void expand_portal(mprpt* P, const mprpt* v4) {
  double v4v0[3];
  cross3(v4v0, v4->v, P[0].v);
  if (dot3(P[1].v, v4v0) > 0) { if (dot3(P[2].v, v4v0) > 0) P[1] = *v4; else P[3] = *v4; }
  else { if (dot3(P[3].v, v4v0) > 0) P[2] = *v4; else P[1] = *v4; }
}
static int reach_tolerance(const mprpt* P, const mprpt* v4, const double* dir, double tol) {
  double dv4 = dot3(v4->v, dir), d1 = dv4 - dot3(P[1].v, dir), d2 = dv4 - dot3(P[2].v, dir), d3 = dv4 - dot3(P[3].v, dir);
  double d = fmin(d1, fmin(d2, d3));
  return ccd_eq(d, tol) || d < tol;
}
/* squared distance from the origin to triangle (a,b,c), closest point in w  (libccd ccdVec3PointTriDist2) */
static double origin_tri_dist2(const double* a, const double* b, const double* c, double* w) {
  double d1[3], d2[3], av[3];
  for (int i = 0; i < 3; i++) { d1[i] = b[i] - a[i]; d2[i] = c[i] - a[i]; av[i] = a[i]; }
  double u = dot3(av, av), v = dot3(d1, d1), ww = dot3(d2, d2), p = dot3(av, d1), q = dot3(av, d2), r = dot3(d1, d2);
  double den = ww * v - r * r, s = 0, t = 0, best = -1;
  if (!ccd_zero(den)) { s = (q * r - ww * p) / den; t = (-s * r - q) / ww; } else s = t = -1;
  if ((ccd_zero(s) || s > 0) && (ccd_eq(s, 1) || s < 1) && (ccd_zero(t) || t > 0) && (ccd_eq(t, 1) || t < 1) && (ccd_eq(t + s, 1) || t + s < 1)) {
    for (int i = 0; i < 3; i++) w[i] = a[i] + s * d1[i] + t * d2[i];
    return u + s * s * v + t * t * ww + 2 * s * p + 2 * t * q + 2 * s * t * r;
  }
  const double* seg[3][2] = {{a, b}, {a, c}, {b, c}};
  for (int e = 0; e < 3; e++) {
    double dd[3], tt, wp[3];
    for (int i = 0; i < 3; i++) dd[i] = seg[e][1][i] - seg[e][0][i];
    double l2 = dot3(dd, dd);
    tt = l2 > 0 ? -dot3(seg[e][0], dd) / l2 : 0;
    tt = tt < 0 ? 0 : (tt > 1 ? 1 : tt);
    for (int i = 0; i < 3; i++) wp[i] = seg[e][0][i] + tt * dd[i];
    double dist = dot3(wp, wp);
    if (best < 0 || dist < best) { best = dist; memcpy(w, wp, 24); }
  }
  return best;
}
/* returns 1 and (depth, dir, pos) if the shapes penetrate */
static int mpr_penetration(const mprctx* c, const double* c0, const double* c1, double* depth, double* pdir, double* pos) {
  const double tol = 1e-6;
  const int maxit = 50;
  mprpt P[4], v4;
  double dir[3], va[3], vb[3], dot;
  for (int i = 0; i < 3; i++) { P[0].a[i] = c0[i]; P[0].b[i] = c1[i]; P[0].v[i] = c0[i] - c1[i]; }
  if (ccd_zero(P[0].v[0]) && ccd_zero(P[0].v[1]) && ccd_zero(P[0].v[2])) P[0].v[0] += CCD_EPS * 10;
  for (int i = 0; i < 3; i++) dir[i] = -P[0].v[i];
  normalize3(dir);
  mpr_support(c, dir, &P[1]);
  dot = dot3(P[1].v, dir);
  if (ccd_zero(dot) || dot < 0) return 0;
  cross3(dir, P[0].v, P[1].v);
  if (ccd_zero(dot3(dir, dir))) {
    /* origin on v1 (touching) or on the v0-v1 segment: libccd findPenetrTouch / findPenetrSegment */
    if (ccd_zero(P[1].v[0]) && ccd_zero(P[1].v[1]) && ccd_zero(P[1].v[2])) {
      *depth = 0; pdir[0] = pdir[1] = pdir[2] = 0;
    } else {
      *depth = norm3(P[1].v);
      for (int i = 0; i < 3; i++) pdir[i] = P[1].v[i];
      normalize3(pdir);
    }
    for (int i = 0; i < 3; i++) pos[i] = 0.5 * (P[1].a[i] + P[1].b[i]);
    return 1;
  }
  normalize3(dir);
  mpr_support(c, dir, &P[2]);
  dot = dot3(P[2].v, dir);
  if (ccd_zero(dot) || dot < 0) return 0;
  for (int i = 0; i < 3; i++) { va[i] = P[1].v[i] - P[0].v[i]; vb[i] = P[2].v[i] - P[0].v[i]; }
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, P[0].v) > 0) { mprpt t = P[1]; P[1] = P[2]; P[2] = t; for (int i = 0; i < 3; i++) dir[i] = -dir[i]; }
  for (int it = 0;; it++) {
    if (it > 100) return 0;
    mpr_support(c, dir, &P[3]);
    dot = dot3(P[3].v, dir);
    if (ccd_zero(dot) || dot < 0) return 0;
    int cont = 0;
    cross3(va, P[1].v, P[3].v);
    dot = dot3(va, P[0].v);
    if (dot < 0 && !ccd_zero(dot)) { P[2] = P[3]; cont = 1; }
    if (!cont) {
      cross3(va, P[3].v, P[2].v);
      dot = dot3(va, P[0].v);
      if (dot < 0 && !ccd_zero(dot)) { P[1] = P[3]; cont = 1; }
    }
    if (!cont) break;
    for (int i = 0; i < 3; i++) { va[i] = P[1].v[i] - P[0].v[i]; vb[i] = P[2].v[i] - P[0].v[i]; }
    cross3(dir, va, vb);
    normalize3(dir);
  }
  /* refine portal */
  for (int it = 0;; it++) {
    portal_dir(P, dir);
    dot = dot3(dir, P[1].v);
    if (ccd_zero(dot) || dot > 0) break; /* origin inside */
    mpr_support(c, dir, &v4);
    dot = dot3(v4.v, dir);
    if (!(ccd_zero(dot) || dot > 0) || reach_tolerance(P, &v4, dir, tol) || it > maxit) return 0;
    expand_portal(P, &v4);
  }
  /* penetration info */
  for (int it = 0;; it++) {
    portal_dir(P, dir);
    mpr_support(c, dir, &v4);
    if (reach_tolerance(P, &v4, dir, tol) || it > maxit) {
      *depth = sqrt(fmax(0.0, origin_tri_dist2(P[1].v, P[2].v, P[3].v, pdir)));
      if (ccd_zero(pdir[0]) && ccd_zero(pdir[1]) && ccd_zero(pdir[2])) memcpy(pdir, dir, 24);
      normalize3(pdir);
      double b[4], vec[3], sum;
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
      if (!(fabs(sum) > 1e-300)) { /* fully degenerate portal (exactly touching faces): use the portal vertex witnesses */
        b[0] = 0; b[1] = b[2] = b[3] = 1; sum = 3;
      }
      for (int i = 0; i < 3; i++) {
        double p1 = 0, p2 = 0;
        for (int k = 0; k < 4; k++) { p1 += b[k] * P[k].a[i]; p2 += b[k] * P[k].b[i]; }
        pos[i] = 0.5 * (p1 + p2) / sum;
      }
      return 1;
    }
    expand_portal(P, &v4);
  }
}

/* broadphase for a non-plane pair: bounding spheres, then the 6 face axes of the two oriented bounding boxes */
static int obb_overlap(const smjo_model* m, const smjo_data* d, int g1, int g2, double margin) {
  const double *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  double c1[3], c2[3], dv[3];
  mulmat3vec(c1, R1, m->geom_aabb + 6 * g1); mulmat3vec(c2, R2, m->geom_aabb + 6 * g2);
  for (int k = 0; k < 3; k++) dv[k] = (d->geom_xpos[3 * g2 + k] + c2[k]) - (d->geom_xpos[3 * g1 + k] + c1[k]);
  if (norm3(dv) > m->geom_rbound[g1] + m->geom_rbound[g2] + margin) return 0;
  const double *h1 = m->geom_aabb + 6 * g1 + 3, *h2 = m->geom_aabb + 6 * g2 + 3;
  for (int s = 0; s < 2; s++) {
    const double *Ra = s ? R2 : R1, *Rb = s ? R1 : R2, *ha = s ? h2 : h1, *hb = s ? h1 : h2;
    for (int k = 0; k < 3; k++) {
      double ax[3] = {Ra[k], Ra[3 + k], Ra[6 + k]}, r = 0;
      for (int j = 0; j < 3; j++) { double bj[3] = {Rb[j], Rb[3 + j], Rb[6 + j]}; r += fabs(dot3(ax, bj)) * hb[j]; }
      if (fabs(dot3(ax, dv)) > ha[k] + r + margin) return 0;
    }
  }
  return 1;
}
/* [MJ] mjc_SphereBox (primitive pairs do not go through the convex solver): sphere s against box b; normal from the
 * sphere to the box */
static int sphere_box(const double* spos, double r, const double* bpos, const double* bmat, const double* bsize, double margin,
                      rawcon* rc) {
  double tmp[3] = {spos[0] - bpos[0], spos[1] - bpos[1], spos[2] - bpos[2]}, cen[3], cl[3], dif[3];
  mulmat3Tvec(cen, bmat, tmp);
  for (int i = 0; i < 3; i++) { cl[i] = cen[i] > bsize[i] ? bsize[i] : (cen[i] < -bsize[i] ? -bsize[i] : cen[i]); dif[i] = cl[i] - cen[i]; }
  double dist = norm3(dif), nl[3], pl[3];
  if (dist - r > margin) return 0;
  if (dist <= MINVAL) { /* centre inside the box: nearest face */
    double closest = 2 * fmax(bsize[0], fmax(bsize[1], bsize[2]));
    int k = 0;
    for (int i = 0; i < 6; i++) {
      double cd = fabs((i % 2 ? 1 : -1) * bsize[i / 2] - cen[i / 2]);
      if (cd < closest) { closest = cd; k = i; }
    }
    nl[0] = nl[1] = nl[2] = 0; nl[k / 2] = k % 2 ? -1 : 1;
    for (int i = 0; i < 3; i++) pl[i] = cen[i] + nl[i] * (r - closest) / 2;
    rc->dist = -closest - r;
  } else {
    for (int i = 0; i < 3; i++) { nl[i] = dif[i] / dist; pl[i] = 0.5 * (cl[i] + cen[i] + dif[i] * (r / dist)); }
    rc->dist = dist - r;
  }
  mulmat3vec(rc->normal, bmat, nl);
  mulmat3vec(rc->pos, bmat, pl);
  for (int i = 0; i < 3; i++) rc->pos[i] += bpos[i];
  return 1;
}
/* [MJ] mjc_SphereSphere */
static int sphere_sphere(const double* p1, double r1, const double* p2, double r2, double margin, rawcon* rc) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, cdist = norm3(dif);
  if (cdist - r1 - r2 > margin) return 0;
  if (cdist < MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; } else for (int i = 0; i < 3; i++) dif[i] /= cdist;
  rc->dist = cdist - r1 - r2;
  for (int i = 0; i < 3; i++) { rc->normal[i] = dif[i]; rc->pos[i] = p1[i] + dif[i] * (r1 + 0.5 * rc->dist); }
  return 1;
}

/* [MJ] mjc_SphereCapsule (engine_collision_primitive.c): the sphere against the sphere of the capsule's radius centred on the
 * nearest point of the capsule's segment.  size = (radius, half length), axis = the frame's z. */
static int sphere_capsule(const double* p1, double r1, const double* p2, const double* mat2, const double* size2, double margin, rawcon* rc) {
  const double ax[3] = {mat2[2], mat2[5], mat2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double x = dot3(ax, vec);
  x = fmax(-size2[1], fmin(size2[1], x));
  const double q[3] = {p2[0] + ax[0] * x, p2[1] + ax[1] * x, p2[2] + ax[2] * x};
  return sphere_sphere(p1, r1, q, size2[0], margin, rc);
}
/* [MJ] mjc_CapsuleCapsule: nearest points of the two segments (axes scaled by the half lengths, parameters in [-1, 1], the
 * clamped one re-solved for the other), then sphere-sphere; PARALLEL axes (det < mjMINVAL): the two ends of capsule 1 against
 * the segment of capsule 2, then -- while fewer than two contacts -- the two ends of capsule 2 against segment 1: up to 2 contacts. */
static int capsule_capsule(const double* p1, const double* mat1, const double* size1, const double* p2, const double* mat2,
                           const double* size2, double margin, rawcon* rc) {
  const double a1[3] = {mat1[2] * size1[1], mat1[5] * size1[1], mat1[8] * size1[1]}, a2[3] = {mat2[2] * size2[1], mat2[5] * size2[1], mat2[8] * size2[1]};
  const double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = ma * mc - mb * mb;
  double v1[3], v2[3];
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > 1) { x1 = 1; x2 = (v - mb) / mc; }
    else if (x1 < -1) { x1 = -1; x2 = (v + mb) / mc; }
    if (x2 > 1) { x2 = 1; x1 = fmax(-1, fmin(1, (u - mb) / ma)); }
    else if (x2 < -1) { x2 = -1; x1 = fmax(-1, fmin(1, (u + mb) / ma)); }
    for (int i = 0; i < 3; i++) { v1[i] = p1[i] + a1[i] * x1; v2[i] = p2[i] + a2[i] * x2; }
    return sphere_sphere(v1, size1[0], v2, size2[0], margin, rc);
  }
  int n = 0;
  for (int e = 0; e < 2; e++) {   /* x1 = +1, -1 */
    const double sg = e ? -1 : 1, x2 = fmax(-1, fmin(1, (v - sg * mb) / mc));
    for (int i = 0; i < 3; i++) { v1[i] = p1[i] + sg * a1[i]; v2[i] = p2[i] + a2[i] * x2; }
    n += sphere_sphere(v1, size1[0], v2, size2[0], margin, rc + n);
  }
  for (int e = 0; e < 2 && n < 2; e++) {   /* x2 = +1, -1 */
    const double sg = e ? -1 : 1, x1 = fmax(-1, fmin(1, (u - sg * mb) / ma));
    for (int i = 0; i < 3; i++) { v1[i] = p1[i] + a1[i] * x1; v2[i] = p2[i] + sg * a2[i]; }
    n += sphere_sphere(v1, size1[0], v2, size2[0], margin, rc + n);
  }
  return n;
}

/* ------------------------------------------------------------------ box-box, multi-point convex contacts
 * [MJ] mjc_BoxBox (engine_collision_box.c) and the multiccd branch of mjc_Convex (engine_collision_convex.c).  MuJoCo's
 * sources are not available here; what follows restates their published behaviour -- separating-axis test over the 15 axes,
 * a face contact clipped to a polygon of up to 8 points with one depth each, a single point for an edge-edge contact; for
 * other convex pairs the first contact is followed by four more penetration queries with the two geoms counter-rotated by a
 * small angle about the tangent axes -- NOT their line-by-line arithmetic (see the header: parity unpinned).  The kernel
 * (csrc/smj_step_impl.h box_box / convex_multi) implements exactly this formulation. */
#define BOXBOX_FUDGE 1.05          /* face axes are preferred over edge axes by this factor (as in ODE's dBoxBox) */
#define MULTICCD_ANGLE 1e-3        /* counter-rotation of the two geoms, radians */
#define MULTICCD_RELTOL 1e-3       /* new contacts closer than this x min(rbound) to an earlier one are duplicates */

static int box_box(const double* p1, const double* R1, const double* A, const double* p2, const double* R2, const double* B,
                   double margin, int maxcon, rawcon* rc) {
  double p[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, R[3][3], Q[3][3], pp[3], pq[3];
  for (int i = 0; i < 3; i++) {
    const double ai[3] = {R1[i], R1[3 + i], R1[6 + i]};
    pp[i] = dot3(p, ai);
    for (int j = 0; j < 3; j++) { const double bj[3] = {R2[j], R2[3 + j], R2[6 + j]}; R[i][j] = dot3(ai, bj); Q[i][j] = fabs(R[i][j]); }
  }
  for (int j = 0; j < 3; j++) { const double bj[3] = {R2[j], R2[3 + j], R2[6 + j]}; pq[j] = dot3(p, bj); }
  /* separating axes: codes 0-2 faces of box 1, 3-5 faces of box 2, 6-14 edge i of box 1 x edge j of box 2 */
  double best = -1e300;
  int code = -1;
  for (int i = 0; i < 3; i++) {
    const double s = fabs(pp[i]) - (A[i] + B[0] * Q[i][0] + B[1] * Q[i][1] + B[2] * Q[i][2]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    const double s = fabs(pq[j]) - (B[j] + A[0] * Q[0][j] + A[1] * Q[1][j] + A[2] * Q[2][j]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = 3 + j; }
  }
  double en[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double l = sqrt(fmax(0.0, 1.0 - R[i][j] * R[i][j]));
      if (l < 1e-6) continue;                 /* parallel edges: covered by the face axes */
      const double e = pp[i2] * R[i1][j] - pp[i1] * R[i2][j];
      const double s = (fabs(e) - (A[i1] * Q[i2][j] + A[i2] * Q[i1][j] + B[j1] * Q[i][j2] + B[j2] * Q[i][j1])) / l;
      if (s > margin) return 0;
      if (s * BOXBOX_FUDGE > best && s > best) {
        best = s; code = 6 + 3 * i + j;
        const double ai[3] = {R1[i], R1[3 + i], R1[6 + i]}, bj[3] = {R2[j], R2[3 + j], R2[6 + j]};
        cross3(en, ai, bj);
        for (int k = 0; k < 3; k++) en[k] /= l;
      }
    }
  if (code >= 6) {   /* edge-edge: one point, midway between the closest points of the two edges */
    double n[3] = {en[0], en[1], en[2]};
    if (dot3(n, p) < 0) for (int k = 0; k < 3; k++) n[k] = -n[k];
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    double pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
    for (int k = 0; k < 3; k++) {
      const double ak[3] = {R1[k], R1[3 + k], R1[6 + k]}, bk[3] = {R2[k], R2[3 + k], R2[6 + k]};
      const double sa = dot3(n, ak) > 0 ? 1.0 : -1.0, sb = dot3(n, bk) > 0 ? -1.0 : 1.0;
      for (int x = 0; x < 3; x++) { pa[x] += sa * A[k] * ak[x]; pb[x] += sb * B[k] * bk[x]; }
    }
    const double ua[3] = {R1[i], R1[3 + i], R1[6 + i]}, ub[3] = {R2[j], R2[3 + j], R2[6 + j]};
    const double dd[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double uaub = dot3(ua, ub), q1 = dot3(ua, dd), q2 = -dot3(ub, dd), den = 1.0 - uaub * uaub;
    double al = 0, be = 0;
    if (den > 1e-12) { al = (q1 + uaub * q2) / den; be = (uaub * q1 + q2) / den; }
    for (int x = 0; x < 3; x++) { rc[0].pos[x] = 0.5 * ((pa[x] + al * ua[x]) + (pb[x] + be * ub[x])); rc[0].normal[x] = n[x]; }
    rc[0].dist = best;
    return 1;
  }
  /* face contact.  Reference box a (the one owning the axis), incident box b; n from a to b */
  const int swap = code >= 3, ia = swap ? code - 3 : code;
  const double *pa = swap ? p2 : p1, *Ra = swap ? R2 : R1, *ha = swap ? B : A, *pb = swap ? p1 : p2, *Rb = swap ? R1 : R2, *hb = swap ? A : B;
  double n[3] = {Ra[ia], Ra[3 + ia], Ra[6 + ia]}, ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  if (dot3(n, ab) < 0) for (int k = 0; k < 3; k++) n[k] = -n[k];
  const int ja = (ia + 1) % 3, ka = (ia + 2) % 3;
  const double u[3] = {Ra[ja], Ra[3 + ja], Ra[6 + ja]}, v[3] = {Ra[ka], Ra[3 + ka], Ra[6 + ka]}, hu = ha[ja], hv = ha[ka];
  double cA[3];
  for (int k = 0; k < 3; k++) cA[k] = pa[k] + n[k] * ha[ia];
  /* incident face: the face of b most anti-parallel to n */
  int ib = 0;
  double bd = -1;
  for (int k = 0; k < 3; k++) { const double bk[3] = {Rb[k], Rb[3 + k], Rb[6 + k]}; const double t = fabs(dot3(bk, n)); if (t > bd) { bd = t; ib = k; } }
  double nb[3] = {Rb[ib], Rb[3 + ib], Rb[6 + ib]};
  if (dot3(nb, n) > 0) for (int k = 0; k < 3; k++) nb[k] = -nb[k];
  const int jb = (ib + 1) % 3, kb = (ib + 2) % 3;
  const double pv[3] = {Rb[jb], Rb[3 + jb], Rb[6 + jb]}, qv[3] = {Rb[kb], Rb[3 + kb], Rb[6 + kb]}, hp = hb[jb], hq = hb[kb];
  double cB[3], w[4][3], wu[4], wv[4];
  for (int k = 0; k < 3; k++) cB[k] = pb[k] + nb[k] * hb[ib];
  static const double sg[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};   /* counter-clockwise corner order */
  for (int c = 0; c < 4; c++) {
    double d[3];
    for (int k = 0; k < 3; k++) { w[c][k] = cB[k] + sg[c][0] * hp * pv[k] + sg[c][1] * hq * qv[k]; d[k] = w[c][k] - cA[k]; }
    wu[c] = dot3(d, u); wv[c] = dot3(d, v);
  }
  const double nnb = dot3(n, nb);   /* < 0 */
  /* 24 candidates in a fixed order: 4 incident corners inside the reference face, 4 reference corners inside the incident
   * face, 16 crossings of an incident edge with a reference edge.  Each is a point x on the incident face with depth below
   * the reference face; the contact sits midway between the two surfaces. */
  double cx[24][3], cdepth[24], cu[24], cv[24];
  int cok[24], nok = 0;
  for (int cand = 0; cand < 24; cand++) {
    double x[3] = {0, 0, 0};
    int ok = 0;
    if (cand < 4) {
      ok = fabs(wu[cand]) <= hu * (1 + 1e-9) && fabs(wv[cand]) <= hv * (1 + 1e-9);   /* (a corner ON the reference face's edge -- boxes of equal size stacked in line -- is inside) */
      memcpy(x, w[cand], 24);
    } else if (cand < 8) {
      const int c = cand - 4;
      double r[3], d[3];
      for (int k = 0; k < 3; k++) { r[k] = cA[k] + sg[c][0] * hu * u[k] + sg[c][1] * hv * v[k]; d[k] = cB[k] - r[k]; }
      const double t = dot3(d, nb) / nnb;
      for (int k = 0; k < 3; k++) { x[k] = r[k] + t * n[k]; d[k] = x[k] - cB[k]; }
      ok = fabs(dot3(d, pv)) <= hp * (1 + 1e-9) && fabs(dot3(d, qv)) <= hq * (1 + 1e-9);   /* (inclusive, as the incident corners: coincident edges) */
    } else {
      const int e = (cand - 8) / 4, r = (cand - 8) % 4, c0 = e, c1 = (e + 1) % 4;   /* incident edge e, reference edge r */
      const int along_u = r < 2;                         /* r = 0,1: the lines u = -hu, +hu; r = 2,3: v = -hv, +hv */
      const double lim = (r & 1 ? 1.0 : -1.0) * (along_u ? hu : hv);
      const double a0 = along_u ? wu[c0] : wv[c0], a1 = along_u ? wu[c1] : wv[c1], b0 = along_u ? wv[c0] : wu[c0], b1 = along_u ? wv[c1] : wu[c1];
      const double den = a1 - a0;
      if (fabs(den) > 1e-12) {
        const double t = (lim - a0) / den, o = b0 + t * (b1 - b0);
        ok = t > 0 && t < 1 && fabs(o) < (along_u ? hv : hu);
        for (int k = 0; k < 3; k++) x[k] = w[c0][k] + t * (w[c1][k] - w[c0][k]);
      }
    }
    const double d[3] = {x[0] - cA[0], x[1] - cA[1], x[2] - cA[2]};
    cdepth[cand] = -dot3(d, n); cu[cand] = dot3(d, u); cv[cand] = dot3(d, v);
    if (-cdepth[cand] > margin) ok = 0;
    cok[cand] = ok; nok += ok;
    memcpy(cx[cand], x, 24);
  }
  /* candidates that coincide (a corner of one face ON an edge or a corner of the other: boxes of equal size stacked in line) are
   * one point: the later one goes, before the count is taken (within 1e-5 m) */
  for (int cand = 1; cand < 24; cand++) {
    if (!cok[cand]) continue;
    const double pc[3] = {cx[cand][0] + 0.5 * cdepth[cand] * n[0], cx[cand][1] + 0.5 * cdepth[cand] * n[1], cx[cand][2] + 0.5 * cdepth[cand] * n[2]};
    for (int c2 = 0; c2 < cand; c2++) {
      if (!cok[c2]) continue;
      const double p2[3] = {cx[c2][0] + 0.5 * cdepth[c2] * n[0], cx[c2][1] + 0.5 * cdepth[c2] * n[1], cx[c2][2] + 0.5 * cdepth[c2] * n[2]};
      if (fabs(pc[0] - p2[0]) < 1e-5 && fabs(pc[1] - p2[1]) < 1e-5 && fabs(pc[2] - p2[2]) < 1e-5) { cok[cand] = 0; nok--; break; }
    }
  }
  /* more points than max_contacts_per_pair: keep the extreme ones along the two axes of the reference face (ties: lowest
   * candidate), a support polygon as wide as the full one */
  if (nok > maxcon) {
    int keep[24] = {0}, pick[4] = {-1, -1, -1, -1};
    for (int cand = 0; cand < 24; cand++) {
      if (!cok[cand]) continue;
      if (pick[0] < 0 || cu[cand] < cu[pick[0]]) pick[0] = cand;
      if (pick[1] < 0 || cu[cand] > cu[pick[1]]) pick[1] = cand;
      if (pick[2] < 0 || cv[cand] < cv[pick[2]]) pick[2] = cand;
      if (pick[3] < 0 || cv[cand] > cv[pick[3]]) pick[3] = cand;
    }
    int nkeep = 0;
    for (int k = 0; k < 4; k++) { nkeep += !keep[pick[k]]; keep[pick[k]] = 1; }
    /* (one point can be extreme in two directions: the places left go to the remaining candidates, lowest first) */
    for (int cand = 0; cand < 24 && nkeep < maxcon; cand++)
      if (cok[cand] && !keep[cand]) { keep[cand] = 1; nkeep++; }
    for (int cand = 0; cand < 24; cand++) cok[cand] = cok[cand] && keep[cand];
  }
  int cnt = 0;
  for (int cand = 0; cand < 24 && cnt < 8; cand++) {
    if (!cok[cand]) continue;
    rc[cnt].dist = -cdepth[cand];
    for (int k = 0; k < 3; k++) { rc[cnt].pos[k] = cx[cand][k] + 0.5 * cdepth[cand] * n[k]; rc[cnt].normal[k] = swap ? -n[k] : n[k]; }
    cnt++;
  }
  return cnt;
}

/* rotate a pose about the point c by the rotation matrix Rm */
static void rotate_pose(double* pos, double* mat, const double* c, const double* Rm) {
  double d[3] = {pos[0] - c[0], pos[1] - c[1], pos[2] - c[2]}, r[3], t[9];
  mulmat3vec(r, Rm, d);
  for (int k = 0; k < 3; k++) pos[k] = c[k] + r[k];
  mulmat3(t, Rm, mat);
  memcpy(mat, t, 72);
}
static void axis_angle_mat(double* Rm, const double* ax, double ang) {
  const double c = cos(ang), s = sin(ang), t = 1 - c, x = ax[0], y = ax[1], z = ax[2];
  Rm[0] = t * x * x + c; Rm[1] = t * x * y - s * z; Rm[2] = t * x * z + s * y;
  Rm[3] = t * x * y + s * z; Rm[4] = t * y * y + c; Rm[5] = t * y * z - s * x;
  Rm[6] = t * x * z - s * y; Rm[7] = t * y * z + s * x; Rm[8] = t * z * z + c;
}
static int convex_pair(const smjo_model* m, const smjo_data* d, int g1, int g2, double margin, rawcon* rc) {
  if (!obb_overlap(m, d, g1, g2, margin)) return 0;
  {
    const int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    const double *x1 = d->geom_xpos + 3 * g1, *x2 = d->geom_xpos + 3 * g2;
    if (t1 == G_SPHERE && t2 == G_SPHERE) return sphere_sphere(x1, m->geom_size[3 * g1], x2, m->geom_size[3 * g2], margin, rc);
    if (t1 == G_SPHERE && t2 == G_BOX) return sphere_box(x1, m->geom_size[3 * g1], x2, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, margin, rc);
    if (t1 == G_BOX && t2 == G_SPHERE) {
      if (!sphere_box(x2, m->geom_size[3 * g2], x1, d->geom_xmat + 9 * g1, m->geom_size + 3 * g1, margin, rc)) return 0;
      for (int i = 0; i < 3; i++) rc->normal[i] = -rc->normal[i]; /* the contact keeps the pair's geom order */
      return 1;
    }
    if (t1 == G_SPHERE && t2 == G_CAPSULE) return sphere_capsule(x1, m->geom_size[3 * g1], x2, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, margin, rc);
    if (t1 == G_CAPSULE && t2 == G_SPHERE) {
      if (!sphere_capsule(x2, m->geom_size[3 * g2], x1, d->geom_xmat + 9 * g1, m->geom_size + 3 * g1, margin, rc)) return 0;
      for (int i = 0; i < 3; i++) rc->normal[i] = -rc->normal[i];
      return 1;
    }
    if (t1 == G_CAPSULE && t2 == G_CAPSULE)
      return capsule_capsule(x1, d->geom_xmat + 9 * g1, m->geom_size + 3 * g1, x2, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, margin, rc);
    if (t1 == G_BOX && t2 == G_BOX && m->multiccd)
      return box_box(x1, d->geom_xmat + 9 * g1, m->geom_size + 3 * g1, x2, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, margin,
                     m->max_con_pair < 4 ? 4 : m->max_con_pair, rc);
  }
  mprctx c;
  c.m = m;
  double cen[2][3];
  for (int k = 0; k < 2; k++) {
    int g = k ? g2 : g1;
    c.g[k] = g; c.type[k] = m->geom_type[g]; c.pos[k] = d->geom_xpos + 3 * g; c.mat[k] = d->geom_xmat + 9 * g;
    c.size[k] = m->geom_size + 3 * g; c.verts[k] = m->hull_vert + 3 * (m->geom_hulladr[g] < 0 ? 0 : m->geom_hulladr[g]);
    c.nvert[k] = m->geom_hullnum[g];
    mulmat3vec(cen[k], c.mat[k], m->geom_ccenter + 3 * g);
    for (int i = 0; i < 3; i++) cen[k][i] += c.pos[k][i];
  }
  double depth, dir[3], pos[3];
  if (!mpr_penetration(&c, cen[0], cen[1], &depth, dir, pos)) return 0;
  if (-depth > margin) return 0;
  rc->dist = -depth; memcpy(rc->normal, dir, 24); memcpy(rc->pos, pos, 24);
  int n = 1;
  if (!m->multiccd || dot3(dir, dir) < 0.5 || c.type[0] == G_SPHERE || c.type[1] == G_SPHERE) return n;
  /* multiccd: counter-rotate the two geoms by +-angle about the two tangent axes through the first contact point and query
   * again; keep contacts that are not duplicates.  Order: axis 1 (+, -), axis 2 (+, -). */
  double fr[9] = {dir[0], dir[1], dir[2], 0, 0, 0, 0, 0, 0};
  make_frame(fr);
  const double tol = MULTICCD_RELTOL * fmin(m->geom_rbound[g1], m->geom_rbound[g2]);
  for (int q = 0; q < 4; q++) {
    const double* ax = fr + 3 * (1 + q / 2);
    const double ang = (q & 1) ? -MULTICCD_ANGLE : MULTICCD_ANGLE;
    double Rp[9], Rn[9], pos2[2][3], mat2[2][9], cen2[2][3];
    axis_angle_mat(Rp, ax, ang); axis_angle_mat(Rn, ax, -ang);
    mprctx c2 = c;
    for (int k = 0; k < 2; k++) {
      memcpy(pos2[k], c.pos[k], 24); memcpy(mat2[k], c.mat[k], 72); memcpy(cen2[k], cen[k], 24);
      rotate_pose(pos2[k], mat2[k], rc[0].pos, k ? Rn : Rp);
      double dum[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      rotate_pose(cen2[k], dum, rc[0].pos, k ? Rn : Rp);
      c2.pos[k] = pos2[k]; c2.mat[k] = mat2[k];
    }
    double dp, dr[3], ps[3];
    if (!mpr_penetration(&c2, cen2[0], cen2[1], &dp, dr, ps)) continue;
    if (-dp > margin || dot3(dr, dr) < 0.5) continue;
    int dup = 0;
    for (int k = 0; k < n; k++) {
      const double e[3] = {ps[0] - rc[k].pos[0], ps[1] - rc[k].pos[1], ps[2] - rc[k].pos[2]};
      if (dot3(e, e) < tol * tol) dup = 1;
    }
    if (dup) continue;
    rc[n].dist = -dp; memcpy(rc[n].normal, dir, 24); memcpy(rc[n].pos, ps, 24);   /* the manifold shares the first normal */
    n++;
  }
  return n;
}

/* [MJ] mj_collision: static pair table (built by the model compiler with MuJoCo's filters) ->
 * bounding-sphere rejection -> narrowphase by type pair */
static void collision(const smjo_model* m, smjo_data* d) {
  d->ncon = 0; d->ncon_dropped = 0;
  if (d->override_ncon >= 0) {   /* contacts given by the test (smjo_set_contacts): one-shot */
    for (int i = 0; i < d->override_ncon && d->ncon < MAXCON; i++) {
      const double* v = d->override_con + 9 * i;
      const int g1 = (int)v[7], g2 = (int)v[8];
      int p = -1;
      for (int k = 0; k < m->npair && p < 0; k++)
        if (m->pair_geom1[k] == g1 && m->pair_geom2[k] == g2) p = k;
      if (p < 0) continue;
      contact_t* c = d->contact + d->ncon++;
      c->dist = v[0]; memcpy(c->pos, v + 1, 24); memcpy(c->frame, v + 4, 24);
      c->frame[3] = c->frame[4] = c->frame[5] = 0;
      make_frame(c->frame);
      c->dim = m->pair_condim[p]; c->geom1 = g1; c->geom2 = g2;
      memcpy(c->friction, m->pair_friction + 5 * p, 40); memcpy(c->solref, m->pair_solref + 2 * p, 16);
      memcpy(c->solimp, m->pair_solimp + 5 * p, 40);
      c->includemargin = m->pair_margin[p] - m->pair_gap[p];
      c->efc_address = -1;
    }
    d->override_ncon = -1;
    return;
  }
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = m->pair_margin[p];
    const double *p1 = d->geom_xpos + 3 * g1, *R1 = d->geom_xmat + 9 * g1, *p2 = d->geom_xpos + 3 * g2, *R2 = d->geom_xmat + 9 * g2;
    const double *s2 = m->geom_size + 3 * g2;
    rawcon rc[8];
    int n = 0;
    double c2[3];
    mulmat3vec(c2, R2, m->geom_center + 3 * g2);
    for (int k = 0; k < 3; k++) c2[k] += p2[k];
    if (t1 == G_PLANE) {
      double nrm[3] = {R1[2], R1[5], R1[8]}, dif[3] = {c2[0] - p1[0], c2[1] - p1[1], c2[2] - p1[2]};
      if (dot3(dif, nrm) - m->geom_rbound[g2] > margin) continue;
      if (t2 == G_SPHERE) n = plane_sphere(p1, R1, p2, s2[0], margin, rc);
      else if (t2 == G_CYLINDER) n = plane_cylinder(p1, R1, p2, R2, s2, margin, rc);
      else if (t2 == G_BOX) n = plane_box(p1, R1, p2, R2, s2, margin, rc);
      else if (t2 == G_CAPSULE) {     /* [MJ] mjc_PlaneCapsule: the two end spheres, +axis end first */
        for (int e = 0; e < 2; e++) {
          double sg = e ? -1 : 1, c[3] = {p2[0] + sg * s2[1] * R2[2], p2[1] + sg * s2[1] * R2[5], p2[2] + sg * s2[1] * R2[8]};
          n += plane_sphere(p1, R1, c, s2[0], margin, rc + n);
        }
      } else if (t2 == G_ELLIPSOID) { /* [MJ] mjc_PlaneEllipsoid: the surface point whose outward normal is -n */
        double nl[3], sv[3], loc[3], wv[3];
        mulmat3Tvec(nl, R2, nrm);
        for (int k = 0; k < 3; k++) sv[k] = -nl[k] * s2[k];
        double len = norm3(sv);
        for (int k = 0; k < 3; k++) loc[k] = len > MINVAL ? s2[k] * sv[k] / len : 0;
        mulmat3vec(wv, R2, loc);
        double df[3] = {p2[0] + wv[0] - p1[0], p2[1] + wv[1] - p1[1], p2[2] + wv[2] - p1[2]}, dist = dot3(df, nrm);
        if (dist <= margin) {
          rc[0].dist = dist; memcpy(rc[0].normal, nrm, 24);
          for (int k = 0; k < 3; k++) rc[0].pos[k] = p2[k] + wv[k] - nrm[k] * dist * 0.5;
          n = 1;
        }
      } else if (t2 == G_MESH)
        n = plane_hull(p1, R1, p2, R2, m->hull_vert + 3 * m->geom_hulladr[g2], m->geom_hullnum[g2], margin, m->max_con_pair, rc);
      else continue;
    } else {
      if (!m->convex_pairs) continue;
      int slot = -1;
      const int tag = m->pair_tag[p];
      n = -1;
      if (m->manifold_keep && tag && t1 != G_SPHERE && t2 != G_SPHERE && !(t1 == G_CAPSULE && t2 == G_CAPSULE)) {
        /* the kernels' mc_entry(): pairs with the static world in the upper half of the slots; the look-up comes after the broadphase */
        if (!obb_overlap(m, d, g1, g2, margin)) continue;
        slot = (int)((((unsigned)tag >> 30) & 1u) << 5 | (((unsigned)tag * 2654435761u) >> 27));
        const int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
        double cur[14];
        memcpy(cur, d->xpos + 3 * b1, 24); memcpy(cur + 3, d->xquat + 4 * b1, 32);
        memcpy(cur + 7, d->xpos + 3 * b2, 24); memcpy(cur + 10, d->xquat + 4 * b2, 32);
        int ok = d->mc_tag[slot] == tag;
        for (int k = 0; k < 14 && ok; k++) ok = fabs(cur[k] - d->mc_pose[slot][k]) <= m->manifold_keep_eps;
        if (ok) {   /* smj_mc_motion / smj_mc_carry: u_b(p) = dx_b + dth_b x (p - x_b), dth = 2 vec(q1 conj(q0)) */
          double dx[2][3], dt[2][3];
          for (int k = 0; k < 2; k++) {
            const double *q0 = d->mc_pose[slot] + 7 * k + 3, *q1 = cur + 7 * k + 3;
            for (int i = 0; i < 3; i++) dx[k][i] = cur[7 * k + i] - d->mc_pose[slot][7 * k + i];
            const double sw = q1[0] * q0[0] + q1[1] * q0[1] + q1[2] * q0[2] + q1[3] * q0[3], sg = sw < 0 ? -2.0 : 2.0;
            dt[k][0] = sg * (q0[0] * q1[1] - q1[0] * q0[1] - (q1[2] * q0[3] - q1[3] * q0[2]));
            dt[k][1] = sg * (q0[0] * q1[2] - q1[0] * q0[2] - (q1[3] * q0[1] - q1[1] * q0[3]));
            dt[k][2] = sg * (q0[0] * q1[3] - q1[0] * q0[3] - (q1[1] * q0[2] - q1[2] * q0[1]));
          }
          n = d->mc_n[slot];
          for (int i = 0; i < n; i++) {
            const double* pp = d->mc_con[slot][i] + 1;
            double u[2][3];
            for (int k = 0; k < 2; k++) {
              const double* x0 = d->mc_pose[slot] + 7 * k;
              const double r[3] = {pp[0] - x0[0], pp[1] - x0[1], pp[2] - x0[2]};
              u[k][0] = dx[k][0] + dt[k][1] * r[2] - dt[k][2] * r[1];
              u[k][1] = dx[k][1] + dt[k][2] * r[0] - dt[k][0] * r[2];
              u[k][2] = dx[k][2] + dt[k][0] * r[1] - dt[k][1] * r[0];
            }
            const double* nr = d->mc_nrm[slot];
            rc[i].dist = d->mc_con[slot][i][0] + nr[0] * (u[1][0] - u[0][0]) + nr[1] * (u[1][1] - u[0][1]) + nr[2] * (u[1][2] - u[0][2]);
            for (int k = 0; k < 3; k++) { rc[i].pos[k] = pp[k] + 0.5 * (u[0][k] + u[1][k]); rc[i].normal[k] = nr[k]; }
          }
          d->mc_hits++;
        }
      }
      if (n < 0) {
        n = convex_pair(m, d, g1, g2, margin, rc);
        if (n && dot3(rc[0].normal, rc[0].normal) < 0.5) continue; /* degenerate touching contact without a direction */
        if (slot >= 0 && n >= 1 && n <= 5 && d->ncon + n <= MAXCON) {   /* keep what the narrowphase found, with the poses it was found at */
          const int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
          d->mc_tag[slot] = tag; d->mc_n[slot] = n;
          memcpy(d->mc_pose[slot], d->xpos + 3 * b1, 24); memcpy(d->mc_pose[slot] + 3, d->xquat + 4 * b1, 32);
          memcpy(d->mc_pose[slot] + 7, d->xpos + 3 * b2, 24); memcpy(d->mc_pose[slot] + 10, d->xquat + 4 * b2, 32);
          memcpy(d->mc_nrm[slot], rc[0].normal, 24);
          for (int i = 0; i < n; i++) { d->mc_con[slot][i][0] = rc[i].dist; memcpy(d->mc_con[slot][i] + 1, rc[i].pos, 24); }
        }
      }
    }
    for (int i = 0; i < n; i++) {
      if (d->ncon >= MAXCON) { d->ncon_dropped++; continue; }
      contact_t* c = d->contact + d->ncon++;
      c->dist = rc[i].dist; memcpy(c->pos, rc[i].pos, 24); memcpy(c->frame, rc[i].normal, 24);
      c->frame[3] = c->frame[4] = c->frame[5] = 0;
      make_frame(c->frame);
      c->dim = m->pair_condim[p]; c->geom1 = g1; c->geom2 = g2;
      memcpy(c->friction, m->pair_friction + 5 * p, 40); memcpy(c->solref, m->pair_solref + 2 * p, 16);
      memcpy(c->solimp, m->pair_solimp + 5 * p, 40);
      c->includemargin = m->pair_margin[p] - m->pair_gap[p];
      c->efc_address = -1;
    }
  }
}

/* ------------------------------------------------------------------ B.4 constraints */
static int add_row(smjo_data* d, int nv, int type, int id, double pos, double margin, double floss, double diag) {
  int i = d->nefc;
  if (i >= MAXEFC) return -1;
  memset(d->efc_J + (size_t)i * nv, 0, sizeof(double) * nv);
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin;
  d->efc_frictionloss[i] = floss; d->efc_diagApprox[i] = diag;
  d->nefc++;
  return i;
}

/* [MJ] getimpedance */
static void get_impedance(const double* solimp, double pos, double margin, double* imp) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(MAXIMP, fmax(MINIMP, dmin)); dmax = fmin(MAXIMP, fmax(MINIMP, dmax));
  width = fmax(MINVAL, width); mid = fmin(MAXIMP, fmax(MINIMP, mid)); power = fmax(1, power);
  if (dmin == dmax) { *imp = 0.5 * (dmin + dmax); return; }
  double x = fabs(pos - margin) / width, y;
  if (x >= 1) { *imp = dmax; return; }
  if (x <= 0) { *imp = dmin; return; }
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  *imp = dmin + y * (dmax - dmin);
}

/* [MJ] mj_makeConstraint + mj_makeImpedance + mj_projectConstraint */
static void make_constraint(const smjo_model* m, smjo_data* d) {
  int nv = m->nv;
  d->nefc = 0;
  /* equality (joint) */
  for (int e = 0; e < m->neq; e++) {
    if (!m->eq_active[e]) continue;
    int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
    const double* a = m->eq_data + 5 * e;
    double p1 = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]], pos, deriv = 0;
    double diag = m->dof_invweight0[m->jnt_dofadr[j1]];
    if (j2 >= 0) {
      double dif = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]];
      pos = p1 - (a[0] + dif * (a[1] + dif * (a[2] + dif * (a[3] + dif * a[4]))));
      deriv = a[1] + dif * (2 * a[2] + dif * (3 * a[3] + dif * 4 * a[4]));
      diag += m->dof_invweight0[m->jnt_dofadr[j2]];
    } else pos = p1 - a[0];
    int i = add_row(d, nv, C_EQUALITY, e, pos, 0, 0, diag);
    d->efc_J[(size_t)i * nv + m->jnt_dofadr[j1]] = 1;
    if (j2 >= 0) d->efc_J[(size_t)i * nv + m->jnt_dofadr[j2]] = -deriv;
  }
  d->ne = d->nefc;
  /* dof friction loss */
  for (int k = 0; k < nv; k++)
    if (m->dof_frictionloss[k] > 0) {
      int i = add_row(d, nv, C_FRICTION_DOF, k, 0, 0, m->dof_frictionloss[k], m->dof_invweight0[k]);
      d->efc_J[(size_t)i * nv + k] = 1;
    }
  d->nf = d->nefc - d->ne;
  /* joint limits (slide / hinge) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == JNT_FREE) continue;
    double q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - q);
      if (dist < margin) {
        int i = add_row(d, nv, C_LIMIT_JOINT, j, dist, margin, 0, m->dof_invweight0[m->jnt_dofadr[j]]);
        d->efc_J[(size_t)i * nv + m->jnt_dofadr[j]] = -side;
      }
    }
  }
  /* contacts */
  double* jp1 = d->scratch; double* jr1 = jp1 + 3 * nv; double* jp2 = jr1 + 3 * nv; double* jr2 = jp2 + 3 * nv;
  for (int c = 0; c < d->ncon; c++) {
    contact_t* con = d->contact + c;
    con->efc_address = -1;
    if (con->dist >= con->includemargin) continue;
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2], dim = con->dim;
    if (d->nefc + dim > MAXEFC) { d->ncon_dropped++; continue; }
    jac_point(m, d, jp1, jr1, con->pos, b1);
    jac_point(m, d, jp2, jr2, con->pos, b2);
    double tran = m->geom_invweight0[2 * con->geom1] + m->geom_invweight0[2 * con->geom2];
    double rot = m->geom_invweight0[2 * con->geom1 + 1] + m->geom_invweight0[2 * con->geom2 + 1];
    con->efc_address = d->nefc;
    for (int r = 0; r < dim; r++) {
      int type = dim == 1 ? C_CONTACT_FRICTIONLESS : C_CONTACT_ELLIPTIC;
      int i = add_row(d, nv, type, c, con->dist, con->includemargin, 0, r < 3 ? tran : rot);
      const double* ax = con->frame + 3 * (r < 3 ? r : r - 3);
      const double *ja = r < 3 ? jp1 : jr1, *jb = r < 3 ? jp2 : jr2;
      for (int k = 0; k < nv; k++) {
        double v = 0;
        for (int x = 0; x < 3; x++) v += ax[x] * (jb[x * nv + k] - ja[x * nv + k]);
        d->efc_J[(size_t)i * nv + k] = v;
      }
    }
  }
  /* impedance, R, D, KBIP */
  for (int i = 0; i < d->nefc; i++) {
    const double *solref, *solimp;
    int id = d->efc_id[i], t = d->efc_type[i];
    if (t == C_EQUALITY) { solref = m->eq_solref + 2 * id; solimp = m->eq_solimp + 5 * id; }
    else if (t == C_FRICTION_DOF) { solref = m->dof_solref + 2 * id; solimp = m->dof_solimp + 5 * id; }
    else if (t == C_LIMIT_JOINT) { solref = m->jnt_solref + 2 * id; solimp = m->jnt_solimp + 5 * id; }
    else { solref = d->contact[id].solref; solimp = d->contact[id].solimp; }
    double imp;
    get_impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp);
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1])), K, B;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * m->timestep), dr = solref[1];
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -solref[0] / fmax(MINVAL, dmax * dmax); B = -solref[1] / fmax(MINVAL, dmax); }
    int friction_row = (t == C_FRICTION_DOF) || (t == C_CONTACT_ELLIPTIC && i != d->contact[id].efc_address);
    if (friction_row) K = 0;
    d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
  }
  /* elliptic cones: friction-row regularisation from the normal row, impratio ([MJ] mj_makeImpedance tail) */
  for (int c = 0; c < d->ncon; c++) {
    contact_t* con = d->contact + c;
    int i = con->efc_address, dim = con->dim;
    if (i < 0 || dim < 3) { con->mu = 0; continue; }
    d->efc_R[i + 1] = d->efc_R[i] / fmax(MINVAL, m->impratio);
    con->mu = con->friction[0] * sqrt(d->efc_R[i + 1] / d->efc_R[i]);
    for (int j = 1; j < dim - 1; j++)
      d->efc_R[i + 1 + j] = d->efc_R[i + 1] * con->friction[0] * con->friction[0] / (con->friction[j] * con->friction[j]);
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
  /* AR = J M^-1 J' + diag(R) */
  int ne = d->nefc;
  double* B = d->scratch; /* nefc x nv : rows M^-1 J_i' */
  for (int i = 0; i < ne; i++) chol_solve(B + (size_t)i * nv, d->qL, d->efc_J + (size_t)i * nv, nv);
  for (int i = 0; i < ne; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * B[(size_t)j * nv + k];
      d->efc_AR[(size_t)i * ne + j] = d->efc_AR[(size_t)j * ne + i] = s;
    }
  for (int i = 0; i < ne; i++) d->efc_AR[(size_t)i * ne + i] += d->efc_R[i];
}

/* ------------------------------------------------------------------ B.5 velocity stage */
/* [MJ] mj_comVel */
static void com_vel(const smjo_model* m, smjo_data* d) {
  memset(d->cvel, 0, 48);
  for (int b = 1; b < m->nbody; b++) {
    double cv[6];
    memcpy(cv, d->cvel + 6 * m->body_parentid[b], 48);
    int da = m->body_dofadr[b];
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int a = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        memset(d->cdof_dot + 6 * a, 0, sizeof(double) * 18);
        for (int k = 0; k < 3; k++)
          for (int x = 0; x < 6; x++) cv[x] += d->cdof[6 * (a + k) + x] * d->qvel[a + k];
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (a + k), cv, d->cdof + 6 * (a + k));
        for (int k = 3; k < 6; k++)
          for (int x = 0; x < 6; x++) cv[x] += d->cdof[6 * (a + k) + x] * d->qvel[a + k];
      } else {
        cross_motion(d->cdof_dot + 6 * a, cv, d->cdof + 6 * a);
        for (int x = 0; x < 6; x++) cv[x] += d->cdof[6 * a + x] * d->qvel[a];
      }
    }
    (void)da;
    memcpy(d->cvel + 6 * b, cv, 48);
  }
}

/* [MJ] mj_passive: springs, dampers, gravity compensation */
static void passive(const smjo_model* m, smjo_data* d) {
  int nv = m->nv;
  for (int k = 0; k < nv; k++) d->qfrc_passive[k] = -m->dof_damping[k] * d->qvel[k];
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] == JNT_FREE || m->jnt_stiffness[j] == 0) continue;
    int qa = m->jnt_qposadr[j];
    d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
  }
  double* jp = d->scratch;
  for (int b = 1; b < m->nbody; b++) {
    if (m->body_gcmass[b] == 0) continue;
    double f[3], pt[3];
    for (int k = 0; k < 3; k++) f[k] = -m->gravity[k] * m->body_gcmass[b];
    mulmat3vec(pt, d->xmat + 9 * b, m->body_gcipos + 3 * b);
    for (int k = 0; k < 3; k++) pt[k] += d->xpos[3 * b + k];
    jac_point(m, d, jp, NULL, pt, b);
    for (int k = 0; k < nv; k++) d->qfrc_passive[k] += jp[k] * f[0] + jp[nv + k] * f[1] + jp[2 * nv + k] * f[2];
  }
}

/* [MJ] mj_rne(flg_acc=0): Coriolis, centrifugal, gravity */
static void rne_bias(const smjo_model* m, smjo_data* d) {
  int nb = m->nbody, nv = m->nv;
  double* cacc = d->cacc; double* cfrc = d->cfrc;
  memset(cacc, 0, 48);
  for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
  for (int b = 1; b < nb; b++) {
    double a[6], t[6], t2[6];
    memcpy(a, cacc + 6 * m->body_parentid[b], 48);
    for (int k = m->body_dofadr[b]; k < m->body_dofadr[b] + m->body_dofnum[b]; k++)
      for (int x = 0; x < 6; x++) a[x] += d->cdof_dot[6 * k + x] * d->qvel[k];
    memcpy(cacc + 6 * b, a, 48);
    mul_inert_vec(t, d->cinert + 10 * b, a);
    mul_inert_vec(t2, d->cinert + 10 * b, d->cvel + 6 * b);
    double cf[6];
    cross_force(cf, d->cvel + 6 * b, t2);
    for (int x = 0; x < 6; x++) cfrc[6 * b + x] = t[x] + cf[x];
  }
  memset(cfrc, 0, 48);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int x = 0; x < 6; x++) cfrc[6 * p + x] += cfrc[6 * b + x];
  }
  for (int k = 0; k < nv; k++) {
    double s = 0;
    for (int x = 0; x < 6; x++) s += d->cdof[6 * k + x] * cfrc[6 * m->dof_bodyid[k] + x];
    d->qfrc_bias[k] = s;
  }
}

/* ------------------------------------------------------------------ B.6 actuation */
static void fwd_actuation(const smjo_model* m, smjo_data* d) {
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(double) * nv);
  for (int a = 0; a < m->nu; a++) {
    double vel = 0, ctrl = d->ctrl[a];
    for (int k = 0; k < nv; k++) vel += d->actuator_moment[a * nv + k] * d->qvel[k];
    d->actuator_velocity[a] = vel;
    if (m->actuator_ctrllimited[a]) ctrl = fmin(m->actuator_ctrlrange[2 * a + 1], fmax(m->actuator_ctrlrange[2 * a], ctrl));
    const double *g = m->actuator_gainprm + 3 * a, *bp = m->actuator_biasprm + 3 * a;
    double f = g[0] * ctrl;
    if (m->actuator_biastype[a] == 1) f += bp[0] + bp[1] * d->actuator_length[a] + bp[2] * vel;
    if (m->actuator_forcelimited[a]) f = fmin(m->actuator_forcerange[2 * a + 1], fmax(m->actuator_forcerange[2 * a], f));
    d->actuator_force[a] = f;
    for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += d->actuator_moment[a * nv + k] * f;
  }
}

/* ------------------------------------------------------------------ B.7 solver (PGS, dual) */
/* [MJ] mju_QCQP: Newton on the multiplier from la = 0, at most 20 iterates.  cap (model option "qcqp_cap", default 20 = MuJoCo)
 * lifts that cap for one purpose: tests of the HIP path's default QCQP, which finds the converged root (smj_step_impl.h qcqp). */
static int qcqp(double* res, const double* Ain, const double* bin, const double* dd, double r, int n, int cap) {
  double A[25], b[5], Ala[25], tmp[5], la = 0;
  for (int i = 0; i < n; i++) {
    b[i] = bin[i] * dd[i];
    for (int j = 0; j < n; j++) A[i * n + j] = Ain[i * n + j] * dd[i] * dd[j];
  }
  for (int it = 0; it < cap; it++) {
    memcpy(Ala, A, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) Ala[i * n + i] += la;
    if (n == 2) { /* [MJ] mju_QCQP2: determinant test */
      double det = Ala[0] * Ala[3] - Ala[1] * Ala[1];
      if (det < 1e-10) { memset(res, 0, sizeof(double) * n); return 0; }
      res[0] = -(Ala[3] * b[0] - Ala[1] * b[1]) / det; res[1] = -(-Ala[1] * b[0] + Ala[0] * b[1]) / det;
      double P[4] = {Ala[3] / det, -Ala[1] / det, -Ala[1] / det, Ala[0] / det};
      tmp[0] = P[0] * res[0] + P[1] * res[1]; tmp[1] = P[2] * res[0] + P[3] * res[1];
    } else {
      if (chol_factor(Ala, n, 1e-10) < n) { memset(res, 0, sizeof(double) * n); return 0; }
      chol_solve(res, Ala, b, n);
      for (int i = 0; i < n; i++) res[i] = -res[i];
      chol_solve(tmp, Ala, res, n);
    }
    double val = -r * r, deriv = 0;
    for (int i = 0; i < n; i++) { val += res[i] * res[i]; deriv -= 2 * res[i] * tmp[i]; }
    if (val < 1e-10) break;
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  for (int i = 0; i < n; i++) res[i] *= dd[i];
  return la != 0;
}

/* [MJ] mj_constraintUpdate (forces from primal residual jar), used only for the warm start */
static void constraint_update_force(const smjo_model* m, smjo_data* d, const double* jar, double* force) {
  for (int i = 0; i < d->nefc; i++) {
    int t = d->efc_type[i];
    double D = d->efc_D[i], R = d->efc_R[i];
    if (t == C_EQUALITY) force[i] = -D * jar[i];
    else if (t == C_FRICTION_DOF) {
      double fl = d->efc_frictionloss[i];
      if (jar[i] <= -R * fl) force[i] = fl;
      else if (jar[i] >= R * fl) force[i] = -fl;
      else force[i] = -D * jar[i];
    } else if (t == C_LIMIT_JOINT || t == C_CONTACT_FRICTIONLESS) force[i] = jar[i] < 0 ? -D * jar[i] : 0;
    else {
      contact_t* con = d->contact + d->efc_id[i];
      int dim = con->dim;
      double mu = con->mu, U[6], N, T = 0;
      U[0] = jar[i] * mu;
      for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * con->friction[j - 1]; T += U[j] * U[j]; }
      N = U[0]; T = sqrt(T);
      if ((T <= 0 && N >= 0) || (T > 0 && N >= mu * T)) { for (int j = 0; j < dim; j++) force[i + j] = 0; }
      else if ((T <= 0 && N < 0) || (T > 0 && mu * N + T <= 0)) { for (int j = 0; j < dim; j++) force[i + j] = -d->efc_D[i + j] * jar[i + j]; }
      else {
        double Dm = d->efc_D[i] / fmax(mu * mu * (1 + mu * mu), MINVAL), NmT = N - mu * T;
        force[i] = -Dm * NmT * mu;
        for (int j = 1; j < dim; j++) force[i + j] = -force[i] / T * U[j] * con->friction[j - 1];
      }
      i += dim - 1;
    }
  }
}

/* identity of a constraint row across steps: (type, equality / dof / joint id) or, for a contact row, (geom pair, ordinal of the contact
 * within its pair's manifold, row within the contact) */
static long long row_key(const smjo_data* d, int i) {
  int t = d->efc_type[i];
  if (t == C_CONTACT_ELLIPTIC || t == C_CONTACT_FRICTIONLESS) {
    int c = d->efc_id[i], ord = 0;
    const contact_t* con = d->contact + c;
    for (int k = c - 1; k >= 0 && d->contact[k].geom1 == con->geom1 && d->contact[k].geom2 == con->geom2; k--) ord++;
    return ((long long)7 << 56) | ((long long)con->geom1 << 36) | ((long long)con->geom2 << 16) | (ord << 8) | (i - con->efc_address);
  }
  return ((long long)t << 56) | (unsigned)d->efc_id[i];
}
static double dual_cost(const smjo_data* d, const double* f, int ne) {
  double cost = 0;
  for (int i = 0; i < ne; i++) {
    double s = 0;
    for (int j = 0; j < ne; j++) s += d->efc_AR[(size_t)i * ne + j] * f[j];
    cost += f[i] * (0.5 * s + d->efc_b[i]);
  }
  return cost;
}

/* [MJ] mj_fwdConstraint: warm start + mj_solPGS + dual->primal */
static void fwd_constraint(const smjo_model* m, smjo_data* d) {
  int nv = m->nv, ne = d->nefc;
  if (ne == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    d->solver_niter = 0;
    return;
  }
  double* f = d->efc_force;
  const double* AR = d->efc_AR;
  /* b = J qacc_smooth - aref */
  for (int i = 0; i < ne; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_smooth[k];
    d->efc_b[i] = s - d->efc_aref[i];
  }
  /* warm start ([MJ] mj_warmstart, PGS branch) */
  memset(f, 0, sizeof(double) * ne);
  if (m->warmstart) {
    double* jar = d->scratch;
    for (int i = 0; i < ne; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k];
      jar[i] = s - d->efc_aref[i];
    }
    constraint_update_force(m, d, jar, f);
    double cost = 0;
    for (int i = 0; i < ne; i++) {
      double s = 0;
      for (int j = 0; j < ne; j++) s += AR[(size_t)i * ne + j] * f[j];
      cost += f[i] * (0.5 * s + d->efc_b[i]);
    }
    if (cost > 0) memset(f, 0, sizeof(double) * ne);
  }
  if (m->pgs_dual_warmstart && d->prev_n > 0) {
    /* NOT MuJoCo: a second candidate -- the forces the rows had at the end of the previous step's solve, matched by row identity and
     * projected onto this step's bounds / cones; taken when its dual cost is below the start MuJoCo's rule gives.  The dual problem is
     * strictly convex (R > 0): where the sweeps converge they converge to the same forces from either start. */
    double* g = d->scratch + ne;
    for (int i = 0; i < ne; i++) {
      long long key = row_key(d, i);
      g[i] = 0;
      for (int j = 0; j < d->prev_n; j++) if (d->prev_key[j] == key) { g[i] = d->prev_force[j]; break; }
    }
    for (int i = 0; i < ne;) {
      int t = d->efc_type[i];
      if (t == C_FRICTION_DOF) { double fl = d->efc_frictionloss[i]; g[i] = fmin(fl, fmax(-fl, g[i])); i++; }
      else if (t == C_LIMIT_JOINT || t == C_CONTACT_FRICTIONLESS) { g[i] = fmax(0, g[i]); i++; }
      else if (t == C_CONTACT_ELLIPTIC) {
        contact_t* con = d->contact + d->efc_id[i];
        int dim = con->dim;
        if (g[i] < MINVAL) { for (int j = 0; j < dim; j++) g[i + j] = 0; }
        else {
          double s2 = 0;
          for (int j = 1; j < dim; j++) s2 += g[i + j] * g[i + j] / (con->friction[j - 1] * con->friction[j - 1]);
          if (s2 > g[i] * g[i]) { double sc = g[i] / sqrt(s2); for (int j = 1; j < dim; j++) g[i + j] *= sc; }
        }
        i += dim;
      } else i++;
    }
    double c0 = dual_cost(d, f, ne), c1 = dual_cost(d, g, ne);
    if (c1 < c0) memcpy(f, g, sizeof(double) * ne);
  }
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  for (; iter < m->iterations; iter++) {
    double improvement = 0;
    for (int i = 0; i < ne;) {
      int t = d->efc_type[i], dim = 1;
      contact_t* con = NULL;
      if (t == C_CONTACT_ELLIPTIC) { con = d->contact + d->efc_id[i]; dim = con->dim; }
      double res[6], old[6], Athis[36];
      for (int r = 0; r < dim; r++) {
        double s = d->efc_b[i + r];
        for (int j = 0; j < ne; j++) s += AR[(size_t)(i + r) * ne + j] * f[j];
        res[r] = s; old[r] = f[i + r];
        for (int c = 0; c < dim; c++) Athis[r * dim + c] = AR[(size_t)(i + r) * ne + i + c];
      }
      if (dim == 1) {
        f[i] -= res[0] / AR[(size_t)i * ne + i];
        if (t == C_FRICTION_DOF) {
          double fl = d->efc_frictionloss[i];
          f[i] = fmin(fl, fmax(-fl, f[i]));
        } else if (t != C_EQUALITY) f[i] = fmax(0, f[i]);
      } else {
        const double* mu = con->friction;
        if (f[i] < MINVAL) { /* normal update */
          f[i] -= res[0] / Athis[0];
          if (f[i] < 0) f[i] = 0;
          for (int j = 1; j < dim; j++) f[i + j] = 0;
        } else { /* ray update */
          double v[6], v1[6], denom = 0, num = 0;
          for (int r = 0; r < dim; r++) v[r] = f[i + r];
          for (int r = 0; r < dim; r++) {
            v1[r] = 0;
            for (int c = 0; c < dim; c++) v1[r] += Athis[r * dim + c] * v[c];
            denom += v[r] * v1[r]; num += v[r] * res[r];
          }
          if (denom >= MINVAL) {
            double x = -num / denom;
            if (f[i] + x * v[0] < 0) x = -f[i] / v[0];
            for (int r = 0; r < dim; r++) f[i + r] += x * v[r];
          }
        }
        /* friction update with the normal fixed */
        double Ac[25], bc[5], v[5];
        int n = dim - 1;
        for (int j = 0; j < n; j++) {
          for (int c = 0; c < n; c++) Ac[j * n + c] = Athis[(j + 1) * dim + c + 1];
          bc[j] = res[j + 1];
          for (int c = 0; c < dim; c++) bc[j] -= Athis[(j + 1) * dim + c] * old[c];
          bc[j] += Athis[(j + 1) * dim] * f[i];
        }
        if (f[i] < MINVAL) { for (int j = 1; j < dim; j++) f[i + j] = 0; }
        else {
          int active = qcqp(v, Ac, bc, mu, f[i], n, m->qcqp_cap);
          if (active) {
            double s = 0;
            for (int j = 0; j < n; j++) s += v[j] * v[j] / (mu[j] * mu[j]);
            s = sqrt(f[i] * f[i] / fmax(MINVAL, s));
            for (int j = 0; j < n; j++) v[j] *= s;
          }
          for (int j = 0; j < n; j++) f[i + 1 + j] = v[j];
        }
      }
      /* cost change ([MJ] costChange) */
      double change = 0;
      for (int r = 0; r < dim; r++) {
        double dr = f[i + r] - old[r], s = 0;
        for (int c = 0; c < dim; c++) s += Athis[r * dim + c] * (f[i + c] - old[c]);
        change += dr * (0.5 * s + res[r]);
      }
      if (change > 1e-10) { for (int r = 0; r < dim; r++) f[i + r] = old[r]; change = 0; }
      improvement -= change;
      i += dim;
    }
    improvement *= scale;
    if (!m->pgs_fixed_iter && improvement < m->tolerance) { iter++; break; }
  }
  d->solver_niter = iter;
  if (m->pgs_dual_warmstart) {
    d->prev_key = (long long*)realloc(d->prev_key, sizeof(long long) * (ne + 1));
    d->prev_force = (double*)realloc(d->prev_force, sizeof(double) * (ne + 1));
    for (int i = 0; i < ne; i++) { d->prev_key[i] = row_key(d, i); d->prev_force[i] = f[i]; }
    d->prev_n = ne;
  }
  for (int k = 0; k < nv; k++) {
    double s = 0;
    for (int i = 0; i < ne; i++) s += d->efc_J[(size_t)i * nv + k] * f[i];
    d->qfrc_constraint[k] = s;
  }
  chol_solve(d->qacc, d->qL, d->qfrc_constraint, nv);
  for (int k = 0; k < nv; k++) d->qacc[k] += d->qacc_smooth[k];
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}


/* ------------------------------------------------------------------ B.7' solver (Newton, primal)
 * [MJ] mj_solNewton: minimise  0.5 (a-a_s)' M (a-a_s) + s(J a - aref)  over qacc with exact Newton steps
 * (H = M + J' D_active J + cone Hessians) and an exact line search.  This is the solver the reference model
 * actually runs (stretch.xml:7 names no solver -> MuJoCo default Newton).  The line search is a safeguarded
 * 1-D Newton root finder on the directional derivative (same minimiser as MuJoCo's CGsearch to its tolerance). */
typedef struct {
  const smjo_model* m;
  smjo_data* d;
  double *Jaref, *Jv, *quad; /* per row: quad[3] */
  double *cq;                /* per contact: u0 v0 uu uv vv Dm mu */
  double quadGauss[3];
} nctx;

/* forces, cost and (optionally) per-contact cone Hessians at residual jar.  coneH: ncon x 36 or NULL */
static double newton_update(const smjo_model* m, smjo_data* d, const double* jar, double* force, int* state, double* coneH) { FL(8LL * d->nefc + 60LL * d->ncon);
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    int t = d->efc_type[i];
    double D = d->efc_D[i], R = d->efc_R[i];
    if (t == C_EQUALITY) { force[i] = -D * jar[i]; state[i] = 1; cost += 0.5 * D * jar[i] * jar[i]; }
    else if (t == C_FRICTION_DOF) {
      double fl = d->efc_frictionloss[i];
      if (jar[i] <= -R * fl) { force[i] = fl; state[i] = 2; cost += -fl * (0.5 * R * fl + jar[i]); }
      else if (jar[i] >= R * fl) { force[i] = -fl; state[i] = 3; cost += -fl * (0.5 * R * fl - jar[i]); }
      else { force[i] = -D * jar[i]; state[i] = 1; cost += 0.5 * D * jar[i] * jar[i]; }
    } else if (t == C_LIMIT_JOINT || t == C_CONTACT_FRICTIONLESS) {
      if (jar[i] >= 0) { force[i] = 0; state[i] = 0; }
      else { force[i] = -D * jar[i]; state[i] = 1; cost += 0.5 * D * jar[i] * jar[i]; }
    } else {
      int c = d->efc_id[i];
      contact_t* con = d->contact + c;
      int dim = con->dim;
      double mu = con->mu, U[6], N, T = 0;
      U[0] = jar[i] * mu;
      for (int j = 1; j < dim; j++) { U[j] = jar[i + j] * con->friction[j - 1]; T += U[j] * U[j]; }
      N = U[0]; T = sqrt(T);
      if (N >= mu * T || (T <= 0 && N >= 0)) { for (int j = 0; j < dim; j++) { force[i + j] = 0; state[i + j] = 0; } }
      else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        for (int j = 0; j < dim; j++) {
          force[i + j] = -d->efc_D[i + j] * jar[i + j]; state[i + j] = 1;
          cost += 0.5 * d->efc_D[i + j] * jar[i + j] * jar[i + j];
        }
      } else {
        double Dm = d->efc_D[i] / fmax(mu * mu * (1 + mu * mu), MINVAL), NT = N - mu * T;
        cost += 0.5 * Dm * NT * NT;
        force[i] = -Dm * NT * mu;
        for (int j = 1; j < dim; j++) force[i + j] = -force[i] / T * U[j] * con->friction[j - 1];
        for (int j = 0; j < dim; j++) state[i + j] = 4;
        if (coneH) { /* Hessian wrt jar: S' Hu S, S = diag(mu, friction), Hu from s(U) = 0.5 Dm (N - mu T)^2 */
          double* H = coneH + 36 * c, S[6];
          S[0] = mu;
          for (int j = 1; j < dim; j++) S[j] = con->friction[j - 1];
          H[0] = 1;
          for (int j = 1; j < dim; j++) H[j] = H[j * dim] = -mu * U[j] / T;
          for (int j = 1; j < dim; j++)
            for (int k = 1; k < dim; k++)
              H[j * dim + k] = mu * N / (T * T * T) * U[j] * U[k] - (j == k ? mu * NT / T : 0);
          for (int j = 0; j < dim; j++)
            for (int k = 0; k < dim; k++) H[j * dim + k] *= Dm * S[j] * S[k];
        }
      }
      i += dim - 1;
    }
  }
  return cost;
}

/* cost and first/second derivative along the search line at step alpha  ([MJ] CGeval) */
static double ls_eval(const nctx* c, double a, double* d1, double* d2) { FL(10LL * c->d->nefc + 40LL * c->d->ncon + 12);
  const smjo_data* d = c->d;
  double q0 = c->quadGauss[0], q1 = c->quadGauss[1], q2 = c->quadGauss[2], cost = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < d->nefc; i++) {
    int t = d->efc_type[i];
    const double* q = c->quad + 3 * i;
    double x = c->Jaref[i] + a * c->Jv[i];
    if (t == C_EQUALITY) { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
    else if (t == C_FRICTION_DOF) {
      double fl = d->efc_frictionloss[i], rf = d->efc_R[i] * fl;
      if (x <= -rf) { q0 += fl * (-0.5 * rf - c->Jaref[i]); q1 += -fl * c->Jv[i]; }
      else if (x >= rf) { q0 += fl * (-0.5 * rf + c->Jaref[i]); q1 += fl * c->Jv[i]; }
      else { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
    } else if (t == C_LIMIT_JOINT || t == C_CONTACT_FRICTIONLESS) {
      if (x < 0) { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
    } else {
      int ci = d->efc_id[i], dim = d->contact[ci].dim;
      const double* k = c->cq + 7 * ci;
      double mu = k[6], Dm = k[5], N = k[0] + a * k[1], Tsq = k[2] + a * (2 * k[3] + a * k[4]);
      if (Tsq <= 0) { if (N < 0) { q0 += q[0]; q1 += q[1]; q2 += q[2]; } }
      else {
        double T = sqrt(Tsq);
        if (N >= mu * T) { /* top: nothing */ }
        else if (mu * N + T <= 0) { q0 += q[0]; q1 += q[1]; q2 += q[2]; }
        else {
          double N1 = k[1], T1 = (k[3] + a * k[4]) / T, T2 = k[4] / T - (k[3] + a * k[4]) * (k[3] + a * k[4]) / (T * T * T);
          double NT = N - mu * T, NT1 = N1 - mu * T1;
          cost += 0.5 * Dm * NT * NT; e1 += Dm * NT * NT1; e2 += Dm * (NT1 * NT1 + NT * (-mu * T2));
        }
      }
      i += dim - 1;
    }
  }
  *d1 = 2 * a * q2 + q1 + e1;
  *d2 = 2 * q2 + e2;
  return a * a * q2 + a * q1 + q0 + cost;
}

static double ls_search(const smjo_model* m, const nctx* c, double snorm, int* nls) {
  double gtol = m->tolerance * m->ls_tolerance * snorm * m->meaninertia * (m->nv > 1 ? m->nv : 1);
  double d1, d2, lo = 0, hi = -1, a = 0, dlo, best_a = 0;
  double c0 = ls_eval(c, 0, &d1, &d2), bestc = c0;
  *nls = 0;
  if (d1 >= 0 || d2 <= 0) return 0; /* search is a descent direction for a convex cost: d1 < 0 */
  dlo = d1;
  a = -d1 / d2;
  for (int it = 0; it < m->ls_iterations; it++) {
    double cc = ls_eval(c, a, &d1, &d2);
    (*nls)++;
    if (cc < bestc) { bestc = cc; best_a = a; }
    if (fabs(d1) < gtol) break;
    if (d1 < 0) { lo = a; dlo = d1; } else hi = a;
    double an = (d2 > 0) ? a - d1 / d2 : -1;
    if (hi < 0) { if (an <= lo) an = 2 * a + 1e-12; }                 /* still unbracketed: Newton or expand */
    else if (!(an > lo && an < hi)) an = 0.5 * (lo + hi);              /* bracketed: Newton or bisect */
    if (hi >= 0 && hi - lo < 1e-14 * fmax(1, hi)) break;
    a = an;
  }
  (void)dlo;
  return best_a;
}

static void fwd_constraint_newton(const smjo_model* m, smjo_data* d) {
  int nv = m->nv, ne = d->nefc, ncon = d->ncon;
  if (ne == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    d->solver_niter = 0;
    return;
  }
  double* buf = (double*)calloc((size_t)ne * 8 + 8 * nv + (size_t)nv * nv * 2 + 36 * (ncon + 1) + 7 * (ncon + 1) + 64, sizeof(double));
  double *Jaref = buf, *Jv = Jaref + ne, *quad = Jv + ne, *force = quad + 3 * ne, *ftmp = force + ne, *spare = ftmp + ne;
  double *Ma = spare + ne, *Mv = Ma + nv, *grad = Mv + nv, *search = grad + nv, *qacc = search + nv, *tmpv = qacc + nv;
  double *H = tmpv + 2 * nv, *Hf = H + nv * nv, *coneH = Hf + nv * nv, *cq = coneH + 36 * (ncon + 1);
  int* state = (int*)calloc(2 * ne + 2, sizeof(int));
  int* state2 = state + ne;
  nctx ctx = {m, d, Jaref, Jv, quad, cq, {0, 0, 0}};
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
#define MATVEC_M(out, x) FL(2LL * nv * nv); for (int i_ = 0; i_ < nv; i_++) { double s_ = 0; for (int k_ = 0; k_ < nv; k_++) s_ += d->qM[i_ * nv + k_] * (x)[k_]; (out)[i_] = s_; }
#define JAREF(out, x) FL(2LL * ne * nv + ne); for (int i_ = 0; i_ < ne; i_++) { double s_ = 0; for (int k_ = 0; k_ < nv; k_++) s_ += d->efc_J[(size_t)i_ * nv + k_] * (x)[k_]; (out)[i_] = s_ - d->efc_aref[i_]; }
#define GAUSS(x, Mx) ({ double g_ = 0; FL(5 * nv); for (int k_ = 0; k_ < nv; k_++) g_ += 0.5 * ((Mx)[k_] - d->qfrc_smooth[k_]) * ((x)[k_] - d->qacc_smooth[k_]); g_; })
  /* warm start ([MJ] mj_warmstart, primal branch): the cheaper of qacc_warmstart and qacc_smooth */
  memcpy(qacc, m->warmstart ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
  MATVEC_M(Ma, qacc);
  JAREF(Jaref, qacc);
  double cost = newton_update(m, d, Jaref, force, state, NULL) + GAUSS(qacc, Ma);
  if (m->warmstart) {
    JAREF(Jv, d->qacc_smooth); /* Gauss term is zero at qacc_smooth */
    double cs = newton_update(m, d, Jv, ftmp, state2, NULL);
    if (cs < cost) { memcpy(qacc, d->qacc_smooth, sizeof(double) * nv); MATVEC_M(Ma, qacc); memcpy(Jaref, Jv, sizeof(double) * ne); }
  }
  int iter = 0, nls_total = 0;
  for (; iter < m->iterations;) {
    cost = newton_update(m, d, Jaref, force, state, coneH) + GAUSS(qacc, Ma);
    /* gradient */
    FL(2LL * ne * nv + 4 * nv);
    for (int k = 0; k < nv; k++) {
      double s = 0;
      for (int i = 0; i < ne; i++) s += d->efc_J[(size_t)i * nv + k] * force[i];
      d->qfrc_constraint[k] = s;
      grad[k] = Ma[k] - d->qfrc_smooth[k] - s;
    }
    double gn = 0;
    for (int k = 0; k < nv; k++) gn += grad[k] * grad[k];
    if (iter > 0 && scale * sqrt(gn) < m->tolerance) break;
    /* Hessian H = M + J' W J */
    memcpy(H, d->qM, sizeof(double) * nv * nv);
    for (int i = 0; i < ne; i++) {
      if (d->efc_type[i] == C_CONTACT_ELLIPTIC && state[i] == 4) {
        int c = d->efc_id[i], dim = d->contact[c].dim;
        for (int r = 0; r < dim; r++)
          for (int q = 0; q < dim; q++) {
            double h = coneH[36 * c + r * dim + q];
            if (h == 0) continue;
            const double *Jr = d->efc_J + (size_t)(i + r) * nv, *Jq = d->efc_J + (size_t)(i + q) * nv;
            for (int a = 0; a < nv; a++)
              if (Jr[a] != 0) {
                FL(3 * nv);
                for (int b = 0; b < nv; b++) H[a * nv + b] += h * Jr[a] * Jq[b];
              }
          }
        i += dim - 1;
      } else if (state[i] == 1) {
        const double* Jr = d->efc_J + (size_t)i * nv;
        double D = d->efc_D[i];
        for (int a = 0; a < nv; a++)
          if (Jr[a] != 0) {
            FL(3 * nv);
            for (int b = 0; b < nv; b++) H[a * nv + b] += D * Jr[a] * Jr[b];
          }
      }
    }
    memcpy(Hf, H, sizeof(double) * nv * nv);
    chol_factor(Hf, nv, MINVAL);
    chol_solve(search, Hf, grad, nv);
    double sn = 0;
    for (int k = 0; k < nv; k++) { search[k] = -search[k]; sn += search[k] * search[k]; }
    sn = sqrt(sn);
    /* line-search preparation ([MJ] CGprepare) */
    FL(2LL * ne * nv + 9 * ne + 6 * nv);
    MATVEC_M(Mv, search);
    for (int i = 0; i < ne; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * search[k];
      Jv[i] = s;
    }
    ctx.quadGauss[0] = GAUSS(qacc, Ma);
    ctx.quadGauss[1] = 0; ctx.quadGauss[2] = 0;
    for (int k = 0; k < nv; k++) { ctx.quadGauss[1] += search[k] * (Ma[k] - d->qfrc_smooth[k]); ctx.quadGauss[2] += 0.5 * search[k] * Mv[k]; }
    for (int i = 0; i < ne; i++) {
      double D = d->efc_D[i];
      quad[3 * i] = 0.5 * D * Jaref[i] * Jaref[i]; quad[3 * i + 1] = D * Jaref[i] * Jv[i]; quad[3 * i + 2] = 0.5 * D * Jv[i] * Jv[i];
    }
    for (int c = 0; c < ncon; c++) {
      contact_t* con = d->contact + c;
      int i = con->efc_address, dim = con->dim;
      if (i < 0 || dim < 3) continue;
      for (int j = 1; j < dim; j++) { quad[3 * i] += quad[3 * (i + j)]; quad[3 * i + 1] += quad[3 * (i + j) + 1]; quad[3 * i + 2] += quad[3 * (i + j) + 2]; }
      double mu = con->mu, *k = cq + 7 * c;
      k[0] = Jaref[i] * mu; k[1] = Jv[i] * mu; k[2] = k[3] = k[4] = 0;
      for (int j = 1; j < dim; j++) {
        double u = Jaref[i + j] * con->friction[j - 1], v = Jv[i + j] * con->friction[j - 1];
        k[2] += u * u; k[3] += u * v; k[4] += v * v;
      }
      k[5] = d->efc_D[i] / fmax(mu * mu * (1 + mu * mu), MINVAL); k[6] = mu;
    }
    int nls;
    double alpha = ls_search(m, &ctx, sn, &nls);
    nls_total += nls;
    iter++;
    if (alpha == 0) break;
    FL(4 * nv + 2 * ne);
    for (int k = 0; k < nv; k++) { qacc[k] += alpha * search[k]; Ma[k] += alpha * Mv[k]; }
    for (int i = 0; i < ne; i++) Jaref[i] += alpha * Jv[i];
    double newcost = newton_update(m, d, Jaref, ftmp, state2, NULL) + GAUSS(qacc, Ma);
    double improvement = scale * (cost - newcost);
    if (improvement < m->tolerance) {
      newton_update(m, d, Jaref, force, state, NULL);
      for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int i = 0; i < ne; i++) s += d->efc_J[(size_t)i * nv + k] * force[i];
        d->qfrc_constraint[k] = s;
      }
      break;
    }
  }
  d->solver_niter = iter;
  (void)nls_total;
  memcpy(d->efc_force, force, sizeof(double) * ne);
  memcpy(d->qacc, qacc, sizeof(double) * nv);
  memcpy(d->qacc_warmstart, qacc, sizeof(double) * nv);
  free(buf); free(state);
#undef MATVEC_M
#undef JAREF
#undef GAUSS
}

/* ------------------------------------------------------------------ forward */
/* option manifold_keep: the kept manifolds as one buffer of smjo_mc_words() doubles (tests: an oracle that evaluates one step of
 * another one's trajectory takes that one's kept manifolds along) */
int smjo_mc_words(void) { return 64 * 39; }
void smjo_mc_get(const smjo_data* d, double* buf) {
  for (int k = 0; k < 64; k++) {
    double* w = buf + 39 * k;
    w[0] = d->mc_tag[k]; w[1] = d->mc_n[k];
    memcpy(w + 2, d->mc_pose[k], 14 * 8); memcpy(w + 16, d->mc_nrm[k], 24); memcpy(w + 19, d->mc_con[k], 20 * 8);
  }
}
void smjo_mc_set(smjo_data* d, const double* buf) {
  for (int k = 0; k < 64; k++) {
    const double* w = buf + 39 * k;
    d->mc_tag[k] = (int)w[0]; d->mc_n[k] = (int)w[1];
    memcpy(d->mc_pose[k], w + 2, 14 * 8); memcpy(d->mc_nrm[k], w + 16, 24); memcpy(d->mc_con[k], w + 19, 20 * 8);
  }
}
void smjo_set_contacts(smjo_data* d, int n, const double* con /* n x (dist, pos3, normal3, geom1, geom2) */) {
  free(d->override_con);
  d->override_con = (double*)malloc(sizeof(double) * 9 * (n > 0 ? n : 1));
  memcpy(d->override_con, con, sizeof(double) * 9 * n);
  d->override_ncon = n;
}

void smjo_forward(const smjo_model* m, smjo_data* d) {
  int nv = m->nv;
  kinematics(m, d);
  com_pos(m, d);
  tendon_transmission(m, d);
  crb_factor(m, d);
  collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  passive(m, d);
  /* [MJ] mj_referenceConstraint */
  for (int i = 0; i < d->nefc; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qvel[k];
    d->efc_vel[i] = s;
    d->efc_aref[i] = -d->efc_KBIP[4 * i + 1] * s - d->efc_KBIP[4 * i] * d->efc_KBIP[4 * i + 2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
  rne_bias(m, d);
  fwd_actuation(m, d);
  for (int k = 0; k < nv; k++)
    d->qfrc_smooth[k] = d->qfrc_passive[k] - d->qfrc_bias[k] + d->qfrc_applied[k] + d->qfrc_actuator[k];
  chol_solve(d->qacc_smooth, d->qL, d->qfrc_smooth, nv);
  if (m->solver == 2) fwd_constraint_newton(m, d);
  else fwd_constraint(m, d);
}

/* ------------------------------------------------------------------ B.8 integrate (implicitfast) */
static void integrate_pos(const smjo_model* m, double* qpos, const double* qvel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[da + k];
      double w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]}, ang = norm3(w) * h;
      if (ang > 0) { /* [MJ] mju_quatIntegrate */
        double ax[3] = {w[0], w[1], w[2]}, dq[4];
        normalize3(ax);
        axisangle2quat(dq, ax, ang);
        quat_mul(qpos + qa + 3, qpos + qa + 3, dq);
      }
      quat_normalize(qpos + qa + 3);
    } else qpos[qa] += h * qvel[da];
  }
}

void smjo_step(const smjo_model* m, smjo_data* d) {
  int nv = m->nv;
  double h = m->timestep;
  smjo_forward(m, d);
  /* qH = M - h*D, D = d(qfrc_passive + qfrc_actuator)/d(qvel), symmetric part ([MJ] mjd_smooth_vel, flg_bias=0) */
  memcpy(d->qH, d->qM, sizeof(double) * nv * nv);
  for (int k = 0; k < nv; k++) d->qH[k * nv + k] += h * m->dof_damping[k];
  for (int a = 0; a < m->nu; a++) {
    if (m->actuator_biastype[a] != 1) continue;
    double bv = m->actuator_biasprm[3 * a + 2];
    if (bv == 0) continue;
    if (m->actuator_forcelimited[a] &&
        (d->actuator_force[a] <= m->actuator_forcerange[2 * a] || d->actuator_force[a] >= m->actuator_forcerange[2 * a + 1]))
      continue;
    const double* mo = d->actuator_moment + a * nv;
    for (int i = 0; i < nv; i++)
      if (mo[i] != 0) {
        FL(3 * nv);
        for (int j = 0; j < nv; j++) d->qH[i * nv + j] -= h * bv * mo[i] * mo[j];
      }
  }
  chol_factor(d->qH, nv, MINVAL);
  double* rhs = d->scratch;
  for (int k = 0; k < nv; k++) rhs[k] = d->qfrc_smooth[k] + d->qfrc_constraint[k];
  chol_solve(rhs, d->qH, rhs, nv);
  FL(4 * nv);
  for (int k = 0; k < nv; k++) d->qvel[k] += h * rhs[k];
  integrate_pos(m, d->qpos, d->qvel, h);
  d->time += h;
}

void smjo_step_n(const smjo_model* m, smjo_data* d, int n) {
  for (int i = 0; i < n; i++) smjo_step(m, d);
}

/* ------------------------------------------------------------------ B.9 sensors */
static double ray_geom(const smjo_model* m, const smjo_data* d, int g, const double* pnt, const double* vec);
static void rb_ensure(smjo_model* m);
static double rb_ray(const smjo_model* m, int mesh, const double* o, const double* dv, double tnear, double best, int cull);

void smjo_sensors(const smjo_model* m, smjo_data* d, int with_lidar) {
  /* expects smjo_forward() state; [MJ] mj_rnePostConstraint for cacc, then mj_objectAcceleration/Velocity */
  int nv = m->nv;
  (void)nv;
  double* cacc = d->cacc;
  memset(cacc, 0, 48);
  for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
  for (int b = 1; b < m->nbody; b++) {
    double a[6];
    memcpy(a, cacc + 6 * m->body_parentid[b], 48);
    for (int k = m->body_dofadr[b]; k < m->body_dofadr[b] + m->body_dofnum[b]; k++)
      for (int x = 0; x < 6; x++) a[x] += d->cdof_dot[6 * k + x] * d->qvel[k] + d->cdof[6 * k + x] * d->qacc[k];
    memcpy(cacc + 6 * b, a, 48);
  }
  if (m->imu_site >= 0) {
    int s = m->imu_site, b = m->site_bodyid[s];
    const double* R = d->site_xmat + 9 * s;
    const double* com = d->subtree_com + 3 * m->body_rootid[b];
    double off[3] = {d->site_xpos[3 * s] - com[0], d->site_xpos[3 * s + 1] - com[1], d->site_xpos[3 * s + 2] - com[2]};
    const double *cv = d->cvel + 6 * b, *ca = cacc + 6 * b;
    double t[3], vlin[3], alin[3], c2[3];
    cross3(t, cv, off);
    for (int k = 0; k < 3; k++) vlin[k] = cv[3 + k] + t[k];
    cross3(t, ca, off);
    for (int k = 0; k < 3; k++) alin[k] = ca[3 + k] + t[k];
    cross3(c2, cv, vlin);
    for (int k = 0; k < 3; k++) alin[k] += c2[k];
    mulmat3Tvec(d->gyro, R, cv);
    mulmat3Tvec(d->accel, R, alin);
  }
  if (with_lidar)
    for (int i = 0; i < m->nlidar; i++) {
      int s = m->lidar_site[i], sb = m->site_bodyid[s];
      const double* R = d->site_xmat + 9 * s;
      /* geoms in the laser's weld group: state-independent hits ray-cast by the model compiler against the triangle
       * meshes (sensor_lidar_static); all other geoms at run time: plane and primitives in closed form, mesh geoms
       * ([MJ] mj_rayMesh: the mesh's own triangles, both sides, any geom group) through the triangle tree */
      double vec[3] = {R[2], R[5], R[8]}, best = m->lidar_static[i];
      const double* pnt = d->site_xpos + 3 * s;
      if (m->rmesh_vert) rb_ensure((smjo_model*)m);
      for (int g = 0; g < m->ngeom; g++) {
        if (m->body_weldid[m->geom_bodyid[g]] == m->body_weldid[sb] || m->geom_rgba[4 * g + 3] == 0) continue;
        if (m->geom_type[g] == G_MESH) {
          if (!m->rmesh_vert || m->geom_rmeshid[g] < 0) continue;
          const double *pos = d->geom_xpos + 3 * g, *Rg = d->geom_xmat + 9 * g;
          double dif[3] = {pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]}, lp[3], lv[3];
          mulmat3Tvec(lp, Rg, dif);
          mulmat3Tvec(lv, Rg, vec);
          double x = rb_ray(m, m->geom_rmeshid[g], lp, lv, 0.0, best >= 0 ? best : 1e300, 0);
          if (x < 1e299 && (best < 0 || x < best)) best = x;
          continue;
        }
        double x = ray_geom(m, d, g, pnt, vec);
        if (x >= 0 && (best < 0 || x < best)) best = x;
      }
      if (best > m->lidar_cutoff && m->lidar_cutoff > 0) best = m->lidar_cutoff;
      d->lidar[i] = best;
    }
}

/* [MJ] mj_rayGeom for plane / sphere / cylinder / box; mesh geoms are tested against their convex hull's
 * bounding box only when they collide (visual triangle meshes are outside the oracle's scope for now). */
static double ray_quad(double a, double b, double c) {
  double det = b * b - a * c;
  if (det < MINVAL) return -1;
  det = sqrt(det);
  double x0 = (-b - det) / a, x1 = (-b + det) / a;
  if (x0 >= 0) return x0;
  if (x1 >= 0) return x1;
  return -1;
}
static double ray_geom(const smjo_model* m, const smjo_data* d, int g, const double* pnt, const double* vec) {
  const double *pos = d->geom_xpos + 3 * g, *R = d->geom_xmat + 9 * g, *size = m->geom_size + 3 * g;
  double dif[3] = {pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]}, lp[3], lv[3];
  mulmat3Tvec(lp, R, dif);
  mulmat3Tvec(lv, R, vec);
  int t = m->geom_type[g];
  if (t == G_PLANE) {
    if (lv[2] > -MINVAL) return -1;
    double x = -lp[2] / lv[2];
    if (x < 0) return -1;
    double px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
    if ((size[0] <= 0 || fabs(px) <= size[0]) && (size[1] <= 0 || fabs(py) <= size[1])) return x;
    return -1;
  }
  if (t == G_SPHERE) return ray_quad(dot3(lv, lv), dot3(lv, lp), dot3(lp, lp) - size[0] * size[0]);
  if (t == G_CYLINDER) {
    double best = -1;
    double a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    if (a > MINVAL) {
      double x = ray_quad(a, b, c);
      if (x >= 0 && fabs(lp[2] + x * lv[2]) <= size[1]) best = x;
    }
    if (fabs(lv[2]) > MINVAL)
      for (int s = -1; s <= 1; s += 2) {
        double x = (s * size[1] - lp[2]) / lv[2];
        if (x >= 0) {
          double px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
          if (px * px + py * py <= size[0] * size[0] && (best < 0 || x < best)) best = x;
        }
      }
    return best;
  }
  if (t == G_BOX) {
    double best = -1;
    for (int ax = 0; ax < 3; ax++) {
      if (fabs(lv[ax]) < MINVAL) continue;
      for (int s = -1; s <= 1; s += 2) {
        double x = (s * size[ax] - lp[ax]) / lv[ax];
        if (x < 0) continue;
        int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
        if (fabs(lp[a1] + x * lv[a1]) <= size[a1] && fabs(lp[a2] + x * lv[a2]) <= size[a2] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (t == G_CAPSULE) {   /* [MJ] mj_rayGeom capsule: the cylinder's side between the caps, then the two end spheres beyond them */
    double best = -1;
    double a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    if (a > MINVAL) {
      double det = b * b - a * c;
      if (det >= MINVAL) {
        det = sqrt(det);
        for (int sgn = -1; sgn <= 1; sgn += 2) {
          double x = (-b + sgn * det) / a;
          if (x >= 0 && fabs(lp[2] + x * lv[2]) <= size[1] && (best < 0 || x < best)) best = x;
        }
      }
    }
    for (int sg = -1; sg <= 1; sg += 2) {
      double q[3] = {lp[0], lp[1], lp[2] - sg * size[1]};
      double aa = dot3(lv, lv), bb = dot3(lv, q), cc = dot3(q, q) - size[0] * size[0], dd = bb * bb - aa * cc;
      if (dd < MINVAL) continue;
      dd = sqrt(dd);
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        double x = (-bb + sgn * dd) / aa;
        if (x >= 0 && sg * (lp[2] + x * lv[2]) >= size[1] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (t == G_ELLIPSOID) {
    double sc[3] = {1 / size[0], 1 / size[1], 1 / size[2]};
    double v[3] = {lv[0] * sc[0], lv[1] * sc[1], lv[2] * sc[2]}, q[3] = {lp[0] * sc[0], lp[1] * sc[1], lp[2] * sc[2]};
    return ray_quad(dot3(v, v), dot3(v, q), dot3(q, q) - 1);
  }
  return -1;
}


/* ------------------------------------------------------------------ depth cameras
 * Restates what the reference obtains from mujoco.Renderer (update_scene + render with depth enabled,
 * stretch_mujoco/mujoco_server_camera_manager.py:127-143) followed by utils.limit_depth_distance (utils.py:87-91):
 * [MJ] geoms of groups 0-2 with alpha > 0 are drawn, the camera looks down -Z with +Y up, projection is the pure-fovy
 * pinhole (cam_sensorsize 0), depth is the distance along the optical axis clipped to [znear, zfar] * extent, back faces
 * are culled.  Ray casting at pixel centres instead of rasterisation: equal away from silhouette pixels.
 * The mesh acceleration structure is a median-split tree walked with a stack -- deliberately not the Morton/heap layout
 * of the HIP renderer (csrc/smj_bvh.h), so that the two share no traversal logic. */
typedef struct { double lo[3], hi[3]; int left, right, first, count; } rnode;
struct rbvh { rnode* node; int nnode; int* tri; /* face ids in leaf order */ };

static void rb_bounds(const smjo_model* m, int mesh, const int* ids, int n, double* lo, double* hi) {
  const double* V = m->rmesh_vert + 3 * (size_t)m->rmesh_vertadr[mesh];
  const int* F = m->rmesh_face + 3 * (size_t)m->rmesh_faceadr[mesh];
  for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) {
      const double* v = V + 3 * F[3 * ids[i] + c];
      for (int k = 0; k < 3; k++) { if (v[k] < lo[k]) lo[k] = v[k]; if (v[k] > hi[k]) hi[k] = v[k]; }
    }
}
static const double* rb_cen; /* qsort context: centroid coordinate per face */
static int rb_cmp(const void* a, const void* b) {
  double x = rb_cen[*(const int*)a], y = rb_cen[*(const int*)b];
  return (x > y) - (x < y);
}
static int rb_build(const smjo_model* m, int mesh, struct rbvh* t, int first, int count, double* cen3) {
  int id = t->nnode++;
  rnode* nd = &t->node[id];
  rb_bounds(m, mesh, t->tri + first, count, nd->lo, nd->hi);
  nd->first = first; nd->count = count; nd->left = nd->right = -1;
  if (count <= 4) return id;
  int ax = 0;
  double ext[3] = {nd->hi[0] - nd->lo[0], nd->hi[1] - nd->lo[1], nd->hi[2] - nd->lo[2]};
  if (ext[1] > ext[ax]) ax = 1;
  if (ext[2] > ext[ax]) ax = 2;
  rb_cen = cen3 + (size_t)ax * m->rmesh_facenum[mesh];
  qsort(t->tri + first, count, sizeof(int), rb_cmp);
  int half = count / 2;
  int l = rb_build(m, mesh, t, first, half, cen3);
  int r = rb_build(m, mesh, t, first + half, count - half, cen3);
  t->node[id].left = l; t->node[id].right = r; t->node[id].count = 0;
  return id;
}
static void rb_ensure(smjo_model* m) {
  if (m->rbvh || !m->rmesh_vert) return;
  m->rbvh = (struct rbvh*)MTRACK(m, calloc(m->nrmesh ? m->nrmesh : 1, sizeof(struct rbvh)));
  for (int mesh = 0; mesh < m->nrmesh; mesh++) {
    int nf = m->rmesh_facenum[mesh];
    struct rbvh* t = &m->rbvh[mesh];
    t->node = (rnode*)MTRACK(m, calloc(2 * (size_t)(nf ? nf : 1), sizeof(rnode)));
    t->tri = (int*)MTRACK(m, malloc((nf ? nf : 1) * sizeof(int)));
    double* cen3 = (double*)malloc(3 * (size_t)(nf ? nf : 1) * sizeof(double));
    const double* V = m->rmesh_vert + 3 * (size_t)m->rmesh_vertadr[mesh];
    const int* F = m->rmesh_face + 3 * (size_t)m->rmesh_faceadr[mesh];
    for (int f = 0; f < nf; f++) {
      t->tri[f] = f;
      for (int k = 0; k < 3; k++) cen3[(size_t)k * nf + f] = (V[3 * F[3 * f] + k] + V[3 * F[3 * f + 1] + k] + V[3 * F[3 * f + 2] + k]) / 3;
    }
    if (nf > 0) rb_build(m, mesh, t, 0, nf, cen3);
    free(cen3);
  }
}
/* nearest front-facing hit with t in [tnear, best) */
static double rb_ray(const smjo_model* m, int mesh, const double* o, const double* dv, double tnear, double best, int cull) {
  const struct rbvh* t = &m->rbvh[mesh];
  if (t->nnode == 0) return best;
  const double* V = m->rmesh_vert + 3 * (size_t)m->rmesh_vertadr[mesh];
  const int* F = m->rmesh_face + 3 * (size_t)m->rmesh_faceadr[mesh];
  int stack[128], sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const rnode* nd = &t->node[stack[--sp]];
    double t0 = tnear, t1 = best;
    int miss = 0;
    for (int k = 0; k < 3 && !miss; k++) {
      if (fabs(dv[k]) < 1e-300) { if (o[k] < nd->lo[k] || o[k] > nd->hi[k]) miss = 1; continue; }
      double a = (nd->lo[k] - o[k]) / dv[k], b = (nd->hi[k] - o[k]) / dv[k];
      if (a > b) { double s = a; a = b; b = s; }
      if (a > t0) t0 = a;
      if (b < t1) t1 = b;
      if (t0 > t1) miss = 1;
    }
    if (miss) continue;
    if (nd->left >= 0) { stack[sp++] = nd->left; stack[sp++] = nd->right; continue; }
    for (int i = 0; i < nd->count; i++) {
      const int f = t->tri[nd->first + i];
      const double *a = V + 3 * F[3 * f], *b = V + 3 * F[3 * f + 1], *c = V + 3 * F[3 * f + 2];
      double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, p[3], q[3];
      cross3(p, dv, e2);
      double det = dot3(e1, p);
      if (cull ? det <= 1e-30 : fabs(det) <= 1e-30) continue; /* back face (cameras only) or edge-on */
      double tv[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
      double u = dot3(tv, p) / det;
      if (u < 0 || u > 1) continue;
      cross3(q, tv, e1);
      double v = dot3(dv, q) / det;
      if (v < 0 || u + v > 1) continue;
      double x = dot3(e2, q) / det;
      if (x >= tnear && x < best) best = x;
    }
  }
  return best;
}
/* entering intersection with a primitive (outside faces only), in the geom frame */
static double ray_prim_front(int type, const double* size, const double* lp, const double* lv, double tnear) {
  if (type == G_PLANE) {
    if (lv[2] > -MINVAL) return -1;
    double x = -lp[2] / lv[2];
    if (x < tnear) return -1;
    double px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
    if ((size[0] <= 0 || fabs(px) <= size[0]) && (size[1] <= 0 || fabs(py) <= size[1])) return x;
    return -1;
  }
  if (type == G_SPHERE) {
    double a = dot3(lv, lv), b = dot3(lv, lp), c = dot3(lp, lp) - size[0] * size[0], det = b * b - a * c;
    if (det < MINVAL) return -1;
    double x = (-b - sqrt(det)) / a;
    return x >= tnear ? x : -1;
  }
  if (type == G_CYLINDER) {
    double best = -1;
    double a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    double det = b * b - a * c;
    if (a > MINVAL && det >= MINVAL) {
      double x = (-b - sqrt(det)) / a;
      if (x >= tnear && fabs(lp[2] + x * lv[2]) <= size[1]) best = x;
    }
    if (fabs(lv[2]) > MINVAL) {
      double sg = lv[2] < 0 ? 1 : -1;
      double x = (sg * size[1] - lp[2]) / lv[2];
      if (x >= tnear) {
        double px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
        if (px * px + py * py <= size[0] * size[0] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == G_BOX) {
    double best = -1;
    for (int ax = 0; ax < 3; ax++) {
      if (fabs(lv[ax]) < MINVAL) continue;
      double sg = lv[ax] < 0 ? 1 : -1;
      double x = (sg * size[ax] - lp[ax]) / lv[ax];
      if (x < tnear) continue;
      int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
      if (fabs(lp[a1] + x * lv[a1]) <= size[a1] && fabs(lp[a2] + x * lv[a2]) <= size[a2] && (best < 0 || x < best)) best = x;
    }
    return best;
  }
  if (type == G_CAPSULE) {
    double best = -1;
    double a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    double det = b * b - a * c;
    if (a > MINVAL && det >= MINVAL) {
      double x = (-b - sqrt(det)) / a;
      if (x >= tnear && fabs(lp[2] + x * lv[2]) <= size[1]) best = x;
    }
    for (int sg = -1; sg <= 1; sg += 2) {
      double q[3] = {lp[0], lp[1], lp[2] - sg * size[1]};
      double aa = dot3(lv, lv), bb = dot3(lv, q), cc = dot3(q, q) - size[0] * size[0], dd = bb * bb - aa * cc;
      if (dd < MINVAL) continue;
      double x = (-bb - sqrt(dd)) / aa;
      if (x >= tnear && sg * (lp[2] + x * lv[2]) >= size[1] && (best < 0 || x < best)) best = x;
    }
    return best;
  }
  if (type == G_ELLIPSOID) {
    double sc[3] = {1 / size[0], 1 / size[1], 1 / size[2]};
    double v[3] = {lv[0] * sc[0], lv[1] * sc[1], lv[2] * sc[2]}, q[3] = {lp[0] * sc[0], lp[1] * sc[1], lp[2] * sc[2]};
    double a = dot3(v, v), b = dot3(v, q), c = dot3(q, q) - 1, det = b * b - a * c;
    if (det < MINVAL) return -1;
    double x = (-b - sqrt(det)) / a;
    return x >= tnear ? x : -1;
  }
  return -1;
}
/* out: float[H][W]; returns 0, or -1 when the blob has no render tables */
static int render_rays(smjo_model* m, const smjo_data* d, int cam, int W, int H, double fovy_deg, double max_depth, float* out, int* gid) {
  if (!m->rmesh_vert || cam < 0 || cam >= m->ncam) return -1;
  rb_ensure(m);
  const double* cp = d->cam_xpos + 3 * cam;
  const double* cm = d->cam_xmat + 9 * cam;
  const double th = tan(fovy_deg * 3.14159265358979323846 / 360.0), aspect = (double)W / (double)H;
  const double tnear = m->znear, tfar = (max_depth > 0 && max_depth < m->zfar) ? max_depth : m->zfar;
  for (int v = 0; v < H; v++)
    for (int u = 0; u < W; u++) {
      double dc[3] = {((u + 0.5) / W * 2 - 1) * th * aspect, (1 - (v + 0.5) / H * 2) * th, -1.0}, dv[3];
      mulmat3vec(dv, cm, dc);
      double best = tfar * (1 + 1e-6);
      int hit = -1;
      for (int g = 0; g < m->ngeom; g++) {
        if (m->geom_group[g] > 2 || m->geom_rgba[4 * g + 3] == 0) continue;
        int type = m->geom_type[g];
        if (type == G_MESH && m->geom_rmeshid[g] < 0) continue;
        const double *pos = d->geom_xpos + 3 * g, *R = d->geom_xmat + 9 * g;
        double dif[3] = {cp[0] - pos[0], cp[1] - pos[1], cp[2] - pos[2]}, lp[3], lv[3];
        mulmat3Tvec(lp, R, dif);
        mulmat3Tvec(lv, R, dv);
        if (type == G_MESH) {
          /* cheap reject against the geom AABB (geom_aabb: centre, half sizes in the geom frame) */
          const double* bb = m->geom_aabb + 6 * g;
          double t0 = tnear, t1 = best;
          int miss = 0;
          for (int k = 0; k < 3 && !miss; k++) {
            const double hs = bb[3 + k] + 1e-6; /* the box was taken before the vertices were rounded to fp32 */
            if (fabs(lv[k]) < 1e-300) { if (fabs(lp[k] - bb[k]) > hs) miss = 1; continue; }
            double a = (bb[k] - hs - lp[k]) / lv[k], b = (bb[k] + hs - lp[k]) / lv[k];
            if (a > b) { double s = a; a = b; b = s; }
            if (a > t0) t0 = a;
            if (b < t1) t1 = b;
            if (t0 > t1) miss = 1;
          }
          if (!miss) {
            const double nb = rb_ray(m, m->geom_rmeshid[g], lp, lv, tnear, best, 1);
            if (nb < best) { best = nb; hit = g; }
          }
        } else {
          double x = ray_prim_front(type, m->geom_size + 3 * g, lp, lv, tnear);
          if (x >= 0 && x < best) { best = x; hit = g; }
        }
      }
      double z = best;
      if (z > tfar) z = max_depth > 0 ? 0 : m->zfar;
      if (max_depth > 0 && z > max_depth) z = 0;
      if (out) out[(size_t)v * W + u] = (float)z;
      if (gid) gid[(size_t)v * W + u] = best <= tfar ? hit : -1;
    }
  return 0;
}
int smjo_render_depth(smjo_model* m, const smjo_data* d, int cam, int W, int H, double fovy_deg, double max_depth, float* out) {
  return render_rays(m, d, cam, W, H, fovy_deg, max_depth, out, NULL);
}
/* The geom each pixel ray hits first (front faces, the geoms a camera draws), -1 = nothing up to the far plane; and its colour
 * as the build's RGB cameras define it: the geom's rgba (material rgba where the MJCF names one) as 8-bit albedo, no lighting,
 * no textures; sky = (169, 224, 255), the colour docs/getting_started.ipynb cell 14 prints for empty pixels.  NOT MuJoCo's
 * OpenGL image: a stand-in with the reference's shapes, resolutions and intrinsics. */
int smjo_render_geomid(smjo_model* m, const smjo_data* d, int cam, int W, int H, double fovy_deg, int* gid, unsigned char* rgb) {
  const int rc = render_rays(m, d, cam, W, H, fovy_deg, 0.0, NULL, gid);
  if (rc) return rc;
  if (rgb)
    for (size_t i = 0; i < (size_t)W * H; i++) {
      const int g = gid[i];
      const unsigned char sky[3] = {169, 224, 255};
      for (int k = 0; k < 3; k++) {
        double c = g >= 0 ? m->geom_rgba[4 * g + k] : 0;
        c = c < 0 ? 0 : (c > 1 ? 1 : c);
        rgb[3 * i + k] = g >= 0 ? (unsigned char)(c * 255.0 + 0.5) : sky[k];
      }
    }
  return 0;
}
