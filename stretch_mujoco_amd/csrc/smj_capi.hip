// C-ABI (include/smj.h) of the batched Stretch physics path for gfx950.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/smj.h"
#include "smj_kernels.h"
#include "smj_model_load.h"
#include "smj_bvh.h"
#include "smj_meshlet.h"
#include "smj_render.h"
#include "smj_comm.h"

struct smj_ctx {
  int device = 0;
  int num_envs = 0;
  DevModel model{};
  DevState state{};
  std::vector<void*> allocs;
  std::string err;
  float* qpos0_dev = nullptr;
  float* stage = nullptr;      // env-major staging copy of the state, [num_envs][layout.stride] (DevState::stage)
  int variant = 0;             // 0: standard step kernel, 1: tall (its 128-row build, three envs per CU), 2 / 3 / 4: big with 38 / 50 / 64 dof columns, 5 / 6: main tree + up to 16 / 32 satellites (smj_model.h, smj_sat.h)
  // capacity escalation (standard variant): the model once more with the tall variant's records, and the list of parked envs
  DevModel model_esc{};
  bool has_esc = false;
  int* redo = nullptr;
  int escalate = 1;
  // launch order: per-env shader time of the last launch and the permutation sorted by it (DevState::order / cost)
  int* cost = nullptr;
  int* order = nullptr;
  int balance = 1;
  int lidar_cull = 1;      // lidar: drop, per env, the geoms that cannot reach the scan plane (smj_render.h lidar_plane_*)
  int balance_min_envs = 1024; // cost-ordered dispatch only above this many envs (option balance_min_envs)
  // pipelined chunks (and with them the pollers) only above this many envs (option pipeline_min_envs).  Round 5: 511, was 1024 -- measured under
  // random actions at 1024 envs: empty scene 5.16 -> 5.52 M, kitchen stand-in 3.97 -> 4.98 M, scene.xml 1.96 -> 2.37 M, the kitchen at Robocasa
  // scale 0.73 -> 1.30 M (an env handed to the large build is finished beside the launch instead of after it); at 512 envs nothing but the
  // last (0.56 -> 0.77 M); settled scenes pay 3-5 % for the chunks' state round trips at these sizes
  int pipe_min_envs = 511;
  int newton_two_waves = 3;   // Newton on the 16-satellite build: 1 = the two-wavefront kernel (smj_kernels_sat2.hip: the second wavefront takes the moving-moving pairs and the satellites' lane-serial stages), 0 = one wavefront per env
  int pgs_two_waves = 1;   // PGS on the 16-satellite build: 1 = the two-wavefront kernel (smj_kernels_satp.hip), 0 = one wavefront per env
  int balance_min = 1;   // steps per launch from which the cost-ordered dispatch is used (round 4: 1 -- a one-step launch is as long as its slowest round of workgroups; was 4)
  int chunk = 0;               // steps per dispatch inside one smj_step (0: the whole launch at once; measured: no gain)
  int pipeline_big = 1;        // pipelined dispatch for the two-envs-per-CU builds of the big variant too (option "pipeline_big")
  int pipeline = 5;            // chunk length of the pipelined dispatch (DevState::pipe_len; 0 = one workgroup per env for the whole launch)
  bool pollers_always = false; // option "pollers" < 0: send the pollers with every launch (tests)
  bool prof_warned = false;    // the one-time warning of smj_step: profiling slot bound, PGS kernel without counters
  int pollers = 2;             // tall-variant workgroups that finish parked envs beside the standard kernel (0: the sweep does it all)
  int* progress = nullptr;     // [B] progress, [B] done_steps, [SMJ_SCHED_WORDS] sched (DevState)
  size_t redo_cap = 0;         // entries the escalation list holds
  hipStream_t aux = nullptr;   // the pollers' stream
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  SmjCaps caps{};              // capacities of the variant in use
  SmjStageLayout layout{};     // staging-row layout of the variant in use
  int debug_floats = 0;
  DevRender render{};
  bool has_render = false;
  float* pose_ws = nullptr;
  // camera-static depth layers: geoms welded to a camera's body always look the same from that camera, so they are
  // rendered once per (camera, image size, field of view) and every later render starts its rays from that layer
  struct Layer { int cam = -1, w = 0, h = 0; float fovy = 0; float* buf = nullptr; };
  std::vector<Layer> layers;
  float* depth_ws = nullptr;   // scratch of the depth renderer's per-env staging pass (allocated at the first render)   // internal [nbody*12][num_envs] body poses when the caller has not bound SMJ_SLOT_XPOSE
  void* slot_ptr[SMJ_SLOT_COUNT] = {};
  long slot_ld[SMJ_SLOT_COUNT] = {};
  // RCCL communicator of the env-sharded job (smj_comm_init); null in a single-GPU run
  SmjComm comm;
};

static_assert(sizeof(ncclUniqueId) == sizeof(SmjNcclId) && (int)ncclFloat == SMJ_NCCL_FLOAT32 && (int)ncclSuccess == SMJ_NCCL_SUCCESS && sizeof(ncclComm_t) == sizeof(void*),
              "smj_comm.h spells out RCCL's ABI by hand");

static int fail(smj_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIPCHK(c, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) return fail(c, -2, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

struct DeviceUploader {
  smj_ctx* c;
  template <class T>
  const T* put(const std::vector<T>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    c->allocs.push_back(d);
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return static_cast<const T*>(d);
  }
  const float* f32(const std::vector<float>& h) { return put(h); }
  const int* i32(const std::vector<int>& h) { return put(h); }
};

// Depth-camera tables: visible-geom list, camera frames and one BVH per render mesh (built here, uploaded once).
static int setup_render(smj_ctx* c, const void* blob, size_t nbytes) {
  SmjBlob b{static_cast<const uint8_t*>(blob), nbytes};
  const SmjBlobEntry* rg = b.find("k_rgeom");
  const SmjBlobEntry* rv = b.find("rmesh_vert");
  const SmjBlobEntry* rf = b.find("rmesh_face");
  const SmjBlobEntry* va = b.find("rmesh_vertadr");
  const SmjBlobEntry* vn = b.find("rmesh_vertnum");
  const SmjBlobEntry* fa = b.find("rmesh_faceadr");
  const SmjBlobEntry* fn = b.find("rmesh_facenum");
  const SmjBlobEntry* gm = b.find("geom_rmeshid");
  const SmjBlobEntry* cb = b.find("cam_bodyid");
  const SmjBlobEntry* cp = b.find("cam_pos");
  const SmjBlobEntry* cm = b.find("k_cam_mat");
  const SmjBlobEntry* vz = b.find("vis_znear_zfar_extent");
  const SmjBlobEntry* nr = b.find("k_nrgeom");
  const SmjBlobEntry* lg = b.find("k_lgeom");
  const SmjBlobEntry* nl = b.find("k_nlgeom");
  const SmjBlobEntry* ls = b.find("sensor_lidar_site");
  const SmjBlobEntry* lt = b.find("sensor_lidar_static");
  if (!rg || !rv || !rf || !va || !vn || !fa || !fn || !gm || !cb || !cp || !cm || !vz || !nr || !lg || !nl || !ls || !lt)
    return 0;   // model without ray-casting tables: smj_render_depth / lidar readout report that
  if (rv->dtype != 3 || rf->dtype != 1 || vz->dtype != 0 || cp->dtype != 0 || cm->dtype != 0)
    return fail(c, -3, "model blob: render tables have unexpected types");
  DeviceUploader up{c};
  DevRender& r = c->render;
  int nrgeom = 0;
  memcpy(&nrgeom, b.p + nr->offset, 4);
  if (nrgeom > SMJ_RGEOM_MAX) return fail(c, -4, "model has %d camera-visible geoms, renderer capacity is %d", nrgeom, SMJ_RGEOM_MAX);
  r.nrgeom = nrgeom;
  r.ncam = (int)(cb->nbytes / 4);
  r.nbody = c->model.nbody_all;
  double z[3];
  memcpy(z, b.p + vz->offset, 24);
  r.znear = (float)(z[0] * z[2]);
  r.zfar = (float)(z[1] * z[2]);
  auto geti = [&](const SmjBlobEntry* e) { std::vector<int> h(e->nbytes / 4 ? e->nbytes / 4 : 1, 0); memcpy(h.data(), b.p + e->offset, e->nbytes); return h; };
  auto getd = [&](const SmjBlobEntry* e) {
    std::vector<float> h(e->nbytes / 8 ? e->nbytes / 8 : 1, 0.f);
    const double* src = reinterpret_cast<const double*>(b.p + e->offset);
    for (size_t i = 0; i < e->nbytes / 8; i++) h[i] = (float)src[i];
    return h;
  };
  r.rgeom = up.i32(geti(rg));
  r.geom_rmeshid = up.i32(geti(gm));
  r.cam_bodyid = up.i32(geti(cb));
  r.cam_pos = up.f32(getd(cp));
  r.cam_mat = up.f32(getd(cm));
  memcpy(&r.nlgeom, b.p + nl->offset, 4);
  r.nlidar = c->model.nlidar;
  r.lidar_cutoff = c->model.lidar_cutoff;
  if (r.nlidar > 384) return fail(c, -4, "model has %d lidar rays, kernel capacity is 384", r.nlidar);
  r.lgeom = up.i32(geti(lg));
  r.lidar_site = up.i32(geti(ls));
  r.lidar_static = up.f32(getd(lt));
  r.site_bodyid = c->model.site_bodyid; r.site_pos = c->model.site_pos; r.site_mat = c->model.k_site_mat;
  if (!r.lgeom || !r.lidar_site || !r.lidar_static) return fail(c, -2, "device allocation failed for the lidar tables");
  {   // the scan plane of the rangefinders (smj_render.h lidar_plane_*): all sites on one body, all rays within 1e-5 of one plane
    r.lidar_plane_body = -1;
    const SmjBlobEntry* sp = b.find("site_pos");
    const SmjBlobEntry* sm = b.find("k_site_mat");
    const SmjBlobEntry* sb = b.find("site_bodyid");
    if (sp && sm && sb && sp->dtype == 0 && sm->dtype == 0 && r.nlidar >= 3) {
      const std::vector<int> sites = geti(ls), sbody = geti(sb);
      const double* P = reinterpret_cast<const double*>(b.p + sp->offset);
      const double* Mx = reinterpret_cast<const double*>(b.p + sm->offset);
      auto dir = [&](int i, double* d) { const double* m9 = Mx + 9 * (size_t)sites[i]; d[0] = m9[2]; d[1] = m9[5]; d[2] = m9[8]; };
      double d0[3], n[3] = {0, 0, 0}, best = 0;
      dir(0, d0);
      for (int i = 1; i < r.nlidar; i++) {   // the normal from the pair of rays closest to a right angle
        double d[3];
        dir(i, d);
        const double cx = d0[1] * d[2] - d0[2] * d[1], cy = d0[2] * d[0] - d0[0] * d[2], cz = d0[0] * d[1] - d0[1] * d[0], l = sqrt(cx * cx + cy * cy + cz * cz);
        if (l > best) { best = l; n[0] = cx / l; n[1] = cy / l; n[2] = cz / l; }
      }
      bool ok = best > 0.5;
      double p0[3] = {0, 0, 0}, slack = 0, slope = 0;
      for (int i = 0; i < r.nlidar; i++) for (int k = 0; k < 3; k++) p0[k] += P[3 * (size_t)sites[i] + k] / r.nlidar;
      for (int i = 0; ok && i < r.nlidar; i++) {
        double d[3];
        dir(i, d);
        ok = sbody[sites[i]] == sbody[sites[0]];
        const double* q = P + 3 * (size_t)sites[i];
        slack = fmax(slack, fabs((q[0] - p0[0]) * n[0] + (q[1] - p0[1]) * n[1] + (q[2] - p0[2]) * n[2]));
        slope = fmax(slope, fabs(d[0] * n[0] + d[1] * n[1] + d[2] * n[2]));
      }
      if (ok && slope < 1e-5) {
        r.lidar_plane_body = sbody[sites[0]];
        for (int k = 0; k < 3; k++) { r.lidar_plane_p[k] = (float)p0[k]; r.lidar_plane_n[k] = (float)n[k]; }
        r.lidar_plane_slack = (float)slack + 1e-4f;          // + fp32 rounding of metre-scale coordinates
        r.lidar_plane_slope = (float)fmax(slope, 2e-6);       // per metre along the ray (fp32 directions)
      }
    }
  }
  r.geom_type = c->model.geom_type; r.geom_bodyid = c->model.geom_bodyid; r.geom_pos = c->model.geom_pos;
  r.geom_mat = c->model.k_geom_mat; r.geom_size = c->model.geom_size; r.geom_rbound = c->model.geom_rbound;
  r.geom_bcenter = c->model.k_geom_bcenter; r.geom_aabb = c->model.geom_aabb;
  {
    const SmjBlobEntry* gc = b.find("geom_rgba");   // colours of the RGB stand-in (smj_render_rgb); older blobs may lack them
    r.geom_rgba = (gc && gc->dtype == 0) ? up.f32(getd(gc)) : nullptr;
  }
  // BVHs
  const std::vector<int> vadr = geti(va), vnum = geti(vn), fadr = geti(fa), fnum = geti(fn);
  const float* verts = reinterpret_cast<const float*>(b.p + rv->offset);
  const int* faces = reinterpret_cast<const int*>(b.p + rf->offset);
  const size_t nmesh = va->nbytes / 4;
  SmjBvhSet set;
  for (size_t i = 0; i < nmesh; i++) {
    // faces index the vertices of their own mesh
    smj_bvh_add_mesh(set, verts + 3 * (size_t)vadr[i], vnum[i], faces + 3 * (size_t)fadr[i], fnum[i]);
  }
  std::vector<int> meshtab(4 * (nmesh ? nmesh : 1), 0);
  for (size_t i = 0; i < nmesh; i++) {
    meshtab[4 * i] = set.mesh[i].nodebase; meshtab[4 * i + 1] = set.mesh[i].tribase; meshtab[4 * i + 2] = set.mesh[i].leaf0;
    meshtab[4 * i + 3] = set.mesh[i].ntri;
  }
  if (set.node.empty()) set.node.assign(16, 0.f);
  if (set.tri.empty()) set.tri.assign(12, 0.f);
  r.node = reinterpret_cast<const float4*>(up.f32(set.node));
  r.tri = reinterpret_cast<const float4*>(up.f32(set.tri));
  r.mesh = reinterpret_cast<const int4*>(up.i32(meshtab));
  if (!r.rgeom || !r.geom_rmeshid || !r.cam_bodyid || !r.cam_pos || !r.cam_mat || !r.node || !r.tri || !r.mesh)
    return fail(c, -2, "device allocation failed for the render tables");
  {   // the mesh rasteriser's tables: meshlets of every render mesh and the work list over the mesh geoms a camera can see
    const SmjBlobEntry* gt = b.find("geom_type");
    std::vector<int> rgh = geti(rg), gmh = geti(gm), gth = gt ? geti(gt) : std::vector<int>();
    SmjMeshletSet ml;
    for (size_t i = 0; i < nmesh; i++) smj_meshlets_add_mesh(ml, verts + 3 * (size_t)vadr[i], faces + 3 * (size_t)fadr[i], set.order[i]);
    const SmjBlobEntry* gs = b.find("geom_size");
    std::vector<float> gsh = (gs && gs->dtype == 0) ? getd(gs) : std::vector<float>();
    std::vector<int> boxlet(nrgeom > 0 ? nrgeom : 1, -1);
    for (int i = 0; i < nrgeom && gt && !gsh.empty(); i++)   // box geoms: one meshlet each, appended after the meshes'
      if (gth[rgh[i]] == 6) boxlet[i] = smj_meshlets_add_box(ml, gsh.data() + 3 * (size_t)rgh[i]);
    std::vector<int> rec(12 * (ml.let.size() ? ml.let.size() : 1), 0), work;
    for (size_t i = 0; i < ml.let.size(); i++) {
      const SmjMeshlet& m = ml.let[i];
      int* q = rec.data() + 12 * i;
      q[0] = m.vbase; q[1] = m.nvert; q[2] = m.tbase; q[3] = m.ntri;
      const float f[8] = {m.cen[0], m.cen[1], m.cen[2], m.rad, m.axis[0], m.axis[1], m.axis[2], m.cosc};
      memcpy(q + 4, f, sizeof f);
    }
    for (int i = 0; i < nrgeom && gt; i++) {   // the work list: one (geom entry, meshlet) item per meshlet, mesh after mesh
      const int g = rgh[i];
      if (boxlet[i] >= 0) { work.push_back(i); work.push_back(boxlet[i]); continue; }
      if (gth[g] != 7 || gmh[g] < 0) continue;
      for (int k = 0; k < ml.mesh_count[gmh[g]]; k++) { work.push_back(i); work.push_back(ml.mesh_first[gmh[g]] + k); }
    }
    r.raster_boxes = gsh.empty() ? 0 : 1;
    r.nmlist = (int)(work.size() / 2);
    if (work.empty()) work.assign(2, 0);
    if (ml.vert.empty()) ml.vert.assign(4, 0.f);
    std::vector<int> tri(ml.tri.begin(), ml.tri.end());
    if (tri.empty()) tri.assign(1, 0);
    r.mlvert = reinterpret_cast<const float4*>(up.f32(ml.vert));
    r.mltri = reinterpret_cast<const unsigned*>(up.i32(tri));
    r.mlrec = reinterpret_cast<const int4*>(up.i32(rec));
    r.mlitem = reinterpret_cast<const int2*>(up.i32(work));
    if (!r.mlvert || !r.mltri || !r.mlrec || !r.mlitem) return fail(c, -2, "device allocation failed for the meshlet tables");
    r.raster = 1;
    r.raster_splits = 8;
  }
  c->has_render = true;
  return 0;
}

extern "C" {

const char* smj_version(void) { return "smj 0.1 (gfx950)"; }

const char* smj_last_error(const smj_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int smj_create(const void* blob, size_t nbytes, int num_envs, int device, smj_ctx** out) {
  if (!out) return -1;
  *out = nullptr;
  if (!blob || nbytes < 16 || memcmp(blob, "SMJB0001", 8) != 0) return -1;
  if (num_envs <= 0) return -1;
  smj_ctx* c = new smj_ctx();
  *out = c;  // returned even on failure so that smj_last_error() can be read; caller must smj_destroy()
  c->device = device;
  c->num_envs = num_envs;
  HIPCHK(c, hipSetDevice(device));
  {
    // The pipelined chunks (and the pollers beside them) lean on workgroups being dispatched in index order: an env's chunk k + 1
    // waits -- bounded, 3 s -- for chunk k, which sits B workgroups earlier in the grid.  That holds on the gfx94x / gfx950
    // command processors this was measured on; on any other architecture the default is the plain launch (one workgroup per env
    // for the whole call).  smj_set_option("pipeline", k) overrides it either way.
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && strncmp(prop.gcnArchName, "gfx94", 5) != 0)) {
      c->pipeline = 0;
      c->pollers = 0;
    }
  }
  DeviceUploader up{c};
  SmjCaps caps[7] = {{NVP, NBP, NENT, NEFC, NCON, 0, 0}, {}, {}, {}, {}, {}, {}};   // standard, tall, big38, big50, big, sat, sat32 (smj_model.h)
  int dbg[7] = {SMJ_DEBUG_FLOATS, 0, 0, 0, 0, 0, 0};
  SmjCaps tall{};   // the 160-row build: escalation target of the standard variant and of the 128-row build
  int dbg_tall = 0;
  smj_tall_caps(&tall.nvp, &tall.nbp, &tall.nent, &tall.nefc, &tall.ncon, &dbg_tall);
  smj_mid_caps(&caps[1].nvp, &caps[1].nbp, &caps[1].nent, &caps[1].nefc, &caps[1].ncon, &dbg[1]);
  smj_big38_caps(&caps[2].nvp, &caps[2].nbp, &caps[2].nent, &caps[2].nefc, &caps[2].ncon, &dbg[2], &caps[2].nvs);
  smj_big50_caps(&caps[3].nvp, &caps[3].nbp, &caps[3].nent, &caps[3].nefc, &caps[3].ncon, &dbg[3], &caps[3].nvs);
  smj_big_caps(&caps[4].nvp, &caps[4].nbp, &caps[4].nent, &caps[4].nefc, &caps[4].ncon, &dbg[4], &caps[4].nvs);
  smj_sat_caps(&caps[5].nvp, &caps[5].nbp, &caps[5].nent, &caps[5].nefc, &caps[5].ncon, &dbg[5], &caps[5].nsat);
  smj_sat32_caps(&caps[6].nvp, &caps[6].nbp, &caps[6].nent, &caps[6].nefc, &caps[6].ncon, &dbg[6], &caps[6].nsat);
  int rc = smj_load_model(blob, nbytes, c->model, up, c->err, caps, 7, &c->variant);
  if (rc) return rc;
  c->caps = caps[c->variant];
  c->layout = smj_stage_layout(c->caps.nvp, c->caps.nbp + c->caps.nsat, c->caps.nsat);
  c->debug_floats = dbg[c->variant];
  if (c->variant <= 3 || c->variant == 5) {
    // escalation target: the same model loaded for the 160-row tall build (standard, 128-row tall) / the 64-column big build (38 / 50 columns) / the 32-satellite build
    int dummy = 0;
    rc = smj_load_model(blob, nbytes, c->model_esc, up, c->err, c->variant <= 1 ? &tall : c->variant == 5 ? caps + 6 : caps + 4, 1, &dummy);
    if (rc) return rc;
    c->has_esc = true;
    void* d = nullptr;
    c->redo_cap = 16 * (size_t)num_envs;   // a 50-step call is 10 chunks per env; longer calls re-allocate (smj_step)
    HIPCHK(c, hipMalloc(&d, sizeof(int) * c->redo_cap));
    c->redo = (int*)d;   // freed in smj_destroy (it can be re-allocated by smj_step)
    HIPCHK(c, hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  DevModel& m = c->model;
  c->qpos0_dev = const_cast<float*>(m.qpos0);
  c->state.B = num_envs;
  c->state.ld = num_envs;
  {
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(int) * 2 * (size_t)num_envs));
    c->allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, sizeof(int) * 2 * (size_t)num_envs));
    c->cost = (int*)d;
    c->order = c->cost + num_envs;
  }
  {
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(int) * (2 * (size_t)num_envs + SMJ_SCHED_WORDS + 1)));
    c->allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, sizeof(int) * (2 * (size_t)num_envs + SMJ_SCHED_WORDS + 1)));
    c->progress = (int*)d;   // + the `hot` word behind sched, which the per-launch memset leaves alone
  }
  {
    void* d = nullptr;
    const size_t bytes = sizeof(float) * (size_t)c->layout.stride * (size_t)num_envs;
    HIPCHK(c, hipMalloc(&d, bytes));
    c->allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, bytes));
    c->stage = (float*)d;
  }
  {   // separating directions of convex pairs, kept between steps (DevState::sepcache)
    void* d = nullptr;
    const size_t bytes = sizeof(float) * 4 * SMJ_SEP_SLOTS * (size_t)num_envs;
    HIPCHK(c, hipMalloc(&d, bytes));
    c->allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, bytes));
    c->state.sepcache = (float*)d;
  }
  {   // contact manifolds of convex pairs, kept between steps (DevState::mcache)
    void* d = nullptr;
    const size_t bytes = sizeof(float) * SMJ_MC_SLOTS * SMJ_MC_WORDS * (size_t)num_envs;
    HIPCHK(c, hipMalloc(&d, bytes));
    c->allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, bytes));
    c->state.mcache = (float*)d;
  }
  {   // PGS: the previous step's rows and forces (DevState::pgsprev), 2.6 KB per env
    void* d = nullptr;
    const size_t bytes = sizeof(float) * SMJ_PGSPREV_STRIDE * (size_t)num_envs;
    HIPCHK(c, hipMalloc(&d, bytes));
    c->allocs.push_back(d);
    HIPCHK(c, hipMemset(d, 0, bytes));
    c->state.pgsprev = (float*)d;
  }
  return setup_render(c, blob, nbytes);
}

int smj_comm_init(smj_ctx* c, int rank, int world, const char* id_path, double timeout_s) {
  if (!c) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  return smj_comm_open(c->comm, rank, world, id_path, timeout_s, c->err);   // binding, id-file rendezvous, ncclCommInitRank: smj_comm.h
}

int smj_allgather_returns(smj_ctx* c, const float* send_dev, float* recv_dev, int count, void* stream) {
  if (!c) return -1;
  if (!send_dev || !recv_dev || count <= 0) return fail(c, -1, "bad buffers / count");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->comm.comm) {   // single-GPU job: the gather is a copy
    if (send_dev != recv_dev)
      HIPCHK(c, hipMemcpyAsync(recv_dev, send_dev, sizeof(float) * (size_t)count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
  }
  return smj_comm_allgather(c->comm, send_dev, recv_dev, count, stream, c->err);
}

int smj_comm_destroy(smj_ctx* c) {
  if (!c) return -1;
  smj_comm_close(c->comm);
  return 0;
}

int smj_destroy(smj_ctx* c) {
  if (!c) return -1;
  smj_comm_destroy(c);
  if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->redo) (void)hipFree(c->redo);
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
  return 0;
}

int smj_dims(const smj_ctx* c, int* out) {
  if (!c || !out) return -1;
  out[SMJ_DIM_NQ] = c->model.nq_all; out[SMJ_DIM_NV] = c->model.nv_all; out[SMJ_DIM_NU] = c->model.nu;
  out[SMJ_DIM_NSAT_MAX] = c->caps.nsat;
  out[SMJ_DIM_NBODY] = c->model.nbody_all; out[SMJ_DIM_NLIDAR] = c->model.nlidar; out[SMJ_DIM_NKEY] = c->model.nkey;
  out[SMJ_DIM_NUM_ENVS] = c->num_envs; out[SMJ_DIM_DEBUG_FLOATS] = c->debug_floats; out[SMJ_DIM_NEFC_MAX] = c->caps.nefc;
  out[SMJ_DIM_NCON_MAX] = c->caps.ncon; out[SMJ_DIM_NV_MAX] = c->caps.nvp; out[SMJ_DIM_NCAM] = c->has_render ? c->render.ncam : 0;
  return 0;
}

int smj_bind(smj_ctx* c, int slot, void* p, long ld) {
  if (!c) return -1;
  if (slot < 0 || slot >= SMJ_SLOT_COUNT) return fail(c, -1, "bad slot %d", slot);
  if (p && ld < c->num_envs) return fail(c, -1, "slot %d: ld %ld < num_envs %d", slot, ld, c->num_envs);
  c->slot_ptr[slot] = p;
  c->slot_ld[slot] = ld;
  DevState& s = c->state;
  switch (slot) {
    case SMJ_SLOT_QPOS: s.qpos = (float*)p; break;
    case SMJ_SLOT_QVEL: s.qvel = (float*)p; break;
    case SMJ_SLOT_CTRL: s.ctrl = (float*)p; break;
    case SMJ_SLOT_WARMSTART: s.warm = (float*)p; break;
    case SMJ_SLOT_NSTEP: s.nstep = (int*)p; break;
    case SMJ_SLOT_ACT_LENGTH: s.act_len = (float*)p; break;
    case SMJ_SLOT_ACT_VELOCITY: s.act_vel = (float*)p; break;
    case SMJ_SLOT_BASE_POSE: s.base = (float*)p; break;
    case SMJ_SLOT_GYRO: s.gyro = (float*)p; break;
    case SMJ_SLOT_ACCEL: s.accel = (float*)p; break;
    case SMJ_SLOT_LIDAR: s.lidar = (float*)p; break;
    case SMJ_SLOT_INFO: s.info = (int*)p; break;
    case SMJ_SLOT_DEBUG: s.debug = (float*)p; break;
    case SMJ_SLOT_PROF: s.prof = (float*)p; break;
    case SMJ_SLOT_XPOSE: s.xpose = (float*)p; break;
    case SMJ_SLOT_BASECTL: s.bctl = (float*)p; break;
  }
  return 0;
}

static int check_bound(smj_ctx* c) {
  static const int need[] = {SMJ_SLOT_QPOS, SMJ_SLOT_QVEL, SMJ_SLOT_CTRL, SMJ_SLOT_WARMSTART, SMJ_SLOT_NSTEP,
                             SMJ_SLOT_ACT_LENGTH, SMJ_SLOT_ACT_VELOCITY, SMJ_SLOT_BASE_POSE, SMJ_SLOT_INFO};
  long ld = -1;
  for (int s : need) {
    if (!c->slot_ptr[s]) return fail(c, -5, "slot %d is not bound", s);
    if (s == SMJ_SLOT_NSTEP) continue;
    if (ld < 0) ld = c->slot_ld[s];
    if (c->slot_ld[s] != ld) return fail(c, -5, "all batch-major slots must share one leading dimension");
  }
  for (int s : {SMJ_SLOT_GYRO, SMJ_SLOT_ACCEL, SMJ_SLOT_LIDAR, SMJ_SLOT_DEBUG, SMJ_SLOT_PROF, SMJ_SLOT_XPOSE, SMJ_SLOT_BASECTL})
    if (c->slot_ptr[s] && c->slot_ld[s] != ld) return fail(c, -5, "slot %d: leading dimension differs", s);
  c->state.ld = ld;
  return 0;
}

int smj_reset(smj_ctx* c, const uint8_t* mask_dev, void* stream) {
  if (!c) return -1;
  int rc = check_bound(c);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  smj_launch_reset(c->model, c->state, mask_dev, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int smj_step(smj_ctx* c, int nsteps, unsigned read_flags, void* stream) {
  if (!c) return -1;
  if (nsteps <= 0) return fail(c, -1, "nsteps must be positive");
  int rc = check_bound(c);
  if (rc) return rc;
  if ((read_flags & SMJ_READ_IMU) && (!c->state.gyro || !c->state.accel)) return fail(c, -5, "IMU readout requested but GYRO/ACCEL not bound");
  if ((read_flags & SMJ_READ_LIDAR) && !c->state.lidar) return fail(c, -5, "lidar readout requested but LIDAR not bound");
  if ((read_flags & SMJ_READ_POSES) && !c->state.xpose) return fail(c, -5, "pose readout requested but XPOSE not bound");
  HIPCHK(c, hipSetDevice(c->device));
  DevState st = c->state;
  long pose_ld = c->slot_ld[SMJ_SLOT_XPOSE];
  if (read_flags & SMJ_READ_LIDAR) {
    // the lidar is ray-cast by its own kernel from the body poses of the last step
    if (!c->has_render) return fail(c, -6, "the model blob carries no ray-casting tables (k_lgeom / rmesh_*): no lidar");
    if (!st.xpose) {
      if (!c->pose_ws) {
        void* d = nullptr;
        HIPCHK(c, hipMalloc(&d, sizeof(float) * 12 * (size_t)c->model.nbody_all * (size_t)c->num_envs));
        c->allocs.push_back(d);
        c->pose_ws = (float*)d;
      }
      st.xpose = c->pose_ws;
      pose_ld = c->num_envs;
      // the step kernel indexes every batch-major array with one leading dimension
      if (st.ld != pose_ld) return fail(c, -5, "bind SMJ_SLOT_XPOSE when the leading dimension differs from num_envs");
    }
    read_flags |= SMJ_READ_POSES;
  }
  // batch-major slots -> env-major staging rows, the step kernel on contiguous rows, and back (smj_model.h, DevState::stage)
  const DevModel& m = c->model;
  StagePlan in, out;
  const SmjStageLayout& Y = c->layout;
  in.add(st.qpos, m.nq_all, Y.qpos); in.add(st.qvel, m.nv_all, Y.qvel); in.add(st.warm, m.nv_all, Y.warm);
  in.add(st.ctrl, m.nu, Y.ctrl); in.add(st.bctl, SMJ_BC_ROWS, Y.bctl); in.add(st.nstep, 1, Y.nstep);
  in.add(st.info, 4, Y.info);
  out = in;
  out.add(st.act_len, m.nu, Y.actlen); out.add(st.act_vel, m.nu, Y.actvel); out.add(st.base, 3, Y.base);
  if (read_flags & SMJ_READ_IMU) { out.add(st.gyro, 3, Y.gyro); out.add(st.accel, 3, Y.accel); }
  if (read_flags & SMJ_READ_POSES) out.add(st.xpose, 12 * m.nbody_all, Y.xpose);
  st.stage = c->stage;
  st.lay = Y;
  const bool esc = c->has_esc && c->escalate;   // standard -> tall, big38 / big50 -> big (has_esc: smj_create)
  st.redo = esc ? c->redo : nullptr;
  st.cost = c->cost;
  if (esc) {   // the escalation target runs with the same options
    DevModel& mb = c->model_esc;
    const DevModel& ms = c->model;
    mb.iterations = ms.iterations; mb.warmstart = ms.warmstart; mb.pgs_fixed_iter = ms.pgs_fixed_iter; mb.qcqp_exact = ms.qcqp_exact; mb.grad_noise = ms.grad_noise; mb.pgs_island_stop = ms.pgs_island_stop; mb.max_con_pair = ms.max_con_pair;
    mb.solver = ms.solver; mb.ls_iterations = ms.ls_iterations; mb.convex_pairs = ms.convex_pairs; mb.multiccd = ms.multiccd; mb.multi_serial = ms.multi_serial; mb.sep_cache = ms.sep_cache; mb.manifold_cache = ms.manifold_cache; mb.pgs_dual_ws = ms.pgs_dual_ws;
    mb.ls_tolerance = ms.ls_tolerance; mb.tolerance = ms.tolerance;
  }
  smj_launch_stage(in, c->stage, Y.stride, c->num_envs, st.ld, false, (hipStream_t)stream);
  // Option `chunk` (default 0 = off; kept for experiments): the n steps go out as separate dispatches of `chunk` steps on the
  // staged rows, each with a fresh longest-first order.  Measured: no gain -- a barrier per chunk keeps the tail's share of a
  // dispatch what it was; the pipelined chunks below have no barrier.  Readout flags go with the last chunk only.
  const int chunk = (c->chunk > 0 && !st.debug && !st.prof && c->num_envs > 1024) ? c->chunk : nsteps;
  // Pipelined chunks (standard variant, batches that need more than one round of workgroups): see DevState::pipe_len
  hipStream_t sm = (hipStream_t)stream;
  // measured: standard +19 % at chunks of 5, tall (2 envs per CU) +7 % at 10, big with 64 columns (1 env per CU, 16 rounds of workgroups) nothing
  // (three envs per CU: chunks of 8 measured 3 % ahead of 10)
  const int pipe_len = c->variant >= 2 ? 2 * c->pipeline : c->variant == 1 ? (3 * c->pipeline + 1) / 2 : c->pipeline;
  const bool pipe = c->variant != 4 && c->variant != 6 && (c->variant < 2 || c->pipeline_big) && c->pipeline > 0 && chunk == nsteps && nsteps > pipe_len && !st.debug && !st.prof && c->num_envs > c->pipe_min_envs;
  st.progress = st.done_steps = st.sched = st.hot = nullptr;
  if (esc || pipe) {
    st.progress = c->progress;
    st.done_steps = c->progress + c->num_envs;
    st.sched = c->progress + 2 * (size_t)c->num_envs;
    st.hot = st.sched + SMJ_SCHED_WORDS;
  }
  st.pipe_len = 0;
  st.pollers = 0;
  // the 32-satellite build exists twice: both solvers on one wavefront per env, and Newton only on two (smj_kernels_sat32n.hip)
  typedef int (*Launch32)(const DevModel&, const DevState&, int, unsigned, hipStream_t);
  auto sat32_of = [&](const DevModel& m) -> Launch32 { return (m.solver == 2 && (c->newton_two_waves & 2) && !st.prof) ? smj_launch_step_sat32n : smj_launch_step_sat32; };
  int lrc = 0;
  for (int done = 0; done < nsteps && !lrc; done += chunk) {
    const int k = nsteps - done < chunk ? nsteps - done : chunk;
    const unsigned fl = done + k >= nsteps ? read_flags : 0u;
    st.redo_worker = 0;
    st.order = nullptr;
    st.pipe_len = pipe ? pipe_len : 0;
    st.pipe_total = c->num_envs * (pipe ? (k + pipe_len - 1) / pipe_len : 1);
    if (esc || pipe) HIPCHK(c, hipMemsetAsync(c->progress, 0, sizeof(int) * (2 * (size_t)c->num_envs + SMJ_SCHED_WORDS), sm));
    if (esc) {
      if ((size_t)st.pipe_total > c->redo_cap) {   // every workgroup of the standard launch can park its env once
        HIPCHK(c, hipDeviceSynchronize());
        (void)hipFree(c->redo);
        c->redo = nullptr;
        c->redo_cap = (size_t)st.pipe_total;
        void* d = nullptr;
        HIPCHK(c, hipMalloc(&d, sizeof(int) * c->redo_cap));
        c->redo = (int*)d;
        st.redo = c->redo;
      }
      // -1 = entry not published yet: only the pollers read entries while the list grows (the sweep runs after the kernel, the count is final)
      if (pipe && c->pollers > 0) HIPCHK(c, hipMemsetAsync(c->redo, 0xff, sizeof(int) * (size_t)st.pipe_total, sm));
    }
    if (c->balance && k >= c->balance_min && c->num_envs > c->balance_min_envs) {
      smj_launch_order(c->cost, c->order, c->num_envs, sm);
      st.order = c->order;
    }
    // A second kernel resident beside the standard one costs a launch in which nobody escalates 5 % (measured: one poller or
    // sixteen, any poll interval, any stream priority), and escalations are rare events (0-2 envs per 50-step launch of 4096
    // under random actions, each worth milliseconds when left to the sweep): the pollers leave at once unless one of the last
    // SMJ_HOT_LAUNCHES launches had an escalation (DevState::hot, kept on the device -- the host runs many launches ahead)
    const bool poll = esc && pipe && c->pollers > 0 && (c->variant == 0 || c->variant == 5);   // (the 16-satellite build hands over to the 32-satellite one the same way)
    if (poll) {
      // pollers first, on their own stream, so that they are resident when the standard kernel fills the device; should they
      // not be (nothing guarantees it), parked envs are given up to the sweep, as without pollers
      DevState sp = st;
      sp.redo_worker = 2;
      const int np = c->variant == 5 ? 6 * c->pollers : c->pollers;   // a kitchen's random-action workload parks ~10 envs per launch, each chunk of the large build takes milliseconds
      sp.pollers = c->pollers_always ? -np : np;
      sp.order = nullptr;
      HIPCHK(c, hipEventRecord(c->ev_fork, sm));
      HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
      lrc = c->variant == 5 ? sat32_of(c->model_esc)(c->model_esc, sp, k, fl, c->aux) : smj_launch_step_tall(c->model_esc, sp, k, fl, c->aux);
      HIPCHK(c, hipEventRecord(c->ev_join, c->aux));
    }
    // The primary builds exist once per solver (smj_step_impl.h newton()): the base name carries Newton, the twin PGS.  In a tools build
    // (csrc/Makefile bigprof) the base name is the profiling copy with BOTH solvers: a PGS launch with the profiling slot bound tries it
    // first and falls back to the twin when the base build refuses the solver (its launcher returns before launching anything).
    typedef int (*Launch)(const DevModel&, const DevState&, int, unsigned, hipStream_t);
    bool prof_dropped = false;
    auto by_solver = [&](Launch base, Launch twin) -> int {
      if (c->model.solver == 2) return base(c->model, st, k, fl, sm);
      if (st.prof) {
        const int r = base(c->model, st, k, fl, sm);
        if (r != SMJ_LAUNCH_REFUSED_SOLVER) return r;
        prof_dropped = true;   // a product build: its base kernel carries Newton only, the PGS twin has no cycle counters compiled in
      }
      return twin(c->model, st, k, fl, sm);
    };
    if (!lrc)
      lrc = c->variant == 6   ? sat32_of(c->model)(c->model, st, k, fl, sm)
            : c->variant == 5 ? by_solver((c->model.solver == 2 && (c->newton_two_waves & 1) && (!st.prof || smj_sat2_profiling())) ? smj_launch_step_sat2 : smj_launch_step_sat, c->pgs_two_waves ? smj_launch_step_satp : smj_launch_step_sat1)
            : c->variant == 4 ? smj_launch_step_big(c->model, st, k, fl, sm)
            : c->variant == 3 ? by_solver(smj_launch_step_big50, smj_launch_step_big50p)
            : c->variant == 2 ? by_solver(smj_launch_step_big38, smj_launch_step_big38p)
            : c->variant == 1 ? by_solver(smj_launch_step_mid, smj_launch_step_midp)
            : st.prof         ? smj_launch_step_prof(c->model, st, k, fl, sm)
                              : by_solver(smj_launch_step, smj_launch_step_pgs);
    if (lrc == SMJ_LAUNCH_REFUSED_SOLVER) return fail(c, -2, "no kernel of this build carries solver %d for variant %d", c->model.solver, c->variant);
    if (prof_dropped && !c->prof_warned) {   // (once: the caller asked for stage cycles and gets zeros -- say so instead of returning them silently)
      c->prof_warned = true;
      fprintf(stderr, "libsmj: the profiling slot is bound but the PGS kernel of this build has no cycle counters (build `make bigprof` and load it through SMJ_LIB_PATH); the counters stay zero\n");
    }
    if (poll) HIPCHK(c, hipStreamWaitEvent(sm, c->ev_join, 0));
    if (!lrc && esc) {
      // the sweep: whatever is left of the envs that ran out of constraint rows / contact slots (parked at the start of the
      // offending step) is finished by the tall variant (160 rows / 48 contacts); an empty list returns at once
      st.redo_worker = 1;
      st.pipe_len = 0;
      lrc = c->variant <= 1 ? smj_launch_step_tall(c->model_esc, st, k, fl, sm) : c->variant == 5 ? sat32_of(c->model_esc)(c->model_esc, st, k, fl, sm) : smj_launch_step_big(c->model_esc, st, k, fl, sm);
    }
  }
  if (lrc) return fail(c, -2, "step kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
  smj_launch_stage(out, c->stage, Y.stride, c->num_envs, st.ld, true, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  if (read_flags & SMJ_READ_LIDAR) {
    DevRender rl = c->render;
    if (!c->lidar_cull) rl.lidar_plane_body = -1;   // option "lidar_cull" = 0: every run-time geom staged for every env (the ranges must not change)
    smj_launch_lidar(rl, st.xpose, pose_ld, c->num_envs, st.lidar, st.ld, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}

int smj_base_controller_tick(smj_ctx* c, void* stream) {
  if (!c) return -1;
  int rc = check_bound(c);
  if (rc) return rc;
  if (!c->state.bctl) return fail(c, -5, "BASECTL slot is not bound");
  HIPCHK(c, hipSetDevice(c->device));
  smj_launch_base_tick(c->state, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int smj_render_depth(smj_ctx* c, int cam, int width, int height, float fovy_deg, float max_depth, void* out_dev, void* stream) {
  if (!c) return -1;
  if (!c->has_render) return fail(c, -6, "the model blob carries no render tables (k_rgeom / rmesh_*)");
  if (cam < 0 || cam >= c->render.ncam) return fail(c, -1, "camera id %d out of range (ncam %d)", cam, c->render.ncam);
  if (width <= 0 || height <= 0 || !(fovy_deg > 0.f && fovy_deg < 180.f)) return fail(c, -1, "bad image size / field of view");
  if (!out_dev) return fail(c, -1, "null output image");
  if (!c->state.xpose) return fail(c, -5, "XPOSE slot is not bound (step with SMJ_READ_POSES first)");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->depth_ws) {
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, smj_depth_workspace_bytes(c->num_envs)));
    c->allocs.push_back(d);
    c->depth_ws = (float*)d;
  }
  smj_ctx::Layer* L = nullptr;
  for (auto& l : c->layers)
    if (l.cam == cam && l.w == width && l.h == height && l.fovy == fovy_deg) L = &l;
  if (!L) {
    smj_ctx::Layer l;
    l.cam = cam; l.w = width; l.h = height; l.fovy = fovy_deg;
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(float) * (size_t)width * (size_t)height));
    c->allocs.push_back(d);
    l.buf = (float*)d;
    smj_launch_depth(c->render, c->state.xpose, c->slot_ld[SMJ_SLOT_XPOSE], c->num_envs, cam, width, height, fovy_deg, 0.f, l.buf,
                     nullptr, 1, c->depth_ws, (hipStream_t)stream);
    HIPCHK(c, hipGetLastError());
    c->layers.push_back(l);
    L = &c->layers.back();
  }
  smj_launch_depth(c->render, c->state.xpose, c->slot_ld[SMJ_SLOT_XPOSE], c->num_envs, cam, width, height, fovy_deg, max_depth,
                   (float*)out_dev, L->buf, 2, c->depth_ws, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int smj_render_rgb(smj_ctx* c, int cam, int width, int height, float fovy_deg, void* rgb_dev, void* gid_dev, void* stream) {
  if (!c) return -1;
  if (!c->has_render) return fail(c, -6, "the model blob carries no render tables (k_rgeom / rmesh_*)");
  if (!c->render.geom_rgba) return fail(c, -6, "the model blob carries no geom colours (geom_rgba)");
  if (cam < 0 || cam >= c->render.ncam) return fail(c, -1, "camera id %d out of range (ncam %d)", cam, c->render.ncam);
  if (width <= 0 || height <= 0 || !(fovy_deg > 0.f && fovy_deg < 180.f)) return fail(c, -1, "bad image size / field of view");
  if (!rgb_dev) return fail(c, -1, "null output image");
  if (!c->state.xpose) return fail(c, -5, "XPOSE slot is not bound (step with SMJ_READ_POSES first)");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->depth_ws) {
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, smj_depth_workspace_bytes(c->num_envs)));
    c->allocs.push_back(d);
    c->depth_ws = (float*)d;
  }
  smj_launch_rgb(c->render, c->state.xpose, c->slot_ld[SMJ_SLOT_XPOSE], c->num_envs, cam, width, height, fovy_deg,
                 (unsigned char*)rgb_dev, (int*)gid_dev, c->depth_ws, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int smj_set_option(smj_ctx* c, const char* name, double v) {
  if (!c || !name) return -1;
  DevModel& m = c->model;
  if (!strcmp(name, "iterations")) m.iterations = (int)v;
  else if (!strcmp(name, "tolerance")) m.tolerance = (float)v;
  else if (!strcmp(name, "warmstart")) m.warmstart = (int)v;
  else if (!strcmp(name, "pgs_fixed_iter")) m.pgs_fixed_iter = (int)v;
  else if (!strcmp(name, "qcqp_exact")) m.qcqp_exact = (int)v;
  else if (!strcmp(name, "grad_noise")) m.grad_noise = (float)v;
  else if (!strcmp(name, "pgs_island_stop")) m.pgs_island_stop = (int)v;
  else if (!strcmp(name, "max_contacts_per_pair")) m.max_con_pair = (int)v;
  else if (!strcmp(name, "solver")) m.solver = (int)v;
  else if (!strcmp(name, "convex_pairs")) m.convex_pairs = (int)v;
  else if (!strcmp(name, "multiccd")) m.multiccd = (int)v;
  else if (!strcmp(name, "sep_cache")) m.sep_cache = (int)v;
  else if (!strcmp(name, "manifold_cache")) m.manifold_cache = (int)v;
  else if (!strcmp(name, "pgs_dual_warmstart")) m.pgs_dual_ws = (int)v;
  else if (!strcmp(name, "lidar_cull")) c->lidar_cull = (int)v;                  // 1 (default): per env, only the geoms whose bounding sphere reaches the rangefinders' scan plane are staged
  else if (!strcmp(name, "depth_raster")) c->render.raster = (int)v;            // 0: ray cast the meshes through their BVHs (the round-2 path)
  else if (!strcmp(name, "depth_raster_splits")) c->render.raster_splits = (int)(v < 1 ? 1 : v > 256 ? 256 : v);
  else if (!strcmp(name, "primary_rows")) m.row_limit = (int)v;   // the escalation variant keeps its full capacity (model_esc is not touched)
  else if (!strcmp(name, "escalate")) c->escalate = (int)v;
  else if (!strcmp(name, "balance")) c->balance = (int)v;
  else if (!strcmp(name, "balance_min")) c->balance_min = (int)v;
  else if (!strcmp(name, "pgs_two_waves")) c->pgs_two_waves = (int)v;
  else if (!strcmp(name, "newton_two_waves")) c->newton_two_waves = (int)v == 1 ? 3 : (int)v;   // 1 (or 3): both satellite builds; 0: neither; tools: 5 = the 16-satellite build only, 2 = the 32-satellite one only
  else if (!strcmp(name, "chunk")) c->chunk = (int)v;
  else if (!strcmp(name, "pipeline")) c->pipeline = (int)v;
  else if (!strcmp(name, "pipeline_big")) c->pipeline_big = (int)v;
  else if (!strcmp(name, "pipeline_min_envs")) c->pipe_min_envs = (int)v;
  else if (!strcmp(name, "balance_min_envs")) c->balance_min_envs = (int)v;
  else if (!strcmp(name, "pollers")) {   // n > 0: n pollers when a recent launch escalated; -n: n pollers with every launch; 0: none
    c->pollers_always = v < 0;
    const int n = (int)(v < 0 ? -v : v);
    c->pollers = n > 256 ? 256 : n;
  }
  else return fail(c, -1, "unknown option '%s'", name);
  return 0;
}

}  // extern "C"
