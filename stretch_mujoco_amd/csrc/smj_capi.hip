// C-ABI (include/smj.h) of the batched Stretch physics path for gfx950.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/smj.h"
#include "smj_kernels.h"
#include "smj_model_load.h"

struct smj_ctx {
  int device = 0;
  int num_envs = 0;
  DevModel model{};
  DevState state{};
  std::vector<void*> allocs;
  std::string err;
  float* qpos0_dev = nullptr;
  void* slot_ptr[SMJ_SLOT_COUNT] = {};
  long slot_ld[SMJ_SLOT_COUNT] = {};
};

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
  DeviceUploader up{c};
  int rc = smj_load_model(blob, nbytes, c->model, up, c->err);
  if (rc) return rc;
  DevModel& m = c->model;
  c->qpos0_dev = const_cast<float*>(m.qpos0);
  c->state.B = num_envs;
  c->state.ld = num_envs;
  return 0;
}

int smj_destroy(smj_ctx* c) {
  if (!c) return -1;
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
  return 0;
}

int smj_dims(const smj_ctx* c, int* out) {
  if (!c || !out) return -1;
  out[SMJ_DIM_NQ] = c->model.nq; out[SMJ_DIM_NV] = c->model.nv; out[SMJ_DIM_NU] = c->model.nu;
  out[SMJ_DIM_NBODY] = c->model.nbody; out[SMJ_DIM_NLIDAR] = c->model.nlidar; out[SMJ_DIM_NKEY] = c->model.nkey;
  out[SMJ_DIM_NUM_ENVS] = c->num_envs; out[SMJ_DIM_DEBUG_FLOATS] = SMJ_DEBUG_FLOATS; out[SMJ_DIM_NEFC_MAX] = NEFC;
  out[SMJ_DIM_NCON_MAX] = NCON;
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
  for (int s : {SMJ_SLOT_GYRO, SMJ_SLOT_ACCEL, SMJ_SLOT_LIDAR, SMJ_SLOT_DEBUG, SMJ_SLOT_PROF})
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
  HIPCHK(c, hipSetDevice(c->device));
  smj_launch_step(c->model, c->state, nsteps, read_flags, (hipStream_t)stream);
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
  else if (!strcmp(name, "max_contacts_per_pair")) m.max_con_pair = (int)v;
  else if (!strcmp(name, "solver")) m.solver = (int)v;
  else if (!strcmp(name, "convex_pairs")) m.convex_pairs = (int)v;
  else return fail(c, -1, "unknown option '%s'", name);
  return 0;
}

}  // extern "C"
