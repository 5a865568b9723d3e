// SMJB blob -> DevModel.  Shared by the HIP library (uploads to device memory) and by the test-only lane
// emulator (host memory); `Up` supplies the two copy routines.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "smj_model.h"

struct SmjBlobEntry {
  char name[48];
  uint32_t dtype, ndim, shape[4];
  uint64_t offset, nbytes;
};

struct SmjBlob {
  const uint8_t* p;
  size_t n;
  const SmjBlobEntry* find(const char* name) const {
    uint32_t cnt;
    memcpy(&cnt, p + 8, 4);
    const SmjBlobEntry* e = reinterpret_cast<const SmjBlobEntry*>(p + 16);
    for (uint32_t i = 0; i < cnt; i++)
      if (strncmp(e[i].name, name, 48) == 0) return &e[i];
    return nullptr;
  }
};

template <class Up>
int smj_load_model(const void* blob, size_t nbytes, DevModel& m, Up& up, std::string& err) {
  if (!blob || nbytes < 16 || memcmp(blob, "SMJB0001", 8) != 0) { err = "not an SMJB model blob"; return -1; }
  SmjBlob b{static_cast<const uint8_t*>(blob), nbytes};
  auto geti = [&](const char* name, int idx, int* out) -> bool {
    const SmjBlobEntry* e = b.find(name);
    if (!e || e->dtype != 1 || e->nbytes < 4u * (idx + 1)) { err = std::string("model blob: missing int ") + name; return false; }
    memcpy(out, b.p + e->offset + 4 * idx, 4);
    return true;
  };
  auto getf = [&](const char* name, int idx, float* out) -> bool {
    const SmjBlobEntry* e = b.find(name);
    if (!e || e->dtype != 0 || e->nbytes < 8u * (idx + 1)) { err = std::string("model blob: missing scalar ") + name; return false; }
    double v;
    memcpy(&v, b.p + e->offset + 8 * idx, 8);
    *out = (float)v;
    return true;
  };
#define GI(field, name, idx) if (!geti(name, idx, &m.field)) return -3;
#define GF(field, name, idx) if (!getf(name, idx, &m.field)) return -3;
  GI(nq, "dims", 0) GI(nv, "dims", 1) GI(nu, "dims", 2) GI(nbody, "dims", 3) GI(njnt, "dims", 4) GI(ngeom, "dims", 5)
  GI(nsite, "dims", 6) GI(neq, "dims", 8) GI(nkey, "dims", 11) GI(npair, "dims", 12)
  GI(nlevel, "k_nlevel", 0) GI(nfric, "k_nfric", 0) GI(nlimit, "k_nlimit", 0) GI(nplanepair, "k_nplanepair", 0)
  GI(nldl, "k_nldl", 0) GI(imu_site, "sensor_imu_site", 0) GI(ngc, "k_ngc", 0) GI(nroot, "k_nroot", 0)
  GI(njump, "k_njump", 0) GI(maxsubtree, "k_maxsubtree", 0) GI(ncgeom, "k_ncgeom", 0) GI(nconvpair, "k_nconvpair", 0) GI(iterations, "opt_iterations", 0)
  GF(timestep, "opt_timestep", 0) GF(gravity[0], "opt_gravity", 0) GF(gravity[1], "opt_gravity", 1)
  GF(gravity[2], "opt_gravity", 2) GF(impratio, "opt_impratio", 0) GF(tolerance, "opt_tolerance", 0)
  GF(meaninertia, "stat_meaninertia", 0) GF(lidar_cutoff, "sensor_lidar_cutoff", 0)
#undef GI
#undef GF
  {
    const SmjBlobEntry* e = b.find("sensor_lidar_site");
    m.nlidar = e ? (int)(e->nbytes / 4) : 0;
  }
  m.warmstart = 1; m.pgs_fixed_iter = 0; m.max_con_pair = 4; m.solver = 0; m.convex_pairs = 1; m.ls_iterations = 50; m.ls_tolerance = 0.01f;
  char buf[256];
  if (m.nv > NVP || m.nbody > NBP || m.nq > NVP + 8 || m.nu > 16) {
    snprintf(buf, sizeof buf, "model exceeds kernel capacity (nv %d<=%d, nbody %d<=%d, nu %d<=16)", m.nv, NVP, m.nbody, NBP, m.nu);
    err = buf;
    return -4;
  }
  if (m.nldl > 5 * 64) { err = "mass-matrix sparsity pattern too large"; return -4; }
  if (m.neq + m.nfric > NEFC) { err = "too many static constraint rows"; return -4; }
  if (2 * m.nlimit > 64) { err = "too many limited joints"; return -4; }
  if (m.ngc > 16) { err = "more than 16 gravity-compensated bodies"; return -4; }
  if (m.njump > 6) { err = "body tree deeper than 64 levels"; return -4; }
  if (m.ncgeom > NCG) { err = "too many geoms in non-plane collision pairs"; return -4; }
#define X(n)                                                                               \
  {                                                                                        \
    const SmjBlobEntry* e = b.find(#n);                                                    \
    if (!e || e->dtype != 1) { err = "model blob: missing i32 array " #n; return -3; }     \
    std::vector<int> h(e->nbytes / 4 ? e->nbytes / 4 : 1, 0);                              \
    memcpy(h.data(), b.p + e->offset, e->nbytes);                                          \
    m.n = up.i32(h);                                                                       \
    if (!m.n) { err = "device allocation failed for " #n; return -2; }                     \
  }
  SMJ_MODEL_I32(X)
#undef X
#define X(n)                                                                               \
  {                                                                                        \
    const SmjBlobEntry* e = b.find(#n);                                                    \
    if (!e || e->dtype != 0) { err = "model blob: missing f64 array " #n; return -3; }     \
    size_t cnt = e->nbytes / 8;                                                            \
    std::vector<float> h(cnt ? cnt : 1, 0.f);                                              \
    const double* src = reinterpret_cast<const double*>(b.p + e->offset);                  \
    for (size_t i = 0; i < cnt; i++) h[i] = (float)src[i];                                 \
    m.n = up.f32(h);                                                                       \
    if (!m.n) { err = "device allocation failed for " #n; return -2; }                     \
  }
  SMJ_MODEL_F32(X)
#undef X
  return 0;
}
