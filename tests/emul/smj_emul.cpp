// TEST INFRASTRUCTURE ONLY -- lane emulator of the HIP step kernel.
//
// Compiles stretch_mujoco_amd/csrc/smj_step_impl.h with SMJ_EMUL (smj_wave.h): every lane region becomes a
// loop over 64 lanes and cross-lane ops act on arrays, in fp32, with the same operation order as the GPU
// code.  It exists so that the kernel LOGIC can be checked against the fp64 oracle on a machine without a
// GPU (`pytest -m "not gpu"`).  It is never linked into libsmj.so and nothing in stretch_mujoco_amd/ loads it.
#define SMJ_EMUL 1
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../stretch_mujoco_amd/csrc/smj_model_load.h"
#include "../../stretch_mujoco_amd/csrc/smj_step_impl.h"

struct HostUploader {
  std::vector<void*>* keep;
  template <class T>
  const T* put(const std::vector<T>& h) {
    T* p = (T*)malloc(h.size() * sizeof(T));
    memcpy(p, h.data(), h.size() * sizeof(T));
    keep->push_back(p);
    return p;
  }
  const float* f32(const std::vector<float>& h) { return put(h); }
  const int* i32(const std::vector<int>& h) { return put(h); }
};

struct emul_ctx {
  DevModel m{};
  DevState s{};
  std::vector<void*> keep;
  std::string err;
  struct { Smem s; unsigned char pgs_tail[sizeof(float) * (NEFC * (NEFC + 1) / 2 + NEFP * NEFP)]; } lds;   // PGS: A runs past the end of Smem
};

extern "C" {

emul_ctx* emul_create(const void* blob, size_t nbytes, int num_envs) {
  emul_ctx* c = new emul_ctx();
  HostUploader up{&c->keep};
  const SmjCaps caps{NVP, NBP, NENT, NEFC, NCON, NVS, NSAT};   // this build's variant (Makefile: -DSMJ_BIG for libsmj_emul_big.so)
  int chosen = 0;
  if (smj_load_model(blob, nbytes, c->m, up, c->err, &caps, 1, &chosen)) {
    fprintf(stderr, "emul_create: %s\n", c->err.c_str());
    delete c;
    return nullptr;
  }
  c->s.B = num_envs;
  c->s.ld = num_envs;
  c->s.lay = smj_stage_layout(NVP, NBT, NSAT);
  c->s.sepcache = (float*)calloc((size_t)num_envs * SMJ_SEP_SLOTS * 4, sizeof(float));
  c->keep.push_back(c->s.sepcache);
  c->s.mcache = (float*)calloc((size_t)num_envs * SMJ_MC_SLOTS * SMJ_MC_WORDS, sizeof(float));
  c->keep.push_back(c->s.mcache);
  c->s.pgsprev = (float*)calloc((size_t)num_envs * SMJ_PGSPREV_STRIDE, sizeof(float));
  c->keep.push_back(c->s.pgsprev);
  return c;
}
// what smj_reset drops of the library's own memory (smj_kernels.hip smj_reset_kernel): kept manifolds, separating directions, PGS second start
void emul_clear_caches(emul_ctx* c) {
  const size_t B = (size_t)c->s.B;
  memset(c->s.sepcache, 0, B * SMJ_SEP_SLOTS * 4 * sizeof(float));
  memset(c->s.mcache, 0, B * SMJ_MC_SLOTS * SMJ_MC_WORDS * sizeof(float));
  memset(c->s.pgsprev, 0, B * SMJ_PGSPREV_STRIDE * sizeof(float));
}
void emul_destroy(emul_ctx* c) {
  for (void* p : c->keep) free(p);
  delete c;
}
// same slot numbering as include/smj.h
int emul_bind(emul_ctx* c, int slot, void* p, long ld) {
  DevState& s = c->s;
  s.ld = ld;
  switch (slot) {
    case 0: s.qpos = (float*)p; break;
    case 1: s.qvel = (float*)p; break;
    case 2: s.ctrl = (float*)p; break;
    case 3: s.warm = (float*)p; break;
    case 4: s.nstep = (int*)p; break;
    case 5: s.act_len = (float*)p; break;
    case 6: s.act_vel = (float*)p; break;
    case 7: s.base = (float*)p; break;
    case 8: s.gyro = (float*)p; break;
    case 9: s.accel = (float*)p; break;
    case 10: s.lidar = (float*)p; break;
    case 11: s.info = (int*)p; break;
    case 12: s.debug = (float*)p; break;
    case 13: s.prof = (float*)p; break;
    case 14: s.xpose = (float*)p; break;
    case 15: s.bctl = (float*)p; break;
    default: return -1;
  }
  return 0;
}
int emul_set_option(emul_ctx* c, const char* name, double v) {
  DevModel& m = c->m;
  if (!strcmp(name, "iterations")) m.iterations = (int)v;
  else if (!strcmp(name, "tolerance")) m.tolerance = (float)v;
  else if (!strcmp(name, "warmstart")) m.warmstart = (int)v;
  else if (!strcmp(name, "pgs_fixed_iter")) m.pgs_fixed_iter = (int)v;
  else if (!strcmp(name, "qcqp_exact")) m.qcqp_exact = (int)v;
  else if (!strcmp(name, "grad_noise")) m.grad_noise = (float)v;
  else if (!strcmp(name, "pgs_island_stop")) m.pgs_island_stop = (int)v;
  else if (!strcmp(name, "pgs_cap")) m.pgs_cap = (int)v;   // what smj_step sets for a PGS launch without dynamic LDS (smj_step_tu.h)
  else if (!strcmp(name, "max_contacts_per_pair")) m.max_con_pair = (int)v;
  else if (!strcmp(name, "solver")) m.solver = (int)v;
  else if (!strcmp(name, "convex_pairs")) m.convex_pairs = (int)v;
  else if (!strcmp(name, "multiccd")) m.multiccd = (int)v;
  else if (!strcmp(name, "multi_serial")) m.multi_serial = (int)v;
  else if (!strcmp(name, "sep_cache")) m.sep_cache = (int)v;
  else if (!strcmp(name, "manifold_cache")) m.manifold_cache = (int)v;
  else if (!strcmp(name, "pgs_dual_warmstart")) m.pgs_dual_ws = (int)v;
  else return -1;
  return 0;
}
// LDS is uninitialised when a workgroup starts: tests poison the emulated LDS to catch reads before writes
long emul_sep_skips() { return smj_emul_sep_skips; }
long emul_ext_steps() { return smj_emul_ext_steps; }
long emul_mc_hits() { return smj_emul_mc_hits; }
long emul_isl_total() { return smj_emul_isl_total; }
long emul_isl_swept() { return smj_emul_isl_swept; }
int emul_poison = -1;
void emul_set_poison(int byte) { emul_poison = byte; }
int emul_step(emul_ctx* c, int nsteps, unsigned read_flags) {
  for (int env = 0; env < c->s.B; env++) {
    if (emul_poison >= 0) memset(&c->lds, emul_poison, sizeof(c->lds));
    StepKernel* k = new StepKernel(c->m, c->s, c->lds.s, env);
    k->run(nsteps, read_flags);
    delete k;
  }
  return 0;
}
int emul_debug_floats() { return SMJ_DEBUG_FLOATS; }
int emul_nvp() { return NVP; }
int emul_lds_bytes() { return (int)sizeof(Smem); }
int emul_ncon_max() { return NCON; }
int emul_nefc_max() { return NEFC; }
}
