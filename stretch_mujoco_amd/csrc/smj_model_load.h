// SMJB blob -> DevModel.  Shared by the HIP library (uploads to device memory) and by the test-only lane
// emulator (host memory); `Up` supplies the two copy routines.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>

#include <map>
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
  // the entry table and every entry's payload lie inside the blob (a truncated / corrupt blob is rejected, not read past)
  bool valid() const {
    if (!p || n < 16) return false;
    uint32_t cnt;
    memcpy(&cnt, p + 8, 4);
    if ((size_t)cnt > (n - 16) / sizeof(SmjBlobEntry)) return false;
    const SmjBlobEntry* e = reinterpret_cast<const SmjBlobEntry*>(p + 16);
    for (uint32_t i = 0; i < cnt; i++)
      if (e[i].offset > n || e[i].nbytes > n - e[i].offset) return false;
    return true;
  }
  const SmjBlobEntry* find(const char* name) const {
    uint32_t cnt;
    memcpy(&cnt, p + 8, 4);
    const SmjBlobEntry* e = reinterpret_cast<const SmjBlobEntry*>(p + 16);
    for (uint32_t i = 0; i < cnt; i++)
      if (strncmp(e[i].name, name, 48) == 0) return &e[i];
    return nullptr;
  }
};

// Host restatement of what the kernel's stage-table loaders (smj_step_impl.h: KinTab, BodyTab, DofTab, EntryTab, ActTab)
// used to gather per lane from the individual tables, one record per lane.
// capacities of a kernel variant (smj_model.h): the loader builds its records for the variant that will run
struct SmjCaps { int nvp, nbp, nent, nefc, ncon, nvs, nsat; };   // nvs: dof columns of the variant's matrices (0: nvp); nsat: satellite capacity (0: a build without satellites)
static inline std::vector<int> smj_build_lanerec(const DevModel& m, std::map<std::string, std::vector<int>>& I,
                                                 std::map<std::string, std::vector<float>>& F, int nent) {
  const int LR_ACT = smj_lr_act(nent), LR_STRIDE = smj_lr_stride(nent);
  std::vector<int> rec(64 * LR_STRIDE, 0);
  auto fb = [](float v) { int b; memcpy(&b, &v, 4); return b; };
  auto gi = [&](const char* n, size_t k) { const std::vector<int>& v = I[n]; return k < v.size() ? v[k] : 0; };
  auto gf = [&](const char* n, size_t k) { const std::vector<float>& v = F[n]; return k < v.size() ? v[k] : 0.f; };
  const int nb = m.nbody, nv = m.nv, nu = m.nu;
  for (int L = 0; L < 64; L++) {
    int* r = rec.data() + L * LR_STRIDE;
    {  // KinTab: parent, level, jump[6], pos[3], quat[4], jtype[2], jqadr[2], jdadr[2], jq0[2], jaxis[6], jpos[6]
      int* k = r + SMJ_LR_KIN;
      const int b = L < nb ? L : 0;
      k[0] = gi("body_parentid", b); k[1] = L < nb ? gi("k_body_level", b) : -1;
      for (int q = 0; q < 6; q++) k[2 + q] = (q < m.njump && L > 0 && L < nb) ? gi("k_body_jump", q * nb + b) : 0;
      for (int q = 0; q < 3; q++) k[8 + q] = fb(gf("body_pos", 3 * b + q));
      for (int q = 0; q < 4; q++) k[11 + q] = fb(gf("body_quat", 4 * b + q));
      const int jn = gi("body_jntnum", b), ja = gi("body_jntadr", b);
      for (int u = 0; u < 2; u++) {
        const int jj = (L < nb && u < jn) ? ja + u : -1, jx = jj >= 0 ? jj : 0, qa = gi("jnt_qposadr", jx);
        k[15 + u] = jj >= 0 ? gi("jnt_type", jx) : -1; k[17 + u] = qa; k[19 + u] = gi("jnt_dofadr", jx); k[21 + u] = fb(gf("qpos0", qa));
        for (int q = 0; q < 3; q++) { k[23 + 3 * u + q] = fb(gf("jnt_axis", 3 * jx + q)); k[29 + 3 * u + q] = fb(gf("jnt_pos", 3 * jx + q)); }
      }
    }
    {  // BodyTab: root, subsize, parent, dofadr, dofnum, jump[6], dofmask lo, hi, inertia_local[10]
      int* k = r + SMJ_LR_BODY;
      const int b = L < nb ? L : 0;
      k[0] = gi("body_rootid", b); k[1] = gi("k_body_subtreesize", b); k[2] = gi("body_parentid", b); k[3] = gi("body_dofadr", b);
      k[4] = L < nb ? gi("body_dofnum", b) : 0;
      for (int q = 0; q < 6; q++) k[5 + q] = (q < m.njump && L > 0 && L < nb) ? gi("k_body_jump", q * nb + b) : 0;
      k[11] = gi("k_body_dofmask_lo", b); k[12] = gi("k_body_dofmask_hi", b);
      for (int q = 0; q < 10; q++) k[13 + q] = fb(gf("k_body_inertia_local", 10 * b + q));
    }
    {  // DofTab: body, jtype, qadr, first, bsub, velmask lo, hi, damp, stiff, spring, act[2], actmom[2]
      int* k = r + SMJ_LR_DOF;
      const int d = L < nv ? L : 0, j = gi("dof_jntid", d), db = gi("dof_bodyid", d), jt = gi("jnt_type", j);
      k[0] = db; k[1] = jt; k[2] = gi("k_dof_qposadr", d); k[3] = gi("jnt_dofadr", j); k[4] = gi("k_body_subtreesize", db);
      k[5] = gi("k_dof_velmask_lo", d); k[6] = gi("k_dof_velmask_hi", d);
      k[7] = fb(gf("dof_damping", d));
      k[8] = fb(jt == 0 ? 0.f : gf("jnt_stiffness", j));
      k[9] = fb(jt == 0 ? 0.f : gf("qpos_spring", gi("jnt_qposadr", j)));
      for (int u = 0; u < 2; u++) { k[10 + u] = gi("k_dof_act", 2 * d + u); k[12 + u] = fb(gf("k_dof_actmom", 2 * d + u)); }
    }
    {  // EntryTab, `nent` mass-matrix pattern slots: i[], j[], arm[] | lact[], damp[], dcoef[], lcoef[] (implicit only)
      int* k = r + SMJ_LR_ENT;
      for (int u = 0; u < nent; u++) {
        const int e = L + 64 * u, ok = e < m.nldl, ex = ok ? e : 0, i = gi("k_ldl_i", ex), j = gi("k_ldl_j", ex);
        k[u] = ok ? i : -1; k[nent + u] = ok ? j : 0; k[2 * nent + u] = fb((ok && i == j) ? gf("dof_armature", i) : 0.f);
        k[3 * nent + u] = ok ? gi("k_ldl_lact", ex) : -1;
        k[4 * nent + u] = fb(ok ? gf("k_ldl_damp", ex) : 0.f); k[5 * nent + u] = fb(ok ? gf("k_ldl_dcoef", ex) : 0.f); k[6 * nent + u] = fb(ok ? gf("k_ldl_lcoef", ex) : 0.f);
      }
    }
    {  // ActTab: dof[4], qadr[4], mom[4], prm[8], flags, gc_body, gc_mlo, gc_mhi, gc_mass, gc_x, gc_y, gc_z
      int* k = r + LR_ACT;
      const int a = L < nu ? L : 0;
      for (int u = 0; u < 4; u++) {
        const int dd = gi("k_act_dof", 4 * a + u);
        k[u] = L < nu ? dd : -1; k[4 + u] = dd >= 0 ? gi("k_dof_qposadr", dd) : 0; k[8 + u] = fb(gf("k_act_mom", 4 * a + u));
      }
      k[12] = fb(gf("actuator_gainprm", 3 * a));
      for (int q = 0; q < 3; q++) k[13 + q] = fb(gf("actuator_biasprm", 3 * a + q));
      k[16] = fb(gf("actuator_ctrlrange", 2 * a)); k[17] = fb(gf("actuator_ctrlrange", 2 * a + 1));
      k[18] = fb(gf("actuator_forcerange", 2 * a)); k[19] = fb(gf("actuator_forcerange", 2 * a + 1));
      k[20] = (gi("actuator_ctrllimited", a) ? 1 : 0) | (gi("actuator_forcelimited", a) ? 2 : 0) | (gi("actuator_biastype", a) == 1 ? 4 : 0);
      const int g = L < m.ngc ? gi("k_gc_body", L) : 0;
      k[21] = g; k[22] = gi("k_body_dofmask_lo", g); k[23] = gi("k_body_dofmask_hi", g);
      k[24] = fb(L < m.ngc ? gf("body_gcmass", g) : 0.f);
      for (int q = 0; q < 3; q++) k[25 + q] = fb(gf("body_gcipos", 3 * g + q));
    }
  }
  return rec;
}

static inline std::vector<int> smj_build_pprec(const DevModel& m, std::map<std::string, std::vector<int>>& I,
                                               std::map<std::string, std::vector<float>>& F) {
  std::vector<int> rec((size_t)(m.nplanepair > 0 ? m.nplanepair : 1) * SMJ_PP_STRIDE, 0);
  auto fb = [](float v) { int b; memcpy(&b, &v, 4); return b; };
  auto gi = [&](const char* n, size_t k) { const std::vector<int>& v = I[n]; return k < v.size() ? v[k] : 0; };
  auto gf = [&](const char* n, size_t k) { const std::vector<float>& v = F[n]; return k < v.size() ? v[k] : 0.f; };
  for (int t = 0; t < m.nplanepair; t++) {
    int* k = rec.data() + (size_t)t * SMJ_PP_STRIDE;
    const int p = gi("k_planepair", t), g1 = gi("pair_geom1", p), g2 = gi("pair_geom2", p);
    k[SMJ_PP_PAIR] = p; k[SMJ_PP_G1] = g1; k[SMJ_PP_G2] = g2; k[SMJ_PP_B1] = gi("geom_bodyid", g1); k[SMJ_PP_B2] = gi("geom_bodyid", g2);
    k[SMJ_PP_T2] = gi("geom_type", g2); k[SMJ_PP_MARGIN] = fb(gf("pair_margin", p)); k[SMJ_PP_RBOUND2] = fb(gf("geom_rbound", g2));
    for (int q = 0; q < 3; q++) {
      k[SMJ_PP_BCEN2 + q] = fb(gf("k_geom_bcenter", 3 * g2 + q));
      k[SMJ_PP_POS1 + q] = fb(gf("geom_pos", 3 * g1 + q)); k[SMJ_PP_POS2 + q] = fb(gf("geom_pos", 3 * g2 + q));
      k[SMJ_PP_SIZE2 + q] = fb(gf("geom_size", 3 * g2 + q));
    }
    for (int q = 0; q < 9; q++) { k[SMJ_PP_MAT1 + q] = fb(gf("k_geom_mat", 9 * g1 + q)); k[SMJ_PP_MAT2 + q] = fb(gf("k_geom_mat", 9 * g2 + q)); }
    k[SMJ_PP_CONDIM] = gi("pair_condim", p); k[SMJ_PP_MG] = fb(gf("pair_margin", p) - gf("pair_gap", p));
    for (int q = 0; q < 5; q++) { k[SMJ_PP_FRIC + q] = fb(gf("pair_friction", 5 * p + q)); k[SMJ_PP_SOLIMP + q] = fb(gf("pair_solimp", 5 * p + q)); }
    k[SMJ_PP_SOLREF] = fb(gf("pair_solref", 2 * p)); k[SMJ_PP_SOLREF + 1] = fb(gf("pair_solref", 2 * p + 1));
    for (int q = 0; q < 6; q++) k[SMJ_PP_BOX2 + q] = fb(gf("geom_aabb", 6 * g2 + q));
  }
  return rec;
}
static inline std::vector<int> smj_build_cgrec(const DevModel& m, std::map<std::string, std::vector<int>>& I,
                                               std::map<std::string, std::vector<float>>& F, bool stat = false) {
  const int ncg = stat ? m.nsgeom : m.ncgeom;
  std::vector<int> rec((size_t)(ncg > 0 ? ncg : 1) * SMJ_CG_STRIDE, 0);
  auto fb = [](float v) { int b; memcpy(&b, &v, 4); return b; };
  auto gi = [&](const char* n, size_t k) { const std::vector<int>& v = I[n]; return k < v.size() ? v[k] : 0; };
  auto gf = [&](const char* n, size_t k) { const std::vector<float>& v = F[n]; return k < v.size() ? v[k] : 0.f; };
  for (int c = 0; c < ncg; c++) {
    int* k = rec.data() + (size_t)c * SMJ_CG_STRIDE;
    const int g = gi(stat ? "k_sgeom" : "k_cgeom", c), adr = gi("geom_hulladr", g);
    k[SMJ_CG_GEOM] = g; k[SMJ_CG_BODY] = gi("geom_bodyid", g);
    k[SMJ_CG_META] = (int)((unsigned)gi("geom_type", g) | ((unsigned)gi("geom_hullnum", g) << 4) | ((unsigned)(adr < 0 ? 0 : adr) << 16));
    for (int q = 0; q < 3; q++) {
      k[SMJ_CG_POS + q] = fb(gf("geom_pos", 3 * g + q)); k[SMJ_CG_LCEN + q] = fb(gf("geom_aabb", 6 * g + q));
      k[SMJ_CG_HALF + q] = fb(gf("geom_aabb", 6 * g + 3 + q)); k[SMJ_CG_CCEN + q] = fb(gf("geom_ccenter", 3 * g + q));
      k[SMJ_CG_SIZE + q] = fb(gf("geom_size", 3 * g + q));
    }
    for (int q = 0; q < 9; q++) k[SMJ_CG_MAT + q] = fb(gf("k_geom_mat", 9 * g + q));
    k[SMJ_CG_RBOUND] = fb(gf("geom_rbound", g));
  }
  return rec;
}

static inline std::vector<int> smj_build_rowrec(const DevModel& m, std::map<std::string, std::vector<int>>& I,
                                                std::map<std::string, std::vector<float>>& F) {
  const int nstat = m.neq + m.nfric, n = nstat + 2 * m.nlimit;
  std::vector<int> rec((size_t)(n > 0 ? n : 1) * SMJ_RR_STRIDE, 0);
  auto fb = [](float v) { int b; memcpy(&b, &v, 4); return b; };
  auto gi = [&](const char* nm, size_t k) { const std::vector<int>& v = I[nm]; return k < v.size() ? v[k] : 0; };
  auto gf = [&](const char* nm, size_t k) { const std::vector<float>& v = F[nm]; return k < v.size() ? v[k] : 0.f; };
  for (int e = 0; e < m.neq; e++) {      // CT_EQUALITY = 0 (joint equalities)
    int* k = rec.data() + (size_t)e * SMJ_RR_STRIDE;
    const int j1 = gi("eq_obj1id", e), j2 = gi("eq_obj2id", e), q1 = gi("jnt_qposadr", j1), d1 = gi("jnt_dofadr", j1);
    k[SMJ_RR_TYPE] = 0; k[SMJ_RR_ID] = e; k[SMJ_RR_D1] = d1; k[SMJ_RR_Q1] = q1; k[SMJ_RR_V1] = fb(gf("qpos0", q1));
    float diag = gf("dof_invweight0", d1);
    if (j2 >= 0) {
      const int q2 = gi("jnt_qposadr", j2), d2 = gi("jnt_dofadr", j2);
      k[SMJ_RR_D2] = d2; k[SMJ_RR_Q2] = q2; k[SMJ_RR_V2] = fb(gf("qpos0", q2));
      diag += gf("dof_invweight0", d2);
    } else { k[SMJ_RR_D2] = -1; k[SMJ_RR_Q2] = 0; }
    for (int q = 0; q < 5; q++) { k[SMJ_RR_DATA + q] = fb(gf("eq_data", 5 * e + q)); k[SMJ_RR_SOLIMP + q] = fb(gf("eq_solimp", 5 * e + q)); }
    k[SMJ_RR_DIAG] = fb(diag);
    k[SMJ_RR_SOLREF] = fb(gf("eq_solref", 2 * e)); k[SMJ_RR_SOLREF + 1] = fb(gf("eq_solref", 2 * e + 1));
  }
  // (satellite of a dof / its index inside it; -1, 0 for dofs of the main tree)
  auto sat_of = [&](int d, int* local) {
    *local = 0;
    for (int sidx = 0; sidx < m.nsat; sidx++) {
      const int da = gi("k_sat_i", 8 * sidx + 3), nd = gi("k_sat_i", 8 * sidx + 4);
      if (d >= da && d < da + nd) { *local = d - da; return sidx; }
    }
    return -1;
  };
  for (int e = 0; e < m.neq; e++) rec[(size_t)e * SMJ_RR_STRIDE + SMJ_RR_SAT] = -1;
  for (int f = 0; f < m.nfric; f++) {    // CT_FRICTION = 1
    int* k = rec.data() + (size_t)(m.neq + f) * SMJ_RR_STRIDE;
    const int d = gi("k_fric_dof", f);
    k[SMJ_RR_SAT] = sat_of(d, &k[SMJ_RR_SDOF]);
    k[SMJ_RR_TYPE] = 1; k[SMJ_RR_ID] = d; k[SMJ_RR_D1] = d; k[SMJ_RR_D2] = -1;
    k[SMJ_RR_DIAG] = fb(gf("dof_invweight0", d)); k[SMJ_RR_FLOSS] = fb(gf("dof_frictionloss", d));
    k[SMJ_RR_SOLREF] = fb(gf("dof_solref", 2 * d)); k[SMJ_RR_SOLREF + 1] = fb(gf("dof_solref", 2 * d + 1));
    for (int q = 0; q < 5; q++) k[SMJ_RR_SOLIMP + q] = fb(gf("dof_solimp", 5 * d + q));
  }
  for (int L = 0; L < 2 * m.nlimit; L++) {   // CT_LIMIT = 3; slot L = (joint, side), lower side first
    int* k = rec.data() + (size_t)(nstat + L) * SMJ_RR_STRIDE;
    const int j = gi("k_limit_jnt", L >> 1), d = gi("jnt_dofadr", j);
    k[SMJ_RR_SAT] = sat_of(d, &k[SMJ_RR_SDOF]);
    k[SMJ_RR_TYPE] = 3; k[SMJ_RR_ID] = j; k[SMJ_RR_D1] = d; k[SMJ_RR_D2] = (L & 1) ? 1 : -1; k[SMJ_RR_Q1] = gi("jnt_qposadr", j);
    k[SMJ_RR_V1] = fb(gf("jnt_range", 2 * j + (L & 1))); k[SMJ_RR_V2] = fb(gf("jnt_margin", j));
    k[SMJ_RR_DIAG] = fb(gf("dof_invweight0", d));
    k[SMJ_RR_SOLREF] = fb(gf("jnt_solref", 2 * j)); k[SMJ_RR_SOLREF + 1] = fb(gf("jnt_solref", 2 * j + 1));
    for (int q = 0; q < 5; q++) k[SMJ_RR_SOLIMP + q] = fb(gf("jnt_solimp", 5 * j + q));
  }
  return rec;
}

static inline std::vector<int> smj_build_cprec(const DevModel& m, std::map<std::string, std::vector<int>>& I,
                                               std::map<std::string, std::vector<float>>& F, bool stat = false) {
  const int np = stat ? m.nstatpair : m.nconvpair;
  std::vector<int> rec((size_t)(np > 0 ? np : 1) * SMJ_CP_STRIDE, 0);
  auto fb = [](float v) { int b; memcpy(&b, &v, 4); return b; };
  auto gi = [&](const char* nm, size_t k) { const std::vector<int>& v = I[nm]; return k < v.size() ? v[k] : 0; };
  auto gf = [&](const char* nm, size_t k) { const std::vector<float>& v = F[nm]; return k < v.size() ? v[k] : 0.f; };
  std::map<int, int> dslot, sslot;   // static pairs: geom -> cache slot of the moving geom / index of the static geom
  if (stat) {
    for (int c = 0; c < m.ncgeom; c++) dslot[gi("k_cgeom", c)] = c;
    for (int c = 0; c < m.nsgeom; c++) sslot[gi("k_sgeom", c)] = c;
  }
  for (int t = 0; t < np; t++) {
    int* k = rec.data() + (size_t)t * SMJ_CP_STRIDE;
    const int p = gi(stat ? "k_statpair" : "k_convpair", t);
    k[SMJ_CP_PAIR] = p; k[SMJ_CP_G1] = gi("pair_geom1", p); k[SMJ_CP_G2] = gi("pair_geom2", p);
    if (stat) {
      k[SMJ_CP_S1] = sslot.count(k[SMJ_CP_G1]) ? -1 - sslot[k[SMJ_CP_G1]] : dslot[k[SMJ_CP_G1]];
      k[SMJ_CP_S2] = sslot.count(k[SMJ_CP_G2]) ? -1 - sslot[k[SMJ_CP_G2]] : dslot[k[SMJ_CP_G2]];
    } else { k[SMJ_CP_S1] = gi("k_convpair_s1", t); k[SMJ_CP_S2] = gi("k_convpair_s2", t); }
    k[SMJ_CP_MARGIN] = fb(gf("pair_margin", p)); k[SMJ_CP_MG] = fb(gf("pair_margin", p) - gf("pair_gap", p)); k[SMJ_CP_CONDIM] = gi("pair_condim", p);
    for (int q = 0; q < 5; q++) { k[SMJ_CP_FRIC + q] = fb(gf("pair_friction", 5 * p + q)); k[SMJ_CP_SOLIMP + q] = fb(gf("pair_solimp", 5 * p + q)); }
    k[SMJ_CP_SOLREF] = fb(gf("pair_solref", 2 * p)); k[SMJ_CP_SOLREF + 1] = fb(gf("pair_solref", 2 * p + 1));
    const float r1 = gf("geom_rbound", k[SMJ_CP_G1]), r2 = gf("geom_rbound", k[SMJ_CP_G2]);
    k[SMJ_CP_RBMIN] = fb(r1 < r2 ? r1 : r2);
    k[SMJ_CP_B1] = gi("geom_bodyid", k[SMJ_CP_G1]); k[SMJ_CP_B2] = gi("geom_bodyid", k[SMJ_CP_G2]);
  }
  return rec;
}

// `caps`: the kernel variants available to the caller, smallest first; the first one the model fits is chosen (*chosen).
template <class Up>
int smj_load_model(const void* blob, size_t nbytes, DevModel& m, Up& up, std::string& err, const SmjCaps* caps, int ncaps, int* chosen) {
  if (!blob || nbytes < 16 || memcmp(blob, "SMJB0001", 8) != 0) { err = "not an SMJB model blob"; return -1; }
  SmjBlob b{static_cast<const uint8_t*>(blob), nbytes};
  if (!b.valid()) { err = "model blob: truncated or corrupt entry table"; return -3; }
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
  GI(nq_all, "dims", 0) GI(nv_all, "dims", 1) GI(nu, "dims", 2) GI(nbody_all, "dims", 3) GI(njnt, "dims", 4) GI(ngeom, "dims", 5)
  m.nq = m.nq_all; m.nv = m.nv_all; m.nbody = m.nbody_all; m.nsat = 0;
  if (b.find("k_nsat")) {   // satellites (model_fuse.find_satellites): the main part's counts index the lane tables
    GI(nsat, "k_nsat", 0)
    if (m.nsat > 0) { GI(nq, "k_main_dims", 0) GI(nv, "k_main_dims", 1) GI(nbody, "k_main_dims", 2) GI(njnt, "k_main_dims", 3) }
  }
  GI(nsite, "dims", 6) GI(neq, "dims", 8) GI(nkey, "dims", 11) GI(npair, "dims", 12)
  GI(nlevel, "k_nlevel", 0) GI(nfric, "k_nfric", 0) GI(nlimit, "k_nlimit", 0) GI(nplanepair, "k_nplanepair", 0)
  GI(nldl, "k_nldl", 0) GI(imu_site, "sensor_imu_site", 0) GI(ngc, "k_ngc", 0) GI(nroot, "k_nroot", 0)
  GI(njump, "k_njump", 0) GI(maxsubtree, "k_maxsubtree", 0) GI(ncgeom, "k_ncgeom", 0) GI(nconvpair, "k_nconvpair", 0) GI(iterations, "opt_iterations", 0)
  GF(timestep, "opt_timestep", 0) GF(gravity[0], "opt_gravity", 0) GF(gravity[1], "opt_gravity", 1)
  GF(gravity[2], "opt_gravity", 2) GF(impratio, "opt_impratio", 0) GF(tolerance, "opt_tolerance", 0)
  GF(meaninertia, "stat_meaninertia", 0) GF(lidar_cutoff, "sensor_lidar_cutoff", 0)
#undef GI
#undef GF
  if (m.nsat > 0 && (m.ngeom >= 65536 || m.npair >= (1 << 19))) { err = "satellite builds: at most 65535 geoms and 524287 collision pairs (16-bit geom ids per contact, the PGS row key)"; return -4; }
  {
    const SmjBlobEntry* e = b.find("sensor_lidar_site");
    m.nlidar = e ? (int)(e->nbytes / 4) : 0;
  }
  m.row_limit = 0; m.pgs_cap = 0; m.warmstart = 1; m.pgs_fixed_iter = 0; m.qcqp_exact = 0; m.grad_noise = 4e-6f; m.pgs_island_stop = 1; m.pgs_dual_ws = 1; m.max_con_pair = 4; m.solver = 0; m.convex_pairs = 1; m.multiccd = 1; m.sep_cache = getenv("SMJ_NO_SEPCACHE") ? 0 : 1; m.manifold_cache = (getenv("SMJ_NO_MCACHE") || m.nv_all <= 32) ? 0 : 1;   /* pays where free objects rest; a robot alone (<= 32 dofs) has no resting convex pair and the lookup cost the headline 1 % */ m.multi_serial = 0; m.ls_iterations = 50; m.ls_tolerance = 0.01f;
  char buf[256];
  int pick = -1, first = 0;
  {   // optional hint of the model compiler: contact-rich scene, start at the big variant (model_fuse.prepare_for_kernels)
    const SmjBlobEntry* e = b.find("k_capacity_hint");
    int hint = 0;
    if (e && e->dtype == 1 && e->nbytes >= 4) memcpy(&hint, b.p + e->offset, 4);
    if (hint > 0 && ncaps > 1) first = 1;   // skip the standard variant: tall if the model fits it, else big
  }
  for (int v = first; v < ncaps && pick < 0; v++)
    if ((m.nsat > 0) == (caps[v].nsat > 0) && m.nsat <= caps[v].nsat &&
        m.nv <= (caps[v].nvs ? caps[v].nvs : caps[v].nvp) && m.nbody <= caps[v].nbp && m.nq <= caps[v].nvp + 8 && m.nldl <= caps[v].nent * 64) pick = v;
  if (pick < 0 && first > 0)   // (the hint skips the standard variant; a satellite model has its own builds)
    for (int v = 0; v < first && pick < 0; v++)
      if ((m.nsat > 0) == (caps[v].nsat > 0) && m.nsat <= caps[v].nsat &&
          m.nv <= (caps[v].nvs ? caps[v].nvs : caps[v].nvp) && m.nbody <= caps[v].nbp && m.nq <= caps[v].nvp + 8 && m.nldl <= caps[v].nent * 64) pick = v;
  if (pick < 0 || m.nu > 16) {
    const SmjCaps& c = caps[ncaps - 1];
    snprintf(buf, sizeof buf, "model exceeds kernel capacity (nv %d<=%d, nbody %d<=%d, nq %d<=%d, nu %d<=16, mass-matrix entries %d<=%d, satellites %d)",
             m.nv, c.nvp, m.nbody, c.nbp, m.nq, c.nvp + 8, m.nu, m.nldl, c.nent * 64, m.nsat);
    err = buf;
    return -4;
  }
  if (chosen) *chosen = pick;
  const int nent = caps[pick].nent;
  if (m.neq + m.nfric > 64) { err = "too many static constraint rows"; return -4; }
  if (2 * m.nlimit > 64) { err = "too many limited joints"; return -4; }
  if (m.ngc > 16) { err = "more than 16 gravity-compensated bodies"; return -4; }
  if (m.njump > 6) { err = "body tree deeper than 64 levels"; return -4; }
  if (m.ncgeom > NCG) { err = "too many geoms in non-plane collision pairs"; return -4; }
  if (m.nconvpair >= 65536) { err = "more than 65535 non-plane collision pairs (the survivor list holds 16-bit pair indices)"; return -4; }
  std::map<std::string, std::vector<int>> hosti;
  std::map<std::string, std::vector<float>> hostf;
#define X(n)                                                                               \
  {                                                                                        \
    const SmjBlobEntry* e = b.find(#n);                                                    \
    if (!e || e->dtype != 1) { err = "model blob: missing i32 array " #n; return -3; }     \
    std::vector<int> h(e->nbytes / 4 ? e->nbytes / 4 : 1, 0);                              \
    memcpy(h.data(), b.p + e->offset, e->nbytes);                                          \
    m.n = up.i32(h);                                                                       \
    hosti[#n] = h;                                                                         \
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
    hostf[#n] = h;                                                                         \
    if (!m.n) { err = "device allocation failed for " #n; return -2; }                     \
  }
  SMJ_MODEL_F32(X)
#undef X
  m.nsgeom = 0; m.nstatpair = 0; m.k_sgrec = m.k_sprec = m.k_spair = m.k_grid_adr = m.k_grid_list = m.k_sg_cell = nullptr; m.k_sg_bound = nullptr; m.k_sgw = nullptr;
  if (b.find("k_nsgeom")) {
    if (!geti("k_nsgeom", 0, &m.nsgeom) || !geti("k_nstatpair", 0, &m.nstatpair)) return -3;
    if (m.nsgeom > 512) { err = "more than 512 static collision geoms (the static broadphase's candidate word holds 9 bits of static geom index)"; return -4; }
    if (m.nstatpair > 0 && m.nsat == 0) { err = "static collision pairs (k_statpair) without satellites: only the satellite builds run collision_static; rebuild the blob (model_fuse.prepare_for_kernels)"; return -4; }
    if (m.nsgeom > 0) {
      for (int q = 0; q < 3; q++) { if (!geti("k_grid", q, &m.grid_dim[q]) || !getf("k_grid_f", q, &m.grid_org[q])) return -3; }
      if (!getf("k_grid_f", 3, &m.grid_h) || !getf("k_grid_f", 4, &m.grid_margin)) return -3;
      auto loadi = [&](const char* name, const int** dst) -> bool {
        const SmjBlobEntry* e = b.find(name);
        if (!e || e->dtype != 1) { err = std::string("model blob: missing i32 array ") + name; return false; }
        std::vector<int> h(e->nbytes / 4 ? e->nbytes / 4 : 1, 0);
        memcpy(h.data(), b.p + e->offset, e->nbytes);
        hosti[name] = h;
        *dst = up.i32(h);
        return *dst != nullptr;
      };
      const int* dummy = nullptr;
      if (!loadi("k_sgeom", &dummy) || !loadi("k_statpair", &dummy) || !loadi("k_spair", &m.k_spair) || !loadi("k_grid_adr", &m.k_grid_adr) ||
          !loadi("k_grid_list", &m.k_grid_list) || !loadi("k_sg_cell", &m.k_sg_cell)) return err.empty() ? -2 : -3;
      {
        const SmjBlobEntry* e = b.find("geom_aabb");   // (already among the float tables: hostf)
        (void)e;
      }
      std::vector<int> sg = smj_build_cgrec(m, hosti, hostf, true), spr = smj_build_cprec(m, hosti, hostf, true);
      m.k_sgrec = up.i32(sg);
      m.k_sprec = up.i32(spr);
      std::vector<float> bound(8 * (size_t)m.nsgeom, 0.f);   // world AABB of the geom's oriented box: lo xyz, pad, hi xyz, pad
      std::vector<float> wcen(3 * (size_t)m.nsgeom, 0.f);
      for (int c = 0; c < m.nsgeom; c++) {
        const int* k = sg.data() + (size_t)c * SMJ_CG_STRIDE;
        float f[SMJ_CG_STRIDE];
        memcpy(f, k, sizeof f);
        for (int i = 0; i < 3; i++) {
          const float cen = f[SMJ_CG_POS + i] + f[SMJ_CG_MAT + 3 * i] * f[SMJ_CG_LCEN] + f[SMJ_CG_MAT + 3 * i + 1] * f[SMJ_CG_LCEN + 1] + f[SMJ_CG_MAT + 3 * i + 2] * f[SMJ_CG_LCEN + 2];
          const float ext = fabsf(f[SMJ_CG_MAT + 3 * i]) * f[SMJ_CG_HALF] + fabsf(f[SMJ_CG_MAT + 3 * i + 1]) * f[SMJ_CG_HALF + 1] + fabsf(f[SMJ_CG_MAT + 3 * i + 2]) * f[SMJ_CG_HALF + 2];
          wcen[3 * c + i] = cen;
          bound[8 * c + i] = cen - ext; bound[8 * c + 4 + i] = cen + ext;
        }
      }
      m.k_sg_bound = up.f32(bound);
      std::vector<float> sgw(32 * (size_t)m.nsgeom, 0.f);
      for (int c = 0; c < m.nsgeom; c++) {
        const int* k = sg.data() + (size_t)c * SMJ_CG_STRIDE;
        float f[SMJ_CG_STRIDE];
        memcpy(f, k, sizeof f);
        float* w = sgw.data() + 32 * (size_t)c;
        for (int i = 0; i < 3; i++) {
          w[i] = f[SMJ_CG_POS + i];
          w[12 + i] = wcen[3 * c + i];
          w[15 + i] = f[SMJ_CG_HALF + i];
          w[18 + i] = f[SMJ_CG_POS + i] + f[SMJ_CG_MAT + 3 * i] * f[SMJ_CG_CCEN] + f[SMJ_CG_MAT + 3 * i + 1] * f[SMJ_CG_CCEN + 1] + f[SMJ_CG_MAT + 3 * i + 2] * f[SMJ_CG_CCEN + 2];
          w[21 + i] = f[SMJ_CG_SIZE + i];
        }
        for (int i = 0; i < 9; i++) w[3 + i] = f[SMJ_CG_MAT + i];
        memcpy(&w[24], &k[SMJ_CG_META], 4);
      }
      m.k_sgw = up.f32(sgw);
      if (!m.k_sgrec || !m.k_sprec || !m.k_sg_bound || !m.k_sgw) { err = "device allocation failed for the static-geometry tables"; return -2; }
    }
  }
  m.nfric_main = m.nfric; m.nlimit_main = m.nlimit; m.k_satrec = nullptr;
  if (m.nsat > 0) {
    const SmjBlobEntry* ei = b.find("k_sat_i");
    const SmjBlobEntry* ef = b.find("k_sat_f");
    if (!ei || !ef || ei->dtype != 1 || ef->dtype != 0 || ei->nbytes < 32u * m.nsat || ef->nbytes < 8u * 44u * m.nsat) { err = "model blob: satellite tables missing"; return -3; }
    std::vector<int> si(8 * (size_t)m.nsat), rec((size_t)m.nsat * SMJ_SR_STRIDE, 0);
    memcpy(si.data(), b.p + ei->offset, 32u * m.nsat);
    hosti["k_sat_i"] = si;
    const double* sf = reinterpret_cast<const double*>(b.p + ef->offset);
    for (int sidx = 0; sidx < m.nsat; sidx++) {
      int* k = rec.data() + (size_t)sidx * SMJ_SR_STRIDE;
      for (int q = 0; q < 8; q++) k[q] = si[8 * sidx + q];
      for (int q = 0; q < 44; q++) { const float v = (float)sf[44 * sidx + q]; memcpy(&k[SMJ_SR_F + q], &v, 4); }
    }
    m.k_satrec = up.i32(rec);
    if (!m.k_satrec) { err = "device allocation failed for k_satrec"; return -2; }
    // static rows / limit slots of the main tree come first in their tables (dof / joint order)
    m.nfric_main = 0;
    for (int f = 0; f < m.nfric; f++) m.nfric_main += hosti["k_fric_dof"][f] < m.nv;
    m.nlimit_main = 0;
    for (int L = 0; L < m.nlimit; L++) m.nlimit_main += hosti["jnt_dofadr"][hosti["k_limit_jnt"][L]] < m.nv;
  }
  {
    std::vector<int> rec = smj_build_lanerec(m, hosti, hostf, nent);
    m.k_lanerec = up.i32(rec);
    if (!m.k_lanerec) { err = "device allocation failed for k_lanerec"; return -2; }
    std::vector<int> pp = smj_build_pprec(m, hosti, hostf), cg = smj_build_cgrec(m, hosti, hostf);
    m.k_pprec = up.i32(pp);
    m.k_cgrec = up.i32(cg);
    if (!m.k_pprec || !m.k_cgrec) { err = "device allocation failed for the collision records"; return -2; }
    std::vector<int> rr = smj_build_rowrec(m, hosti, hostf);
    m.k_rowrec = up.i32(rr);
    if (!m.k_rowrec) { err = "device allocation failed for k_rowrec"; return -2; }
    std::vector<int> cp = smj_build_cprec(m, hosti, hostf);
    m.k_cprec = up.i32(cp);
    if (!m.k_cprec) { err = "device allocation failed for k_cprec"; return -2; }
  }
  return 0;
}
