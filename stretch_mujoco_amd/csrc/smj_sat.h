// Satellites -- members of StepKernel in the satellite builds (NSAT > 0; included inside the struct by smj_step_impl.h).
//
// A kitchen is the robot plus many small mechanisms that meet it only through contacts: free objects (one body, 6 dofs) and the
// doors / drawers / knobs of fixtures (one body on a hinge or slide, welded to the world otherwise).  MuJoCo treats them as
// more trees of the same model (reference: whatever `scene_xml_path` holds, mujoco_server.py:248-252; the kitchens of
// robocasa_gen.py:129-239); dense 64-column kernels (the big38 / big50 builds) pay for them in every stage.  Here:
//   * the MAIN tree keeps the standard variant's mapping (lane = body, lane = dof, 32 dense columns);
//   * satellite s is ONE lane (32 + s): pose, 6 x 6 mass block, bias and passive forces in closed form from its record
//     (DevModel::k_satrec) -- no tree stages;
//   * constraint rows that touch the main tree come first (rows 0 .. nd-1) and are the only ones with a dense Jacobian row; every
//     row has two 6-column satellite slots (SatMem::Js);
//   * Newton: H = M + J' W J is block diagonal over the satellites except where a contact couples a satellite to the main tree or
//     to another satellite.  Uncoupled satellites (an object resting on a counter, a door nobody touches) solve their 6 x 6 block
//     on their own lane; coupled ones join a dense extension of the main system for that step (up to NXS of them).
// Same algorithm as the monolithic solver (one line search, one termination test over all dofs): [MJ] mj_solNewton on the whole
// model, which is what the oracle runs.

SMJ_DEV const int* satrec(int si) const { return M.k_satrec + si * SMJ_SR_STRIDE; }
SMJ_DEV static int tri6(int i, int j) { return i >= j ? (i * (i + 1)) / 2 + j : (j * (j + 1)) / 2 + i; }
// the six satellite columns of slot u of a row
SMJ_DEV const float* jsp(int row, int u) const { return u ? s.sat.Js2[s.sat.e2[row]] : s.sat.Js[row]; }
SMJ_DEV float* jspw(int row, int u) { return u ? s.sat.Js2[s.sat.e2[row]] : s.sat.Js[row]; }

// ------------------------------------------------------------------ state
SMJ_DEV void sat_load_state() {
  const long ld = S.ld;
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      const int* r = satrec(si);
      const int qa = r[SMJ_SR_QADR], da = r[SMJ_SR_DADR], nd = r[SMJ_SR_NDOF], nqs = nd == 6 ? 7 : 1;
      s.sat.jtype[si] = r[SMJ_SR_JTYPE]; s.sat.ndof[si] = nd; s.sat.body[si] = r[SMJ_SR_BODY];
      if (S.stage) {
        const float* st = stage_row();
        for (int k = 0; k < 7; k++) s.sat.q[si][k] = k < nqs ? ld_coh(&st[S.lay.qpos + qa + k]) : 0.f;
        for (int k = 0; k < 6; k++) {
          s.sat.x[SX_V][si][k] = k < nd ? ld_coh(&st[S.lay.qvel + da + k]) : 0.f;
          s.sat.x[SX_QA][si][k] = k < nd ? ld_coh(&st[S.lay.warm + da + k]) : 0.f;
        }
      } else {
        for (int k = 0; k < 7; k++) s.sat.q[si][k] = k < nqs ? S.qpos[(qa + k) * ld + env] : 0.f;
        for (int k = 0; k < 6; k++) {
          s.sat.x[SX_V][si][k] = k < nd ? S.qvel[(da + k) * ld + env] : 0.f;
          s.sat.x[SX_QA][si][k] = k < nd ? S.warm[(da + k) * ld + env] : 0.f;
        }
      }
    }
  }
  SYNC();
}
SMJ_DEV void sat_store_state() {
  const long ld = S.ld;
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      const int* r = satrec(si);
      const int qa = r[SMJ_SR_QADR], da = r[SMJ_SR_DADR], nd = r[SMJ_SR_NDOF], nqs = nd == 6 ? 7 : 1;
      if (S.stage) {
        float* st = stage_row();
        for (int k = 0; k < nqs; k++) st_coh(&st[S.lay.qpos + qa + k], s.sat.q[si][k]);
        for (int k = 0; k < nd; k++) { st_coh(&st[S.lay.qvel + da + k], s.sat.x[SX_V][si][k]); st_coh(&st[S.lay.warm + da + k], s.sat.x[SX_QA][si][k]); }
      } else {
        for (int k = 0; k < nqs; k++) S.qpos[(qa + k) * ld + env] = s.sat.q[si][k];
        for (int k = 0; k < nd; k++) { S.qvel[(da + k) * ld + env] = s.sat.x[SX_V][si][k]; S.warm[(da + k) * ld + env] = s.sat.x[SX_QA][si][k]; }
      }
    }
  }
}

// motion axis k of satellite si as a spatial vector about the satellite's centre of mass [angular; linear]  ([MJ] cdof)
SMJ_DEV void sat_cdof(int si, int k, const float* R, const float* off /* com - anchor */, float* c) const {
  const int jt = s.sat.jtype[si];
  if (jt == JT_FREE) {
    if (k < 3) { c[0] = c[1] = c[2] = 0.f; c[3] = k == 0; c[4] = k == 1; c[5] = k == 2; }
    else { const float ax[3] = {R[k - 3], R[3 + k - 3], R[6 + k - 3]}; c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2]; cross3(c + 3, ax, off); }
  } else if (jt == JT_SLIDE) { c[0] = c[1] = c[2] = 0.f; for (int x = 0; x < 3; x++) c[3 + x] = s.sat.wax[si][x]; }
  else { for (int x = 0; x < 3; x++) c[x] = s.sat.wax[si][x]; cross3(c + 3, s.sat.wax[si], off); }
}

// ------------------------------------------------------------------ B.1 .. B.6 for a single body, one satellite per lane
// [MJ] mj_kinematics, mj_comPos, mj_crb (the body's own 6 x 6 block), mj_comVel, mj_passive, mj_rne(flg_acc = 0): the same
// formulas as the tree stages, specialised to a tree of one body whose parent is the world.
SMJ_DEV void sat_forward() {
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      const int* r = satrec(si);
      const int b = r[SMJ_SR_BODY], jt = r[SMJ_SR_JTYPE], nd = r[SMJ_SR_NDOF];
      float pos[3], quat[4], Rm[9], anc[3];
      if (jt == JT_FREE) {
        for (int k = 0; k < 3; k++) pos[k] = s.sat.q[si][k];
        for (int k = 0; k < 4; k++) quat[k] = s.sat.q[si][3 + k];
        quat_normalize(quat);
        for (int k = 0; k < 3; k++) anc[k] = pos[k];
      } else {
        float ax[3], jp[3], xa[3], a[3];
        for (int k = 0; k < 3; k++) { pos[k] = asf(r[SMJ_SR_POS + k]); ax[k] = asf(r[SMJ_SR_JAXIS + k]); jp[k] = asf(r[SMJ_SR_JPOS + k]); }
        for (int k = 0; k < 4; k++) quat[k] = asf(r[SMJ_SR_QUAT + k]);
        quat2mat(Rm, quat);
        mulmat3vec(xa, Rm, ax);
        mulmat3vec(a, Rm, jp);
        for (int k = 0; k < 3; k++) { anc[k] = pos[k] + a[k]; s.sat.wax[si][k] = xa[k]; s.sat.wanc[si][k] = anc[k]; }
        const float dq = s.sat.q[si][0] - asf(r[SMJ_SR_Q0]);
        if (jt == JT_SLIDE) { for (int k = 0; k < 3; k++) pos[k] += xa[k] * dq; }
        else {
          float sn, cs;
          sincosf(0.5f * dq, &sn, &cs);
          const float dqv[4] = {cs, ax[0] * sn, ax[1] * sn, ax[2] * sn};
          quat_mul(quat, quat, dqv);
          quat2mat(Rm, quat);
          mulmat3vec(a, Rm, jp);
          for (int k = 0; k < 3; k++) pos[k] = anc[k] - a[k];
        }
        quat_normalize(quat);
      }
      quat2mat(Rm, quat);
      for (int k = 0; k < 3; k++) s.xpos[b][k] = pos[k];
      for (int k = 0; k < 4; k++) s.xquat[b][k] = quat[k];
      for (int k = 0; k < 9; k++) s.xmat[b][k] = Rm[k];
      // centre of mass, inertia about it in world axes
      float ip[3], com[3], off[3];
      const float lip[3] = {asf(r[SMJ_SR_INL + 6]), asf(r[SMJ_SR_INL + 7]), asf(r[SMJ_SR_INL + 8])};
      mulmat3vec(ip, Rm, lip);
      for (int k = 0; k < 3; k++) { com[k] = pos[k] + ip[k]; s.com[b][k] = com[k]; off[k] = com[k] - anc[k]; }
      const float I0 = asf(r[SMJ_SR_INL]), I1 = asf(r[SMJ_SR_INL + 1]), I2 = asf(r[SMJ_SR_INL + 2]), I3 = asf(r[SMJ_SR_INL + 3]), I4 = asf(r[SMJ_SR_INL + 4]), I5 = asf(r[SMJ_SR_INL + 5]);
      const float Il[9] = {I0, I3, I4, I3, I1, I5, I4, I5, I2}, mass = asf(r[SMJ_SR_INL + 9]);
      float RI[9], Rt[9], T[9];
      mulmat3(RI, Rm, Il);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[3 * i + j] = Rm[3 * j + i];
      mulmat3(T, RI, Rt);
      const float cin[10] = {T[0], T[4], T[8], T[1], T[2], T[5], 0.f, 0.f, 0.f, mass};
      // motion axes, the 6 x 6 mass block, velocity
      float cd[6][6], Icd[6][6], cvel[6] = {0, 0, 0, 0, 0, 0};
      for (int k = 0; k < 6; k++) {
        if (k < nd) sat_cdof(si, k, Rm, off, cd[k]); else for (int x = 0; x < 6; x++) cd[k][x] = 0.f;
        mul_inert_vec(Icd[k], cin, cd[k]);
        const float vk = s.sat.x[SX_V][si][k];
        for (int x = 0; x < 6; x++) cvel[x] += cd[k][x] * vk;
      }
      for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
          float v = 0;
          for (int x = 0; x < 6; x++) v += cd[j][x] * Icd[i][x];
          if (i == j) v = i < nd ? v + asf(r[SMJ_SR_ARM + i]) : 1.f;   // identity padding beyond the satellite's dofs
          s.sat.Mb[si][tri6(i, j)] = v;
        }
      // bias force: body acceleration without qacc = sum cdof_dot qvel - gravity, f = I a + v x* (I v)   ([MJ] mj_rne)
      float cacc[6] = {0, 0, 0, -M.gravity[0], -M.gravity[1], -M.gravity[2]};
      if (jt == JT_FREE) {
        // the rotational axes see the velocity of the translational dofs only ([MJ] mj_comVel: the free joint's special case)
        const float cv[6] = {0, 0, 0, s.sat.x[SX_V][si][0], s.sat.x[SX_V][si][1], s.sat.x[SX_V][si][2]};
        for (int k = 3; k < 6; k++) {
          float cdd[6];
          cross_motion(cdd, cv, cd[k]);
          const float vk = s.sat.x[SX_V][si][k];
          for (int x = 0; x < 6; x++) cacc[x] += cdd[x] * vk;
        }
      }
      float t1[6], t2[6], cf[6];
      mul_inert_vec(t1, cin, cacc);
      mul_inert_vec(t2, cin, cvel);
      cross_force(cf, cvel, t2);
      // passive: damper, spring (hinge / slide), gravity compensation at the body's gravcomp point
      const float gcm = asf(r[SMJ_SR_GCMASS]);
      float go[3] = {0, 0, 0}, gF[3] = {0, 0, 0};
      if (gcm != 0.f) {
        const float lp[3] = {asf(r[SMJ_SR_GCIPOS]), asf(r[SMJ_SR_GCIPOS + 1]), asf(r[SMJ_SR_GCIPOS + 2])};
        float pt[3];
        mulmat3vec(pt, Rm, lp);
        for (int k = 0; k < 3; k++) { go[k] = pt[k] + pos[k] - com[k]; gF[k] = -M.gravity[k] * gcm; }
      }
      for (int k = 0; k < 6; k++) {
        float g = 0.f;
        if (k < nd) {
          float bias = 0;
          for (int x = 0; x < 6; x++) bias += cd[k][x] * (t1[x] + cf[x]);
          g = -asf(r[SMJ_SR_DAMP + k]) * s.sat.x[SX_V][si][k] - bias;
          if (jt != JT_FREE && asf(r[SMJ_SR_STIFF]) != 0.f) g -= asf(r[SMJ_SR_STIFF]) * (s.sat.q[si][0] - asf(r[SMJ_SR_SPRING]));
          if (gcm != 0.f) {
            float tv[3];
            cross3(tv, cd[k], go);
            g += (cd[k][3] + tv[0]) * gF[0] + (cd[k][4] + tv[1]) * gF[1] + (cd[k][5] + tv[2]) * gF[2];
          }
        }
        s.sat.x[SX_G][si][k] = g;
      }
    }
  }
  SYNC();
}

// out = Mb in, both satellite vector fields (lane = satellite)
SMJ_DEV void sat_matM(int fout, int fin) {
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      float xin[6];
      for (int k = 0; k < 6; k++) xin[k] = s.sat.x[fin][si][k];
      for (int i = 0; i < 6; i++) {
        float v = 0;
        for (int j = 0; j < 6; j++) v += s.sat.Mb[si][tri6(i, j)] * xin[j];
        s.sat.x[fout][si][i] = i < s.sat.ndof[si] ? v : 0.f;
      }
    }
  }
}
// the satellite part of row `row` times the satellite field `fld`  (lane = row code)
SMJ_DEV float sat_jdot(int row, int fld) const {
  float v = 0;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int si = s.sat.esat[row][u];
    if (si >= 0)
      { const float* js = jsp(row, u); for (int k = 0; k < 6; k++) v += js[k] * s.sat.x[fld][si][k]; }
  }
  return v;
}
// x[SX_TMP] = J_s' f : constraint force on the satellite's dofs (forces from s.ef)
SMJ_DEV void sat_JTf() {
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      float acc[6] = {0, 0, 0, 0, 0, 0};
      for (int it = 0; it < s.sat.nitem[si]; it++) {
        const int r0 = s.sat.irow[si][it], inf = s.sat.iinf[si][it], n = inf & ITEM_N, u = (inf & ITEM_SLOT) ? 1 : 0;
        for (int p = 0; p < n; p++) {
          const float f = s.ef[r0 + p];
          { const float* js = jsp(r0 + p, u); for (int k = 0; k < 6; k++) acc[k] += js[k] * f; }
        }
      }
      for (int k = 0; k < 6; k++) s.sat.x[SX_TMP][si][k] = acc[k];
    }
  }
}

// in-register LDL' solve of a 6 x 6 symmetric positive definite block (packed lower triangle), x <- A^-1 x
SMJ_DEV static void sat_solve6(const float* A, float* x) {
  float L[21], D[6], Dv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float d = A[tri6(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[tri6(j, k)] * L[tri6(j, k)] * Dv[k];
    d = fmaxf(d, 1e-30f);
    Dv[j] = d;
    D[j] = fast_rcp(d);   // (kept as the reciprocal)
    const float inv = D[j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float v = A[tri6(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) v -= L[tri6(i, k)] * L[tri6(j, k)] * Dv[k];
      L[tri6(i, j)] = v * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int k = 0; k < i; k++) x[i] -= L[tri6(i, k)] * x[k];
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] *= D[i];
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int k = i + 1; k < 6; k++) x[i] -= L[tri6(k, i)] * x[k];
}

// Newton blocks of the satellites: Hb = Mb + sum over the satellite's rows of J_s' W J_s  (W = D for rows in the quadratic
// zone, the cone Hessian for contacts in the middle zone -- s.ediag / s.u.n.cH as staged by solve_newton), lane = satellite
SMJ_DEV void sat_hessian(uint64_t conemask) {
  // 64 / NSAT lanes per satellite (4 in the 16-satellite build, 2 in the 32-satellite one): each takes every LPS-th item of the
  // satellite's list, the partial blocks are summed over the lane group with quad permutes (round 4: 10 k -> 3 k cycles per call)
  constexpr int LPS = 64 / NSAT;
  static_assert(LPS == 2 || LPS == 4, "satellite Hessian: 2 or 4 lanes per satellite");
  PL<float[21]> Hp;
  LANES {
    const int si = lane / LPS, sub = lane % LPS;
    float H[21];
    for (int k = 0; k < 21; k++) H[k] = (sub == 0 && si < M.nsat) ? s.sat.Mb[si][k] : 0.f;
    if (si < M.nsat)
      for (int it = sub; it < s.sat.nitem[si]; it += LPS) {
        const int r0 = s.sat.irow[si][it], inf = s.sat.iinf[si][it], n = inf & ITEM_N, u = (inf & ITEM_SLOT) ? 1 : 0;
        const int c = s.sat.icon[si][it];
        if ((inf & ITEM_CONTACT) && ((conemask >> c) & 1)) {
          const float* Hc = s.u.n.cH[s.sat.chs[c]];
          float Jc[6][6];   // the contact's rows, satellite columns (zero beyond the contact's condim)
          for (int p = 0; p < 6; p++) {
            const float* js = jsp(r0 + (p < n ? p : 0), u);
            for (int k = 0; k < 6; k++) Jc[p][k] = p < n ? js[k] : 0.f;
          }
#pragma unroll
          for (int p = 0; p < 6; p++) {   // H += Jc' (Hc Jc), one row of T = Hc Jc at a time (rows beyond the condim: Jc[p] = 0)
            float T[6];
            for (int k = 0; k < 6; k++) {
              float v = 0;
              for (int q = 0; q < 6; q++) v += Hc[6 * p + q] * Jc[q][k];   // (cH is stored 6 x 6, zero beyond the condim)
              T[k] = v;
            }
            for (int i = 0; i < 6; i++)
              for (int j = 0; j <= i; j++) H[tri6(i, j)] += Jc[p][i] * T[j];
          }
        } else {
          for (int p = 0; p < n; p++) {
            const float w = s.ediag[r0 + p];
            if (w != 0.f) {
              const float* js = jsp(r0 + p, u);
              for (int i = 0; i < 6; i++) {
                const float wi = w * js[i];
                for (int j = 0; j <= i; j++) H[tri6(i, j)] += wi * js[j];
              }
            }
          }
        }
      }
    for (int k = 0; k < 21; k++) Hp[lane][k] = H[k];
  }
#pragma unroll
  for (int k = 0; k < 21; k++) {
    PL<float> t;
    LANES { t[lane] = Hp[lane][k]; }
    wave_group_sum<LPS>(t);
    LANES { Hp[lane][k] = t[lane]; }
  }
  LANES {
    const int si = lane / LPS;
    if (lane % LPS == 0 && si < M.nsat)
      for (int k = 0; k < 21; k++) s.sat.Hb[si][k] = Hp[lane][k];
  }
}

// search direction of the satellites that are solved on their own lane: srch = -Hb^-1 grad
SMJ_DEV void sat_solve_own() {
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat && s.sat.ext[si] < 0) {
      float A[21], x[6];
      for (int k = 0; k < 21; k++) A[k] = s.sat.Hb[si][k];
      for (int k = 0; k < 6; k++) x[k] = s.sat.x[SX_GRAD][si][k];
      sat_solve6(A, x);
      for (int k = 0; k < 6; k++) s.sat.x[SX_SRCH][si][k] = k < s.sat.ndof[si] ? -x[k] : 0.f;
    }
  }
}

// Dense extension of the Newton system for the satellites coupled to the main tree or to each other in this step: columns
// NVS + 6 e + k of s.u.n.H (e = extension slot).  Called after the main block H[0..NVS)[0..NVS) has been stored.
SMJ_DEV void sat_extend_hessian(int next, uint64_t conemask) {
  const int n = NVS + 6 * next;
  // clear everything outside the main block up to order n (identity on the diagonal: padding of 1-dof satellites)
  LANES {
    for (int idx = lane; idx < n * n; idx += 64) {
      const int i = idx / n, j = idx - i * n;
      if (i >= NVS || j >= NVS) s.u.n.H[i][j] = 0.f;
    }
  }
  SYNC();
  // own blocks
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat && s.sat.ext[si] >= 0) {
      const int o = NVS + 6 * s.sat.ext[si];
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) s.u.n.H[o + i][o + j] = s.sat.Hb[si][tri6(i, j)];
    }
  }
  SYNC();
  // main-satellite coupling: lane = main dof d, H[o+k][d] = sum over the rows of contacts between the main tree and the satellite
  for (int e = 0; e < next; e++) {
    const int si = uni(s.sat.xs[e]), o = NVS + 6 * e, ni = uni(s.sat.nitem[si]);
    for (int it = 0; it < ni; it++) {
      const int inf = uni((int)s.sat.iinf[si][it]);
      if (!(inf & ITEM_MAIN)) continue;
      const int r0 = uni((int)s.sat.irow[si][it]), nr = inf & ITEM_N, u = (inf & ITEM_SLOT) ? 1 : 0, c = uni((int)s.sat.icon[si][it]);
      const bool cone = (conemask >> c) & 1;
      LANES {
        if (lane < NVS) {
          float acc[6] = {0, 0, 0, 0, 0, 0};
          for (int p = 0; p < nr; p++) {
            float wj;   // (W J)[p][lane]
            if (cone) {
              wj = 0.f;
              for (int q = 0; q < nr; q++) wj += s.u.n.cH[s.sat.chs[c]][6 * p + q] * s.J[r0 + q][lane];
            } else wj = s.ediag[r0 + p] * s.J[r0 + p][lane];
            const float* js = jsp(r0 + p, u);
            for (int k = 0; k < 6; k++) acc[k] += wj * js[k];
          }
          for (int k = 0; k < 6; k++) { s.u.n.H[o + k][lane] += acc[k]; s.u.n.H[lane][o + k] += acc[k]; }
        }
      }
      SYNC();
    }
  }
  // satellite-satellite coupling: lane = (k, l) entry of the 6 x 6 block between the two satellites of the contact
  for (int t = 0; t < NSS; t++) {
    const int c = uni(s.sat.sscon[t]);
    if (c < 0) break;
    const int r0 = uni(s.cefc[c]);
    if (r0 < 0) continue;
    const int nr = uni(s.cdim[c]), sa = uni((int)s.sat.esat[r0][0]), sb = uni((int)s.sat.esat[r0][1]);
    if (sa < 0 || sb < 0) continue;
    const int ea = uni(s.sat.ext[sa]), eb = uni(s.sat.ext[sb]);
    if (ea < 0 || eb < 0) continue;
    const int oa = NVS + 6 * ea, ob = NVS + 6 * eb;
    const bool cone = (conemask >> c) & 1;
    LANES {
      if (lane < 36) {
        const int k = lane / 6, l = lane - 6 * k;
        float v = 0;
        for (int p = 0; p < nr; p++) {
          float wj;
          if (cone) {
            wj = 0.f;
            for (int q = 0; q < nr; q++) wj += s.u.n.cH[s.sat.chs[c]][6 * p + q] * jsp(r0 + q, 1)[l];
          } else wj = s.ediag[r0 + p] * jsp(r0 + p, 1)[l];
          v += s.sat.Js[r0 + p][k] * wj;
        }
        s.u.n.H[oa + k][ob + l] += v;
        s.u.n.H[ob + l][oa + k] += v;
      }
    }
    SYNC();
  }
}

// ------------------------------------------------------------------ B.8 for the satellites: (Mb + h diag(damping)) x = qfrc, then
// the semi-implicit position update ([MJ] mj_implicit, fast variant: no actuators on satellites, damping is the only
// velocity derivative), and mj_checkPos / mj_checkVel.  Returns (per lane) whether the satellite's state went bad.
SMJ_DEV void sat_integrate(PL<int>& bad) {
  const float h = M.timestep;
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      const int* r = satrec(si);
      const int nd = s.sat.ndof[si];
      float A[21], x[6];
      for (int k = 0; k < 21; k++) A[k] = s.sat.Mb[si][k];
      for (int k = 0; k < 6; k++) { if (k < nd) A[tri6(k, k)] += h * asf(r[SMJ_SR_DAMP + k]); x[k] = s.sat.x[SX_TMP][si][k]; }
      sat_solve6(A, x);
      int b = 0;
      for (int k = 0; k < 6; k++) {
        if (k < nd) { s.sat.x[SX_V][si][k] += h * x[k]; b |= !(fabsf(s.sat.x[SX_V][si][k]) < 1e10f); }
      }
      if (s.sat.jtype[si] == JT_FREE) {
        for (int k = 0; k < 3; k++) s.sat.q[si][k] += h * s.sat.x[SX_V][si][k];
        const float wv[3] = {s.sat.x[SX_V][si][3], s.sat.x[SX_V][si][4], s.sat.x[SX_V][si][5]};
        const float nrm = sqrtf(dot3(wv, wv)), ang = nrm * h;
        float q[4] = {s.sat.q[si][3], s.sat.q[si][4], s.sat.q[si][5], s.sat.q[si][6]};
        if (ang > 0) {
          const float sn = sinf(0.5f * ang) / nrm, dq[4] = {cosf(0.5f * ang), wv[0] * sn, wv[1] * sn, wv[2] * sn};
          quat_mul(q, q, dq);
        }
        quat_normalize(q);
        for (int k = 0; k < 4; k++) s.sat.q[si][3 + k] = q[k];
        for (int k = 0; k < 7; k++) b |= !(fabsf(s.sat.q[si][k]) < 1e10f);
      } else {
        s.sat.q[si][0] += h * s.sat.x[SX_V][si][0];
        b |= !(fabsf(s.sat.q[si][0]) < 1e10f);
      }
      bad[lane] |= b;
    }
  }
}
SMJ_DEV void sat_reset_state() {
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < M.nsat) {
      const int* r = satrec(si);
      const int qa = r[SMJ_SR_QADR], nqs = r[SMJ_SR_NDOF] == 6 ? 7 : 1;
      for (int k = 0; k < nqs; k++) s.sat.q[si][k] = M.qpos0[qa + k];
      for (int k = 0; k < 6; k++) { s.sat.x[SX_V][si][k] = 0.f; s.sat.x[SX_QA][si][k] = 0.f; }
    }
  }
}

// ------------------------------------------------------------------ B.4 constraint rows with satellites
// Same rows as make_constraint ([MJ] mj_makeConstraint / mj_makeImpedance), in another ORDER: the rows that touch the main tree
// first -- equalities, friction loss and limits of main dofs, contacts with a main body on either side: rows 0 .. nd-1, the
// ones with a dense Jacobian row -- then the satellites' friction-loss rows, their active limits and the contacts among
// satellites and static geoms.  (Newton's result does not depend on the order of the rows; the oracle keeps MuJoCo's.)
int nd = 0, nd_prev = NDR, next_sat = 0;
// enter rows r0 .. r0 + n - 1 (slot `slot` of their satellite columns) in satellite si's item list; called from divergent lanes
SMJ_DEV void sat_item(int si, int r0, int n, int slot, int inf, int c = 0) {
  const int at = lds_atomic_inc(&s.sat.nitem[si]);
  if (at < NIT) { s.sat.irow[si][at] = (unsigned short)r0; s.sat.iinf[si][at] = (unsigned char)(n | (slot ? ITEM_SLOT : 0) | inf); s.sat.icon[si][at] = (unsigned char)c; }
}
SMJ_DEV void make_constraint_sat(float* pc, bool prof) {
  long long tq = prof ? smj_clock() : 0;
#define QTICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - tq); tq = t1; }
  const int nv = M.nv, neq = M.neq, nfm = M.nfric_main, nlm = M.nlimit_main, nbm = M.nbody, nsat = M.nsat;
  const int nfs = M.nfric - nfm, nls = M.nlimit - nlm;
  // clear: the dense rows the previous step used, row metadata and satellite columns of its rows
  LANES {
    for (int k = lane; k < nd_prev * JS; k += 64) (&s.J[0][0])[k] = 0.f;
  }
  ROWPASS(rb, nefc) LANES {
    const int row = lane + rb;
    if (row < NEFC) {
      s.etype[row] = CT_NONE; s.efloss[row] = 0; s.eid[row] = 0; s.epos[row] = 0; s.emargin[row] = 0; s.ediag[row] = 0;
      s.sat.esat[row][0] = -1; s.sat.esat[row][1] = -1; s.sat.erec[row] = 0; s.sat.e2[row] = 0;
      for (int k = 0; k < 6; k++) s.sat.Js[row][k] = 0.f;
    }
  }
  SYNC();
  const int cap = (M.row_limit > 0 && M.row_limit < NEFC) ? M.row_limit : NEFC;
  // ---- main static rows and limits (as make_constraint, on the main part of the row records)
  const int nstat = neq + nfm, nstat_all = neq + M.nfric;
  PL<int> act, lrow[5];
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
      s.sat.erec[e] = (short)e;
    }
    int a = 0;
    float q = 0;
    for (int k = 0; k < 5; k++) lrow[k][lane] = 0;
    if (lane < 2 * nlm) {
      const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_rowrec + (nstat_all + ol) * SMJ_RR_STRIDE, 16));
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
  uint64_t lm = wave_ballot(act);
  int row0 = nstat;
  LANES {
    if (act[lane]) {
      const int r = row0 + popc64(lm & ((1ull << lane) - 1));
      if (r < cap && r < NDR) {
        const int side = lrow[1][lane];
        s.J[r][lrow[0][lane]] = (float)(-side);
        s.etype[r] = CT_LIMIT; s.eid[r] = lane; s.sat.erec[r] = (short)(nstat_all + lane);
        s.epos[r] = side * (asf(lrow[2][lane]) - lq[lane]);
        s.emargin[r] = asf(lrow[3][lane]); s.ediag[r] = asf(lrow[4][lane]);
      }
    }
  }
  row0 += popc64(lm);
  if (row0 > cap || row0 > NDR) { row0 = cap < NDR ? cap : NDR; flags |= SMJ_FLAG_EFC_OVERFLOW | 0x1000; }
  SYNC();
  QTICK(SMJ_PROF_MC_ROWS)
  // ---- contacts, phase 1 (lane = contact): bodies, classes, dof masks, diagonal approximations
  PL<int> cact, cdimv, crow, cmain, csa, csb;   // csa / csb: satellite of geom1's / geom2's body, or -1
  PL<float> ctran, crot;
  LANES {
    int act = 0, dim = 0, mainc = 0, sa = -1, sb = -1;
    float tran = 0, rot = 0;
    if (lane < ncon) {
      const int c = lane;
      dim = s.cdim[c];
      act = s.cdist[c] < s.cmargin[c];
      const int g1 = s.cgeom1[c], g2 = s.cgeom2[c], b1 = M.geom_bodyid[g1], b2 = M.geom_bodyid[g2];
      const bool m1 = b1 > 0 && b1 < nbm, m2 = b2 > 0 && b2 < nbm;
      s.u.k.b1[c] = b1; s.u.k.b2[c] = b2;
      s.u.k.m1lo[c] = m1 ? M.k_body_dofmask_lo[b1] : 0; s.u.k.m1hi[c] = m1 ? M.k_body_dofmask_hi[b1] : 0;
      s.u.k.m2lo[c] = m2 ? M.k_body_dofmask_lo[b2] : 0; s.u.k.m2hi[c] = m2 ? M.k_body_dofmask_hi[b2] : 0;
      mainc = m1 || m2;
      sa = b1 >= nbm ? b1 - nbm : -1; sb = b2 >= nbm ? b2 - nbm : -1;
      tran = M.geom_invweight0[2 * g1] + M.geom_invweight0[2 * g2];
      rot = M.geom_invweight0[2 * g1 + 1] + M.geom_invweight0[2 * g2 + 1];
    }
    cact[lane] = act; cdimv[lane] = dim; ctran[lane] = tran; crot[lane] = rot; crow[lane] = -1; cmain[lane] = mainc; csa[lane] = sa; csb[lane] = sb;
    if (lane < NVP)
      for (int x = 0; x < 6; x++) s.u.k.cd[lane][x] = lane < nv ? cdof[lane][x] : 0.f;
  }
  // rows of the contacts that touch the main tree (dense rows): offsets by ballot prefix of the contacts' row counts; the serial
  // loop only when the rows run out (it degrades contacts one by one, as make_constraint does)
  {
    PL<int> bit;
    LANES { bit[lane] = (cact[lane] && cmain[lane]) ? cdimv[lane] & 1 : 0; }
    const uint64_t m0 = wave_ballot(bit);
    LANES { bit[lane] = (cact[lane] && cmain[lane]) ? cdimv[lane] & 2 : 0; }
    const uint64_t m1 = wave_ballot(bit);
    LANES { bit[lane] = (cact[lane] && cmain[lane]) ? cdimv[lane] & 4 : 0; }
    const uint64_t m2 = wave_ballot(bit);
    const int total = popc64(m0) + 2 * popc64(m1) + 4 * popc64(m2), lim = cap < NDR ? cap : NDR;
    if (row0 + total <= lim) {
      LANES {
        const uint64_t lt = (1ull << lane) - 1;
        if (cact[lane] && cmain[lane]) crow[lane] = row0 + popc64(m0 & lt) + 2 * popc64(m1 & lt) + 4 * popc64(m2 & lt);
      }
      row0 += total;
    } else {
      for (int c = 0; c < ncon; c++) {
        int d = wave_read(cdimv, c);
        if (!wave_read(cact, c) || !wave_read(cmain, c)) continue;
        if (row0 + d > lim) {
          flags |= SMJ_FLAG_EFC_OVERFLOW | 0x1000;
          if (d > 3 && row0 + 3 <= lim) d = 3;
          else if (row0 + 1 <= lim) d = 1;
          else continue;
          LANES { if (lane == c) { cdimv[lane] = d; s.cdim[c] = d; } }
        }
        LANES { if (lane == c) crow[lane] = row0; }
        row0 += d;
      }
    }
  }
  nd = row0;
  // ---- the satellites' friction-loss rows and active limits (every row is entered in its satellite's item list as it is made)
  LANES { if (lane >= 32 && lane - 32 < nsat) { s.sat.nitem[lane - 32] = 0; s.sat.ext[lane - 32] = -1; } }
  SYNC();
  LANES {
    if (lane < nfs) {
      const int rec = neq + nfm + lane, e = row0 + lane;
      const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_rowrec + opaque(rec) * SMJ_RR_STRIDE, 16));
      if (e < cap) {
        const int si = r[SMJ_RR_SAT], k = r[SMJ_RR_SDOF];
        s.sat.esat[e][0] = (signed char)si; s.sat.Js[e][k] = 1.f;
        sat_item(si, e, 1, 0, 0);
        s.etype[e] = CT_FRICTION; s.eid[e] = r[SMJ_RR_ID]; s.efloss[e] = asf(r[SMJ_RR_FLOSS]); s.ediag[e] = asf(r[SMJ_RR_DIAG]);
        s.sat.erec[e] = (short)rec;
      }
    }
  }
  row0 += nfs;
  if (row0 > cap) { row0 = cap; flags |= SMJ_FLAG_EFC_OVERFLOW | 0x2000; }
  LANES {
    int a = 0;
    float q = 0;
    for (int k = 0; k < 5; k++) lrow[k][lane] = 0;
    if (lane < 2 * nls) {
      const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_rowrec + (nstat_all + 2 * nlm + opaque(lane)) * SMJ_RR_STRIDE, 16));
      const int side = r[SMJ_RR_D2], si = r[SMJ_RR_SAT];
      q = s.sat.q[si][0];
      const float dist = side * (asf(r[SMJ_RR_V1]) - q);
      a = dist < asf(r[SMJ_RR_V2]);
      lrow[0][lane] = si; lrow[1][lane] = side; lrow[2][lane] = r[SMJ_RR_V1]; lrow[3][lane] = r[SMJ_RR_V2]; lrow[4][lane] = r[SMJ_RR_DIAG];
    }
    act[lane] = a; lq[lane] = q;
  }
  lm = wave_ballot(act);
  LANES {
    if (act[lane]) {
      const int r = row0 + popc64(lm & ((1ull << lane) - 1));
      if (r < cap) {
        const int side = lrow[1][lane];
        s.sat.esat[r][0] = (signed char)lrow[0][lane]; s.sat.Js[r][0] = (float)(-side);
        sat_item(lrow[0][lane], r, 1, 0, 0);
        s.etype[r] = CT_LIMIT; s.eid[r] = 2 * nlm + lane; s.sat.erec[r] = (short)(nstat_all + 2 * nlm + lane);
        s.epos[r] = side * (asf(lrow[2][lane]) - lq[lane]);
        s.emargin[r] = asf(lrow[3][lane]); s.ediag[r] = asf(lrow[4][lane]);
      }
    }
  }
  row0 += popc64(lm);
  if (row0 > cap) { row0 = cap; flags |= SMJ_FLAG_EFC_OVERFLOW | 0x2000; }
  // ---- rows of the contacts that touch no main body (same scheme)
  {
    PL<int> bit;
    LANES { bit[lane] = (cact[lane] && !cmain[lane]) ? cdimv[lane] & 1 : 0; }
    const uint64_t m0 = wave_ballot(bit);
    LANES { bit[lane] = (cact[lane] && !cmain[lane]) ? cdimv[lane] & 2 : 0; }
    const uint64_t m1 = wave_ballot(bit);
    LANES { bit[lane] = (cact[lane] && !cmain[lane]) ? cdimv[lane] & 4 : 0; }
    const uint64_t m2 = wave_ballot(bit);
    const int total = popc64(m0) + 2 * popc64(m1) + 4 * popc64(m2);
    if (row0 + total <= cap) {
      LANES {
        const uint64_t lt = (1ull << lane) - 1;
        if (cact[lane] && !cmain[lane]) crow[lane] = row0 + popc64(m0 & lt) + 2 * popc64(m1 & lt) + 4 * popc64(m2 & lt);
      }
      row0 += total;
    } else {
      for (int c = 0; c < ncon; c++) {
        int d = wave_read(cdimv, c);
        if (!wave_read(cact, c) || wave_read(cmain, c)) continue;
        if (row0 + d > cap) {
          flags |= SMJ_FLAG_EFC_OVERFLOW | 0x2000;
          if (d > 3 && row0 + 3 <= cap) d = 3;
          else if (row0 + 1 <= cap) d = 1;
          else continue;
          LANES { if (lane == c) { cdimv[lane] = d; s.cdim[c] = d; } }
        }
        LANES { if (lane == c) crow[lane] = row0; }
        row0 += d;
      }
    }
  }
  QTICK(SMJ_PROF_MC_CON)
  PL<int> over2;
  LANES { over2[lane] = 0; if (lane == 0) s.sat.ns2 = 0; }
  SYNC();
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
        // satellite columns: slot 0 = the first satellite of the pair, slot 1 = the second (both bodies satellites)
        const int sa = csa[lane], sb = csb[lane];
        int slot = 0;
        if (sa >= 0 && sb >= 0) {   // a contact between two satellites: rows of the second-slot pool
          int base = lds_atomic_add(&s.sat.ns2, dim);
          if (base + dim > NS2) { base = 0; over2[lane] = 1; }   // (pool exhausted: flagged below)
          for (int r = 0; r < dim; r++) s.sat.e2[r0 + r] = (short)(base + r);
        }
#pragma unroll
        for (int w = 0; w < 2; w++) {
          const int si = w ? sb : sa;
          if (si < 0) continue;
          const float sg = w ? 1.f : -1.f;   // geom2's body moves the contact point with +, geom1's with -  ([MJ] mj_jacDifPair)
          const int b = s.sat.body[si], jt = s.sat.jtype[si], ndof = s.sat.ndof[si];
          const float* anc = jt == JT_FREE ? s.xpos[b] : s.sat.wanc[si];
          const float rel[3] = {s.cpos[c][0] - anc[0], s.cpos[c][1] - anc[1], s.cpos[c][2] - anc[2]};
          for (int k = 0; k < 6; k++) {
            float jp[3] = {0, 0, 0}, jr[3] = {0, 0, 0};
            if (k < ndof) {
              if (jt == JT_FREE) {
                if (k < 3) jp[k] = 1.f;
                else { for (int x = 0; x < 3; x++) jr[x] = s.xmat[b][3 * x + (k - 3)]; cross3(jp, jr, rel); }
              } else if (jt == JT_SLIDE) { for (int x = 0; x < 3; x++) jp[x] = s.sat.wax[si][x]; }
              else { for (int x = 0; x < 3; x++) jr[x] = s.sat.wax[si][x]; cross3(jp, jr, rel); }
            }
#pragma unroll
            for (int r = 0; r < 6; r++)
              if (r < dim) {
                const float* ax = s.cframe[c] + 3 * (r < 3 ? r : r - 3);
                jspw(r0 + r, slot)[k] = sg * (r < 3 ? dot3(ax, jp) : dot3(ax, jr));
              }
          }
#pragma unroll
          for (int r = 0; r < 6; r++)
            if (r < dim) s.sat.esat[r0 + r][slot] = (signed char)si;
          // the satellite's item; a contact with anything but the static world couples it (dense extension of this step)
          const int ob = w ? s.u.k.b1[c] : s.u.k.b2[c];
          sat_item(si, r0, dim, slot, ITEM_CONTACT | ((ob > 0 && ob < nbm) ? ITEM_MAIN : 0), c);
          if (ob > 0) s.sat.ext[si] = 0;   // (marker; slots are handed out below)
          slot++;
        }
      }
    }
  }
  if (wave_ballot(over2)) flags |= SMJ_FLAG_EFC_OVERFLOW | 0x400;
  SYNC();
  // contacts phase 2: the main columns of the dense rows (as make_constraint: lanes = dofs, two contacts per pass)
  constexpr int CPP = 64 / NVP;
  // (only the contacts with dense rows -- the ones that touch the main tree: a handful of a kitchen's 20-40 -- two per pass)
  PL<int> hasd;
  LANES { hasd[lane] = lane < ncon && s.cefc[lane] >= 0 && s.cefc[lane] < nd; }
  uint64_t dm = wave_ballot(hasd);
  while (dm) {
    int cc[CPP];
#pragma unroll
    for (int q = 0; q < CPP; q++) { cc[q] = dm ? ffs64(dm) : -1; if (dm) dm &= dm - 1; }
    LANES {
      const int c = cc[lane / NVP < CPP ? lane / NVP : 0], d = lane % NVP;
      if (c >= 0 && d < nv) {
        const int r0 = s.cefc[c];
        if (r0 >= 0 && r0 < nd) {
          const int dim = s.cdim[c], b1 = s.u.k.b1[c], b2 = s.u.k.b2[c];
          const uint64_t m1 = mk64(s.u.k.m1lo[c], s.u.k.m1hi[c]), m2 = mk64(s.u.k.m2lo[c], s.u.k.m2hi[c]);
          const int in1 = (int)((m1 >> d) & 1), in2 = (int)((m2 >> d) & 1);
          const float sg = (float)(in2 - in1);
          if (sg != 0.f) {
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
  nd_prev = nd;
#ifdef SMJ_EMUL
  if ((flags & SMJ_FLAG_EFC_OVERFLOW) && getenv("SMJ_SAT_TRACE")) fprintf(stderr, "row overflow: nd %d nefc %d ncon %d\n", nd, nefc, ncon);
#endif
  SYNC();
  QTICK(SMJ_PROF_MC_JAC)
  // ---- extension slots of the coupled satellites, satellite-satellite contacts
  PL<int> wantx;
  LANES {
    const int si = lane - 32;
    int want = 0;
    if (lane >= 32 && si < nsat) {
      want = s.sat.ext[si] == 0;
      if (s.sat.nitem[si] > NIT) { s.sat.nitem[si] = NIT; want |= 2; }
    }
    wantx[lane] = want;
  }
  {
    PL<int> b0, b1;
    LANES { b0[lane] = wantx[lane] & 1; b1[lane] = wantx[lane] & 2; }
    const uint64_t xm = wave_ballot(b0);
    if (wave_ballot(b1)) {   // more row items on one satellite than NIT: its later rows are not in its block
      flags |= SMJ_FLAG_EFC_OVERFLOW | 0x100;
#ifdef SMJ_EMUL
      if (getenv("SMJ_SAT_TRACE")) fprintf(stderr, "item overflow\n");
#endif
    }
    next_sat = popc64(xm);
    LANES {
      const int si = lane - 32;
      if (lane >= 32 && si < nsat) {
        const int e = (xm >> lane) & 1 ? popc64(xm & ((1ull << lane) - 1)) : -1;
        s.sat.ext[si] = e < NXS ? e : -1;
        if (e >= 0 && e < NXS) s.sat.xs[e] = si;
      }
    }
#ifdef SMJ_EMUL
    if (next_sat > 0) smj_emul_ext_steps++;   // (tests: the dense extension is exercised)
    if (popc64(wave_ballot(b0)) > 0 && getenv("SMJ_SAT_TRACE")) fprintf(stderr, "ext step: %d satellites coupled, nd %d nefc %d\n", next_sat, nd, row0);
#endif
    if (next_sat > NXS) { next_sat = NXS; flags |= SMJ_FLAG_EFC_OVERFLOW | 0x200; }   // beyond the extension's capacity the coupling blocks of the surplus satellites are dropped: flagged
    PL<int> ss;
    LANES { ss[lane] = lane < ncon && s.cefc[lane] >= 0 && csa[lane] >= 0 && csb[lane] >= 0; }
    const uint64_t sm = wave_ballot(ss);
    LANES {
      if (lane < NSS) s.sat.sscon[lane] = -1;
    }
    SYNC();
    LANES {
      if (ss[lane]) {
        const int at = popc64(sm & ((1ull << lane) - 1));
        if (at < NSS) s.sat.sscon[at] = lane;
      }
    }
    if (popc64(sm) > NSS) {
      flags |= SMJ_FLAG_EFC_OVERFLOW | 0x400;
#ifdef SMJ_EMUL
      if (getenv("SMJ_SAT_TRACE")) fprintf(stderr, "sat-sat contact overflow %d\n", popc64(sm));
#endif
    }
  }
  SYNC();
  QTICK(SMJ_PROF_MC_ITEMS)
  // ---- impedance, R, K, B  [MJ] mj_makeImpedance (row records through erec: rows do not sit at their record's index here)
  ROWPASS(rb, nefc) LANES {
    const int i = lane + rb;
    if (i < nefc) {
      const int t = s.etype[i], id = s.eid[i];
      float solref[2], solimp[5];
      if (t == CT_CONTACT_FRICTIONLESS || t == CT_CONTACT_ELLIPTIC) {
        solref[0] = s.csolref[id][0]; solref[1] = s.csolref[id][1];
        for (int k = 0; k < 5; k++) solimp[k] = s.csolimp[id][k];
      } else {
        const int* r = M.k_rowrec + (int)s.sat.erec[i] * SMJ_RR_STRIDE;
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
  QTICK(SMJ_PROF_MC_IMP)
#undef QTICK
}

// x <- H^-1 x for the extended system in s.u.n.H: main block A (NVS x NVS), extension block C (nx = 6 next rows / columns from
// NVS on), coupling B.  The extension is eliminated first -- Gauss-Jordan on the rows [C | Bt | rhs], lane = COLUMN, so that the
// nx pivots cost nx - 1 fused multiply-adds per lane each -- which leaves Y = C^-1 Bt and z = C^-1 x_ext; the main block takes the
// Schur complement A - B Y, is solved in registers like any other step (solve_H), and x_ext = z - Y x_main.  (The first version
// ran the LDS Gauss-Jordan of the 64-column builds over the whole order-50 system: 60-70 k cycles per solve, this: under 20 k.)
SMJ_DEV void solve_ext_schur(PL<float>& x, int next) {
  const int nx = 6 * next, o = NVS;
  LANES { if (lane >= o && lane < o + nx) s.u.n.H[lane][NXV] = x[lane]; }
  SYNC();
  for (int k = 0; k < nx; k++) {
    const float rp = fast_rcp(fmaxf(uni(s.u.n.H[o + k][o + k]), 1e-30f));
    LANES {
      // lane -> column: the main columns, the extension columns beyond the pivot's, the right-hand side
      const int col = lane < NVS ? lane : (lane - NVS < nx ? o + (lane - NVS) : (lane == 63 ? NXV : -1));
      if (col >= 0 && (col < o || col > o + k)) {
        const float pv = s.u.n.H[o + k][col];
        if (pv != 0.f)
          for (int i = 0; i < nx; i++)
            if (i != k) s.u.n.H[o + i][col] -= s.u.n.H[o + i][o + k] * rp * pv;
      }
    }
    SYNC();
  }
  // Schur complement and right-hand side of the main block: lane = main row d
  LANES {
    if (lane < NVS) {
      float bd[6 * NXS], acc = 0.f;
#pragma unroll
      for (int i = 0; i < 6 * NXS; i++) bd[i] = i < nx ? s.u.n.H[lane][o + i] * fast_rcp(fmaxf(s.u.n.H[o + i][o + i], 1e-30f)) : 0.f;   // B[d][i] / C_ii
#pragma unroll
      for (int i = 0; i < 6 * NXS; i++) acc += i < nx ? bd[i] * s.u.n.H[o + i][NXV] : 0.f;
      x[lane] -= acc;
      for (int e = 0; e < NVS; e++) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 6 * NXS; i++) v += i < nx ? bd[i] * s.u.n.H[o + i][e] : 0.f;
        s.u.n.H[lane][e] -= v;
      }
    }
  }
  SYNC();
  PL<float> xm;
  LANES { xm[lane] = lane < M.nv ? x[lane] : 0.f; }
  solve_H(xm);
  LANES { if (lane < NVS) s.tmp[lane] = lane < M.nv ? xm[lane] : 0.f; }
  SYNC();
  LANES {
    if (lane < NVS) x[lane] = s.tmp[lane];
    else if (lane - NVS < nx) {
      const int i = lane - NVS;
      float v = s.u.n.H[o + i][NXV];
      for (int d = 0; d < NVS; d++) v -= s.u.n.H[o + i][d] * s.tmp[d];
      x[lane] = v * fast_rcp(fmaxf(s.u.n.H[o + i][o + i], 1e-30f));
    }
  }
  SYNC();
}

// ------------------------------------------------------------------ collision against the static world (a kitchen's fixtures)
// The world body's collision geoms never move: their frames are model constants (DevModel::k_sgrec), they take no slot in the
// collision stage's LDS cache and no pair of theirs is in the table the kernels scan.  Broadphase: lane = moving geom (cache
// slot), which visits the cells of the uniform grid (DevModel::k_grid_*) its bounding sphere overlaps and tests the static geoms
// listed there -- a pair is reported from the FIRST cell the two cell ranges share, so never twice -- with the same two tests as
// the moving-moving pairs (bounding spheres, then the six face axes of the two oriented boxes) and MuJoCo's pair filters
// (k_spair: -1 = filtered).  The survivors are put in pair-table order and go through the same narrowphase, the static geom staged
// in cache slot NCG.  Same contacts as the oracle's scan of the whole table (oracle/smj_oracle.c collision()).
SMJ_DEV void stage_static(int sg) {
  LANES {
    if (lane < 25) {   // one word per lane: the loader's world-frame record (DevModel::k_sgw) into the cache arrays' staging slot
      const float v = M.k_sgw[sg * 32 + lane];
      float* dst = lane < 3 ? &s.u.c.pos[NCG][lane] : lane < 12 ? &s.u.c.mat[NCG][lane - 3] : lane < 15 ? &s.u.c.cen[NCG][lane - 12]
                 : lane < 18 ? &s.u.c.half[NCG][lane - 15] : lane < 21 ? &s.u.c.ccen[NCG][lane - 18] : lane < 24 ? &s.u.c.size[NCG][lane - 21]
                 : reinterpret_cast<float*>(&s.u.c.meta[NCG]);
      *dst = v;
    }
  }
  SYNC();
}
SMJ_DEV void collision_static(float* pc, bool prof) {
  const int nsg = M.nsgeom;
  if (nsg == 0 || M.nstatpair == 0) return;
  LANES { if (lane == 0) s.u.c.sl_n = 0; }
  SYNC();
  const long long tb0 = prof ? smj_clock() : 0;
  // Broadphase in two phases (below).  The first version walked the grid cells of every moving geom with lane = moving geom
  // (dependent, uncoalesced loads per cell: 600 k cycles per step), the second ran sphere test, filter look-up and box test in one
  // loop with lane = static geom (the lanes that passed the sphere test paid a global load inside the loop, the others waited:
  // 260 k).  The uniform grid (k_grid_*) stays in the blob for scenes with thousands of static geoms.
  const float gm = M.grid_margin;
  const int ncg = M.ncgeom;
  // phase 1 (only when a moving geom has travelled SMJ_SB_SLACK since the list was built, or at the start of a launch), lane =
  // static geom: its world AABB against every moving geom's bounding sphere inflated by the slack.  22 k tests: 116 k cycles when
  // it ran every step -- a launch-long LDS list of the ~200 near pairs makes it a once-in-ten-steps cost.
  {
    PL<int> mv;
    LANES {
      int m = 0;
      for (int c = lane; c < ncg; c += 64) {
        const float d[3] = {s.u.c.cen[c][0] - s.sat.refcen[c][0], s.u.c.cen[c][1] - s.sat.refcen[c][1], s.u.c.cen[c][2] - s.sat.refcen[c][2]};
        m |= !(dot3(d, d) < SMJ_SB_SLACK * SMJ_SB_SLACK);
      }
      mv[lane] = m;
    }
    if (!uni(s.sat.cand_ok) || wave_ballot(mv) != 0) {
      // bounding radii of the moving geoms, staged for this rebuild only (round 5: they were staged every step -- two global-load
      // round trips per step for a list that is rebuilt once in ten)
      for (int c0 = 0; c0 < ncg; c0 += 64) {
        LANES { if (c0 + lane < ncg) s.u.c.mc_r[c0 + lane] = asf(M.k_cgrec[opaque(c0 + lane) * SMJ_CG_STRIDE + SMJ_CG_RBOUND]); }
      }
      LANES {
        if (lane == 0) { s.sat.ncand = 0; s.sat.cand_ok = 1; }
        for (int c = lane; c < ncg; c += 64)
          for (int k = 0; k < 3; k++) s.sat.refcen[c][k] = s.u.c.cen[c][k];
      }
      SYNC();
      for (int g0 = 0; g0 < nsg; g0 += 64) {
        LANES {
          const int sg = g0 + lane;
          if (sg < nsg) {
            const Vec4 blo = *reinterpret_cast<const Vec4*>(M.k_sg_bound + 8 * opaque(sg)), bhi = *reinterpret_cast<const Vec4*>(M.k_sg_bound + 8 * opaque(sg) + 4);
            // eight moving geoms per round: centres / radii fetched first, the tests leave a bit mask, a round with a hit enters the append loop
            for (int c0 = 0; c0 < ncg; c0 += 8) {
              float cx[8], cy[8], cz[8], cr[8];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const int c = c0 + u < ncg ? c0 + u : ncg - 1;
                cx[u] = s.u.c.cen[c][0]; cy[u] = s.u.c.cen[c][1]; cz[u] = s.u.c.cen[c][2]; cr[u] = s.u.c.mc_r[c];
              }
              unsigned mask = 0;
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const float dx = fmaxf(0.f, fmaxf(blo.x - cx[u], cx[u] - bhi.x)), dy = fmaxf(0.f, fmaxf(blo.y - cy[u], cy[u] - bhi.y)),
                            dz = fmaxf(0.f, fmaxf(blo.z - cz[u], cz[u] - bhi.z)), rr = cr[u] + gm + 2.f * SMJ_SB_SLACK;
                mask |= (c0 + u < ncg && dx * dx + dy * dy + dz * dz <= rr * rr) ? 1u << u : 0u;
              }
              while (mask) {
                const int u = __builtin_ctz(mask);
                mask &= mask - 1;
                const int at = lds_atomic_inc(&s.sat.ncand);
                if (at < NCAND) s.sat.cand[at] = (unsigned short)(sg | ((c0 + u) << 9));   // (512 static geoms x 128 cache slots)
              }
            }
          }
        }
      }
      SYNC();
      if (uni(s.sat.ncand) > NCAND) {   // more near pairs than the list holds: flagged, and the list is rebuilt next step
        LANES { if (lane == 0) { s.sat.ncand = NCAND; s.sat.cand_ok = 0; } }
        flags |= SMJ_FLAG_CON_OVERFLOW | 0x800;
        SYNC();
      }
    }
  }
  if (prof) pc[SMJ_PROF_PROJECT] += (float)(smj_clock() - tb0);
  const int ncand = uni(s.sat.ncand);
  // phase 2, lane = candidate: MuJoCo's pair filter (k_spair), then the six face axes of the two oriented boxes
  // (the filter look-ups of all rounds are issued first -- independent global loads, one round trip for the lot instead of one per round)
  constexpr int NRND = (NCAND + 63) / 64;
  PL<int> cw[NRND], csid[NRND];
  LANES {
#pragma unroll
    for (int r = 0; r < NRND; r++) {
      const int k = 64 * r + lane;
      int w = 0, sid = -1;
      if (k < ncand) { w = s.sat.cand[k]; sid = M.k_spair[(w >> 9) * nsg + (w & 511)]; }
      cw[r][lane] = w; csid[r][lane] = sid;
    }
  }
#pragma unroll
  for (int r = 0; r < NRND; r++) {
    if (64 * r >= ncand) break;   // uniform
    LANES {
      {
        const int w = cw[r][lane], sg = w & 511, c = w >> 9;
        const int sid = csid[r][lane];
        if (sid >= 0) {
          const float* sw = M.k_sgw + 32 * sg;
          float Rb[9], hb[3], Ra[9], ha[3], Rm[3][3], ta[3], tb[3];
          for (int q = 0; q < 9; q++) { Rb[q] = sw[3 + q]; Ra[q] = s.u.c.mat[c][q]; }
          for (int q = 0; q < 3; q++) { hb[q] = sw[15 + q]; ha[q] = s.u.c.half[c][q]; }
          const float dv[3] = {sw[12] - s.u.c.cen[c][0], sw[13] - s.u.c.cen[c][1], sw[14] - s.u.c.cen[c][2]};
          bool hit = true;
          for (int i = 0; i < 3; i++) {
            ta[i] = Ra[i] * dv[0] + Ra[3 + i] * dv[1] + Ra[6 + i] * dv[2];
            tb[i] = Rb[i] * dv[0] + Rb[3 + i] * dv[1] + Rb[6 + i] * dv[2];
            for (int j = 0; j < 3; j++) Rm[i][j] = fabsf(Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]);
          }
          for (int q = 0; q < 3; q++) {
            if (fabsf(ta[q]) > ha[q] + (Rm[q][0] * hb[0] + Rm[q][1] * hb[1] + Rm[q][2] * hb[2]) + gm) hit = false;
            if (fabsf(tb[q]) > hb[q] + (Rm[0][q] * ha[0] + Rm[1][q] * ha[1] + Rm[2][q] * ha[2]) + gm) hit = false;
          }
          if (hit) {
            const int at = lds_atomic_inc(&s.u.c.sl_n);
            if (at < NSURV) { s.u.c.sl_sid[at] = sid; s.u.c.sl_c[at] = (unsigned char)c; }
          }
        }
      }
    }
  }
  SYNC();
  if (prof) pc[SMJ_PROF_S_BROAD] += (float)(smj_clock() - tb0);
  int n = uni(s.u.c.sl_n);
  if (n > NSURV) { n = NSURV; flags |= SMJ_FLAG_CON_OVERFLOW | 0x800; }
  if (prof) pc[SMJ_PROF_C_NSPHERE] += (float)n;
#ifdef SMJ_EMUL
  if (getenv("SMJ_SAT_TRACE")) { fprintf(stderr, "static candidates %d survivors %d:", ncand, n); for (int i = 0; i < n; i++) fprintf(stderr, " %d", M.k_sprec[s.u.c.sl_sid[i] * SMJ_CP_STRIDE + SMJ_CP_PAIR]); fprintf(stderr, "\n"); }
#endif
  if (n == 0) return;
  // pair-table order: rank of every survivor among the pair indices (all different)
  LANES {
    for (int i = lane; i < n; i += 64) {
      const int key = s.u.c.sl_sid[i];
      int rank = 0;
      for (int j = 0; j < n; j++) rank += s.u.c.sl_sid[j] < key;
      s.u.c.sl_ord[rank] = (unsigned char)i;
    }
  }
  SYNC();
  float* const sepbase = (S.sepcache && M.sep_cache) ? S.sepcache + (size_t)env * (SMJ_SEP_SLOTS * 4) : nullptr;
  const long long tn0 = prof ? smj_clock() : 0;
  for (int k0 = 0; k0 < n; k0 += 64) {
    // the survivors' separating directions, fetched lane-parallel ahead of the serial loop (the upper half of the env's slots: the moving-moving
    // pairs -- worked by the other wavefront in the two-wavefront builds -- keep to the lower half; the tag tells whose entry it is)
    PL<float> sdx, sdy, sdz;
    PL<int> stag, ssid;
    LANES {
      Vec4 e = {0.f, 0.f, 0.f, 0.f};
      int sid = -1;
      if (k0 + lane < n) {
        sid = s.u.c.sl_sid[s.u.c.sl_ord[k0 + lane]];
        if (sepbase) e = *reinterpret_cast<const Vec4*>(sepbase + 4 * (SMJ_SEP_SLOTS / 2 + ((sid * 7 + 29) & (SMJ_SEP_SLOTS / 2 - 1))));
      }
      sdx[lane] = e.x; sdy[lane] = e.y; sdz[lane] = e.z; stag[lane] = __builtin_bit_cast(int, e.w); ssid[lane] = sid;
    }
    const int m = n - k0 < 64 ? n - k0 : 64;
    // stored manifolds (DevState::mcache), all survivors of the batch at once: lane = pair compares the poses of the pair's two
    // bodies with the entry's and, where neither has moved, writes the entry's contacts (ballot prefix of the counts)
    PL<int> done;
    LANES { done[lane] = 0; }
    if (S.mcache && M.manifold_cache) {
      PL<int> cnt;
      LANES {
        int c = 0;
        if (lane < m) {
          const int sid = ssid[lane], tag = 0x40000000 | sid;
          const int* r = M.k_sprec + sid * SMJ_CP_STRIDE;
          const float* mc = mc_entry(tag);
          const Vec4* m4 = reinterpret_cast<const Vec4*>(mc);
          const Vec4 a = m4[0], b = m4[1], cc = m4[2], d = m4[3];
          const float w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, cc.x, cc.y, cc.z, cc.w, d.x, d.y, d.z, d.w};
          bool ok = __builtin_bit_cast(int, w[0]) == tag;
          const int b1 = r[SMJ_CP_B1], b2 = r[SMJ_CP_B2];
          for (int k = 0; k < 7; k++) {
            ok = ok && fabsf((k < 3 ? s.xpos[b1][k] : s.xquat[b1][k - 3]) - w[1 + k]) <= SMJ_MC_EPS;
            ok = ok && fabsf((k < 3 ? s.xpos[b2][k] : s.xquat[b2][k - 3]) - w[8 + k]) <= SMJ_MC_EPS;
          }
          if (ok) { c = (int)w[15]; c = c < 0 ? 0 : c > 5 ? 5 : c; done[lane] = 1; }
#ifdef SMJ_EMUL
          else if (getenv("SMJ_SAT_TRACE")) {
            float dmax = 0.f;
            for (int k = 0; k < 7; k++) dmax = fmaxf(dmax, fabsf((k < 3 ? s.xpos[b1][k] : s.xquat[b1][k - 3]) - w[1 + k])), dmax = fmaxf(dmax, fabsf((k < 3 ? s.xpos[b2][k] : s.xquat[b2][k - 3]) - w[8 + k]));
            fprintf(stderr, "  mcache miss: pair %d tag %s stored n %d pose delta %.2e\n", r[SMJ_CP_PAIR], __builtin_bit_cast(int, w[0]) == tag ? "ok" : "OTHER", (int)w[15], dmax);
          }
#endif
        }
        cnt[lane] = c;
      }
      PL<int> bit;
      LANES { bit[lane] = cnt[lane] & 1; }
      const uint64_t m0 = wave_ballot(bit);
      LANES { bit[lane] = cnt[lane] & 2; }
      const uint64_t m1 = wave_ballot(bit);
      LANES { bit[lane] = cnt[lane] & 4; }
      const uint64_t m2 = wave_ballot(bit);
      const int total = popc64(m0) + 2 * popc64(m1) + 4 * popc64(m2);
      if (total) {
        // (side by side with the second wavefront the slots are claimed first, and a batch that does not fit is not written at all)
        const bool fits = (SMJ_SPLIT_COLLIDE && split_on) ? con_claim(total) : ncon + total <= NCON;
        LANES {
          const int c = (fits || !(SMJ_SPLIT_COLLIDE && split_on)) ? cnt[lane] : 0;
          if (c) {
            const uint64_t lt = (1ull << lane) - 1;
            const int off = ncon + popc64(m0 & lt) + 2 * popc64(m1 & lt) + 4 * popc64(m2 & lt);
            const int sid = ssid[lane];
            const int* r = M.k_sprec + sid * SMJ_CP_STRIDE;
            const float* mc = mc_entry(0x40000000 | sid);
            const float nrm[3] = {mc[16], mc[17], mc[18]};
            const int b1 = r[SMJ_CP_B1], b2 = r[SMJ_CP_B2];
            float x0a[3], q0a[4], x0b[3], q0b[4], x1a[3], q1a[4], x1b[3], q1b[4], dxa[3], dta[3], dxb[3], dtb[3];
            for (int k = 0; k < 3; k++) { x0a[k] = mc[1 + k]; x0b[k] = mc[8 + k]; x1a[k] = s.xpos[b1][k]; x1b[k] = s.xpos[b2][k]; }
            for (int k = 0; k < 4; k++) { q0a[k] = mc[4 + k]; q0b[k] = mc[11 + k]; q1a[k] = s.xquat[b1][k]; q1b[k] = s.xquat[b2][k]; }
            smj_mc_motion(x0a, q0a, x1a, q1a, dxa, dta);
            smj_mc_motion(x0b, q0b, x1b, q1b, dxb, dtb);
            for (int k = 0; k < c; k++)
              if (off + k < NCON) {
                float p3[3] = {mc[20 + 4 * k], mc[21 + 4 * k], mc[22 + 4 * k]};
                float dist = mc[19 + 4 * k];
                smj_mc_carry(x0a, dxa, dta, x0b, dxb, dtb, nrm, dist, p3);
                write_contact(off + k, r, dist, p3, nrm);
              }
          }
        }
        if (!fits) { flags |= SMJ_FLAG_CON_OVERFLOW | 0x4000; if (!(SMJ_SPLIT_COLLIDE && split_on)) ncon = NCON; }
        else ncon += total;
        SYNC();
#ifdef SMJ_EMUL
        smj_emul_mc_hits += popc64(wave_ballot(done));
#endif
      }
    }
    for (int l = 0; l < m; l++) {
      if (wave_read(done, l)) continue;
      const int sid = wave_read(ssid, l);
      const int* r = static_cast<const int*>(__builtin_assume_aligned(M.k_sprec + sid * SMJ_CP_STRIDE, 16));
      const int S1 = uni(r[SMJ_CP_S1]), S2 = uni(r[SMJ_CP_S2]);
      stage_static(S1 < 0 ? -1 - S1 : -1 - S2);
      const float sd[3] = {wave_read(sdx, l), wave_read(sdy, l), wave_read(sdz, l)};
      const int tag = 0x40000000 | sid;
      narrow_pair(r, S1 < 0 ? NCG : S1, S2 < 0 ? NCG : S2, sepbase ? sepbase + 4 * (SMJ_SEP_SLOTS / 2 + ((sid * 7 + 29) & (SMJ_SEP_SLOTS / 2 - 1))) : nullptr, tag,
                  sepbase && wave_read(stag, l) == tag, sd, pc, prof, false);
      SYNC();
    }
  }
  if (prof) pc[SMJ_PROF_S_NARROW] += (float)(smj_clock() - tn0);
}
