// PGS with satellites: constraint islands.  Included inside StepKernel (NSAT > 0), after pgs_block / residual_refresh.
//
// [MJ] mj_solPGS sweeps every row in order; rows that share no dof do not see each other (A = J M^-1 J' + R is block diagonal
// over the islands of the constraint graph), so the sweeps of different islands commute and can run side by side:
//   * the DENSE system -- the rows that touch the main tree (rows 0 .. nd-1) and every row of a satellite that is coupled to the
//     main tree or to another satellite in this step (SatMem::ext >= 0: the same set Newton's dense extension holds) -- is swept
//     as in the other builds: A as a packed triangle in LDS (s.u.pa, NEFC_P rows), residuals and forces in registers, lane = row;
//   * every other satellite (an object resting on a counter, a door against its stop) is an island of its own and is swept BY ITS
//     OWN LANE, all of them at once, in velocity space: with M_s = L D L' the satellite's columns of its rows are whitened in
//     place, y_i = D^-1/2 L^-1 J_i', so that A_ij = y_i . y_j (+ R_i), and the lane keeps z = sum_i y_i f_i: the residual of a
//     row is y_i . z + R_i f_i + b_i (6 multiply-adds, no matrix), a contact's DIM x DIM block of A is DIM (DIM + 1) / 2 dots.
// One sweep = the dense rows, then the satellite lanes; the improvement is summed over all of them and tested against
// opt.tolerance as MuJoCo does (one iteration count for the whole system: the result equals the serial sweep's up to rounding,
// the cost of a sweep is the dense island's plus the LARGEST satellite island's instead of the sum of all).
int ndp = 0;   // rows of the dense system of this step
static_assert(sizeof(short) * (NEFC + NEFC_P) <= sizeof(SatMem::Hb), "row <-> dense-index tables live in the (Newton-only) Hb blocks");
SMJ_DEV short* pgs_didx() { return reinterpret_cast<short*>(&s.sat.Hb[0][0]); }   // row -> index in the dense system, or -1
SMJ_DEV short* pgs_drow() { return pgs_didx() + NEFC; }                            // index in the dense system -> row

struct SatFac { float L[15], rs[6], sd[6]; };   // M_s = L D L': unit lower L (strict part, row major), D^-1/2, D^1/2
SMJ_DEV static int sl6(int i, int k) { return (i * (i - 1)) / 2 + k; }
SMJ_DEV static void sat_factor6(const float* A, SatFac& F) {
  float Dv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float d = A[tri6(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d -= F.L[sl6(j, k)] * F.L[sl6(j, k)] * Dv[k];
    d = fmaxf(d, 1e-30f);
    Dv[j] = d;
    F.rs[j] = fast_rsqrt(d); F.sd[j] = d * F.rs[j];
    const float inv = F.rs[j] * F.rs[j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float v = A[tri6(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) v -= F.L[sl6(i, k)] * F.L[sl6(j, k)] * Dv[k];
      F.L[sl6(i, j)] = v * inv;
    }
  }
}
SMJ_DEV static void sat_whiten(const SatFac& F, float* x) {   // x <- D^-1/2 L^-1 x
#pragma unroll
  for (int i = 1; i < 6; i++)
#pragma unroll
    for (int k = 0; k < i; k++) x[i] -= F.L[sl6(i, k)] * x[k];
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] *= F.rs[i];
}

// one elliptic contact of a satellite island, by the satellite's lane  [MJ] mj_solPGS elliptic branch, as pgs_block
template <int DIM>
SMJ_DEV float pgs_block_lane(int r0, int c, float* z) {
  constexpr int NF = DIM - 1;
  float res[DIM], old[DIM], f[DIM], v1[DIM], At[DIM * DIM], mu[NF], Ac[NF * NF], a0[NF];
#pragma unroll
  for (int p = 0; p < DIM; p++) {
    const float* yp = s.sat.Js[r0 + p];
    float a = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) a += yp[k] * z[k];
    old[p] = s.ef[r0 + p]; f[p] = old[p];
    res[p] = a + s.eR[r0 + p] * old[p] + s.eb[r0 + p];
#pragma unroll
    for (int q = 0; q <= p; q++) {
      const float* yq = s.sat.Js[r0 + q];
      float t = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) t += yp[k] * yq[k];
      if (q == p) t += s.eR[r0 + p];
      At[p * DIM + q] = t; At[q * DIM + p] = t;
    }
  }
  float denom = 0, num = 0;
#pragma unroll
  for (int p = 0; p < DIM; p++) {
    float a = 0;
#pragma unroll
    for (int q = 0; q < DIM; q++) a += At[p * DIM + q] * old[q];
    v1[p] = a; denom += old[p] * a; num += old[p] * res[p];
  }
#pragma unroll
  for (int j = 0; j < NF; j++) {
    mu[j] = s.cfric[c][j]; a0[j] = At[(j + 1) * DIM];
#pragma unroll
    for (int q = 0; q <= j; q++) Ac[j * NF + q] = At[(j + 1) * DIM + q + 1];
  }
  if (f[0] < SMJ_MINVAL) {   // normal update
    f[0] -= res[0] * fast_rcp(At[0]);
    if (f[0] < 0) f[0] = 0;
#pragma unroll
    for (int j = 1; j < DIM; j++) f[j] = 0;
  } else if (denom >= SMJ_MINVAL) {   // ray update
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
  float bc[NF], v[NF];
#pragma unroll
  for (int j = 0; j < NF; j++) bc[j] = res[j + 1] - v1[j + 1] + a0[j] * f[0];
  if (f[0] < SMJ_MINVAL) {
#pragma unroll
    for (int j = 1; j < DIM; j++) f[j] = 0;
  } else {
    float la = s.eK[c];   // (eK is free once aref is known: the contact's QCQP multiplier of the last sweep)
    const int active = qcqp<NF>(v, Ac, bc, mu, f[0], la, M.qcqp_exact != 0);
    s.eK[c] = la;
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
#pragma unroll
  for (int p = 0; p < DIM; p++) {
    float a = 0;
#pragma unroll
    for (int q = 0; q < DIM; q++) a += At[p * DIM + q] * delta[q];
    change += delta[p] * (0.5f * a + res[p]);
  }
  if (change > SMJ_PGS_GUARD) return 0.f;
#pragma unroll
  for (int p = 0; p < DIM; p++) {
    const float* yp = s.sat.Js[r0 + p];
    s.ef[r0 + p] = old[p] + delta[p];
#pragma unroll
    for (int k = 0; k < 6; k++) z[k] += yp[k] * delta[p];
  }
  return -change;
}

// z = sum over the satellite's rows of y_i f_i (both slots), by the satellite's lane
SMJ_DEV void sat_yf(int si, float* z) const {
  for (int k = 0; k < 6; k++) z[k] = 0.f;
  for (int it = 0; it < s.sat.nitem[si]; it++) {
    const int r0 = s.sat.irow[si][it], inf = s.sat.iinf[si][it], n = inf & ITEM_N, u = (inf & ITEM_SLOT) ? 1 : 0;
    for (int p = 0; p < n; p++) {
      const float f = s.ef[r0 + p];
      const float* y = jsp(r0 + p, u);
      for (int k = 0; k < 6; k++) z[k] += y[k] * f;
    }
  }
}

// One sweep of the satellite islands, each by its own lane (lane = 32 + satellite); returns the sum of their improvements.
// zs: z = sum y_i f_i of the lane's island (refreshed from the forces every 8th sweep), frozen: islands that have stopped.
SMJ_DEV float sat_lanes_sweep(PL<float[6]>& zs, PL<int>& frozen, int iter, float scale, float istop) {
  const int nsat = M.nsat;
  if (iter > 0 && (iter & 7) == 0) {
    LANES { const int si = lane - 32; if (lane >= 32 && si < nsat && s.sat.ext[si] < 0 && !frozen[lane]) sat_yf(si, zs[lane]); }
  }
    PL<float> imp;
  LANES {
    float im = 0;
    const int si = lane - 32;
    if (lane >= 32 && si < nsat && s.sat.ext[si] < 0 && !frozen[lane]) {
      float* z = zs[lane];
      for (int it = 0; it < s.sat.nitem[si]; it++) {
        const int r0 = s.sat.irow[si][it], inf = s.sat.iinf[si][it], n = inf & ITEM_N;
        if ((inf & ITEM_CONTACT) && n >= 3) {
          const int c = s.sat.icon[si][it];
          if (n == 3) im += pgs_block_lane<3>(r0, c, z);
          else if (n == 4) im += pgs_block_lane<4>(r0, c, z);
          else im += pgs_block_lane<6>(r0, c, z);
        } else {
          const float* y = s.sat.Js[r0];
          const int t = s.etype[r0];
          const float R = s.eR[r0], old = s.ef[r0], fl = s.efloss[r0];
          float a = 0, aii = R;
          for (int k = 0; k < 6; k++) { a += y[k] * z[k]; aii += y[k] * y[k]; }
          const float res = a + R * old + s.eb[r0];
          const float lo = t == CT_EQUALITY ? -INFINITY : t == CT_FRICTION ? -fl : 0.f, hi = t == CT_FRICTION ? fl : INFINITY;
          const float fn = fminf(hi, fmaxf(lo, old - res * fast_rcp(aii)));
          float delta = fn - old;
          float change = delta * (0.5f * aii * delta + res);
          if (change > SMJ_PGS_GUARD) { delta = 0; change = 0; }
          im -= change;
          s.ef[r0] = old + delta;
          for (int k = 0; k < 6; k++) z[k] += y[k] * delta;
        }
      }
    }
    imp[lane] = im;
#ifdef SMJ_EMUL
    if (lane >= 32 && si < nsat && s.sat.ext[si] < 0 && s.sat.nitem[si] > 0) { smj_emul_isl_total++; smj_emul_isl_swept += !frozen[lane]; }
#endif
    if (im * scale < istop) frozen[lane] = 1;
  }
  return wave_sum(imp);
}

#ifdef SMJ_TWO_WAVES
// The second wavefront of the env (smj_kernels_satp.hip): waits for the first one to reach the sweeps of a step, sweeps the
// satellite islands beside its dense sweeps -- one barrier per sweep, where the two exchange their improvements and take the same
// decision --, and leaves when the first wavefront is through with the launch.  Mailbox: SatMem::x[SX_MV][0] (Newton only):
// [0..3] improvements (sweep parity x wavefront), [4] command; z of the islands comes over in x[SX_SRCH].
enum { PGS2_RUN = 1, PGS2_EXIT = 2 };
SMJ_DEV void pgs_helper() {
  for (;;) {
    WG_BARRIER();
    if (uni(__builtin_bit_cast(int, s.sat.x[SX_MV][0][4])) != PGS2_RUN) return;
    PL<float[6]> zs;
    PL<int> frozen;
    LANES {
      frozen[lane] = 0;
      for (int k = 0; k < 6; k++) zs[lane][k] = (lane >= 32 && lane - 32 < M.nsat) ? s.sat.x[SX_SRCH][lane - 32][k] : 0.f;
    }
    const float scale = 1.0f / (M.meaninertia * (float)(M.nv_all > 1 ? M.nv_all : 1));
    const float istop = M.pgs_island_stop ? M.tolerance * (1.f / 64) : -1.f;
    for (int iter = 0; iter < M.iterations; iter++) {
      const float mine = sat_lanes_sweep(zs, frozen, iter, scale, istop);
      LANES { if (lane == 0) s.sat.x[SX_MV][0][2 * (iter & 1) + 1] = mine; }
      WG_BARRIER();
      const float improvement = (uni(s.sat.x[SX_MV][0][2 * (iter & 1)]) + mine) * scale;
      if (!M.pgs_fixed_iter && improvement < M.tolerance) break;
    }
  }
}
#endif

// Identity of a constraint row from one step to the next (option pgs_dual_ws): a static / limit row is its record (DevModel::k_rowrec:
// one per equality, friction-loss dof and joint-limit side); a contact row is (collision pair, ordinal of the contact within the pair's
// manifold -- a pair's contacts are contiguous in the list --, row within the contact).
SMJ_DEV int pgs_row_key(int row) const {
  const int t = s.etype[row];
  if (t == CT_CONTACT_ELLIPTIC || t == CT_CONTACT_FRICTIONLESS) {
    const int c = s.eid[row], pr = s.cpair[c];
    int ord = 0;
    for (int k = c - 1; k >= 0 && s.cpair[k] == pr; k--) ord++;
    return (int)(0x80000000u | ((unsigned)pr << 6) | ((unsigned)(ord & 7) << 3) | (unsigned)((row - s.cefc[c]) & 7));
  }
  return 0x40000000 | (int)s.sat.erec[row];
}

#define PSETS(p, ne) _Pragma("unroll") for (int p = 0; p < NP; p++) if (p == 0 || (ne) > 64 * p)
#define PSETS_ALL(p) _Pragma("unroll") for (int p = 0; p < NP; p++)
SMJ_DEV void solve_pgs_sat(bool dbg, float* pc, long long& t0, bool prof) {
#define TICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - t0); t0 = t1; }
  const int nv = M.nv, nsat = M.nsat, ne = nefc;
  short* const didx = pgs_didx();
  short* const drow = pgs_drow();
  // ---- satellites: item lists in MuJoCo's row order (friction loss, limits, contacts by index); they were appended from divergent lanes
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < nsat) {
      const int n = s.sat.nitem[si];
      for (int a = 1; a < n; a++) {
        const int r = s.sat.irow[si][a], inf = s.sat.iinf[si][a], c = s.sat.icon[si][a];
        const int key = (inf & ITEM_CONTACT) ? (2 << 16 | c) : ((s.etype[r] == CT_FRICTION ? 0 : 1) << 16 | (s.eid[r] & 0xffff));
        int b = a - 1;
        for (; b >= 0; b--) {
          const int rb = s.sat.irow[si][b], ib = s.sat.iinf[si][b], cb = s.sat.icon[si][b];
          const int kb = (ib & ITEM_CONTACT) ? (2 << 16 | cb) : ((s.etype[rb] == CT_FRICTION ? 0 : 1) << 16 | (s.eid[rb] & 0xffff));
          if (kb <= key) break;
          s.sat.irow[si][b + 1] = (unsigned short)rb; s.sat.iinf[si][b + 1] = (unsigned char)ib; s.sat.icon[si][b + 1] = (unsigned char)cb;
        }
        s.sat.irow[si][b + 1] = (unsigned short)r; s.sat.iinf[si][b + 1] = (unsigned char)inf; s.sat.icon[si][b + 1] = (unsigned char)c;
      }
    }
  }
  // ---- efc_vel, aref, the warm start's J qacc_warmstart - aref (every row, before the Jacobians are transformed)
  ROWPASS(rb, ne) LANES {
    const int row = lane + rb;
    if (row < ne) {
      const float* jr = jrow(row);
      float vel = 0, jw = 0;
      if (row < nd)
        for (int k = 0; k < nv; k++) { vel += jr[k] * s.qvel[k]; jw += jr[k] * s.warm[k]; }
      vel += sat_jdot(row, SX_V); jw += sat_jdot(row, SX_QA);
      const float ar = -s.eBv[row] * vel - s.eK[row] * s.eimp[row] * (s.epos[row] - s.emargin[row]);
      s.earef[row] = ar; s.ediag[row] = jw - ar;
    }
  }
  // u = L^-T phase of qfrc_smooth (dof lanes)
  PL<float> u;
  LANES { u[lane] = lane < nv ? g_r[lane] : 0.f; }
  solve_LT(u);
  LANES { if (lane < NVP) s.uu[lane] = lane < nv ? u[lane] : 0.f; }
  SYNC();
  LANES { if (lane < NCON) s.eK[lane] = 0.f; }   // (from here on: QCQP multipliers of the satellite islands' contacts)
  // Y = J L^-1 on the rows that touch the main tree
  for (int i = nv - 1; i > 0; i--) {
    const int na = uni(M.k_dof_anc_num[i]), adr = uni(M.k_dof_anc_adr[i]);
    if (na == 0) continue;
    ROWPASS(rb, nd) LANES {
      const int row = lane + rb;
      if (row < nd) {
        const float xi = s.J[row][i];
        if (xi != 0.f)
          for (int a = 0; a < na; a++) {
            const int j = uni(M.k_dof_anc[adr + a]);
            s.J[row][j] -= s.MM[i][j] * xi;
          }
      }
    }
  }
  // the satellites' columns, whitened in place by each satellite's lane; y0 = D^-1/2 L^-1 qfrc_smooth
  LANES {
    const int si = lane - 32;
    if (lane >= 32 && si < nsat) {
      SatFac F;
      sat_factor6(s.sat.Mb[si], F);
      float x[6];
      for (int k = 0; k < 6; k++) x[k] = s.sat.x[SX_G][si][k];
      sat_whiten(F, x);
      for (int k = 0; k < 6; k++) s.sat.x[SX_GRAD][si][k] = x[k];
      for (int it = 0; it < s.sat.nitem[si]; it++) {
        const int r0 = s.sat.irow[si][it], inf = s.sat.iinf[si][it], n = inf & ITEM_N, uu = (inf & ITEM_SLOT) ? 1 : 0;
        for (int p = 0; p < n; p++) {
          float* y = jspw(r0 + p, uu);
          for (int k = 0; k < 6; k++) x[k] = y[k];
          sat_whiten(F, x);
          for (int k = 0; k < 6; k++) y[k] = x[k];
        }
      }
    }
  }
  SYNC();
  // b = J M^-1 qfrc_smooth - aref; the dense system's rows
  int cnt = 0, kept = 0;
  ROWPASS(rb, ne) {
    PL<int> isd;
    LANES {
      const int row = lane + rb;
      int d = 0;
      if (row < ne) {
        float v = 0;
        if (row < nd)
          for (int k = 0; k < nv; k++) v += s.J[row][k] * s.Dinv[k] * s.uu[k];
        v += sat_jdot(row, SX_GRAD);
        s.eb[row] = v - s.earef[row];
        const int sa = s.sat.esat[row][0], sb = s.sat.esat[row][1];
        d = row < nd || (sa >= 0 && s.sat.ext[sa] >= 0) || (sb >= 0 && s.sat.ext[sb] >= 0);
      }
      isd[lane] = d;
    }
    const uint64_t m = wave_ballot(isd);
    PL<int> keep;
    LANES {
      const int row = lane + rb;
      int k = 0;
      if (row < NEFC) {
        int idx = -1;
        if (isd[lane]) {
          idx = cnt + popc64(m & ((1ull << lane) - 1));
          // a row beyond the dense system's capacity is left out, and so is the whole contact whose block would straddle it (the step is flagged)
          const int t = s.etype[row];
          int last = idx;
          if (t == CT_CONTACT_ELLIPTIC) { const int c = s.eid[row]; last = idx - (row - s.cefc[c]) + s.cdim[c] - 1; }
          if (last >= NEFC_P) idx = -1;
        }
        didx[row] = (short)idx;
        if (idx >= 0) { drow[idx] = (short)row; k = 1; }
      }
      keep[lane] = k;
    }
    cnt += popc64(m);
    kept += popc64(wave_ballot(keep));
  }
  ndp = kept;
  if (cnt > kept) flags |= SMJ_FLAG_EFC_OVERFLOW | 0x1000;
  SYNC();
  // The sweep order of the dense system is MuJoCo's row order -- equalities, friction loss by dof, limits by joint, contacts by
  // pair -- not the order the rows are stored in here (main tree first): 100 sweeps leave a remainder that depends on it.
  if (cnt == kept) {
    int* const keyb = reinterpret_cast<int*>(s.emargin);   // (free once aref is known)
    for (int b = 0; b < NEFC_P; b += 64) {
      if (b >= ndp) break;
      LANES {
        const int i = lane + b;
        if (i < ndp) {
          const int row = drow[i], t = s.etype[row];
          int key;
          if (t == CT_CONTACT_ELLIPTIC || t == CT_CONTACT_FRICTIONLESS) { const int c = s.eid[row]; key = (3 << 28) | (s.cpair[c] << 9) | (c << 3) | (row - s.cefc[c]); }
          else {
            const int* r = M.k_rowrec + (int)s.sat.erec[row] * SMJ_RR_STRIDE;
            key = t == CT_EQUALITY ? r[SMJ_RR_ID] : t == CT_FRICTION ? (1 << 28) | r[SMJ_RR_ID] : (2 << 28) | (2 * r[SMJ_RR_ID] + (r[SMJ_RR_D2] > 0 ? 1 : 0));
          }
          keyb[i] = key;
        }
      }
    }
    SYNC();
    PL<int> nrow[NPS], nidx[NPS];
#pragma unroll
    for (int p = 0; p < NPS; p++) {
      if (p > 0 && ndp <= 64 * p) break;
      LANES {
        const int i = lane + 64 * p;
        int rk = 0, row = 0;
        if (i < ndp) {
          const int key = keyb[i];
          row = drow[i];
          for (int j = 0; j < ndp; j++) rk += keyb[j] < key;
        }
        nrow[p][lane] = row; nidx[p][lane] = rk;
      }
    }
    SYNC();
#pragma unroll
    for (int p = 0; p < NPS; p++) {
      if (p > 0 && ndp <= 64 * p) break;
      LANES {
        if (lane + 64 * p < ndp) { drow[nidx[p][lane]] = (short)nrow[p][lane]; didx[nrow[p][lane]] = (short)nidx[p][lane]; }
      }
    }
    SYNC();
  }
  // at most 64 dense rows (the usual step: the robot's own rows and its wheels' contacts): one register set and A as a 64 x 64
  // square (row i = column i, conflict-free reads) -- the sweeps cost a third of the packed triangle's
  if (ndp > NEFP) pgs_sat_core<true>(u, dbg, pc, t0, prof);
  else pgs_sat_core<false>(u, dbg, pc, t0, prof);
#undef TICK
}
template <bool WIDE>
SMJ_DEV void pgs_sat_core(PL<float>& u, bool dbg, float* pc, long long& t0, bool prof) {
#define TICK(slot) if (prof) { const long long t1 = smj_clock(); pc[slot] += (float)(t1 - t0); t0 = t1; }
  constexpr int NP = WIDE ? NPS : 1;
  const int nv = M.nv, nsat = M.nsat, ne = nefc;
  float* const A = Amat<WIDE>();
  short* const didx = pgs_didx();
  short* const drow = pgs_drow();
  PL<int> drow_r[NP];
  PL<float> bb[NP];
  PSETS_ALL(p) LANES { const int i = lane + 64 * p; drow_r[p][lane] = i < ndp ? (int)drow[i] : -1; bb[p][lane] = i < ndp ? s.eb[drow[i]] : 0.f; }
  // A = Y Dinv Y' on the matrix cores over the dense system's rows (a row without main columns reads the zero row) ...
  {
    const int ntile = (ndp + 15) >> 4, ksteps = (nv + 3) >> 2;
    for (int tr = 0; tr < ntile; tr++)
      for (int tc = 0; tc <= tr; tc++) {
        PL<F4v> acc;
        PL<int> ja, jb;
        LANES {
          for (int r = 0; r < 4; r++) acc[lane].r[r] = 0.f;
          const int ia = 16 * tr + (lane & 15), ib = 16 * tc + (lane & 15);
          const int ra = ia < ndp ? (int)drow[ia] : NDR, rbw = ib < ndp ? (int)drow[ib] : NDR;
          ja[lane] = ra < nd ? ra : NDR; jb[lane] = rbw < nd ? rbw : NDR;
        }
        for (int ks = 0; ks < ksteps; ks++) {
          PL<float> a, b;
          LANES {
            const int k = 4 * ks + (lane >> 4);
            const float dk = k < nv ? s.Dinv[k] : 0.f;
            a[lane] = k < nv ? s.J[ja[lane]][k] * dk : 0.f;
            b[lane] = k < nv ? s.J[jb[lane]][k] : 0.f;
          }
          mfma16x16x4(acc, a, b);
        }
        LANES {
          for (int r = 0; r < 4; r++) {
            const int row = 16 * tr + (lane >> 4) * 4 + r, col = 16 * tc + (lane & 15);
            float v = acc[lane].r[r];
            if (row == col) v += row < ndp ? s.eR[drow[row]] : 1.f;
            if (WIDE) { if (col <= row && row < NEFC_P) A[tri(row, col)] = v; }
            else if (row < NEFP && col < NEFP) { A[row * NEFP + col] = v; if (tr != tc) A[col * NEFP + row] = v; }
          }
        }
      }
  }
  SYNC();
  // ... plus the blocks of the coupled satellites: A_ij += y_i . y_j over the rows of each satellite (lane = row i, j over the satellite's items)
  PSETS(p, ndp) LANES {
    const int i = lane + 64 * p;
    if (i < ndp) {
      const int row = drow_r[p][lane];
#pragma unroll
      for (int uu = 0; uu < 2; uu++) {
        const int si = s.sat.esat[row][uu];
        if (si < 0) continue;
        float yi[6];
        { const float* y = jsp(row, uu); for (int k = 0; k < 6; k++) yi[k] = y[k]; }
        for (int it = 0; it < s.sat.nitem[si]; it++) {
          const int r0 = s.sat.irow[si][it], inf = s.sat.iinf[si][it], n = inf & ITEM_N, uj = (inf & ITEM_SLOT) ? 1 : 0;
          for (int q = 0; q < n; q++) {
            const int j = didx[r0 + q];
            if (j < 0 || j > i) continue;
            const float* y = jsp(r0 + q, uj);
            float v = 0;
            for (int k = 0; k < 6; k++) v += yi[k] * y[k];
            A[ai<WIDE>(i, j)] += v;
            if (!WIDE && j != i) A[ai<WIDE>(j, i)] += v;
          }
        }
      }
    }
  }
  SYNC();
  TICK(SMJ_PROF_PROJECT)
  // ---- warm start  [MJ] mj_warmstart (PGS branch), every row (s.ediag holds J qacc_warmstart - aref)
  ROWPASS(rb, ne) LANES {
    const int i = lane + rb;
    if (i < ne) {
      float f = 0;
      if (M.warmstart) {
        const int t = s.etype[i];
        const float R = s.eR[i], D = 1.0f / R, jr = s.ediag[i];
        if (t == CT_EQUALITY) f = -D * jr;
        else if (t == CT_FRICTION) {
          const float fl = s.efloss[i];
          f = (jr <= -R * fl) ? fl : (jr >= R * fl) ? -fl : -D * jr;
        } else if (t == CT_LIMIT || t == CT_CONTACT_FRICTIONLESS) f = jr < 0 ? -D * jr : 0.f;
      }
      s.ef[i] = f;
    }
  }
  SYNC();
  LANES {
    if (lane < ncon && M.warmstart) {
      const int c = lane, i = s.cefc[c], dim = s.cdim[c];
      if (i >= 0 && dim >= 3) {
        const float mu = contact_mu(c);
        float U[6], T = 0;
        U[0] = s.ediag[i] * mu;
        for (int j = 1; j < dim; j++) { U[j] = s.ediag[i + j] * s.cfric[c][j - 1]; T += U[j] * U[j]; }
        const float N = U[0];
        T = sqrtf(T);
        if ((T <= 0 && N >= 0) || (T > 0 && N >= mu * T)) { for (int j = 0; j < dim; j++) s.ef[i + j] = 0; }
        else if ((T <= 0 && N < 0) || (T > 0 && mu * N + T <= 0)) { for (int j = 0; j < dim; j++) s.ef[i + j] = -s.ediag[i + j] / s.eR[i + j]; }
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
  // ---- NOT MuJoCo (option pgs_dual_ws, default on): a second start -- the forces the rows had at the end of the PREVIOUS step's
  // solve (DevState::pgsprev), matched by row identity, projected onto this step's bounds and cones -- taken when its dual cost is
  // below that of MuJoCo's start.  The dual problem is strictly convex (R > 0), so where the sweeps converge they converge to the same
  // forces from either start; what changes is how many sweeps that takes: a kitchen's resting contacts carry the same forces from
  // step to step, while MuJoCo's start (the primal residual at qacc_warmstart pushed through the constraint update) lands far enough
  // from them that 100 sweeps never got there (fp64 oracle, Robocasa-scale kitchen: 100 sweeps on every step -> 19 settled / 71
  // under random actions; oracle option pgs_dual_warmstart).
  bool have_prev = false;
  if (M.pgs_dual_ws && M.warmstart && S.pgsprev) {
    const int* const pk = reinterpret_cast<const int*>(S.pgsprev + (size_t)env * SMJ_PGSPREV_STRIDE);
    const float* const pf = S.pgsprev + (size_t)env * SMJ_PGSPREV_STRIDE + 1 + SMJ_PGSPREV_ROWS;
    int np = uni(pk[0]);
    np = np < 0 ? 0 : np > SMJ_PGSPREV_ROWS ? SMJ_PGSPREV_ROWS : np;
    if (np > 0) {
      have_prev = true;
      // the stored rows into LDS, lane-parallel (keys -> emargin, forces -> eimp: both dead once aref is known); a row usually sits
      // where it sat a step ago, so its own index is looked at first and the scan of the whole list runs only for the lanes that miss
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
          s.ediag[row] = f;   // (free: the warm start above was the last reader of J qacc_warmstart - aref)
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
  // residual r = A f + b and the dual cost of a start: the dense system (lane = row) + the satellite islands (lane = satellite).
  // Pass 0: MuJoCo's start (s.ef).  Pass 1: the previous step's forces, swapped into s.ef; kept when cheaper.  Pass 2: MuJoCo's once
  // more when it was the better one (the registers hold the state of the start evaluated last).
  PL<float> cost;
  PL<float[6]> zs;   // lane = satellite: z = sum y_i f_i of its island
  float wcost = 0.f;
  for (int pass = 0;; pass++) {
    LANES { cost[lane] = 0.f; for (int k = 0; k < 6; k++) zs[lane][k] = 0.f; }
    PSETS_ALL(p) LANES {
      const int i = lane + 64 * p;
      f_r[p][lane] = i < ndp ? s.ef[drow_r[p][lane]] : 0.f;
      r_r[p][lane] = 0.f;
    }
    residual_refresh<WIDE>(bb);
    PSETS(p, ndp) LANES { cost[lane] += lane + 64 * p < ndp ? f_r[p][lane] * 0.5f * (r_r[p][lane] + bb[p][lane]) : 0.f; }
    LANES {
      const int si = lane - 32;
      if (lane >= 32 && si < nsat && s.sat.ext[si] < 0) {
        float z[6], cs = 0;
        sat_yf(si, z);
        for (int it = 0; it < s.sat.nitem[si]; it++) {
          const int r0 = s.sat.irow[si][it], n = s.sat.iinf[si][it] & ITEM_N;
          for (int p = 0; p < n; p++) {
            const float* y = s.sat.Js[r0 + p];
            const float f = s.ef[r0 + p], b = s.eb[r0 + p];
            float r = s.eR[r0 + p] * f + b;
            for (int k = 0; k < 6; k++) r += y[k] * z[k];
            cs += f * 0.5f * (r + b);
          }
        }
        cost[lane] += cs;
        for (int k = 0; k < 6; k++) zs[lane][k] = z[k];
      }
    }
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
  if (wcost > 0) {
    PSETS(p, ndp) LANES { f_r[p][lane] = 0.f; r_r[p][lane] = bb[p][lane]; }
    ROWPASS(rb, ne) LANES { if (lane + rb < ne) s.ef[lane + rb] = 0.f; }
    LANES { for (int k = 0; k < 6; k++) zs[lane][k] = 0.f; }
    SYNC();
  }
  TICK(SMJ_PROF_WARM)
  // ---- PGS sweeps  [MJ] mj_solPGS
  PL<int> type_r[NP], dimc_r[NP];
  PL<float> aii_r[NP], lo_r[NP], hi_r[NP];
  PSETS_ALL(p) LANES {
    const int i = lane + 64 * p, row = i < ndp ? drow_r[p][lane] : 0;
    const int t = i < ndp ? s.etype[row] : CT_NONE;
    type_r[p][lane] = t;
    int dc = 0;
    if (t == CT_CONTACT_ELLIPTIC) { const int c = s.eid[row]; dc = s.cdim[c] | (c << 8); }
    dimc_r[p][lane] = dc;
    const float aii = i < ndp ? A[ai<WIDE>(i, i)] : 1.f;
    aii_r[p][lane] = aii; ARinv_r[p][lane] = 1.0f / aii;
    const float fl = i < ndp ? s.efloss[row] : 0.f;
    lo_r[p][lane] = t == CT_EQUALITY ? -INFINITY : t == CT_FRICTION ? -fl : 0.f;
    hi_r[p][lane] = t == CT_FRICTION ? fl : INFINITY;
  }
  LANES { qla_r[lane] = 0.f; }
  PL<int> frozen;   // lane = satellite: its island has stopped sweeping (option pgs_island_stop)
  LANES { frozen[lane] = 0; }
  const float scale = 1.0f / (M.meaninertia * (float)(M.nv_all > 1 ? M.nv_all : 1));
  // An island sees no row of another, so its improvement falls on its own schedule; once it is below 1/64 of the tolerance the
  // SUM over all islands is tested against, further sweeps of it cannot decide the test and move its forces by less than the
  // solver's own tolerance: the island stops (its lane idles) while the dense system goes on.  An island whose updates are all
  // rejected by the costChange guard (improvement exactly 0) would repeat itself anyway.
  const float istop = M.pgs_island_stop ? M.tolerance * (1.f / 64) : -1.f;
  const int nD = ndp;
#ifdef SMJ_TWO_WAVES
  LANES {
    if (lane >= 32 && lane - 32 < nsat) for (int k = 0; k < 6; k++) s.sat.x[SX_SRCH][lane - 32][k] = zs[lane][k];
    if (lane == 0) s.sat.x[SX_MV][0][4] = __builtin_bit_cast(float, (int)PGS2_RUN);
  }
  WG_BARRIER();   // the second wavefront starts its sweeps
#endif
  int iter = 0;
  for (; iter < M.iterations; iter++) {
    float improvement = 0;
    if (iter > 0 && (iter & 7) == 0) {
      residual_refresh<WIDE>(bb);
    }
    ppc = prof ? pc : nullptr;
    long long tp = prof ? smj_clock() : 0;
    for (int i = 0; i < nD;) {
      const int t = prow<NP>(type_r, i);
      if (t != CT_CONTACT_ELLIPTIC) {
        PL<float> arow[NP];
        PSETS(p, nD) LANES { const int col = lane + 64 * p; arow[p][lane] = col < nD ? A[ai<WIDE>(i, col)] : 0.f; }
        int nrun = 0;
        for (;;) {
          const bool more = i + 1 < nD;
          const int inext = more ? i + 1 : i;
          PL<float> anext[NP];
          PSETS(p, nD) LANES { const int col = lane + 64 * p; anext[p][lane] = col < nD ? A[ai<WIDE>(inext, col)] : 0.f; }
          const int tnext = more ? prow<NP>(type_r, inext) : CT_CONTACT_ELLIPTIC;
          const float res = prow<NP>(r_r, i), old = prow<NP>(f_r, i), ainv = prow<NP>(ARinv_r, i);
          const float aii = prow<NP>(aii_r, i), lo = prow<NP>(lo_r, i), hi = prow<NP>(hi_r, i);
          const float fn = fminf(hi, fmaxf(lo, old - res * ainv));
          float delta = fn - old;
          float change = delta * (0.5f * aii * delta + res);
          if (change > SMJ_PGS_GUARD) { delta = 0; change = 0; }
          improvement -= change;
          PSETS(p, nD) LANES {
            r_r[p][lane] += arow[p][lane] * delta;
            if (lane + 64 * p == i) f_r[p][lane] += delta;
          }
          i += 1; nrun++;
          if (tnext == CT_CONTACT_ELLIPTIC) break;
          PSETS(p, nD) LANES { arow[p][lane] = anext[p][lane]; }
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
#ifdef SMJ_TWO_WAVES
    // the satellite islands are swept by the env's second wavefront meanwhile (pgs_helper): exchange the improvements
    LANES { if (lane == 0) s.sat.x[SX_MV][0][2 * (iter & 1)] = improvement; }
    WG_BARRIER();
    improvement += uni(s.sat.x[SX_MV][0][2 * (iter & 1) + 1]);
#else
    improvement += sat_lanes_sweep(zs, frozen, iter, scale, istop);   // the satellite islands, each by its own lane
#endif
    if (prof) { const long long t1 = smj_clock(); pc[SMJ_PROF_SAT_H] += (float)(t1 - tp); tp = t1; }
    improvement *= scale;
    if (!M.pgs_fixed_iter && improvement < M.tolerance) { iter++; break; }
  }
  niter = iter;
  TICK(SMJ_PROF_PGS)
  // ---- forces back to their rows; qacc = M^-1 (qfrc_smooth + J' f)
  PSETS_ALL(p) LANES { const int i = lane + 64 * p; if (i < ndp) s.ef[drow_r[p][lane]] = f_r[p][lane]; }
  SYNC();
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
  PL<float> w, qc;
  LANES {
    float v = 0;
    if (lane < nv)
      for (int r = 0; r < nd; r++) v += s.J[r][lane] * s.ef[r];
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
    const int si = lane - 32;
    if (lane >= 32 && si < nsat) {   // qacc_s = L^-T D^-1/2 (y0 + z); qfrc_smooth + J_s' f = g + L D^1/2 z (the integrator's right-hand side)
      SatFac F;
      sat_factor6(s.sat.Mb[si], F);
      float z[6], x[6], t[6];
      sat_yf(si, z);
      for (int k = 0; k < 6; k++) { x[k] = F.rs[k] * (s.sat.x[SX_GRAD][si][k] + z[k]); t[k] = F.sd[k] * z[k]; }
#pragma unroll
      for (int i = 5; i >= 0; i--)
#pragma unroll
        for (int k = i + 1; k < 6; k++) x[i] -= F.L[sl6(k, i)] * x[k];
      const int ndof = s.sat.ndof[si];
      for (int i = 0; i < 6; i++) {
        float v = t[i];
        for (int k = 0; k < i; k++) v += F.L[sl6(i, k)] * t[k];
        s.sat.x[SX_QA][si][i] = i < ndof ? x[i] : 0.f;
        s.sat.x[SX_TMP][si][i] = i < ndof ? s.sat.x[SX_G][si][i] + v : 0.f;
      }
    }
  }
  SYNC();
  if (dbg && S.debug) {   // the debug layout holds the first 64 rows (row order of make_constraint_sat)
    LANES {
      if (lane < nv) S.debug[(SMJ_DBG_QACC + lane) * S.ld + env] = qacc_r[lane];
      S.debug[(SMJ_DBG_EFC_FORCE + lane) * S.ld + env] = lane < ne ? s.ef[lane] : 0.f;
      S.debug[(SMJ_DBG_EFC_B + lane) * S.ld + env] = lane < ne ? s.eb[lane] : 0.f;
      S.debug[(SMJ_DBG_EFC_R + lane) * S.ld + env] = lane < ne ? s.eR[lane] : 0.f;
      S.debug[(SMJ_DBG_EFC_AREF + lane) * S.ld + env] = lane < ne ? s.earef[lane] : 0.f;
    }
  }
#undef TICK
}
#undef PSETS
#undef PSETS_ALL
