// mre_solver.h -- collision driver, constraint assembly and the PGS solve for one
// environment per wavefront.  Included by mre_kernels.hip after `struct Sm`.
//
// Constraint rows (MuJoCo order): 7 equality rows (2 connect x 3 + 1 joint),
// active joint limits, then 3 rows (normal + 2 tangents, elliptic cone) per
// active contact.  Row storage exploits the scene's block structure: a row
// touches at most the 15 robot dofs ("robot part", Jr/Br pools, only for rows
// that involve a robot body) and at most two cubes ("prop parts", 6 dofs each).
// The cube blocks of the mass matrix are diagonal in MuJoCo's c-frame
// (diag(m,m,m,Ixx,Iyy,Izz)), so M^-1 J' needs storage for the robot part only.
//
// PGS is run matrix-free: lanes own dofs and carry a = M^-1 J' f and w = J' f in
// registers; a row update is one wave-wide dot product (DPP reduction) plus a
// rank-1 update.  Iterates are identical (in exact arithmetic) to MuJoCo's PGS
// on the explicit A = J M^-1 J' + R that the CPU oracle builds.
#pragma once

namespace mre {

constexpr int HDR_NONE = 0xFF;

MRE_DEV float prop_invM(const Sm& s, int p, int k) {
  return 1.0f / ((k < 3) ? s.prop_mass[p] : s.prop_inertia[p][k - 3]);
}

MRE_DEV void geom_pose(const DevModel* M, const Sm& s, int g, float* p, float* R, float* size,
                       float* rbound) {
  const int b = M->geom_body[g];
  float tmp[3], q[4];
  m3mulv(tmp, s.xmat[b], M->geom_pos[g]);
  v3add(p, s.xpos[b], tmp);
  qmul(q, s.xquat[b], M->geom_quat[g]);
  q2mat(R, q);
  const int pid = M->geom_propid[g];
  if (pid >= 0) {
    v3copy(size, s.prop_size[pid]);
    *rbound = v3norm(size);
  } else {
    v3copy(size, M->geom_size[g]);
    *rbound = M->geom_rbound[g];
  }
}

// ------------------------------------------------------------------ mj_collision
// lane = entry of the static pair table; keeps ACTIVE contacts only
// (dist < margin - gap), in pair order, capped at NCON_MAX.
MRE_DEV void collide(const DevModel* M, Sm& s, int l) {
  PairContacts pc;
  pc.n = 0;
  const int g1 = M->pair_g1[l], g2 = M->pair_g2[l];
  if (g1 >= 0) {
    const int b1 = M->geom_body[g1], b2 = M->geom_body[g2];
    if (body_is_active(M, s, b1) && body_is_active(M, s, b2)) {
      float p1[3], R1[9], s1[3], rb1, p2[3], R2[9], s2[3], rb2;
      geom_pose(M, s, g1, p1, R1, s1, &rb1);
      geom_pose(M, s, g2, p2, R2, s2, &rb2);
      const float inc = M->pair_margin[l] - M->pair_gap[l];
      if (M->geom_type[g1] == 0) {
        float n[3] = {R1[2], R1[5], R1[8]}, df[3];
        v3sub(df, p2, p1);
        if (v3dot(df, n) - rb2 <= inc) plane_box(p1, R1, p2, R2, s2, inc, pc);
      } else {
        float df[3];
        v3sub(df, p2, p1);
        const float r = rb1 + rb2 + inc;
        if (v3dot(df, df) <= r * r) box_box(p1, R1, s1, p2, R2, s2, inc, pc);
      }
      // instantiate only contacts with dist < includemargin
      int m = 0;
      for (int c = 0; c < pc.n; c++)
        if (pc.dist[c] < inc) {
          if (m != c) { v3copy(pc.pos[m], pc.pos[c]); pc.dist[m] = pc.dist[c]; }
          m++;
        }
      pc.n = m;
    }
  }
  s.iscr[l] = pc.n;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < l; k++) off += s.iscr[k];
  if (l == 63) {
    const int tot = off + pc.n;
    s.ncon = tot < NCON_MAX ? tot : NCON_MAX;
    if (tot > NCON_MAX) s.overflow = 1;
  }
  for (int c = 0; c < pc.n; c++) {
    const int id = off + c;
    if (id >= NCON_MAX) break;
    v3copy(s.con_pos[id], pc.pos[c]);
    float f[9];
    v3copy(f, pc.normal);
    make_frame(f);
    for (int k = 0; k < 9; k++) s.con_frame[id][k] = f[k];
    s.con_dist[id] = pc.dist[c];
    s.con_pair[id] = l;
  }
  __syncthreads();
}

// getimpedance (MuJoCo engine_core_constraint.c)
MRE_DEV float impedance(const float* solimp, float pos, float margin) {
  const float dmin = clampf(solimp[0], 0.0001f, 0.9999f), dmax = clampf(solimp[1], 0.0001f, 0.9999f);
  const float width = fmaxf(solimp[2], kMinVal), mid = clampf(solimp[3], 0.0001f, 0.9999f);
  const float power = fmaxf(solimp[4], 1.0f);
  if (dmin == dmax) return 0.5f * (dmin + dmax);
  const float x = fabsf((pos - margin) / width);
  if (x >= 1.0f) return dmax;
  if (x <= 0.0f) return dmin;
  float y;
  if (power == 1.0f) y = x;
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.0f);
  else y = 1.0f - powf(1.0f - x, power) / powf(1.0f - mid, power - 1.0f);
  return dmin + y * (dmax - dmin);
}

// accumulate sg * ax . (d point / d q) of a point on robot body b into Jr[rs]
MRE_DEV void jac_robot(const DevModel* M, Sm& s, int rs, int b, const float* p, const float* ax, float sg) {
  float off[3];
  v3sub(off, p, s.com_robot);
  const int n = M->chain_len[b];
  for (int k = 0; k < n; k++) {
    const int j = M->chain_dof[b][k];
    const float* c = s.cdof[j];
    float t[3];
    v3cross(t, c, off);
    s.Jr[rs][j] += sg * (ax[0] * (c[3] + t[0]) + ax[1] * (c[4] + t[1]) + ax[2] * (c[5] + t[2]));
  }
}
// same for a cube body into Jp[i][slot*6 ..]
MRE_DEV void jac_prop(Sm& s, int i, int slot, int b, const float* p, const float* ax, float sg) {
  float off[3];
  v3sub(off, p, s.xpos[b]);
  float* J = &s.Jp[i][slot * 6];
  for (int k = 0; k < 3; k++) {
    J[k] = sg * ax[k];
    float axk[3] = {s.xmat[b][k], s.xmat[b][3 + k], s.xmat[b][6 + k]}, t[3];
    v3cross(t, axk, off);
    J[3 + k] = sg * v3dot(ax, t);
  }
}

// J_i . vec for an nv-vector in LDS (lane = row helper)
MRE_DEV float row_dot(const Sm& s, int i, const float* vec) {
  const int h = s.hdr[i];
  const int rs = h & 0xFF, pa = (h >> 8) & 0xF, pb = (h >> 12) & 0xF;
  float acc = 0.f;
  if (rs != HDR_NONE)
    for (int j = 0; j < NRV; j++) acc += s.Jr[rs][j] * vec[j];
  if (pa < NPROP)
    for (int k = 0; k < 6; k++) acc += s.Jp[i][k] * vec[NRV + 6 * pa + k];
  if (pb < NPROP)
    for (int k = 0; k < 6; k++) acc += s.Jp[i][6 + k] * vec[NRV + 6 * pb + k];
  return acc;
}

// ------------------------- mj_makeConstraint + mj_makeImpedance + reference + project
MRE_DEV void assemble_constraints(const DevModel* M, Sm& s, int l) {
  // ---- joint limits: lane = robot body; at most one side can be violated
  int lim = 0, lim_side = 0;
  if (l >= 1 && l < NRB && M->jnt_limited[l]) {
    const float q = s.qpos[l - 1];
    if (q - M->jnt_range[l][0] < 0.f) { lim = 1; lim_side = -1; }
    else if (M->jnt_range[l][1] - q < 0.f) { lim = 1; lim_side = 1; }
  }
  s.iscr[l] = lim;
  __syncthreads();
  int lim_idx = 0;
  for (int k = 0; k < l; k++) lim_idx += s.iscr[k];
  if (l == 63) s.nl = lim_idx + lim;
  if (lim) s.lim_info[lim_idx] = l | ((lim_side > 0 ? 1 : 0) << 8);
  __syncthreads();
  // ---- robot-slot assignment and capacity (serial, lane 0)
  if (l == 0) {
    const int base = 7 + s.nl;
    int rnext = base, ncon = s.ncon, kept = 0;
    for (int c = 0; c < ncon; c++) {
      const int pr = s.con_pair[c];
      const bool rob = M->geom_body[M->pair_g1[pr]] < NRB && M->geom_body[M->pair_g1[pr]] > 0;
      const bool rob2 = M->geom_body[M->pair_g2[pr]] < NRB && M->geom_body[M->pair_g2[pr]] > 0;
      if (base + 3 * (c + 1) > NEFC_MAX || ((rob || rob2) && rnext + 3 > NRROW_MAX)) {
        s.overflow = 1;
        break;
      }
      s.con_rslot[c] = (rob || rob2) ? rnext : HDR_NONE;
      if (rob || rob2) rnext += 3;
      kept++;
    }
    s.ncon = kept;
    s.nefc = base + 3 * kept;
    s.nrrow = rnext;
  }
  __syncthreads();
  const int nefc = s.nefc, nl = s.nl;
  // ---- zero robot slots
  for (int e = l; e < s.nrrow * NRV; e += 64) (&s.Jr[0][0])[e] = 0.f;
  __syncthreads();
  // ---- rows (lane = row)
  for (int i = l; i < nefc; i += 64) {
    int rs = HDR_NONE, pa = 0xF, pb = 0xF;
    float pos = 0.f, margin = 0.f, diag = 0.f, imp_pos = 0.f;
    const float *solref, *solimp;
    bool fric_row = false;
    float R0scale = 1.0f;
    for (int k = 0; k < 12; k++) s.Jp[i][k] = 0.f;
    if (i < 7) {
      rs = i;
      const int e = i < 6 ? i / 3 : 2;
      solref = M->eq_solref[e]; solimp = M->eq_solimp[e];
      if (e < 2) {
        const int b1 = M->eq_obj[e][0], b2 = M->eq_obj[e][1], k = i % 3;
        float p1[3], p2[3], cp[3], ax[3] = {0.f, 0.f, 0.f};
        m3mulv(p1, s.xmat[b1], M->eq_data[e]); v3add(p1, p1, s.xpos[b1]);
        m3mulv(p2, s.xmat[b2], M->eq_data[e] + 3); v3add(p2, p2, s.xpos[b2]);
        v3sub(cp, p1, p2);
        ax[k] = 1.f;
        jac_robot(M, s, rs, b1, p1, ax, 1.f);
        jac_robot(M, s, rs, b2, p2, ax, -1.f);
        pos = cp[k];
        imp_pos = v3norm(cp);
        diag = M->body_invweight0[b1][0] + M->body_invweight0[b2][0];
      } else {
        const int b1 = M->eq_obj[e][0], b2 = M->eq_obj[e][1];
        const int d1 = b1 - 1, d2 = b2 - 1;
        const float* pc = M->eq_data[e];
        const float dif = s.qpos[d2] - M->qpos0[d2];
        pos = s.qpos[d1] - M->qpos0[d1] -
              (pc[0] + dif * (pc[1] + dif * (pc[2] + dif * (pc[3] + dif * pc[4]))));
        const float deriv = pc[1] + dif * (2.f * pc[2] + dif * (3.f * pc[3] + dif * 4.f * pc[4]));
        s.Jr[rs][d1] += 1.f;
        s.Jr[rs][d2] -= deriv;
        imp_pos = pos;
        diag = M->dof_invweight0[d1] + M->dof_invweight0[d2];
      }
    } else if (i < 7 + nl) {
      rs = i;
      const int info = s.lim_info[i - 7], b = info & 0xFF, hi = (info >> 8) & 1;
      const float q = s.qpos[b - 1];
      pos = hi ? (M->jnt_range[b][1] - q) : (q - M->jnt_range[b][0]);
      s.Jr[rs][b - 1] = hi ? -1.f : 1.f;
      imp_pos = pos;
      diag = M->dof_invweight0[b - 1];
      solref = M->jnt_solref[b]; solimp = M->jnt_solimp[b];
    } else {
      const int c = (i - 7 - nl) / 3, r = (i - 7 - nl) % 3;
      const int pr = s.con_pair[c];
      const int b1 = M->geom_body[M->pair_g1[pr]], b2 = M->geom_body[M->pair_g2[pr]];
      const float* ax = &s.con_frame[c][3 * r];
      const float* p = s.con_pos[c];
      const int rs0 = s.con_rslot[c];
      if (rs0 != HDR_NONE) rs = rs0 + r;
      int slot = 0;
      float dg = 0.f;
      if (b1 > 0) {
        if (b1 < NRB) { jac_robot(M, s, rs, b1, p, ax, -1.f); dg += M->body_invweight0[b1][0]; }
        else { pa = b1 - NRB; jac_prop(s, i, slot++, b1, p, ax, -1.f); dg += 1.0f / s.prop_mass[pa]; }
      }
      if (b2 > 0) {
        if (b2 < NRB) { jac_robot(M, s, rs, b2, p, ax, 1.f); dg += M->body_invweight0[b2][0]; }
        else {
          const int pid = b2 - NRB;
          if (slot == 0) pa = pid; else pb = pid;
          jac_prop(s, i, slot++, b2, p, ax, 1.f);
          dg += 1.0f / s.prop_mass[pid];
        }
      }
      diag = dg;
      solref = M->pair_solref[pr]; solimp = M->pair_solimp[pr];
      margin = M->pair_margin[pr] - M->pair_gap[pr];
      imp_pos = s.con_dist[c];
      if (r == 0) pos = s.con_dist[c];
      else {
        fric_row = true;
        pos = 0.f;
        // R1 = R0/impratio ; R2 = R1 * mu0^2/mu1^2 (mu0 == mu1 for condim 3)
        R0scale = 1.0f / fmaxf(M->impratio, kMinVal);
      }
    }
    s.hdr[i] = rs | (pa << 8) | (pb << 12);
    // impedance of the block's leading row, stiffness / damping from solref
    const float imp = impedance(solimp, imp_pos, margin);
    float tc = solref[0];
    const float dr = solref[1], dmax = clampf(solimp[1], 0.0001f, 0.9999f);
    float K, B;
    if (tc > 0.f) {
      tc = fmaxf(tc, 2.f * M->timestep);
      K = 1.0f / (dmax * dmax * tc * tc * dr * dr);
      B = 2.0f / (dmax * tc);
    } else { K = -tc / (dmax * dmax); B = -dr / dmax; }
    if (fric_row) K = 0.f;
    float R = fmaxf((1.f - imp) * diag / imp, kMinVal) * R0scale;
    // reference acceleration (mj_referenceConstraint)
    const float vel = row_dot(s, i, s.qvel);
    const float efc_margin = fric_row ? 0.f : margin;
    const float aref = -B * vel - K * imp * (pos - efc_margin);
    s.rowdata[i] = make_float4(R, aref, 0.f, 0.f);
  }
  __syncthreads();
  // ---- Br = M^-1 Jr' (lane = robot slot, serial sparse solve in place)
  for (int rs = l; rs < s.nrrow; rs += 64) {
    for (int j = 0; j < NRV; j++) s.Br[rs][j] = s.Jr[rs][j];
    solve_robot_serial(M, s.qLD, s.qLDinv, s.Br[rs]);
  }
  __syncthreads();
  // ---- diagonal blocks of A = J M^-1 J' + R
  for (int i = l; i < nefc; i += 64) {
    const int h = s.hdr[i];
    const int rs = h & 0xFF, pa = (h >> 8) & 0xF, pb = (h >> 12) & 0xF;
    const int nb = (i < 7 + nl) ? 1 : 3;
    const int r = (nb == 3) ? (i - 7 - nl) % 3 : 0;
    const int i0 = i - r;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int cc = 0; cc < nb; cc++) {
      float a = 0.f;
      if (rs != HDR_NONE) {
        const int rs2 = rs - r + cc;
        for (int j = 0; j < NRV; j++) a += s.Jr[rs][j] * s.Br[rs2][j];
      }
      if (pa < NPROP)
        for (int k = 0; k < 6; k++) a += s.Jp[i][k] * s.Jp[i0 + cc][k] * prop_invM(s, pa, k);
      if (pb < NPROP)
        for (int k = 0; k < 6; k++) a += s.Jp[i][6 + k] * s.Jp[i0 + cc][6 + k] * prop_invM(s, pb, k);
      acc[cc] = a;
    }
    float4 rd = s.rowdata[i];
    if (nb == 1) {
      rd.w = 1.0f / (acc[0] + rd.x);
    } else {
      const int c = (i - 7 - nl) / 3;
      acc[r] += rd.x;
      for (int cc = 0; cc < 3; cc++) s.Ablk[c][3 * r + cc] = acc[cc];
      rd.w = 1.0f / acc[r];
    }
    s.rowdata[i] = rd;
  }
  __syncthreads();
}

// mju_QCQP2
MRE_DEV bool qcqp2(float* res, const float* A, const float* b, float d0, float d1, float r) {
  const float b1 = b[0] * d0, b2 = b[1] * d1;
  const float A11 = A[0] * d0 * d0, A22 = A[3] * d1 * d1, A12 = A[1] * d0 * d1;
  float la = 0.f, v1 = 0.f, v2 = 0.f;
  for (int iter = 0; iter < 20; iter++) {
    const float det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10f) { res[0] = res[1] = 0.f; return false; }
    const float di = 1.0f / det;
    const float P11 = (A22 + la) * di, P22 = (A11 + la) * di, P12 = -A12 * di;
    v1 = -P11 * b1 - P12 * b2;
    v2 = -P12 * b1 - P22 * b2;
    const float val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10f) break;
    const float deriv = -2.0f * (P11 * v1 * v1 + 2.0f * P12 * v1 * v2 + P22 * v2 * v2);
    const float delta = -val / deriv;
    if (delta < 1e-10f) break;
    la += delta;
  }
  res[0] = v1 * d0;
  res[1] = v2 * d1;
  return la != 0.f;
}

// per-lane J / B entry of row i for the dof this lane owns
MRE_DEV void lane_JB(const Sm& s, int i, int l, int lp, int lk, float linvM, float& j, float& b) {
  const int h = s.hdr[i];
  const int rs = h & 0xFF, pa = (h >> 8) & 0xF, pb = (h >> 12) & 0xF;
  j = 0.f; b = 0.f;
  if (l < NRV) {
    if (rs != HDR_NONE) { j = s.Jr[rs][l]; b = s.Br[rs][l]; }
  } else if (l < NV) {
    if (lp == pa) j = s.Jp[i][lk];
    else if (lp == pb) j = s.Jp[i][6 + lk];
    b = j * linvM;
  }
}

// ------------------------------------------------------------- mj_fwdConstraint
// On exit: s.qacc = qacc_smooth + M^-1 J' f, s.qfrc_con = J' f.
MRE_DEV void solve_constraints(const DevModel* M, Sm& s, int l) {
  const int nefc = s.nefc, nl = s.nl;
  const int lp = (l >= NRV && l < NV) ? (l - NRV) / 6 : -1;
  const int lk = (l >= NRV && l < NV) ? (l - NRV) % 6 : 0;
  const float linvM = (lp >= 0) ? prop_invM(s, lp, lk) : 0.f;
  // ---- efc_b and warm-start forces (mj_constraintUpdate on J*qacc_warmstart - aref)
  for (int i = l; i < nefc; i += 64) {
    float4 rd = s.rowdata[i];
    const float aref = rd.y;
    s.jar[i] = row_dot(s, i, s.qacc_ws) - aref;
    rd.y = row_dot(s, i, s.qacc_smooth) - aref;
    s.rowdata[i] = rd;
  }
  __syncthreads();
  for (int i = l; i < nefc; i += 64) {
    const float D = 1.0f / s.rowdata[i].x;
    if (i < 7) s.rowdata[i].z = -D * s.jar[i];
    else if (i < 7 + nl) s.rowdata[i].z = s.jar[i] < 0.f ? -D * s.jar[i] : 0.f;
    else if ((i - 7 - nl) % 3 == 0) {
      const int c = (i - 7 - nl) / 3, pr = s.con_pair[c];
      const float fr0 = M->pair_friction[pr][0];
      const float D1 = 1.0f / s.rowdata[i + 1].x, D2 = 1.0f / s.rowdata[i + 2].x;
      const float mu = fr0 * sqrtf(s.rowdata[i + 1].x / s.rowdata[i].x);
      const float j0 = s.jar[i], j1 = s.jar[i + 1], j2 = s.jar[i + 2];
      const float U0 = j0 * mu, U1 = j1 * fr0, U2 = j2 * fr0;
      const float N = U0, T = sqrtf(U1 * U1 + U2 * U2);
      float f0, f1, f2;
      if (mu * N >= T || (T <= 0.f && N >= 0.f)) { f0 = f1 = f2 = 0.f; }
      else if (N + mu * T <= 0.f || (T <= 0.f && N < 0.f)) { f0 = -D * j0; f1 = -D1 * j1; f2 = -D2 * j2; }
      else {
        const float Dm = D / fmaxf(mu * mu * (1.f + mu * mu), kMinVal);
        const float NT = N - mu * T;
        f0 = -Dm * NT * mu;
        f1 = -f0 / T * U1 * fr0;
        f2 = -f0 / T * U2 * fr0;
      }
      s.rowdata[i].z = f0; s.rowdata[i + 1].z = f1; s.rowdata[i + 2].z = f2;
    }
  }
  __syncthreads();
  // ---- a = M^-1 J' f, w = J' f for the warm start (lane = dof)
  float a = 0.f, w = 0.f;
  for (int i = 0; i < nefc; i++) {
    const float fi = s.rowdata[i].z;
    if (fi != 0.f) {
      float j, b;
      lane_JB(s, i, l, lp, lk, linvM, j, b);
      a += b * fi;
      w += j * fi;
    }
  }
  if (l < NVP) s.scratch[l] = (l < NV) ? a : 0.f;
  __syncthreads();
  // dual cost 0.5 f'ARf + f'b ; cold start if positive
  float part = 0.f;
  for (int i = l; i < nefc; i += 64) {
    const float4 rd = s.rowdata[i];
    const float Af = row_dot(s, i, s.scratch) + rd.x * rd.z;
    part += rd.z * (0.5f * Af + rd.y);
  }
  const float cost = wave_sum(part);
  __syncthreads();
  if (cost > 0.f) {
    a = 0.f; w = 0.f;
    for (int i = l; i < nefc; i += 64) s.rowdata[i].z = 0.f;
  }
  __syncthreads();
  // ---- PGS sweeps
  const int nva = NRV + 6 * s.nprops;
  float msum = M->M0_diag_robot_sum;
  for (int p = 0; p < s.nprops; p++)
    msum += 3.f * s.prop_mass[p] + s.prop_inertia[p][0] + s.prop_inertia[p][1] + s.prop_inertia[p][2];
  const float scale = 1.0f / ((msum / nva) * nva);
  int iters = 0;
  for (int iter = 0; iter < M->iterations; iter++) {
    float improvement = 0.f;
    int i = 0;
    for (; i < 7 + nl; i++) {
      float j, b;
      lane_JB(s, i, l, lp, lk, linvM, j, b);
      const float4 rd = s.rowdata[i];
      const float res = wave_sum(j * a) + rd.x * rd.z + rd.y;
      float fn = rd.z - res * rd.w;
      if (i >= 7 && fn < 0.f) fn = 0.f;
      float delta = fn - rd.z;
      float change = delta * (0.5f * delta / rd.w + res);
      if (change > 1e-10f) { delta = 0.f; change = 0.f; }
      improvement -= change;
      a += b * delta;
      w += j * delta;
      if (l == 0) s.rowdata[i].z = rd.z + delta;
    }
    for (; i < nefc; i += 3) {
      const int c = (i - 7 - nl) / 3;
      float j0, b0, j1, b1, j2, b2;
      lane_JB(s, i, l, lp, lk, linvM, j0, b0);
      lane_JB(s, i + 1, l, lp, lk, linvM, j1, b1);
      lane_JB(s, i + 2, l, lp, lk, linvM, j2, b2);
      const float4 r0 = s.rowdata[i], r1 = s.rowdata[i + 1], r2 = s.rowdata[i + 2];
      float res[3];
      res[0] = wave_sum(j0 * a) + r0.x * r0.z + r0.y;
      res[1] = wave_sum(j1 * a) + r1.x * r1.z + r1.y;
      res[2] = wave_sum(j2 * a) + r2.x * r2.z + r2.y;
      float At[9];
      for (int k = 0; k < 9; k++) At[k] = s.Ablk[c][k];
      const float fr = M->pair_friction[s.con_pair[c]][0];
      const float old[3] = {r0.z, r1.z, r2.z};
      float f[3] = {old[0], old[1], old[2]};
      if (f[0] < kMinVal) {
        f[0] -= res[0] / At[0];
        if (f[0] < 0.f) f[0] = 0.f;
        f[1] = f[2] = 0.f;
      } else {
        float v1[3];
        m3mulv(v1, At, f);
        const float denom = v3dot(f, v1);
        if (denom >= kMinVal) {
          float x = -v3dot(f, res) / denom;
          if (f[0] + x * f[0] < 0.f) x = -1.0f;
          const float v[3] = {f[0], f[1], f[2]};
          for (int k = 0; k < 3; k++) f[k] += x * v[k];
        }
      }
      const float Ac[4] = {At[4], At[5], At[7], At[8]};
      float bc[2];
      for (int k = 0; k < 2; k++) {
        bc[k] = res[1 + k] - Ac[2 * k] * old[1] - Ac[2 * k + 1] * old[2] + At[3 * (k + 1)] * (f[0] - old[0]);
      }
      if (f[0] < kMinVal) {
        f[1] = f[2] = 0.f;
      } else {
        float v[2];
        const bool active = qcqp2(v, Ac, bc, fr, fr, f[0]);
        if (active) {
          float sq = (v[0] / fr) * (v[0] / fr) + (v[1] / fr) * (v[1] / fr);
          sq = sqrtf(f[0] * f[0] / fmaxf(sq, kMinVal));
          v[0] *= sq; v[1] *= sq;
        }
        f[1] = v[0]; f[2] = v[1];
      }
      float d[3] = {f[0] - old[0], f[1] - old[1], f[2] - old[2]}, Ad[3];
      m3mulv(Ad, At, d);
      float change = 0.5f * v3dot(d, Ad) + v3dot(d, res);
      if (change > 1e-10f) { d[0] = d[1] = d[2] = 0.f; change = 0.f; }
      improvement -= change;
      a += b0 * d[0] + b1 * d[1] + b2 * d[2];
      w += j0 * d[0] + j1 * d[1] + j2 * d[2];
      if (l < 3) s.rowdata[i + l].z = old[l] + d[l];
    }
    iters = iter + 1;
    __syncthreads();
    if (improvement * scale < M->tolerance) break;
  }
  if (l < NVP) {
    s.qacc[l] = (l < NV) ? s.qacc_smooth[l] + a : 0.f;
    s.qfrc_con[l] = (l < NV) ? w : 0.f;
  }
  if (l == 0) s.solver_iters = iters;
  __syncthreads();
}

}  // namespace mre
