// mre_osc.h -- operational-space controller evaluated inside the step kernel at
// every control tick (mujoco_controllers.osc.OSC.compute_control_output, called
// from reference models/robot_arm.py:71; law restated in the reference at
// tasks/rearrangement_mjx.py:59-135; gains config/robots/arm/controller_config/osc.yaml).
//
//   J    = [jacp; jacr] of the controller site (arm attachment_site), arm dofs   (6x7)
//   M    = arm block of the full mass matrix (incl. reflected gripper inertia)    (7x7)
//   L^-1 = J M^-1 J' ;  L = inv(L^-1) if |det| >= 1e-2 else pinv(L^-1, rcond 1e-2)
//   F    = [kp_p e_p + kd_p (v* - J_p qd) ; kp_o e_o + kd_o (w* - J_r qd)]
//   tau  = J' L F + (I - J' Jbar') tau0 + qfrc_bias[arm],  Jbar = M^-1 J' L
// Lane mappings: lane = matrix entry for every small product; the two in-place
// Gauss-Jordan inversions run 7 / 6 pivot steps with lane = (row, col).
// Included by mre_kernels.hip after `struct Sm`.
#pragma once

namespace mre {

// in-place inverse of an SPD n x n matrix (row stride ld) by Gauss-Jordan without
// pivoting; lane = (i, j).  Returns the determinant (product of pivots), uniform.
template <int N, int LD>
MRE_DEV float gauss_jordan_inplace(float* A, int l) {
  const int i = l / N, j = l % N;
  const bool on = l < N * N;
  float det = 1.f;
  for (int k = 0; k < N; k++) {
    const float piv = A[k * LD + k];
    const float aik = on ? A[i * LD + k] : 0.f;
    const float akj = on ? A[k * LD + j] : 0.f;
    const float aij = on ? A[i * LD + j] : 0.f;
    MRE_SYNC();
    det *= piv;
    const float p = 1.0f / piv;
    if (on) {
      float v;
      if (i == k && j == k) v = p;
      else if (i == k) v = akj * p;
      else if (j == k) v = -aik * p;
      else v = aij - aik * akj * p;
      A[i * LD + j] = v;
    }
    MRE_SYNC();
  }
  return det;
}

// pinv of a symmetric 6x6 with relative cutoff rcond (cyclic Jacobi, single lane; rare path)
MRE_DEV void sym_pinv6_serial(const float* A_in, float* out, float* V, float* w, float* A, float rcond) {
  for (int k = 0; k < 36; k++) { A[k] = A_in[k]; V[k] = (k % 7 == 0) ? 1.f : 0.f; }
  for (int sweep = 0; sweep < 30; sweep++) {
    float off = 0.f;
    for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++) off += A[i * 6 + j] * A[i * 6 + j];
    if (off < 1e-20f) break;
    for (int p = 0; p < 6; p++)
      for (int q = p + 1; q < 6; q++) {
        const float apq = A[p * 6 + q];
        if (fabsf(apq) < 1e-30f) continue;
        const float th = (A[q * 6 + q] - A[p * 6 + p]) / (2.f * apq);
        const float t = (th >= 0.f ? 1.f : -1.f) / (fabsf(th) + sqrtf(th * th + 1.f));
        const float c = 1.0f / sqrtf(t * t + 1.f), sn = t * c;
        for (int k = 0; k < 6; k++) {
          const float akp = A[k * 6 + p], akq = A[k * 6 + q];
          A[k * 6 + p] = c * akp - sn * akq; A[k * 6 + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 6; k++) {
          const float apk = A[p * 6 + k], aqk = A[q * 6 + k];
          A[p * 6 + k] = c * apk - sn * aqk; A[q * 6 + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 6; k++) {
          const float vkp = V[k * 6 + p], vkq = V[k * 6 + q];
          V[k * 6 + p] = c * vkp - sn * vkq; V[k * 6 + q] = sn * vkp + c * vkq;
        }
      }
  }
  float wmax = 0.f;
  for (int k = 0; k < 6; k++) { w[k] = A[k * 6 + k]; wmax = fmaxf(wmax, fabsf(w[k])); }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      float sum = 0.f;
      for (int k = 0; k < 6; k++) {
        if (fabsf(w[k]) <= rcond * wmax) continue;
        sum += V[i * 6 + k] * V[j * 6 + k] / w[k];
      }
      out[i * 6 + j] = sum;
    }
}

// position / orientation error of the controller site w.r.t. the target
MRE_DEV void osc_errors(ModelP M, const Sm& s, const float* tgt, float* ep, float* eo) {
  const int st = M->eef_site;
  v3sub(ep, tgt, s.site_xpos[st]);
  float q[4], qc[4], qe[4];
  (void)st;
  mat2q(q, s.site_xmat[0]);
  qc[0] = q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = -q[3];
  qmul(qe, tgt + 3, qc);
  const float sg = qe[0] > 0.f ? 1.f : (qe[0] < 0.f ? -1.f : 0.f);
  eo[0] = sg * qe[1]; eo[1] = sg * qe[2]; eo[2] = sg * qe[3];
}

MRE_DEV bool osc_converged(ModelP M, const Sm& s, const OscConfig* cp, const float* tgt) {
  const OscConfig& c = *cp;
  float ep[3], eo[3];
  osc_errors(M, s, tgt, ep, eo);
  return v3norm(ep) < c.pos_thresh && v3norm(eo) < c.ori_thresh;
}

// writes s.ctrl[0..6]; tgt = [pos3 quat4 vel3 angvel3] (uniform pointer into LDS)
MRE_PHASE_FN void osc_compute(ModelP M, Sm& s, OscSm& o, const OscConfig* cp, const float* tgt, int l) {
  const OscConfig& c = *cp;
  const int st = M->eef_site;
  // J (lane = r*7+a) and dense arm mass block (lane = i*7+j)
  if (l < 42) {
    const int r = l / 7, a = l % 7;
    const float* cd = s.cdof[a];
    float off[3], t[3];
    v3sub(off, s.site_xpos[st], s.com_robot);
    v3cross(t, cd, off);
    o.J[r][a] = (r < 3) ? (cd[3 + r] + t[r]) : cd[r - 3];
  }
  if (l < 49) {
    const int i = l / 7, j = l % 7;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    o.M[i][j] = s.qM[M->dof_Madr[hi] + (hi - lo)];
  }
  if (l == 0) osc_errors(M, s, tgt, o.ep, o.eo);
  MRE_SYNC();
  gauss_jordan_inplace<7, 7>(&o.M[0][0], l);  // o.M <- M^-1
  if (l < 42) {
    const int i = l / 6, cc = l % 6;
    float sum = 0.f;
    for (int k = 0; k < 7; k++) sum += o.M[i][k] * o.J[cc][k];
    o.MiJt[i][cc] = sum;
  }
  if (l >= 48 && l < 54) {
    const int r = l - 48;
    float sum = 0.f;
    for (int a = 0; a < 7; a++) sum += o.J[r][a] * s.qvel[a];
    o.xd[r] = sum;
  }
  if (l >= 54 && l < 61) {
    const int a = l - 54;
    o.tn[a] = c.kp_null * (c.null_q[a] - s.qpos[a]) + c.kd_null * (0.f - s.qvel[a]);
  }
  MRE_SYNC();
  if (l < 36) {
    const int r = l / 6, cc = l % 6;
    float sum = 0.f;
    for (int k = 0; k < 7; k++) sum += o.J[r][k] * o.MiJt[k][cc];
    o.Li[r][cc] = sum;
    o.Lam[r][cc] = sum;
  }
  if (l >= 36 && l < 42) {
    const int k = l - 36;
    o.F[k] = (k < 3) ? c.kp_pos * o.ep[k] + c.kd_pos * (tgt[7 + k] - o.xd[k])
                     : c.kp_ori * o.eo[k - 3] + c.kd_ori * (tgt[10 + k - 3] - o.xd[k]);
  }
  MRE_SYNC();
  const float det = gauss_jordan_inplace<6, 6>(&o.Lam[0][0], l);  // o.Lam <- inv(L^-1)
  const bool use_pinv = c.pinv_always || !(fabsf(det) >= 1e-2f);
  if (use_pinv) {
    if (l == 0) sym_pinv6_serial(&o.Li[0][0], &o.Lam[0][0], &o.V[0][0], o.w, &o.Jbar[0][0] /* free until Jbar is formed */, 1e-2f);
    MRE_SYNC();
  }
  if (l < 6) {
    float sum = 0.f;
    for (int k = 0; k < 6; k++) sum += o.Lam[l][k] * o.F[k];
    o.LF[l] = sum;
  }
  if (l >= 8 && l < 50) {
    const int i = (l - 8) / 6, cc = (l - 8) % 6;
    float sum = 0.f;
    for (int k = 0; k < 6; k++) sum += o.MiJt[i][k] * o.Lam[k][cc];
    o.Jbar[i][cc] = sum;
  }
  MRE_SYNC();
  if (l < 6) {
    float sum = 0.f;
    for (int a = 0; a < 7; a++) sum += o.Jbar[a][l] * o.tn[a];
    o.Jbt[l] = sum;
  }
  MRE_SYNC();
  if (l < 7) {
    float t = 0.f, pj = 0.f;
    for (int r = 0; r < 6; r++) { t += o.J[r][l] * o.LF[r]; pj += o.J[r][l] * o.Jbt[r]; }
    s.ctrl[l] = t + (o.tn[l] - pj) + s.qfrc_bias[l];
  }
  MRE_SYNC();
}

}  // namespace mre
