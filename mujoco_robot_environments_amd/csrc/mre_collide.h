// mre_collide.h -- per-lane narrow phase (box-box, plane-box), fp32.
// One lane owns one collision pair; all scratch is lane-private.
// Box-box: separating-axis test over 15 axes (faces preferred over edges by a
// 5 % fudge), face contact = incident face clipped against the reference face
// (Sutherland-Hodgman, <= 8 points), edge contact = closest points of the two
// supporting edges.  Contact position is midway between the surfaces, normal
// points from geom1 to geom2, dist < 0 is penetration (MuJoCo conventions).
#pragma once
#include "mre_math.h"

namespace mre {

struct PairContacts {
  int n;
  float normal[3];
  float pos[8][3];
  float dist[8];
};

// clip polygon (x, y, depth) against |x| <= sx, |y| <= sy; in/out p, returns count
MRE_DEV int clip_poly(float (*p)[3], int n, float sx, float sy) {
  float q[16][3];
  for (int e = 0; e < 4; e++) {
    const int ax = e >> 1;
    const float sg = (e & 1) ? -1.0f : 1.0f;
    const float lim = ax ? sy : sx;
    int nq = 0;
    for (int i = 0; i < n; i++) {
      const float* a = p[i];
      const float* b = p[(i + 1) % n];
      const float da = sg * a[ax] - lim, db = sg * b[ax] - lim;
      if (da <= 0) { q[nq][0] = a[0]; q[nq][1] = a[1]; q[nq][2] = a[2]; nq++; }
      if ((da <= 0) != (db <= 0)) {
        const float t = da / (da - db);
        for (int k = 0; k < 3; k++) q[nq][k] = a[k] + t * (b[k] - a[k]);
        nq++;
      }
      if (nq >= 15) break;
    }
    n = nq;
    for (int i = 0; i < n; i++) { p[i][0] = q[i][0]; p[i][1] = q[i][1]; p[i][2] = q[i][2]; }
    if (n == 0) return 0;
  }
  return n;
}

// R1/R2: 3x3 row-major geom frames (columns = box axes)
MRE_DEV void box_box(const float* p1, const float* R1, const float* s1, const float* p2,
                     const float* R2, const float* s2, float margin, PairContacts& out) {
  out.n = 0;
  float A[3][3], B[3][3], dv[3], Cm[3][3], aC[3][3];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; B[i][k] = R2[3 * k + i]; }
  v3sub(dv, p2, p1);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { Cm[i][j] = v3dot(A[i], B[j]); aC[i][j] = fabsf(Cm[i][j]); }
  float tA[3], tB[3];
  for (int i = 0; i < 3; i++) { tA[i] = v3dot(dv, A[i]); tB[i] = v3dot(dv, B[i]); }
  float best_face = -1e30f;
  int face_code = -1;
  for (int i = 0; i < 3; i++) {
    const float sep = fabsf(tA[i]) - (s1[i] + s2[0] * aC[i][0] + s2[1] * aC[i][1] + s2[2] * aC[i][2]);
    if (sep > best_face) { best_face = sep; face_code = i; }
  }
  for (int j = 0; j < 3; j++) {
    const float sep = fabsf(tB[j]) - (s2[j] + s1[0] * aC[0][j] + s1[1] * aC[1][j] + s1[2] * aC[2][j]);
    if (sep > best_face) { best_face = sep; face_code = 3 + j; }
  }
  if (best_face > margin) return;
  float best_edge = -1e30f;
  int ei = -1, ej = -1;
  float en[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float L[3];
      v3cross(L, A[i], B[j]);
      const float len = v3norm(L);
      if (len < 1e-6f) continue;
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float rA = s1[i1] * aC[i2][j] + s1[i2] * aC[i1][j];
      const float rB = s2[j1] * aC[i][j2] + s2[j2] * aC[i][j1];
      const float t = v3dot(dv, L);
      const float sep = (fabsf(t) - rA - rB) / len;
      if (sep > margin) return;
      if (sep > best_edge) {
        best_edge = sep; ei = i; ej = j;
        const float sg = (t >= 0 ? 1.0f : -1.0f) / len;
        en[0] = L[0] * sg; en[1] = L[1] * sg; en[2] = L[2] * sg;
      }
    }
  const bool use_edge = (ei >= 0) && (best_edge - best_face > 0.05f * fabsf(best_face) + 1e-7f);
  if (use_edge) {
    float c1[3], c2[3];
    v3copy(c1, p1);
    v3copy(c2, p2);
    for (int k = 0; k < 3; k++) {
      if (k != ei) v3addscl(c1, A[k], (v3dot(en, A[k]) >= 0 ? 1.0f : -1.0f) * s1[k]);
      if (k != ej) v3addscl(c2, B[k], (v3dot(en, B[k]) >= 0 ? -1.0f : 1.0f) * s2[k]);
    }
    float w[3];
    v3sub(w, c1, c2);
    const float uv = Cm[ei][ej], uw = v3dot(A[ei], w), vw = v3dot(B[ej], w);
    const float den = 1.0f - uv * uv;
    float a = 0.f, bb = 0.f;
    if (den > 1e-12f) { a = (uv * vw - uw) / den; bb = (vw - uv * uw) / den; }
    float q1[3], q2[3], df[3];
    v3copy(q1, c1); v3addscl(q1, A[ei], a);
    v3copy(q2, c2); v3addscl(q2, B[ej], bb);
    v3copy(out.normal, en);
    for (int k = 0; k < 3; k++) out.pos[0][k] = 0.5f * (q1[k] + q2[k]);
    v3sub(df, q2, q1);
    out.dist[0] = v3dot(df, en);
    out.n = out.dist[0] <= margin ? 1 : 0;
    return;
  }
  const float *pr, *pi, *sr, *si;
  float (*Ar)[3];
  float (*Ai)[3];
  int a;
  float nr[3];
  if (face_code < 3) {
    a = face_code; pr = p1; pi = p2; sr = s1; si = s2; Ar = A; Ai = B;
    const float sg = tA[a] >= 0 ? 1.0f : -1.0f;
    for (int k = 0; k < 3; k++) { nr[k] = A[a][k] * sg; out.normal[k] = nr[k]; }
  } else {
    a = face_code - 3; pr = p2; pi = p1; sr = s2; si = s1; Ar = B; Ai = A;
    const float sg = tB[a] >= 0 ? -1.0f : 1.0f;
    for (int k = 0; k < 3; k++) { nr[k] = B[a][k] * sg; out.normal[k] = -nr[k]; }
  }
  int kk = 0;
  float bestd = -1.f;
  for (int k = 0; k < 3; k++) {
    const float dd = fabsf(v3dot(nr, Ai[k]));
    if (dd > bestd) { bestd = dd; kk = k; }
  }
  const float isg = v3dot(nr, Ai[kk]) >= 0 ? -1.0f : 1.0f;
  const int ku = (kk + 1) % 3, kv = (kk + 2) % 3;
  float ci[3], cr[3];
  v3copy(ci, pi);
  v3addscl(ci, Ai[kk], isg * si[kk]);
  const int au = (a + 1) % 3, av = (a + 2) % 3;
  v3copy(cr, pr);
  v3addscl(cr, nr, sr[a]);
  float poly[16][3];
  for (int v = 0; v < 4; v++) {
    const float su = (v == 0 || v == 3) ? 1.0f : -1.0f;
    const float sv = (v < 2) ? 1.0f : -1.0f;
    float w[3];
    v3copy(w, ci);
    v3addscl(w, Ai[ku], su * si[ku]);
    v3addscl(w, Ai[kv], sv * si[kv]);
    v3sub(w, w, cr);
    poly[v][0] = v3dot(w, Ar[au]);
    poly[v][1] = v3dot(w, Ar[av]);
    poly[v][2] = v3dot(w, nr);
  }
  const int n = clip_poly(poly, 4, sr[au], sr[av]);
  int nc = 0;
  for (int v = 0; v < n && nc < 8; v++) {
    const float dep = poly[v][2];
    if (dep > margin) continue;
    for (int k = 0; k < 3; k++)
      out.pos[nc][k] = cr[k] + poly[v][0] * Ar[au][k] + poly[v][1] * Ar[av][k] + 0.5f * dep * nr[k];
    out.dist[nc] = dep;
    nc++;
  }
  out.n = nc;
}

// plane (geom1, normal = +z of its frame) vs box: corners within margin, at most 4
MRE_DEV void plane_box(const float* pp, const float* Rp, const float* pb, const float* Rb,
                       const float* sb, float margin, PairContacts& out) {
  out.n = 0;
  float n[3] = {Rp[2], Rp[5], Rp[8]};
  v3copy(out.normal, n);
  int cnt = 0;
  for (int c = 0; c < 8 && cnt < 4; c++) {
    float loc[3] = {(c & 1) ? sb[0] : -sb[0], (c & 2) ? sb[1] : -sb[1], (c & 4) ? sb[2] : -sb[2]};
    float w[3], wd[3];
    m3mulv(w, Rb, loc);
    v3add(w, w, pb);
    v3sub(wd, w, pp);
    const float ds = v3dot(wd, n);
    if (ds > margin) continue;
    v3addscl(w, n, -0.5f * ds);
    v3copy(out.pos[cnt], w);
    out.dist[cnt] = ds;
    cnt++;
  }
  out.n = cnt;
}

// mju_makeFrame: f[0:3] unit normal given, builds the two tangents
MRE_DEV void make_frame(float* f) {
  float y[3] = {0.f, 0.f, 0.f};
  if (f[1] < 0.5f && f[1] > -0.5f) y[1] = 1.f; else y[2] = 1.f;
  const float t = v3dot(f, y);
  v3addscl(y, f, -t);
  v3normalize(y);
  v3copy(f + 3, y);
  v3cross(f + 6, f, f + 3);
}

}  // namespace mre
