// mre_collide.h -- per-lane narrow phase (box-box, plane-box), fp32.
// One lane owns one collision pair.  No private-memory arrays with runtime indices:
// the clip polygons live in a per-lane LDS buffer (56 floats, aliased by the caller
// on the Jacobian pools that are not in use yet) and small vectors are picked with
// select chains, so the kernel needs no scratch memory.
// Box-box: separating-axis test over 15 axes (faces preferred over edges by a
// 5 % fudge), face contact = incident face clipped against the reference face
// (Sutherland-Hodgman, <= 8 points), edge contact = closest points of the two
// supporting edges.  Contact position is midway between the surfaces, normal
// points from geom1 to geom2, dist < 0 is penetration (MuJoCo conventions).
// Output: candidate k < n = position cand_xyz(buf,k)[0..2] and distance cand_dist(buf,k).
#pragma once
#include "mre_math.h"

namespace mre {

constexpr int COLL_BUF = 48;   // floats per lane: poly[8][3] | q[8][3]; candidates: xyz in q, dist in poly[.][2]
MRE_DEV float* cand_xyz(float* buf, int k) { return buf + 24 + 3 * k; }
MRE_DEV float& cand_dist(float* buf, int k) { return buf[3 * k + 2]; }

// runtime pick of one of three values by arithmetic masks (a select chain over a local
// array is turned back into an indexed stack access by the compiler -> scratch memory)
MRE_DEV float sel3(const float* v, int i) {
  const float w0 = i == 0 ? 1.f : 0.f, w1 = i == 1 ? 1.f : 0.f, w2 = i == 2 ? 1.f : 0.f;
  return w0 * v[0] + w1 * v[1] + w2 * v[2];
}
MRE_DEV void row3(float* o, const float (*m)[3], int i) {
  const float w0 = i == 0 ? 1.f : 0.f, w1 = i == 1 ? 1.f : 0.f, w2 = i == 2 ? 1.f : 0.f;
  o[0] = w0 * m[0][0] + w1 * m[1][0] + w2 * m[2][0];
  o[1] = w0 * m[0][1] + w1 * m[1][1] + w2 * m[2][1];
  o[2] = w0 * m[0][2] + w1 * m[1][2] + w2 * m[2][2];
}

// clip polygon p (x, y, depth) against |x| <= sx, |y| <= sy using q as the second buffer;
// result ends in p.  p, q are LDS pointers private to this lane.
MRE_DEV int clip_poly(float* p, float* q, int n, float sx, float sy) {
  // the two buffers swap roles from edge to edge (four edges: the result is back in p), and a vertex that was
  // the end of one polygon edge stays in registers as the start of the next
  float* src = p;
  float* dst = q;
  for (int e = 0; e < 4; e++) {
    const int ax = e >> 1;
    const float sg = (e & 1) ? -1.0f : 1.0f;
    const float lim = ax ? sy : sx;
    int nq = 0;
    const float f0 = src[0], f1 = src[1], f2 = src[2];
    float a0 = f0, a1 = f1, a2 = f2;
    for (int i = 0; i < n; i++) {
      const bool last = i + 1 == n;
      const int i2 = last ? 0 : i + 1;
      const float b0 = last ? f0 : src[3 * i2], b1 = last ? f1 : src[3 * i2 + 1], b2 = last ? f2 : src[3 * i2 + 2];
      const float da = sg * (ax ? a1 : a0) - lim, db = sg * (ax ? b1 : b0) - lim;
      if (da <= 0 && nq < 8) { dst[3 * nq] = a0; dst[3 * nq + 1] = a1; dst[3 * nq + 2] = a2; nq++; }
      if ((da <= 0) != (db <= 0) && nq < 8) {
        const float t = da / (da - db);
        dst[3 * nq] = a0 + t * (b0 - a0); dst[3 * nq + 1] = a1 + t * (b1 - a1); dst[3 * nq + 2] = a2 + t * (b2 - a2);
        nq++;
      }
      a0 = b0; a1 = b1; a2 = b2;
    }
    n = nq;
    if (n == 0) return 0;
    float* t = src; src = dst; dst = t;
  }
  return n;
}

// R1/R2: 3x3 row-major geom frames (columns = box axes); buf: this lane's LDS buffer.
// Returns the number of candidates written (cand_xyz / cand_dist).
MRE_DEV int box_box(const float* p1, const float* R1, const float* s1, const float* p2,
                    const float* R2, const float* s2, float margin, float* normal, float* buf) {
  float* poly = buf;
  float* qbuf = buf + 24;
  float A[3][3], B[3][3], dv[3], Cm[3][3], aC[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; B[i][k] = R2[3 * k + i]; }
  v3sub(dv, p2, p1);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { Cm[i][j] = v3dot(A[i], B[j]); aC[i][j] = fabsf(Cm[i][j]); }
  float tA[3], tB[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { tA[i] = v3dot(dv, A[i]); tB[i] = v3dot(dv, B[i]); }
  float best_face = -1e30f;
  int face_code = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float sep = fabsf(tA[i]) - (s1[i] + s2[0] * aC[i][0] + s2[1] * aC[i][1] + s2[2] * aC[i][2]);
    if (sep > best_face) { best_face = sep; face_code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const float sep = fabsf(tB[j]) - (s2[j] + s1[0] * aC[0][j] + s1[1] * aC[1][j] + s1[2] * aC[2][j]);
    if (sep > best_face) { best_face = sep; face_code = 3 + j; }
  }
  if (best_face > margin) return 0;
  float best_edge = -1e30f;
  int ei = -1, ej = -1;
  float en[3] = {0.f, 0.f, 0.f};
  bool separated = false;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float L[3];
      v3cross(L, A[i], B[j]);
      const float len = v3norm(L);
      if (len >= 1e-6f) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const float rA = s1[i1] * aC[i2][j] + s1[i2] * aC[i1][j];
        const float rB = s2[j1] * aC[i][j2] + s2[j2] * aC[i][j1];
        const float t = v3dot(dv, L);
        const float sep = (fabsf(t) - rA - rB) / len;
        if (sep > margin) separated = true;
        if (sep > best_edge) {
          best_edge = sep; ei = i; ej = j;
          const float sg = (t >= 0 ? 1.0f : -1.0f) / len;
          en[0] = L[0] * sg; en[1] = L[1] * sg; en[2] = L[2] * sg;
        }
      }
    }
  if (separated) return 0;
  const bool use_edge = (ei >= 0) && (best_edge - best_face > 0.05f * fabsf(best_face) + 1e-7f);
  if (use_edge) {
    float c1[3], c2[3];
    v3copy(c1, p1);
    v3copy(c2, p2);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k != ei) v3addscl(c1, A[k], (v3dot(en, A[k]) >= 0 ? 1.0f : -1.0f) * s1[k]);
      if (k != ej) v3addscl(c2, B[k], (v3dot(en, B[k]) >= 0 ? -1.0f : 1.0f) * s2[k]);
    }
    float Ae[3], Be[3], w[3];
    row3(Ae, A, ei);
    row3(Be, B, ej);
    v3sub(w, c1, c2);
    const float uv = v3dot(Ae, Be), uw = v3dot(Ae, w), vw = v3dot(Be, w);
    const float den = 1.0f - uv * uv;
    float a = 0.f, bb = 0.f;
    if (den > 1e-12f) { a = (uv * vw - uw) / den; bb = (vw - uv * uw) / den; }
    float q1[3], q2[3], df[3];
    v3copy(q1, c1); v3addscl(q1, Ae, a);
    v3copy(q2, c2); v3addscl(q2, Be, bb);
    v3copy(normal, en);
    v3sub(df, q2, q1);
    const float d = v3dot(df, en);
    float* c0 = cand_xyz(buf, 0);
    c0[0] = 0.5f * (q1[0] + q2[0]); c0[1] = 0.5f * (q1[1] + q2[1]); c0[2] = 0.5f * (q1[2] + q2[2]);
    cand_dist(buf, 0) = d;
    return d <= margin ? 1 : 0;
  }
  // face contact: the reference box owns the separating face
  const bool ref1 = face_code < 3;
  const int a = ref1 ? face_code : face_code - 3;
  float pr[3], pi[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { pr[k] = ref1 ? p1[k] : p2[k]; pi[k] = ref1 ? p2[k] : p1[k]; }
  float Ar[3][3], Ai[3][3], srv3[3], siv3[3];  // element-wise selects keep everything in registers
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int k = 0; k < 3; k++) { Ar[i][k] = ref1 ? A[i][k] : B[i][k]; Ai[i][k] = ref1 ? B[i][k] : A[i][k]; }
    srv3[i] = ref1 ? s1[i] : s2[i];
    siv3[i] = ref1 ? s2[i] : s1[i];
  }
  float nr[3], Ara[3];
  row3(Ara, Ar, a);
  {
    const float ta = ref1 ? sel3(tA, a) : sel3(tB, a);
    const float sg = ref1 ? (ta >= 0 ? 1.0f : -1.0f) : (ta >= 0 ? -1.0f : 1.0f);
    for (int k = 0; k < 3; k++) { nr[k] = Ara[k] * sg; normal[k] = ref1 ? nr[k] : -nr[k]; }
  }
  int kk = 0;
  float bestd = -1.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float dd = fabsf(v3dot(nr, Ai[k]));
    if (dd > bestd) { bestd = dd; kk = k; }
  }
  const int ku = (kk + 1) % 3, kv = (kk + 2) % 3, au = (a + 1) % 3, av = (a + 2) % 3;
  float Aik[3], Aiu[3], Aiv[3], Aru[3], Arv[3];
  row3(Aik, Ai, kk); row3(Aiu, Ai, ku); row3(Aiv, Ai, kv);
  row3(Aru, Ar, au); row3(Arv, Ar, av);
  const float sik = sel3(siv3, kk), siu = sel3(siv3, ku), siv = sel3(siv3, kv);
  const float sra = sel3(srv3, a), sru = sel3(srv3, au), srv = sel3(srv3, av);
  const float isg = v3dot(nr, Aik) >= 0 ? -1.0f : 1.0f;
  float ci[3], cr[3];
  v3copy(ci, pi);
  v3addscl(ci, Aik, isg * sik);
  v3copy(cr, pr);
  v3addscl(cr, nr, sra);
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const float su = (v == 0 || v == 3) ? 1.0f : -1.0f;
    const float sv = (v < 2) ? 1.0f : -1.0f;
    float w[3];
    v3copy(w, ci);
    v3addscl(w, Aiu, su * siu);
    v3addscl(w, Aiv, sv * siv);
    v3sub(w, w, cr);
    poly[3 * v] = v3dot(w, Aru);
    poly[3 * v + 1] = v3dot(w, Arv);
    poly[3 * v + 2] = v3dot(w, nr);
  }
  const int n = clip_poly(poly, qbuf, 4, sru, srv);
  int nc = 0;
  for (int v = 0; v < n; v++) {
    const float x = poly[3 * v], y = poly[3 * v + 1], dep = poly[3 * v + 2];
    if (dep > margin) continue;
    float* cx = cand_xyz(buf, nc);  // nc <= v: q is free after clipping, poly[nc][2] already consumed
    cx[0] = cr[0] + x * Aru[0] + y * Arv[0] + 0.5f * dep * nr[0];
    cx[1] = cr[1] + x * Aru[1] + y * Arv[1] + 0.5f * dep * nr[1];
    cx[2] = cr[2] + x * Aru[2] + y * Arv[2] + 0.5f * dep * nr[2];
    cand_dist(buf, nc) = dep;
    nc++;
  }
  return nc;
}

// plane (geom1, normal = +z of its frame) vs box: corners within margin, at most 4
MRE_DEV int plane_box(const float* pp, const float* Rp, const float* pb, const float* Rb,
                      const float* sb, float margin, float* normal, float* buf) {
  float n[3] = {Rp[2], Rp[5], Rp[8]};
  v3copy(normal, n);
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
    float* cx = cand_xyz(buf, cnt);
    cx[0] = w[0]; cx[1] = w[1]; cx[2] = w[2];
    cand_dist(buf, cnt) = ds;
    cnt++;
  }
  return cnt;
}

// mju_makeFrame: f[0:3] unit normal given, builds the two tangents
// Cylinder (axis = local z of Rc, radius r, half height h; geom 2) against a box (geom 1): ONE contact, the normal from
// the box to the cylinder -- the float32 form of the oracle's mro_cylbox (same candidate directions in the same order:
// box face normals, cylinder axis, axis x box edges, closest points of the axis segment and the box; same contact
// point rule).  MuJoCo hands this pair to its general convex collider, whose iterative penetration query has no
// closed form to restate; tasks/push.py:153-160, tasks/lasa_draw.py:115-122 attach the cylinder.
MRE_DEV int cyl_box(const float* pb, const float* Rb, const float* sb, const float* pc, const float* Rc,
                    float r, float h, float margin, float* normal, float* buf) {
  float df[3], c[3], a[3];
  v3sub(df, pc, pb);
  for (int k = 0; k < 3; k++) {
    c[k] = Rb[k] * df[0] + Rb[3 + k] * df[1] + Rb[6 + k] * df[2];
    a[k] = Rb[k] * Rc[2] + Rb[3 + k] * Rc[5] + Rb[6 + k] * Rc[8];
  }
  float best = -1e30f, b0 = 0.f, b1 = 0.f, b2 = 1.f;
  auto tryd = [&](float dx, float dy, float dz) {
    float dc = dx * c[0] + dy * c[1] + dz * c[2];
    if (dc < 0.f) { dx = -dx; dy = -dy; dz = -dz; dc = -dc; }
    const float ad = a[0] * dx + a[1] * dy + a[2] * dz, rad = fmaxf(1.f - ad * ad, 0.f);
    const float sep = dc - (sb[0] * fabsf(dx) + sb[1] * fabsf(dy) + sb[2] * fabsf(dz)) - (h * fabsf(ad) + r * sqrtf(rad));
    if (sep > best + 1e-7f) { best = sep; b0 = dx; b1 = dy; b2 = dz; }
  };
  tryd(1.f, 0.f, 0.f); tryd(0.f, 1.f, 0.f); tryd(0.f, 0.f, 1.f);
  tryd(a[0], a[1], a[2]);
  {
    const float l0 = sqrtf(a[2] * a[2] + a[1] * a[1]), l1 = sqrtf(a[2] * a[2] + a[0] * a[0]), l2 = sqrtf(a[1] * a[1] + a[0] * a[0]);
    if (l0 > 1e-6f) tryd(0.f, a[2] / l0, -a[1] / l0);
    if (l1 > 1e-6f) tryd(-a[2] / l1, 0.f, a[0] / l1);
    if (l2 > 1e-6f) tryd(a[1] / l2, -a[0] / l2, 0.f);
  }
  {
    // closest points of the axis segment and the box (golden section on the convex f(t) = dist^2(c + t a, box))
    float lo = -h, hi = h;
    const float g = 0.61803399f;
    float t1 = hi - g * (hi - lo), t2 = lo + g * (hi - lo);
    for (int it = 0; it < 26; it++) {
      float f1 = 0.f, f2 = 0.f;
      for (int k = 0; k < 3; k++) {
        const float u1 = fabsf(c[k] + t1 * a[k]) - sb[k], u2 = fabsf(c[k] + t2 * a[k]) - sb[k];
        if (u1 > 0.f) f1 += u1 * u1;
        if (u2 > 0.f) f2 += u2 * u2;
      }
      if (f1 <= f2) { hi = t2; t2 = t1; t1 = hi - g * (hi - lo); }
      else { lo = t1; t1 = t2; t2 = lo + g * (hi - lo); }
    }
    const float t = 0.5f * (lo + hi);
    float v[3];
    for (int k = 0; k < 3; k++) {
      const float q = c[k] + t * a[k];
      v[k] = q - fminf(fmaxf(q, -sb[k]), sb[k]);
    }
    const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (vn > 1e-7f) tryd(v[0] / vn, v[1] / vn, v[2] / vn);
  }
  // refinement for rim-against-edge / corner poses (mro_cylbox): the direction from the box to the cylinder's deepest
  // point along the best direction so far is one more candidate
  for (int it = 0; it < 3; it++) {
    const float adr = a[0] * b0 + a[1] * b1 + a[2] * b2;
    float wr[3] = {b0 - adr * a[0], b1 - adr * a[1], b2 - adr * a[2]};
    const float wrn = sqrtf(wr[0] * wr[0] + wr[1] * wr[1] + wr[2] * wr[2]);
    const float sh = adr > 0.f ? -h : h;
    float v[3];
    for (int k = 0; k < 3; k++) {
      const float xr = c[k] + sh * a[k] - (wrn > 1e-6f ? r * wr[k] / wrn : 0.f);
      v[k] = xr - fminf(fmaxf(xr, -sb[k]), sb[k]);
    }
    const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (vn < 1e-7f) break;
    const float before = best;
    tryd(v[0] / vn, v[1] / vn, v[2] / vn);
    if (!(best > before)) break;
  }
  if (best > margin) return 0;
  const float bd[3] = {b0, b1, b2};
  const float ad = a[0] * b0 + a[1] * b1 + a[2] * b2;
  float w[3] = {b0 - ad * a[0], b1 - ad * a[1], b2 - ad * a[2]};
  const float wn = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  float x[3] = {c[0], c[1], c[2]};
  if (fabsf(ad) > 1e-4f) {   // (parallel within 1e-4 rad: mro_cylbox)
    const float sh = ad > 0.f ? -h : h;
    for (int k = 0; k < 3; k++) x[k] += sh * a[k];
  } else {
    const float tau0 = -(c[0] * a[0] + c[1] * a[1] + c[2] * a[2]);
    const float E = sb[0] * fabsf(a[0]) + sb[1] * fabsf(a[1]) + sb[2] * fabsf(a[2]);
    const float lo = fmaxf(tau0 - E, -h), hi = fminf(tau0 + E, h);
    const float t = lo <= hi ? 0.5f * (lo + hi) : fminf(fmaxf(tau0, -h), h);
    for (int k = 0; k < 3; k++) x[k] += t * a[k];
  }
  if (wn > 1e-4f) {
    for (int k = 0; k < 3; k++) x[k] -= r * w[k] / wn;
  } else {
    float o[3];
    for (int k = 0; k < 3; k++) o[k] = fminf(fmaxf(x[k], -sb[k]), sb[k]) - x[k];
    const float oa = o[0] * a[0] + o[1] * a[1] + o[2] * a[2];
    for (int k = 0; k < 3; k++) o[k] -= oa * a[k];
    const float on = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    const float sc = on > r ? r / on : 1.f;
    for (int k = 0; k < 3; k++) x[k] += sc * o[k];
  }
  for (int k = 0; k < 3; k++) x[k] -= 0.5f * best * bd[k];
  float* c0 = cand_xyz(buf, 0);
  for (int k = 0; k < 3; k++) {
    c0[k] = pb[k] + Rb[3 * k] * x[0] + Rb[3 * k + 1] * x[1] + Rb[3 * k + 2] * x[2];
    normal[k] = Rb[3 * k] * bd[0] + Rb[3 * k + 1] * bd[1] + Rb[3 * k + 2] * bd[2];
  }
  cand_dist(buf, 0) = best;
  return 1;
}

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
