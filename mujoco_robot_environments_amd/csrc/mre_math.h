// mre_math.h -- small fp32 vector / quaternion / spatial-algebra helpers for
// the gfx950 kernels (per-lane scalar code; cross-lane work lives in the kernels).
#pragma once
#include <hip/hip_runtime.h>

#define MRE_DEV __device__ __forceinline__
// whole phases are real functions: register allocation is scoped per phase instead of across
// the fused step loop (the inlined kernel needed 332 registers -> 1 wave per SIMD)
// Diagnostic builds (-DMRE_PHASE_STAMPS=4, 5, ...: tools/phase_stamps.py): time inside a phase function by part,
// summed by lane 0 into four LDS words that the kernel returns through the stats rows.
#if defined(MRE_PHASE_STAMPS) && MRE_PHASE_STAMPS >= 4
#define MRE_DBG_T0() unsigned long long dbg_t_ = __builtin_amdgcn_s_memtime()
#define MRE_DBG_STAMP(SET, K)                                                      \
  do {                                                                             \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();                    \
    if (MRE_PHASE_STAMPS == (SET) && threadIdx.x == 0) dbg_acc[K] += n_ - dbg_t_;  \
    dbg_t_ = n_;                                                                   \
  } while (0)
#else
#define MRE_DBG_T0() do {} while (0)
#define MRE_DBG_STAMP(SET, K) do {} while (0)
#endif
#define MRE_PHASE_FN __device__ __attribute__((noinline))
// Ordering point between lanes of the ONE wave that steps an env.  A workgroup is a single wavefront, whose
// LDS instructions issue and execute in order, so lanes see each other's earlier LDS writes without waiting for
// them: all that is needed is that the compiler keeps the memory operations on their side of the point.
// __syncthreads() would add `s_waitcnt lgkmcnt(0)` (drain every LDS operation in flight, writes included) to
// each of the few hundred ordering points of a step.
#define MRE_SYNC()                                        \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
  } while (0)
// Pointer to the uploaded model, typed as a GLOBAL-address-space pointer.  A plain `const DevModel*` that
// reaches a noinline phase function (or comes out of the by-value launch arguments) is a generic pointer to
// the compiler: its loads become flat_load, which count on BOTH memory counters, so every LDS wait also
// waits for the model constants in flight.  With the address space in the type the loads are global_load
// (vmcnt only) and overlap the LDS traffic.
// (the host pass of hipcc parses the device functions too but has no address spaces to convert between)
#if defined(__HIP_DEVICE_COMPILE__)
#define MRE_MODEL_PTR(T) const T __attribute__((address_space(1)))*
#else
#define MRE_MODEL_PTR(T) const T*
#endif

namespace mre {

constexpr float kMinVal = 1e-15f;

MRE_DEV void v3copy(float* r, const float* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
MRE_DEV void v3zero(float* r) { r[0] = r[1] = r[2] = 0.f; }
MRE_DEV void v3add(float* r, const float* a, const float* b) {
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
}
MRE_DEV void v3sub(float* r, const float* a, const float* b) {
  r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2];
}
MRE_DEV float v3dot(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MRE_DEV void v3cross(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
MRE_DEV void v3addscl(float* r, const float* a, float s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
MRE_DEV float v3norm(const float* a) { return sqrtf(v3dot(a, a)); }
MRE_DEV float v3normalize(float* a) {
  float n = v3norm(a);
  if (n < kMinVal) { a[0] = 1.f; a[1] = 0.f; a[2] = 0.f; return n; }
  float inv = 1.0f / n;
  a[0] *= inv; a[1] *= inv; a[2] *= inv;
  return n;
}
// r = M(3x3 row-major) v
MRE_DEV void m3mulv(float* r, const float* m, const float* v) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  float y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  float z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MRE_DEV void m3tmulv(float* r, const float* m, const float* v) {
  float x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  float y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  float z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MRE_DEV void qmul(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
MRE_DEV void qnormalize(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < kMinVal) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; return; }
  float inv = 1.0f / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
MRE_DEV void q2mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
MRE_DEV void qrotv(float* r, const float* q, const float* v) {
  float m[9];
  q2mat(m, q);
  m3mulv(r, m, v);
}
// sin and cos of an angle of a few radians (half joint angles, half rotation steps): three-constant
// Cody-Waite reduction by pi/2 and the single-precision minimax kernels on [-pi/4, pi/4] (about 1 ulp).
// sincosf() carries a Payne-Hanek reduction for arguments up to 1e38 -- a hundred instructions that the
// kinematics would execute on each of its nine tree levels.
MRE_DEV void sincos_small(float x, float* sn, float* cs) {
  const float j = rintf(x * 0.6366197723675814f);
  float r = fmaf(-j, 1.5703125f, x);
  r = fmaf(-j, 4.837512969970703125e-4f, r);
  r = fmaf(-j, 7.54978995489188192e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                        fmaf(-0.5f, z, 1.0f));
  const int q = (int)j & 3;
  const float a = (q & 1) ? pc : ps, b = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}
MRE_DEV void axisangle2q(float* q, const float* axis, float angle) {
  float s, c;
  sincos_small(0.5f * angle, &s, &c);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
MRE_DEV void mat2q(float* q, const float* m) {
  float t = m[0] + m[4] + m[8];
  if (t > 0) {
    float s = sqrtf(t + 1.0f) * 2;
    q[0] = 0.25f * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    float s = sqrtf(1.0f + m[0] - m[4] - m[8]) * 2;
    q[0] = (m[7] - m[5]) / s; q[1] = 0.25f * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    float s = sqrtf(1.0f + m[4] - m[0] - m[8]) * 2;
    q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25f * s; q[3] = (m[5] + m[7]) / s;
  } else {
    float s = sqrtf(1.0f + m[8] - m[0] - m[4]) * 2;
    q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25f * s;
  }
  qnormalize(q);
}
// spatial inertia (10) times motion vector (6): [rot; lin]
MRE_DEV void mul_inert_vec(float* r, const float* i, const float* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
MRE_DEV void cross_motion(float* r, const float* vel, const float* v) {
  float a[3], b[3];
  v3cross(r, vel, v);
  v3cross(a, vel, v + 3);
  v3cross(b, vel + 3, v);
  v3add(r + 3, a, b);
}
MRE_DEV void cross_force(float* r, const float* vel, const float* f) {
  float a[3], b[3];
  v3cross(a, vel, f);
  v3cross(b, vel + 3, f + 3);
  v3add(r, a, b);
  v3cross(r + 3, vel, f + 3);
}
MRE_DEV float dot6(const float* a, const float* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// wave64 sum, result uniform in every lane: DPP row reduction (quad_perm, row_half_mirror,
// row_mirror), row_bcast15 / row_bcast31 across the four 16-lane rows, readlane 63.
template <int CTRL, int ROW_MASK>
MRE_DEV float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, t);
}
MRE_DEV float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 -> rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 -> rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Three wave sums at once: the three DPP chains interleave, so the two wait states a DPP read needs after
// the write of its source are filled with the other chains' adds instead of s_nop (the compiler serialises
// three wave_sum calls: 6 adds + 6 s_nop each).  Same additions as wave_sum, same bits.
MRE_DEV void wave_sum3(float& a, float& b, float& c) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "+v"(c));
  a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
  b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
  c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
}
template <int CTRL, int ROW_MASK>
MRE_DEV float dpp_max(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return fmaxf(v, __builtin_bit_cast(float, t));
}
// wave64 maximum of non-negative values (lanes a DPP step does not reach contribute 0), uniform result
MRE_DEV float wave_max(float v) {
  v = dpp_max<0xB1, 0xF>(v);
  v = dpp_max<0x4E, 0xF>(v);
  v = dpp_max<0x141, 0xF>(v);
  v = dpp_max<0x140, 0xF>(v);
  v = dpp_max<0x142, 0xA>(v);
  v = dpp_max<0x143, 0xC>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
MRE_DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// fp64 reciprocal and reciprocal square root (PGS: robot-contact block update; integrator: the cubes' quaternions): the hardware estimates (v_rcp_f64 /
// v_rsq_f64, ~2^-26 relative) + two Newton steps = 1e-16, a third of the instructions of the IEEE division / sqrt
// sequences with their scaling and fix-up steps (operands here are never denormal, zero or infinite: guarded by the callers)
MRE_DEV double rcp64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  return r * (2.0 - x * r);
}
MRE_DEV double rsq64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  return y * (1.5 - 0.5 * x * y * y);
}


}  // namespace mre
