/* flop_count.h -- TEST INFRASTRUCTURE.  Counts the floating-point operations the CPU restatement executes.
 *
 * SURVEY.md section 8(d) asks for "counted FLOPs from the instrumented CPU restatement" as the algorithmic work per
 * env-step.  Instead of hand-placed counters, mre_oracle.c is compiled a second time AS C++ with `double` replaced by a
 * one-member class whose arithmetic operators bump a per-stage counter:
 *     g++ -x c++ -fpermissive -include flop_count.h -DMRO_COUNT_FLOPS mre_oracle.c -> libmre_oracle_flops.so
 * Same source, same results (the class holds a double and nothing else), every + - * / counted where it is executed.
 * Counted: add / sub / mul / div as 1 each (compound assignments too), sqrt and the transcendental calls as 1
 * "special" each (reported separately); comparisons, negation, abs, min / max and conversions are not FLOPs.
 * mre_oracle.c marks the pipeline stage with MRO_STAGE(k) (a no-op in the plain C build).
 */
#ifndef MRO_FLOP_COUNT_H
#define MRO_FLOP_COUNT_H
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MRO_NSTAGE 16
extern "C" {
extern __thread unsigned long long mro_flop_n[MRO_NSTAGE][2]; /* [stage][0: + - * /, 1: sqrt / trig / pow] */
extern __thread int mro_flop_stage;
}
#define MRO_CNT(k) (mro_flop_n[mro_flop_stage][k]++)

struct mro_cd {
  double v;
  mro_cd() = default;
  constexpr mro_cd(double x) : v(x) {}
  constexpr mro_cd(float x) : v(x) {}
  constexpr mro_cd(int x) : v(x) {}
  constexpr mro_cd(unsigned x) : v(x) {}
  constexpr mro_cd(long x) : v((double)x) {}
  constexpr mro_cd(unsigned long x) : v((double)x) {}
  constexpr mro_cd(long long x) : v((double)x) {}
  constexpr mro_cd(unsigned long long x) : v((double)x) {}
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  explicit operator long() const { return (long)v; }
  explicit operator unsigned() const { return (unsigned)v; }
  explicit operator long long() const { return (long long)v; }
  explicit operator unsigned long long() const { return (unsigned long long)v; }
  explicit operator bool() const { return v != 0.0; }
  mro_cd& operator+=(mro_cd b) { MRO_CNT(0); v += b.v; return *this; }
  mro_cd& operator-=(mro_cd b) { MRO_CNT(0); v -= b.v; return *this; }
  mro_cd& operator*=(mro_cd b) { MRO_CNT(0); v *= b.v; return *this; }
  mro_cd& operator/=(mro_cd b) { MRO_CNT(0); v /= b.v; return *this; }
};
static_assert(sizeof(mro_cd) == sizeof(double), "layout of the C ABI");
inline mro_cd operator+(mro_cd a, mro_cd b) { MRO_CNT(0); return a.v + b.v; }
inline mro_cd operator-(mro_cd a, mro_cd b) { MRO_CNT(0); return a.v - b.v; }
inline mro_cd operator*(mro_cd a, mro_cd b) { MRO_CNT(0); return a.v * b.v; }
inline mro_cd operator/(mro_cd a, mro_cd b) { MRO_CNT(0); return a.v / b.v; }
inline mro_cd operator-(mro_cd a) { return -a.v; }
inline mro_cd operator+(mro_cd a) { return a; }
inline bool operator<(mro_cd a, mro_cd b) { return a.v < b.v; }
inline bool operator>(mro_cd a, mro_cd b) { return a.v > b.v; }
inline bool operator<=(mro_cd a, mro_cd b) { return a.v <= b.v; }
inline bool operator>=(mro_cd a, mro_cd b) { return a.v >= b.v; }
inline bool operator==(mro_cd a, mro_cd b) { return a.v == b.v; }
inline bool operator!=(mro_cd a, mro_cd b) { return a.v != b.v; }
inline bool operator!(mro_cd a) { return a.v == 0.0; }
#define MRO_F1(name) inline mro_cd name(mro_cd x) { MRO_CNT(1); return ::name(x.v); }
MRO_F1(sqrt) MRO_F1(sin) MRO_F1(cos) MRO_F1(tan) MRO_F1(acos) MRO_F1(asin) MRO_F1(atan) MRO_F1(exp) MRO_F1(log)
inline mro_cd atan2(mro_cd y, mro_cd x) { MRO_CNT(1); return ::atan2(y.v, x.v); }
inline mro_cd pow(mro_cd x, mro_cd y) { MRO_CNT(1); return ::pow(x.v, y.v); }
inline mro_cd fabs(mro_cd x) { return ::fabs(x.v); }
inline mro_cd floor(mro_cd x) { return ::floor(x.v); }
inline mro_cd ceil(mro_cd x) { return ::ceil(x.v); }
inline mro_cd fmax(mro_cd a, mro_cd b) { return ::fmax(a.v, b.v); }
inline mro_cd fmin(mro_cd a, mro_cd b) { return ::fmin(a.v, b.v); }
inline mro_cd copysign(mro_cd a, mro_cd b) { return ::copysign(a.v, b.v); }
inline int mro_isfinite(mro_cd x) { return std::isfinite(x.v); }
inline int mro_isnan(mro_cd x) { return std::isnan(x.v); }
#undef isfinite
#undef isnan
#define isfinite(x) mro_isfinite(x)
#define isnan(x) mro_isnan(x)
#define double mro_cd
#endif
