// nlsolver_amd/csrc/nlsg_math.h — deterministic fp64 log / cos for the kernels.
//
// rnorm (nlsolver.h:2479-2485) needs log and cos; the device library's versions
// differ from glibc's in the last bits, which would make positions (and then
// selection indices) diverge from the CPU restatement. These are built from
// +,-,*,/, EXPLICIT fused multiply-adds (the Horner chains and the argument reductions: one
// v_fma_f64 instead of a multiply and an add — the kernels that draw normal variates are bound
// by exactly these instructions) and integer ops only, in a fixed operation order (the TU is
// compiled with -ffp-contract=off, so nothing else is ever fused), so the CPU restatement, which
// carries its own copy of the same algorithms with C's fma(), agrees bit for bit; vs libm they
// are within 1 ulp
// (tests/test_oracle_pso_golden.py). Algorithms: argument reduction + the
// classic minimax kernels (Sun fdlibm coefficient sets).
#pragma once

#ifndef __HIPCC_RTC__  // also compiled at run time for user objectives (nlsg_rtc.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace nlsg {

// a * b + k for a compile-time constant k: one v_fma_f64 with k in a scalar register pair. Left to
// itself hipcc writes a Horner step as two v_mov_b32 (the 64-bit constant into the destination)
// plus a v_fmac_f64 — three issue slots of the fp64 pipe instead of one, a fifth of the vector
// instructions of the kernels that draw normal variates. Same fused multiply-add, same bits.
__device__ inline double fma_k(double a, double b, double k) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
}

__device__ inline double det_log(double x) {
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                   Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                   Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                   Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  if (x != x || x < 0.0) return __builtin_nan("");
  if (x == 0.0) return -__builtin_inf();
  if (x == __builtin_inf()) return x;
  int k = 0;
  uint64_t u = static_cast<uint64_t>(__double_as_longlong(x));
  if ((u >> 52) == 0) {  // subnormal: scale up by 2^54
    x = x * 0x1p54;
    u = static_cast<uint64_t>(__double_as_longlong(x));
    k -= 54;
  }
  uint32_t hx = static_cast<uint32_t>(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  k += static_cast<int>(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  const double m = __longlong_as_double(
      static_cast<long long>((static_cast<uint64_t>(hx) << 32) | (u & 0xffffffffull)));
  const double f = m - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma_k(w, __builtin_fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma_k(w, fma_k(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double dk = static_cast<double>(k);
  return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

__device__ inline double det_kernel_cos(double x) {  // |x| <= pi/4
  constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                   C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                   C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double r = z * fma_k(z, fma_k(z, fma_k(z, fma_k(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + __builtin_fma(z, r, (1.0 - w) - hz);
}
__device__ inline double det_kernel_sin(double x) {  // |x| <= pi/4
  constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                   S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                   S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x;
  const double v = z * x;
  const double r = fma_k(z, fma_k(z, fma_k(z, __builtin_fma(z, S6, S5), S4), S3), S2);
  return __builtin_fma(v, fma_k(z, r, S1), x);
}

// cosine for |y| <= 64 (two-term Cody-Waite reduction by pi/2); NaN outside
__device__ inline double det_cos(double y) {
  constexpr double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                   pio2_1t = 6.07710050650619224932e-11;
  if (!(y >= -64.0 && y <= 64.0)) return __builtin_nan("");
  const double fn = floor(__builtin_fma(y, invpio2, 0.5));
  const double r = __builtin_fma(-fn, pio2_1t, __builtin_fma(-fn, pio2_1, y));
  const int q = static_cast<int>(static_cast<long long>(fn) & 3);
  const double c = det_kernel_cos(r), s = det_kernel_sin(r);
  return q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}

// n / d and sqrt(x) as the compiler expands them for gfx950 (v_rcp_f64 / v_rsq_f64 seed, the same
// Newton and residual steps, hence the same correctly rounded results — tests/test_math_gpu.py
// checks them against the CPU's division and sqrt on millions of arguments) minus the operand
// scaling and the special-case fix-up, for operands that need neither: the division for
// 1 < d < 4 and |n| < 1 (or n = 0), the root for x = 0, x = +inf or x >= 2^-767.
__device__ inline double div_midrange(double n, double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = n * y;
  const double r = __builtin_fma(-d, q, n);
  return __builtin_fma(r, y, q);
}
__device__ inline double sqrt_unscaled(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return (x == 0.0 || x == __builtin_inf()) ? x : g;
}

// One normal variate from one 64-bit draw z (rnorm, nlsolver.h:2479-2485):
// sqrt(-2 log u1) * cos(2 * 3.141593 * u2) with u1 = z 2^-64 and u2 = (z mod 2^32) 2^-32 — the
// values det_log and det_cos give (the CPU restatement calls exactly those), written for the
// arguments that occur here so that none of their other cases is paid for: u1 is 0 or a normal
// number in [2^-64, 1] (no NaN, sign, infinity or subnormal path; u1 = 0 -> log = -inf as a
// final select), the cosine's argument lies in [0, 6.3] (no range test, the quadrant from a
// 32-bit conversion, the sign of the result set by integer arithmetic instead of nested selects).
// The kernels that draw normal variates are bound by the vector unit's instruction count:
// ~30 fewer instructions per variate.
__device__ inline double det_rnorm(uint64_t zbits) {
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                   Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                   Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                   Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  constexpr double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                   pio2_1t = 6.07710050650619224932e-11;
  // u1 = (double)z 2^-64 rounded once, as the conversion rounds: hi 2^-32 + lo 2^-64 in one fma
  // (both terms exact); the angle y = 2 pi_ u2 = lo (2 pi_ 2^-32) — scaling by 2^-32 is exact, so
  // the one rounding is that of 2 pi_ u2
  const double hi_d = static_cast<double>(static_cast<uint32_t>(zbits >> 32));
  const double lo_d = static_cast<double>(static_cast<uint32_t>(zbits));
  const double u1 = __builtin_fma(hi_d, 0x1p-32, lo_d * 0x1p-64);
  // log u1 (det_log's normal-number path)
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(u1));
  uint32_t hx = static_cast<uint32_t>(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  const int k = static_cast<int>(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  const double m = __longlong_as_double(
      static_cast<long long>((static_cast<uint64_t>(hx) << 32) | (u & 0xffffffffull)));
  const double f = m - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = div_midrange(f, 2.0 + f);  // = f / (2.0 + f): 1.7 < 2 + f < 2.42
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma_k(w, __builtin_fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma_k(w, fma_k(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double dk = static_cast<double>(k);
  double lg = s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
  lg = u1 == 0.0 ? -__builtin_inf() : lg;
  // cos(2 pi_ u2) (det_cos's path for 0 <= y <= 64)
  const double y = lo_d * (2 * 3.141593 * 0x1p-32);
  const double fn = floor(__builtin_fma(y, invpio2, 0.5));
  const double r = __builtin_fma(-fn, pio2_1t, __builtin_fma(-fn, pio2_1, y));
  const uint32_t q = static_cast<uint32_t>(static_cast<int>(fn));  // 0 .. 4
  const double c = det_kernel_cos(r), sn = det_kernel_sin(r);
  // q mod 4 = 0: c, 1: -s, 2: -c, 3: s
  const uint64_t mag = static_cast<uint64_t>(__double_as_longlong((q & 1u) ? sn : c));
  const uint64_t flip = static_cast<uint64_t>((q + 1u) & 2u) << 62;
  const double cs = __longlong_as_double(static_cast<long long>(mag ^ flip));
  return sqrt_unscaled(-2 * lg) * cs;  // -2 lg is 0, +inf or at least 2^-53
}

// cos(2 pi x) the way the reference's Rastrigin writes it (test_functions.h:74-76): the product
// t = (2 M_PI) x is rounded first, then its cosine is taken. Beyond det_cos's range the period
// is taken off x itself — x - rint(x) is exact — which differs from a cosine of the rounded
// product by less than the rounding of that product (|x| > 10: 1e-15 absolute on a value
// added to x^2 > 100).
__device__ inline double det_cos_2pi(double x) {
  constexpr double two_pi = 2 * 3.14159265358979323846;
  double t = two_pi * x;
  if (!(t >= -64.0 && t <= 64.0)) t = two_pi * (x - rint(x));
  return det_cos(t);
}

// exp / tanh for the NLLS residual models (same algorithms as oracle_lm.c orc_exp/orc_tanh)
__device__ inline double det_exp(double x) {
  constexpr double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                   invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                   P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                   P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.0) return __builtin_inf();
  if (x < -708.0) return 0.0;
  const int k = static_cast<int>(__builtin_fma(invln2, x, x < 0 ? -0.5 : 0.5));
  const double hi = __builtin_fma(-static_cast<double>(k), ln2HI, x), lo = static_cast<double>(k) * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = __builtin_fma(-t, fma_k(t, fma_k(t, fma_k(t, __builtin_fma(t, P5, P4), P3), P2), P1), r);
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  return y * __longlong_as_double(static_cast<long long>(static_cast<uint64_t>(1023 + k) << 52));
}
__device__ inline double det_tanh(double x) {
  if (x != x) return x;
  const double ax = fabs(x);
  if (ax > 22.0) return x < 0 ? -1.0 : 1.0;
  const double e = det_exp(2.0 * ax);
  const double t = 1.0 - 2.0 / (e + 1.0);
  return x < 0 ? -t : t;
}

}  // namespace nlsg
