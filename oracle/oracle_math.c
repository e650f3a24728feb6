/* oracle/oracle_math.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Deterministic log / cos built from +,-,*,/ and explicit fma() only (no libm
 * transcendental; fma is exactly specified), so that the
 * synchronous restatements and the HIP kernels — which carry their own copy of
 * the same algorithms in nlsolver_amd/csrc/nlsg_math.h — agree bit for bit.
 * Accuracy ~1 ulp; the serial restatements keep libm, like the reference
 * (rnorm, nlsolver.h:2479-2485).
 *
 * Algorithms: classic argument reduction + minimax kernels (Sun fdlibm
 * e_log.c / k_cos.c / k_sin.c coefficient sets, public since 1993).
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

static uint64_t bits_of(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
static double from_bits(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

/* natural logarithm for finite x >= 0 (x == 0 -> -inf, x < 0 or NaN -> NaN) */
double orc_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                      Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                      Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                      Lg7 = 1.479819860511658591e-01;
  if (x != x || x < 0.0) return NAN;
  if (x == 0.0) return -INFINITY;
  if (x == INFINITY) return x;
  int k = 0;
  uint64_t u = bits_of(x);
  if ((u >> 52) == 0) { /* subnormal: scale up by 2^54 */
    x = x * 0x1p54;
    u = bits_of(x);
    k -= 54;
  }
  /* x = 2^k * m with m in [sqrt(1/2), sqrt(2)) */
  uint32_t hx = (uint32_t)(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  k += (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  const double m = from_bits(((uint64_t)hx << 32) | (u & 0xffffffffull));
  const double f = m - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double dk = (double)k;
  return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

static double kernel_cos(double x) { /* |x| <= pi/4 */
  static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                      C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                      C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + fma(z, r, (1.0 - w) - hz);
}
static double kernel_sin(double x) { /* |x| <= pi/4 */
  static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                      S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                      S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x;
  const double v = z * x;
  const double r = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
  return fma(v, fma(z, r, S1), x);
}

/* cos(2 pi x) as nlsg_math.h det_cos_2pi (the device's Rastrigin term) */
double orc_cos_2pi(double x) {
  const double two_pi = 2 * 3.14159265358979323846;
  double t = two_pi * x;
  if (!(t >= -64.0 && t <= 64.0)) t = two_pi * (x - rint(x));
  return orc_cos(t);
}

/* cosine for |y| <= 64 (two-term Cody-Waite reduction by pi/2); NaN outside */
double orc_cos(double y) {
  static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                      pio2_1t = 6.07710050650619224932e-11;
  if (!(y >= -64.0 && y <= 64.0)) return NAN;
  const double fn = floor(fma(y, invpio2, 0.5));
  const double r = fma(-fn, pio2_1t, fma(-fn, pio2_1, y));
  const int q = (int)((long long)fn & 3);
  switch (q) {
    case 0: return kernel_cos(r);
    case 1: return -kernel_sin(r);
    case 2: return -kernel_cos(r);
    default: return kernel_sin(r);
  }
}

/* One normal variate from one 64-bit draw (rnorm, nlsolver.h:2479-2485): u1 (the radius) from
 * all of the draw, u2 (the angle) from its low 32 bits — what nlsg_math.h's det_rnorm computes
 * on the device. */
double orc_rnorm(uint64_t z1) {
  const double pi_ = 3.141593; /* the reference's literal, :2480 */
  const double u1 = orc_u01(z1), u2 = (double)(uint32_t)z1 * 0x1p-32;
  return sqrt(-2 * orc_log(u1)) * orc_cos(2 * pi_ * u2);
}

/* The primitives on arrays of bit patterns (the CPU side of nlsg_probe_math; fn = nlsg_probe_fn) */
void orc_probe_math(int fn, const uint64_t *in, uint64_t *out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    const double x = from_bits(in[i]);
    double r;
    switch (fn) {
      case 0: r = orc_log(x); break;
      case 1: r = orc_cos(x); break;
      case 2: r = orc_exp(x); break;
      case 3: r = orc_tanh(x); break;
      case 4: r = orc_cos_2pi(x); break;
      case 5: r = orc_u01(in[i]); break;
      default: r = orc_rnorm(in[i]); break;
    }
    out[i] = bits_of(r);
  }
}
