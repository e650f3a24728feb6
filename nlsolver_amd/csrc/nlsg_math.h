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

// sqrt(x) as the compiler expands it for gfx950 (v_rsq_f64 seed, the same Newton and residual
// steps, hence the same correctly rounded result — tests/test_math_gpu.py checks det_rnorm against
// the CPU's sqrt on millions of arguments) minus the operand scaling, for x = 0, x = +inf or
// x >= 2^-767.
// n / d as the compiler expands it (v_rcp_f64 seed, two Newton steps, quotient, residual, final
// fma: the correctly rounded quotient) minus v_div_scale / v_div_fixup, for operands that need
// neither: d and the quotient well inside the normal range (say 2^-500 < |d|, |n / d| < 2^500), or
// n = 0; NaN operands give NaN.
__device__ inline double div_unscaled(double n, double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = n * y;
  const double r = __builtin_fma(-d, q, n);
  return __builtin_fma(r, y, q);
}
template <bool SPECIAL = true>  // SPECIAL = false: x is known to be a positive normal number
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
  if constexpr (!SPECIAL) return g;
  return (x == 0.0 || x == __builtin_inf()) ? x : g;
}

// The logarithm of det_rnorm, table-driven (no division): u = 2^k m with m in [sqrt(1/2), sqrt(2))
// — det_log's reduction, so that u ~ 1 has k = 0 — the entry of m's subinterval (128 of them, by
// the leading bits of m) gives invc ~ 1 / m and logc = -log(invc) = logc_hi + logc_lo, logc_hi a
// multiple of 2^-40; r = m invc - 1 is one fma (|r| < 2^-8); log u = k ln2 + logc + log1p(r) with
// log1p by its series up to r^7 (remainder < 2^-66). k ln2_hi + logc_hi is exact, the sum with r
// carries its rounding error along, everything small is added last: <= 0.62 ulp against 200-bit
// arithmetic on [2^-64, 1] (fdlibm's kernel above: < 1 ulp). oracle_math.c's orc_log_unit is the
// same arithmetic with the same table (scripts/gen_rnorm_log_table.py writes both).
// generated by scripts/gen_rnorm_log_table.py: max |m invc - 1| = 0.003890 (2^-8.01); columns: invc, logc_hi, logc_lo, pad
__constant__ static const double kRnormLogTab[128][4] __attribute__((aligned(32))) = {
    {0x1.690aa14c2f61dp+0, -0x1.60112dbc1c000p-2, 0x1.e1adca9895a55p-43, 0.0},
    {0x1.67103c7e0340fp+0, -0x1.5a70f9db58000p-2, 0x1.d9cda231d98f3p-42, 0.0},
    {0x1.651b5c793d42dp+0, -0x1.54d8a47c78000p-2, -0x1.8c9a3679e8fe8p-42, 0.0},
    {0x1.632bea459c7d5p+0, -0x1.4f4817ba7c000p-2, 0x1.fb517dddc18c5p-43, 0.0},
    {0x1.6141cf69a8eb0p+0, -0x1.49bf3e0b34000p-2, 0x1.6d53b96f79bd8p-42, 0.0},
    {0x1.5f5cf5e74d59dp+0, -0x1.443e023d68000p-2, 0x1.b987aab3c88e1p-42, 0.0},
    {0x1.5d7d48388d303p+0, -0x1.3ec44f76e4000p-2, 0x1.94f1e46beca0cp-43, 0.0},
    {0x1.5ba2b14c5500dp+0, -0x1.39521132a4000p-2, 0x1.d01fdeeb93230p-44, 0.0},
    {0x1.59cd1c8364ef7p+0, -0x1.33e7333f00000p-2, -0x1.1a4082588b979p-42, 0.0},
    {0x1.57fc75ad53f2dp+0, -0x1.2e83a1bbf4000p-2, -0x1.c7f93512b2855p-48, 0.0},
    {0x1.5630a905ab0cbp+0, -0x1.292749195c000p-2, -0x1.469fc272971d1p-42, 0.0},
    {0x1.5469a3311797cp+0, -0x1.23d216155c000p-2, -0x1.d30e3925fe725p-44, 0.0},
    {0x1.52a7513ab3d5ep+0, -0x1.1e83f5bab0000p-2, -0x1.736311d250bc3p-43, 0.0},
    {0x1.50e9a09164f25p+0, -0x1.193cd55f24000p-2, -0x1.87341a47075b1p-44, 0.0},
    {0x1.4f307f054db28p+0, -0x1.13fca2a204000p-2, 0x1.2c9b51384d711p-42, 0.0},
    {0x1.4d7bdac555190p+0, -0x1.0ec34b6a98000p-2, -0x1.21fddeb17ca9dp-43, 0.0},
    {0x1.4bcba25cc0461p+0, -0x1.0990bde6bc000p-2, -0x1.f6a4fea4fcb20p-43, 0.0},
    {0x1.4a1fc4b0dee7cp+0, -0x1.0464e88964000p-2, -0x1.8625d69a5602cp-42, 0.0},
    {0x1.487830fec992fp+0, -0x1.fe7f741288000p-3, 0x1.e8cb082f38cbep-42, 0.0},
    {0x1.46d4d6d931650p+0, -0x1.f44242bec8000p-3, 0x1.b7ae277ee1b0fp-42, 0.0},
    {0x1.4535a62640555p+0, -0x1.ea121b8bc8000p-3, 0x1.6930de9f258b9p-43, 0.0},
    {0x1.439a8f1d89a16p+0, -0x1.dfeedd6d50000p-3, 0x1.d60dc860ff6a0p-42, 0.0},
    {0x1.4203824609c7ap+0, -0x1.d5d867d420000p-3, 0x1.d97ba092ce9a1p-42, 0.0},
    {0x1.407070743586ep+0, -0x1.cbce9aab90000p-3, 0x1.0260d70b58f4dp-42, 0.0},
    {0x1.3ee14ac81760ap+0, -0x1.c1d1565728000p-3, 0x1.31582395bc837p-42, 0.0},
    {0x1.3d5602ab7b200p+0, -0x1.b7e07bb040000p-3, 0x1.1a4a327455d50p-43, 0.0},
    {0x1.3bce89d026ebfp+0, -0x1.adfbec03c8000p-3, 0x1.ebdedeae71acap-43, 0.0},
    {0x1.3a4ad22e2170ap+0, -0x1.a423891000000p-3, 0x1.da98ef72a934ep-44, 0.0},
    {0x1.38cace0204b00p+0, -0x1.9a57350258000p-3, -0x1.173f0be63b010p-42, 0.0},
    {0x1.374e6fcb5d0dep+0, -0x1.9096d27558000p-3, 0x1.f685ce310f535p-43, 0.0},
    {0x1.35d5aa4b142f9p+0, -0x1.86e2446e70000p-3, 0x1.9d754b6a2d821p-43, 0.0},
    {0x1.34607081e74c0p+0, -0x1.7d396e5c18000p-3, 0x1.497df16dbf740p-44, 0.0},
    {0x1.32eeb5aee88b9p+0, -0x1.739c3413c8000p-3, 0x1.91fb71ea30e5bp-42, 0.0},
    {0x1.31806d4e0b1bap+0, -0x1.6a0a79d000000p-3, 0x1.1e9a5b2449f03p-42, 0.0},
    {0x1.30158b16b99d3p+0, -0x1.6084242e78000p-3, -0x1.44e7ed3d44830p-42, 0.0},
    {0x1.2eae02fa7697cp+0, -0x1.5709182e50000p-3, 0x1.e422c49bb9be5p-44, 0.0},
    {0x1.2d49c923869f9p+0, -0x1.4d993b2e20000p-3, 0x1.9f446785847f3p-44, 0.0},
    {0x1.2be8d1f3a3de1p+0, -0x1.443472ea60000p-3, 0x1.01af444dba68bp-42, 0.0},
    {0x1.2a8b1202bab0cp+0, -0x1.3adaa57b98000p-3, 0x1.e2e362376f9e3p-44, 0.0},
    {0x1.29307e1daf14dp+0, -0x1.318bb954c0000p-3, -0x1.f1f2e91d1b6e0p-43, 0.0},
    {0x1.27d90b452a980p+0, -0x1.28479541a0000p-3, 0x1.97115b6b671a1p-44, 0.0},
    {0x1.2684aeac72899p+0, -0x1.1f0e206520000p-3, 0x1.1d599fdb9dc6ap-43, 0.0},
    {0x1.25335db8462a9p+0, -0x1.15df4237d0000p-3, -0x1.ca903e9630bb8p-46, 0.0},
    {0x1.23e50dfdc49c4p+0, -0x1.0cbae28658000p-3, -0x1.20fe94f6a7df8p-42, 0.0},
    {0x1.2299b5415a4fdp+0, -0x1.03a0e97000000p-3, 0x1.6e6896c1a8867p-42, 0.0},
    {0x1.21514975b5bbfp+0, -0x1.f5227eca30000p-4, -0x1.f420961574c0bp-42, 0.0},
    {0x1.200bc0bac31edp+0, -0x1.e3179a4ba0000p-4, 0x1.7945798473cbdp-43, 0.0},
    {0x1.1ec9115caf152p+0, -0x1.d120f780f0000p-4, -0x1.f43f83a3dc2aep-42, 0.0},
    {0x1.1d8931d2efd1bp+0, -0x1.bf3e6920f0000p-4, -0x1.e5fba56d03e27p-42, 0.0},
    {0x1.1c4c18bf54c08p+0, -0x1.ad6fc27980000p-4, -0x1.cfc95f2d7211ap-46, 0.0},
    {0x1.1b11bced1c64fp+0, -0x1.9bb4d76d10000p-4, 0x1.97320261e8e87p-43, 0.0},
    {0x1.19da15501042dp+0, -0x1.8a0d7c7020000p-4, 0x1.266760237300dp-43, 0.0},
    {0x1.18a51903a6a35p+0, -0x1.78798686c0000p-4, 0x1.c20cc4b9fe770p-42, 0.0},
    {0x1.1772bf4a2a09ap+0, -0x1.66f8cb41f0000p-4, -0x1.56bfc2b6e7095p-42, 0.0},
    {0x1.1642ff8be62bcp+0, -0x1.558b20bd90000p-4, -0x1.e7ed6c8bd7b14p-43, 0.0},
    {0x1.1515d1565a45fp+0, -0x1.44305d9da0000p-4, -0x1.78fb9f6429825p-42, 0.0},
    {0x1.13eb2c5b70a01p+0, -0x1.32e8590c40000p-4, -0x1.a2ce0e7a64813p-45, 0.0},
    {0x1.12c30870bb1dfp+0, -0x1.21b2eab740000p-4, 0x1.88889f74f5d61p-43, 0.0},
    {0x1.119d5d8eb4b51p+0, -0x1.108feace00000p-4, -0x1.312e516123776p-44, 0.0},
    {0x1.107a23d007a34p+0, -0x1.fefe63fec0000p-5, 0x1.641f5f471c820p-42, 0.0},
    {0x1.0f595370d842ap+0, -0x1.dd0132eec0000p-5, 0x1.e287b464f23d8p-44, 0.0},
    {0x1.0e3ae4ce14593p+0, -0x1.bb27f5bac0000p-5, 0x1.f2d81b09c0061p-42, 0.0},
    {0x1.0d1ed064c6c2fp+0, -0x1.997260a380000p-5, -0x1.01f4e98dd082cp-42, 0.0},
    {0x1.0c050ed16f565p+0, -0x1.77e028d8a0000p-5, 0x1.27aa19dc41079p-46, 0.0},
    {0x1.0aed98cf5ee48p+0, -0x1.56710473e0000p-5, 0x1.7fd1740384be4p-42, 0.0},
    {0x1.09d867381737ap+0, -0x1.3524aa75c0000p-5, 0x1.6f791c08279fep-42, 0.0},
    {0x1.08c57302aef1cp+0, -0x1.13fad2c1c0000p-5, -0x1.21a7888ddf8acp-43, 0.0},
    {0x1.07b4b54339310p+0, -0x1.e5e66c35c0000p-6, 0x1.91ff66479b555p-42, 0.0},
    {0x1.06a6272a30dd5p+0, -0x1.a41b1c3ec0000p-6, -0x1.8f34a1c4c9a44p-43, 0.0},
    {0x1.0599c203e7862p+0, -0x1.62932a8c80000p-6, 0x1.8ba32bfa74f27p-42, 0.0},
    {0x1.048f7f37f7b66p+0, -0x1.214e0db580000p-6, 0x1.bbafa3f817b27p-42, 0.0},
    {0x1.03875848baa63p+0, -0x1.c0967be700000p-7, 0x1.0d69b180c41d5p-42, 0.0},
    {0x1.028146d2c1326p+0, -0x1.3f146a3880000p-7, -0x1.394aa93ad43ecp-42, 0.0},
    {0x1.017d448c50034p+0, -0x1.7c29ba6e00000p-8, 0x1.a3be9c9331f41p-49, 0.0},
    {0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0, 0.0},
    {0x1.fdee6607c8aa7p-1, 0x1.09564e8c00000p-8, -0x1.e1337a8cbaaaap-44, 0.0},
    {0x1.f9fe7fcf63b4fp-1, 0x1.82a5ba1380000p-7, 0x1.2693a36fddc00p-42, 0.0},
    {0x1.f61e0b5e77662p-1, 0x1.3f561d0400000p-6, -0x1.d004aa20ab58fp-43, 0.0},
    {0x1.f24cae8520b85p-1, 0x1.bc6324ae80000p-6, -0x1.4e0f5966647dep-42, 0.0},
    {0x1.ee8a11cc60d64p-1, 0x1.1c3ed77900000p-5, 0x1.b5f38904618e2p-44, 0.0},
    {0x1.ead5e05c04446p-1, 0x1.59d4b09720000p-5, -0x1.208fef7c3fb80p-42, 0.0},
    {0x1.e72fc7e1b406dp-1, 0x1.96f4e5eec0000p-5, -0x1.64768bdca0c53p-44, 0.0},
    {0x1.e3977879215f4p-1, 0x1.d3a1359a20000p-5, -0x1.24640210f9e49p-42, 0.0},
    {0x1.e00ca4953da63p-1, 0x1.07eda9ee30000p-4, 0x1.477a31e232219p-42, 0.0},
    {0x1.dc8f00ea70998p-1, 0x1.25d275b5d0000p-4, 0x1.80841491ddc40p-42, 0.0},
    {0x1.d91e4459c0442p-1, 0x1.437fcedbb0000p-4, -0x1.de598a24d2f96p-45, 0.0},
    {0x1.d5ba27dcde604p-1, 0x1.60f6819670000p-4, 0x1.035944afe800ap-44, 0.0},
    {0x1.d26266730fc58p-1, 0x1.7e3755bcb0000p-4, -0x1.685eac6019853p-43, 0.0},
    {0x1.cf16bd0ee3195p-1, 0x1.9b430ee4a0000p-4, -0x1.26f330be4a4e3p-42, 0.0},
    {0x1.cbd6ea84ac94fp-1, 0x1.b81a6c82c0000p-4, 0x1.62b14df93ffd9p-44, 0.0},
    {0x1.c8a2af79bd42cp-1, 0x1.d4be2a0780000p-4, 0x1.ff586ed6fb3aap-42, 0.0},
    {0x1.c579ce544c9f1p-1, 0x1.f12efefbd0000p-4, -0x1.aceaa0a7e3c95p-42, 0.0},
    {0x1.c25c0b2c0c07fp-1, 0x1.06b6cf8e30000p-3, 0x1.686ee8ebd35e6p-43, 0.0},
    {0x1.bf492bbb5bdeap-1, 0x1.14bd5d3a68000p-3, 0x1.7a8e06d80c3cfp-42, 0.0},
    {0x1.bc40f7511aae8p-1, 0x1.22ab7ebc80000p-3, 0x1.deab0b09487f8p-46, 0.0},
    {0x1.b94336c307176p-1, 0x1.3081888ef8000p-3, 0x1.c2d7f181a7df2p-42, 0.0},
    {0x1.b64fb460ad9c1p-1, 0x1.3e3fcd7908000p-3, -0x1.96f3e4762715ep-42, 0.0},
    {0x1.b3663be6dbd40p-1, 0x1.4be69e9a00000p-3, -0x1.22a32aff75760p-42, 0.0},
    {0x1.b0869a7392d58p-1, 0x1.59764b74b8000p-3, 0x1.7a680925a6edep-42, 0.0},
    {0x1.adb09e7a73033p-1, 0x1.66ef21fa60000p-3, -0x1.6869a0209182cp-44, 0.0},
    {0x1.aae417b99bb29p-1, 0x1.74516e94d8000p-3, 0x1.e7b7306d48b8ep-42, 0.0},
    {0x1.a820d72ef96cap-1, 0x1.819d7c3118000p-3, 0x1.79aa33dc7bfc5p-44, 0.0},
    {0x1.a566af0dfdce8p-1, 0x1.8ed39448c8000p-3, 0x1.40a6654f62733p-42, 0.0},
    {0x1.a2b572b5bc4fap-1, 0x1.9bf3feebf8000p-3, -0x1.e987cfbfc3311p-43, 0.0},
    {0x1.a00cf6a767735p-1, 0x1.a8ff02ca28000p-3, -0x1.da58632273616p-46, 0.0},
    {0x1.9d6d107d2a21fp-1, 0x1.b5f4e53b60000p-3, -0x1.72af7ecec8449p-44, 0.0},
    {0x1.9ad596e1591fep-1, 0x1.c2d5ea48b8000p-3, -0x1.f3f698b236b80p-42, 0.0},
    {0x1.98466185f8c9dp-1, 0x1.cfa254b4b8000p-3, -0x1.ada95bb615eb1p-42, 0.0},
    {0x1.95bf491c936fap-1, 0x1.dc5a660380000p-3, 0x1.74e1701e519a3p-42, 0.0},
    {0x1.9340274e5cd4dp-1, 0x1.e8fe5e82b0000p-3, 0x1.01caa6a905219p-43, 0.0},
    {0x1.90c8d6b49f894p-1, 0x1.f58e7d50e0000p-3, -0x1.64ed15e0d206fp-48, 0.0},
    {0x1.8e5932d170f5bp-1, 0x1.0105803290000p-2, 0x1.888d4ed9ea13ep-42, 0.0},
    {0x1.8bf11808a91e9p-1, 0x1.073a124b14000p-2, 0x1.f4d49161f34afp-43, 0.0},
    {0x1.899063991b448p-1, 0x1.0d6512d098000p-2, 0x1.ade17bb7326abp-42, 0.0},
    {0x1.8736f3960cacep-1, 0x1.13869f1864000p-2, 0x1.5540f4efcbefdp-42, 0.0},
    {0x1.84e4a6e0e6fd0p-1, 0x1.199ed3f1a8000p-2, 0x1.10a8d4286f800p-42, 0.0},
    {0x1.82995d2323b23p-1, 0x1.1fadcda8ac000p-2, 0x1.c473c04940d46p-42, 0.0},
    {0x1.8054f6c86e5f2p-1, 0x1.25b3a809e8000p-2, 0x1.155686bb3f876p-43, 0.0},
    {0x1.7e1754f8fb71bp-1, 0x1.2bb07e64f8000p-2, 0x1.7d1c0474c19aap-46, 0.0},
    {0x1.7be05994115fap-1, 0x1.31a46b8f8c000p-2, -0x1.f735a4de38e2bp-46, 0.0},
    {0x1.79afe72ac2320p-1, 0x1.378f89e834000p-2, 0x1.c49ff45f79a99p-42, 0.0},
    {0x1.7785e0fad37e4p-1, 0x1.3d71f35928000p-2, -0x1.01f98b256c248p-42, 0.0},
    {0x1.75622ae9d2f2ep-1, 0x1.434bc15ad8000p-2, 0x1.0a15752eadb86p-42, 0.0},
    {0x1.7344a98055b3ap-1, 0x1.491d0cf6a4000p-2, -0x1.8b6e970f3f59bp-43, 0.0},
    {0x1.712d41e560d4ap-1, 0x1.4ee5eec93c000p-2, 0x1.95ab38de39ae9p-42, 0.0},
    {0x1.6f1bd9d9f957ep-1, 0x1.54a67f0530000p-2, 0x1.ab80ba9640c2ap-42, 0.0},
    {0x1.6d1057b4da225p-1, 0x1.5a5ed57538000p-2, 0x1.f34d69fc89ce4p-42, 0.0},
    {0x1.6b0aa25e4e709p-1, 0x1.600f097e90000p-2, 0x1.310cd48817e0cp-44, 0.0},
};
// The kernels read the table from LDS (kRnormTabDoubles doubles, filled by rnorm_table_to_lds at
// their start): a table lookup then waits on the LDS counter only. Gathered from global memory it
// would sit behind the kernel's row loads in the in-order vector-memory counter and stall the
// first variate until the rows have arrived from HBM (measured: the PSO move lost 8 %).
constexpr int kRnormTabDoubles = 128 * 4;
__device__ inline void rnorm_table_to_lds(double *lds) {  // every thread of the workgroup
  const double *src = &kRnormLogTab[0][0];
  for (int e = threadIdx.x; e < kRnormTabDoubles; e += blockDim.x) lds[e] = src[e];
  __syncthreads();
}
__device__ inline double det_log_unit(double x, const double *tab) {  // x in [2^-64, 1]
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                   A0 = 0x1.5555555555555p-2, A1 = -0x1p-2, A2 = 0x1.999999999999ap-3,
                   A3 = -0x1.5555555555555p-3, A4 = 0x1.2492492492492p-3;
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(x));
  uint32_t hx = static_cast<uint32_t>(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  const int k = static_cast<int>(hx >> 20) - 0x3ff;
  const uint32_t t20 = hx & 0x000fffffu;
  const double m = __longlong_as_double(static_cast<long long>(
      (static_cast<uint64_t>(t20 + 0x3fe6a09eu) << 32) | (u & 0xffffffffull)));
  const double *te = tab + 4 * (t20 >> 13);
  const double invc = te[0], logc_hi = te[1], logc_lo = te[2];
  const double r = __builtin_fma(m, invc, -1.0);
  const double r2 = r * r;
  const double p = fma_k(r, fma_k(r, fma_k(r, __builtin_fma(r, A4, A3), A2), A1), A0);
  const double lo = __builtin_fma(r2, -0.5, (r2 * r) * p);
  const double dk = static_cast<double>(k);
  const double t1 = __builtin_fma(dk, ln2_hi, logc_hi);  // exact
  const double hi = t1 + r;
  const double err = (t1 - hi) + r;
  return hi + (err + __builtin_fma(dk, ln2_lo, lo + logc_lo));
}

// One normal variate from one 64-bit draw z (rnorm, nlsolver.h:2479-2485):
// sqrt(-2 log u1) * cos(2 * 3.141593 * u2) with u1 = z 2^-64 and u2 = (z mod 2^32) 2^-32, written
// for the arguments that occur here: u1 is 0 or a normal number in [2^-64, 1] — the table-driven
// logarithm above, u1 = 0 -> -inf as a final select; the cosine's argument lies in [0, 6.3] —
// one sine polynomial on [-pi/2, pi/2] (below).
// oracle_math.c's orc_rnorm is the CPU mirror. The kernels that draw normal variates are bound
// by the vector unit's instruction count: 265 -> ~120 vector instructions per variate since
// round 1.
// SPECIAL = false leaves out the two selects that only the draws z = 0 (u1 = 0: the logarithm is
// -inf) and z >= 2^64 - 2^10 (u1 rounds to 1: the square root of -0) need — det_rnorm below takes
// that path when no lane of the wave holds such a draw (all but 2^-53 of the time), six vector
// instructions per variate fewer; same bits either way.
template <bool SPECIAL>
__device__ inline double det_rnorm_impl(uint64_t zbits, const double *tab) {  // tab: the LDS table
  constexpr double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  // u1 = (double)z 2^-64 rounded once, as the conversion rounds: hi 2^-32 + lo 2^-64 in one fma
  // (both terms exact); the angle y = 2 pi_ u2 = lo (2 pi_ 2^-32) — scaling by 2^-32 is exact, so
  // the one rounding is that of 2 pi_ u2
  const double hi_d = static_cast<double>(static_cast<uint32_t>(zbits >> 32));
  const double lo_d = static_cast<double>(static_cast<uint32_t>(zbits));
  const double u1 = __builtin_fma(hi_d, 0x1p-32, lo_d * 0x1p-64);
  double lg = det_log_unit(u1, tab);  // (u1 = 0: garbage, replaced below)
  if constexpr (SPECIAL) lg = u1 == 0.0 ? -__builtin_inf() : lg;
  // cos(2 pi_ u2), y in [0, 6.3]: cos y = -(-1)^g sin(r) with g = rint(y / pi - 1/2) in {0, 1, 2}
  // and r = y - (g + 1/2) pi in [-pi/2, pi/2] (two-term Cody-Waite, (2g + 1) pio2_1 exact), and
  // ONE odd polynomial for the sine there, sin r = r + r^3 Q(r^2) (degree 8: Chebyshev fit of
  // (sin x - x) / x^3, truncation 5e-22) — fdlibm's pair of kernels on [-pi/4, pi/4] costs a
  // lane both although it needs one. Within 2.3e-16 (absolute) of libm's cos; oracle_math.c's
  // orc_cos_unit is the same arithmetic.
  constexpr double invpi = 3.18309886183790671538e-01,
                   Q8 = -0x1.275ecac266a3ap-57, Q7 = 0x1.9507fb692a94ap-49,
                   Q6 = -0x1.ae7ee39a09ceap-41, Q5 = 0x1.612460b690375p-33,
                   Q4 = -0x1.ae64567e73021p-26, Q3 = 0x1.71de3a556b9b2p-19,
                   Q2 = -0x1.a01a01a01a00dp-13, Q1 = 0x1.1111111111111p-7,
                   Q0 = -0x1.5555555555555p-3;
  const double y = lo_d * (2 * 3.141593 * 0x1p-32);
  const double g = rint(__builtin_fma(y, invpi, -0.5));
  const double h = 2.0 * g + 1.0;
  // (the polynomial is evaluated at -r: every operation below is odd in r, so the result is -sin r
  // bit for bit and the sign handling is one shift, one mask and one exclusive-or)
  const double nr = __builtin_fma(h, pio2_1t, __builtin_fma(h, pio2_1, -y));
  const double z = nr * nr;
  const double q = fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, __builtin_fma(z, Q8, Q7), Q6), Q5), Q4), Q3), Q2), Q1), Q0);
  const double nsn = __builtin_fma(z * nr, q, nr);
  // cos y = sin r for g odd, -sin r for g even: h = 2g + 1 is 1.0, 3.0 or 5.0 and only 3.0 has bit 19
  // of its high word set, so that bit moved to the sign position flips -sin r back for odd g
  const uint64_t hbits = static_cast<uint64_t>(__double_as_longlong(h));
  const uint64_t flip = (hbits << 12) & 0x8000000000000000ull;
  const double cs = __longlong_as_double(static_cast<long long>(
      static_cast<uint64_t>(__double_as_longlong(nsn)) ^ flip));
  return sqrt_unscaled<SPECIAL>(-2 * lg) * cs;  // -2 lg is 0, +inf or at least 2^-53
}
__device__ inline double det_rnorm(uint64_t zbits, const double *tab) {
  const bool special = zbits - 1 >= 0xFFFFFFFFFFFFFBFFull;  // z = 0 or z >= 2^64 - 2^10
  if (__builtin_expect(__ballot(special) != 0, 0)) return det_rnorm_impl<true>(zbits, tab);  // wave-uniform
  return det_rnorm_impl<false>(zbits, tab);
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
// tanh from the rational form of the exponential: with 2|x| = k ln2 + r and fdlibm's
// c = r - r^2 P(r^2), exp(r) = (2 + 2r - c) / (2 - c), so with s = 2^k and B = 2 - c
//   tanh|x| = (e - 1) / (e + 1) = ((s - 1) B + 2 s r) / ((s + 1) B + 2 s r)
// — ONE division (1 - 2 / (exp(2|x|) + 1) takes two: the NLLS evaluation pays for every fp64
// vector instruction of this chain), no cancellation for small |x| (k = 0:
// r / (B + r)), 1 exactly from |x| = 22 on; within 2.3e-16 (absolute) of libm's tanh.
// oracle_lm.c's orc_tanh is the same arithmetic.
__device__ inline double det_tanh(double x) {
  constexpr double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                   invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                   P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                   P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double ax = fabs(x);
  ax = ax < 22.0 ? ax : 22.0;  // a NaN takes the cap too: the integer conversion below stays defined
  const double X = 2.0 * ax;
  const int k = static_cast<int>(__builtin_fma(invln2, X, 0.5));
  const double dk = static_cast<double>(k);
  const double hi = __builtin_fma(-dk, ln2HI, X), lo = dk * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = __builtin_fma(-t, fma_k(t, fma_k(t, fma_k(t, __builtin_fma(t, P5, P4), P3), P2), P1), r);
  const double B = 2.0 - c;
  const double s = __builtin_ldexp(1.0, k), sr2 = __builtin_ldexp(r, k + 1);  // 2^k, 2 s r: exact
  const double num = __builtin_fma(s - 1.0, B, sr2), den = __builtin_fma(s + 1.0, B, sr2);
  const double tt = num / den;
  return x != x ? x : __builtin_copysign(tt, x);  // tanh(-0) = -0
}

}  // namespace nlsg
