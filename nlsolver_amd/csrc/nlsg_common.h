// nlsolver_amd/csrc/nlsg_common.h — device-side building blocks shared by the
// gfx950 solver kernels: counter RNG, wave64 reductions, built-in objectives,
// the fixed block-tree reduction, host-side error plumbing.
//
// Arithmetic contract: every floating-point expression here is written in the
// order the CPU restatement (oracle/*.c) uses, and the translation unit is
// compiled with -ffp-contract=off, so device results are bit-identical to it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/nlsg_c_api.h"

namespace nlsg {

// ---------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------
inline char *err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
#define NLSG_HIP(call)                                                          \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess)                                                       \
      return ::nlsg::fail(e_ == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP, \
                          "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                          __FILE__, __LINE__);                                  \
  } while (0)

// ---------------------------------------------------------------------------
// counter-based RNG: random access into splitmix64 streams
// (rng::splitmix::yield_init, nlsolver.h:1273-1278; constants :1274-1276)
// ---------------------------------------------------------------------------
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// child key / draw `index` under `parent`
__host__ __device__ inline uint64_t ctr_key(uint64_t parent, uint64_t index) {
  return mix64(parent + kGolden * (index + 1));
}
// U[0,1] inclusive, as rng::xorshift::yield scales its output (nlsolver.h:1358):
// (double)bits / 2^64.
__host__ __device__ inline double u01(uint64_t bits) {
  return static_cast<double>(bits) * 0x1p-64;
}
// generate_index (nlsolver.h:2325-2329) with the u == 1.0 corner clamped (B10).
__host__ __device__ inline uint64_t clamp_index(double u, uint64_t n) {
  uint64_t p = static_cast<uint64_t>(u * static_cast<double>(n));
  return p >= n ? n - 1 : p;
}

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
__device__ inline int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

// xor butterfly (32,16,8,4,2,1): every lane ends with the same bit pattern.
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// The fixed 256-thread block tree of DESIGN.md §Reductions: caller passes the
// per-thread sequential partial; returns the block total in every thread.
// `red` is a 4-double LDS scratch.
__device__ inline double block_tree_256(double partial, double *red) {
  const double w = wave_sum(partial);
  const int wid = static_cast<int>(threadIdx.x) >> 6;
  __syncthreads();  // protect `red` from a previous use
  if (lane_id() == 0) red[wid] = w;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

// ---------------------------------------------------------------------------
// built-in objectives. A wave owns one point x[0..D): element e = c*128 + 2*l
// and e+1 live in lane l as xv[c][0], xv[c][1]. Term i may read x[i], x[i+1].
// ---------------------------------------------------------------------------
template <int OBJ>
struct Objective;

template <>
struct Objective<NLSG_OBJ_ROSENBROCK> {
  static constexpr bool kChain = true;
  __device__ static inline double term(double xi, double xn) {
    const double t1 = 1 - xi;
    const double t2 = (xn - xi * xi);
    return t1 * t1 + 100 * t2 * t2;  // example.cpp:43-47
  }
  __device__ static inline uint64_t n_terms(uint64_t D) { return D ? D - 1 : 0; }
  __device__ static inline double finish(double s, uint64_t) { return s; }
};
template <>
struct Objective<NLSG_OBJ_SPHERE> {
  static constexpr bool kChain = false;
  __device__ static inline double term(double xi, double) { return xi * xi; }
  __device__ static inline uint64_t n_terms(uint64_t D) { return D; }
  __device__ static inline double finish(double s, uint64_t) { return s; }
};
template <>
struct Objective<NLSG_OBJ_STYBLINSKI_TANG> {
  static constexpr bool kChain = false;
  __device__ static inline double term(double xi, double) {
    const double x2 = xi * xi;
    return x2 * x2 - 16 * x2 + 5 * xi;  // test_functions.h:255-257
  }
  __device__ static inline uint64_t n_terms(uint64_t D) { return D; }
  __device__ static inline double finish(double s, uint64_t) { return s / 2.0; }
};
template <>
struct Objective<NLSG_OBJ_RASTRIGIN> {
  static constexpr bool kChain = false;
  __device__ static inline double term(double xi, double) {
    return xi * xi - 10 * cos(2 * 3.14159265358979323846 * xi);  // test_functions.h:74-76
  }
  __device__ static inline uint64_t n_terms(uint64_t D) { return D; }
  __device__ static inline double finish(double s, uint64_t D) {
    return 10.0 * static_cast<double>(D) + s;
  }
};

// f(x) for the point held by the wave; all lanes return the same bits.
template <int OBJ, int CHUNKS>
__device__ inline double wave_objective(const double (&xv)[CHUNKS][2], uint64_t D) {
  using O = Objective<OBJ>;
  const int lane = lane_id();
  const uint64_t nt = O::n_terms(D);
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    double xn = 0.0;
    if (O::kChain) {
      // x[e0+2]: lane+1's first element, or lane 0 of the next chunk for lane 63
      const double same = __shfl_down(xv[c][0], 1, 64);
      double next = 0.0;
      if (c + 1 < CHUNKS) next = __shfl(xv[c + 1][0], 0, 64);
      xn = (lane == 63) ? next : same;
    }
    if (e0 < nt) acc = acc + O::term(xv[c][0], xv[c][1]);
    if (e0 + 1 < nt) acc = acc + O::term(xv[c][1], xn);
  }
  return O::finish(wave_sum(acc), D);
}

}  // namespace nlsg
