// nlsolver_amd/csrc/nlsg_common.h — device-side building blocks shared by the
// gfx950 solver kernels: counter RNG, wave64 reductions, built-in objectives,
// the fixed block-tree reduction, host-side error plumbing.
//
// Arithmetic contract: every floating-point expression here is written in the
// order the CPU restatement (oracle/*.c) uses, and the translation unit is
// compiled with -ffp-contract=off, so device results are bit-identical to it.
#pragma once

// This header (with nlsg_de_kernels.h) is also compiled at run time by hiprtc for user-supplied
// objectives (nlsg_rtc.hip): everything host-only sits behind !__HIPCC_RTC__, and nothing
// from the standard library is needed on the device side.
#ifdef __HIPCC_RTC__
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned int uint32_t;
typedef int int32_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#endif

#include "../../include/nlsg_c_api.h"
#include "nlsg_math.h"
#ifndef __HIPCC_RTC__
#include "nlsg_pool.h"
#endif

namespace nlsg {

#ifndef __HIPCC_RTC__
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
// Host wall-clock of the calling thread's last create / minimize / destroy, read back through
// nlsg_call_timing (include/nlsg_c_api.h): what a call through the drop-in header costs beyond its
// kernels. Phases are host-side laps, without extra synchronisation: device work queued by `init`
// that is still running when the lap is taken is counted under `iterate`; `upload` is
// nlsg_lm_set_data (the model's design matrices crossing PCIe).
struct CallTiming {
  double create_ms, upload_ms, init_ms, iterate_ms, readback_ms, destroy_ms;
};
inline CallTiming &call_timing() {
  static thread_local CallTiming t = {0, 0, 0, 0, 0, 0};
  return t;
}
struct PhaseClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double lap() {  // milliseconds since construction or the previous lap
    const auto t1 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    t0 = t1;
    return ms;
  }
};
#define NLSG_HIP(call)                                                          \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess)                                                       \
      return ::nlsg::fail(e_ == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP, \
                          "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                          __FILE__, __LINE__);                                  \
  } while (0)
// every engine's first question: is `device` a gfx950 this process can see?
inline int check_device(int device) {
  // the verdict on a device does not change while the process lives: asked once (the property
  // query costs about a millisecond, every minimize() through the header creates an engine)
  static std::atomic<unsigned long long> ok_mask{0};
  if (device >= 0 && device < 64 && ((ok_mask.load(std::memory_order_relaxed) >> device) & 1ull)) return NLSG_OK;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(NLSG_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n)
    return fail(NLSG_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
  hipDeviceProp_t prop;
  NLSG_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NLSG_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                device, prop.gcnArchName);
  if (device < 64) ok_mask.fetch_or(1ull << device, std::memory_order_relaxed);
  return NLSG_OK;
}

// A caller's stream handle as the engines use it. The C-ABI spells "the null stream" as
// hipStreamLegacy ((void *)1, because NULL means "create a private stream"); inside the library
// that handle is replaced by the null handle itself — the same stream — once, at engine creation:
// with HIP 7.0 an event recorded on the hipStreamLegacy handle crashes the next
// hipStreamWaitEvent on it, and RCCL records and waits on the stream it is given.
inline hipStream_t borrowed_stream(void *handle) {
  hipStream_t s = static_cast<hipStream_t>(handle);
  return s == hipStreamLegacy ? nullptr : s;
}


// Launches of run-time compiled kernels (module API). The launch helpers of the engines return
// nothing — an entry point enqueues many launches and asks once at its end — so a failed module
// launch is remembered here and reported by launches_status() together with hipGetLastError().
inline hipError_t &module_launch_error() {
  static thread_local hipError_t err = hipSuccess;
  return err;
}
inline void launch_module_kernel(hipFunction_t fn, unsigned grid, unsigned block, unsigned lds_bytes,
                                 hipStream_t stream, void **args, unsigned grid_y = 1) {
  const hipError_t r = hipModuleLaunchKernel(fn, grid, grid_y, 1, block, 1, 1, lds_bytes, stream, args, nullptr);
  if (r != hipSuccess && module_launch_error() == hipSuccess) module_launch_error() = r;
}
inline hipError_t launches_status() {
  const hipError_t m = module_launch_error();
  module_launch_error() = hipSuccess;
  const hipError_t r = hipGetLastError();
  return m != hipSuccess ? m : r;
}

#endif  // !__HIPCC_RTC__


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
// The two uniforms of ONE normal variate (rnorm, nlsolver.h:2479-2485: sqrt(-2 ln u1) cos(2 pi u2))
// from one 64-bit draw: the radius' uniform is the draw as every other uniform takes it (all 64
// bits into the fp64), the angle's uniform its low 32 bits (an angular resolution of 1.5e-9 rad; the
// bits it shares with u1 sit below u1's 32 leading bits). Half the mixing work of two draws — the
// kernels that draw normal variates are bound by exactly this arithmetic.
__host__ __device__ inline double u01_low32(uint64_t bits) {
  return static_cast<double>(static_cast<uint32_t>(bits)) * 0x1p-32;
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

// lane_xor<OFF>(v): the value lane (l ^ OFF) holds, without the LDS crossbar (`__shfl_xor`
// compiles to ds_bpermute, ~100 cycles of latency per level of a butterfly; these are VALU
// moves): OFF 1, 2 quad permutes, 4 two bank-masked row shifts, 8 a row rotate (DPP),
// 16 / 32 the gfx950 row / half swaps. Checked lane by lane against l ^ OFF on the device.
template <int CTRL, int BANK>
__device__ inline uint32_t dpp_mov32(uint32_t old, uint32_t src) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(old), static_cast<int>(src),
                                                           CTRL, 0xF, BANK, false));
}
template <int OFF>
__device__ inline uint32_t lane_xor32(uint32_t v) {
  static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "");
  if constexpr (OFF == 1) return dpp_mov32<0xB1, 0xF>(v, v);        // quad_perm [1,0,3,2]
  if constexpr (OFF == 2) return dpp_mov32<0x4E, 0xF>(v, v);        // quad_perm [2,3,0,1]
  if constexpr (OFF == 4)                                            // row_shl:4 -> banks 0,2
    return dpp_mov32<0x114, 0xA>(dpp_mov32<0x104, 0x5>(v, v), v);    // row_shr:4 -> banks 1,3
  if constexpr (OFF == 8) return dpp_mov32<0x128, 0xF>(v, v);        // row_ror:8
  if constexpr (OFF == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((threadIdx.x >> 4) & 1) ? r[0] : r[1];
  }
  if constexpr (OFF == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return ((threadIdx.x >> 5) & 1) ? r[0] : r[1];
  }
  return v;
}
template <int OFF>
__device__ inline uint64_t lane_xor(uint64_t v) {
  const uint32_t lo = lane_xor32<OFF>(static_cast<uint32_t>(v));
  const uint32_t hi = lane_xor32<OFF>(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
template <int OFF>
__device__ inline double lane_xor(double v) {
  return __longlong_as_double(static_cast<long long>(
      lane_xor<OFF>(static_cast<uint64_t>(__double_as_longlong(v)))));
}
// lane l+1's value (lane 63 keeps its own): DPP wave_shl:1
__device__ inline double lane_down1(double v) {
  const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t lo = dpp_mov32<0x130, 0xF>(static_cast<uint32_t>(b), static_cast<uint32_t>(b));
  const uint32_t hi = dpp_mov32<0x130, 0xF>(static_cast<uint32_t>(b >> 32), static_cast<uint32_t>(b >> 32));
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
// lane l-1's value (lane 0 keeps its own): DPP wave_shr:1
__device__ inline double lane_up1(double v) {
  const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t lo = dpp_mov32<0x138, 0xF>(static_cast<uint32_t>(b), static_cast<uint32_t>(b));
  const uint32_t hi = dpp_mov32<0x138, 0xF>(static_cast<uint32_t>(b >> 32), static_cast<uint32_t>(b >> 32));
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
// lane 0's value through scalar registers
__device__ inline double lane_first(double v) {
  const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<int>(b & 0xffffffffu));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32));
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
// ---- scalar-issue relief. A CU has ONE scalar unit for its four SIMDs: per SIMD it issues at
// most one scalar instruction every four cycles, the same rate as the SIMD's vector instructions,
// and wave-uniform 64-bit integer work (the keyed RNG of a whole agent: ~25 scalar instructions
// per mix64) lands there. Kernels that are otherwise memory bound become bound by it (the DE
// generation: 419 scalar vs 238 vector instructions per wave). These helpers move uniform
// integer work to the vector unit: on_valu() hands the compiler the same value in vector
// registers (what follows is computed per lane, redundantly or — better — one draw per lane),
// first64() / readlane64() bring results back to scalar registers.
__device__ inline uint64_t on_valu(uint64_t s) {
  const uint32_t lo = static_cast<uint32_t>(s), hi = static_cast<uint32_t>(s >> 32);
  uint32_t vlo, vhi;
  asm("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(lo));
  asm("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(hi));
  return (static_cast<uint64_t>(vhi) << 32) | vlo;
}
__device__ inline uint64_t first64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v & 0xffffffffu));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ inline uint64_t readlane64(uint64_t v, int src) {  // src wave-uniform
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffffu), src);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), src);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
// value of lane `src` (wave-uniform index) through scalar registers
__device__ inline double lane_broadcast(double v, int src) {
  return __longlong_as_double(static_cast<long long>(
      readlane64(static_cast<uint64_t>(__double_as_longlong(v)), src)));
}
// f(int_c<OFF>) for OFF = FIRST, FIRST/2, ..., 1
template <int N>
struct int_c {
  static constexpr int value = N;
};
template <int FIRST, typename F>
__device__ inline void butterfly_levels(F &&f) {
  f(int_c<FIRST>{});
  if constexpr (FIRST > 1) butterfly_levels<FIRST / 2>(f);
}

// xor butterfly (32,16,8,4,2,1): every lane ends with the same bit pattern.
__device__ inline double wave_sum(double v) {
  butterfly_levels<32>([&](auto off) { v = v + lane_xor<decltype(off)::value>(v); });
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
  static constexpr bool kWhole = false;
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
  static constexpr bool kWhole = false;
  __device__ static inline double term(double xi, double) { return xi * xi; }
  __device__ static inline uint64_t n_terms(uint64_t D) { return D; }
  __device__ static inline double finish(double s, uint64_t) { return s; }
};
template <>
struct Objective<NLSG_OBJ_STYBLINSKI_TANG> {
  static constexpr bool kChain = false;
  static constexpr bool kWhole = false;
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
  static constexpr bool kWhole = false;
  __device__ static inline double term(double xi, double) {
    return xi * xi - 10 * det_cos_2pi(xi);  // test_functions.h:74-76, deterministic cosine
  }
  __device__ static inline uint64_t n_terms(uint64_t D) { return D; }
  __device__ static inline double finish(double s, uint64_t D) {
    return 10.0 * static_cast<double>(D) + s;
  }
};

// What a whole-vector user objective (nlsg_custom_objective.chain == NLSG_CUSTOM_VECTOR) sees of
// the point its wave — or its group of G lanes — holds:
//   x(i)      coordinate i, for an index that is the same in every lane (literals, loop counters)
//   x.size()  D
//   x.sum(g)  sum over all coordinates of g(x_i, i), in the kernels' lane-tree order (the order
//             of the built-in objectives): per lane its coordinates in ascending order, then the
//             xor butterfly; g is any callable double(double, uint64_t)
// Every lane evaluates the body and must return the same value.
template <int CHUNKS>
struct WavePoint {
  const double (&v)[CHUNKS][2];
  uint64_t D;
  __device__ inline uint64_t size() const { return D; }
  __device__ inline double operator()(uint64_t i) const {  // i wave-uniform
    double own = 0.0;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) own = (static_cast<uint64_t>(c) == (i >> 7) && static_cast<uint64_t>(k) == (i & 1)) ? v[c][k] : own;
    return lane_broadcast(own, static_cast<int>((i & 127) >> 1));
  }
  template <typename F>
  __device__ inline double sum(F g) const {
    const uint64_t l2 = 2 * static_cast<uint64_t>(lane_id());
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = static_cast<uint64_t>(c) * 128 + l2 + k;
        if (e < D) acc = acc + g(v[c][k], e);
      }
    return wave_sum(acc);
  }
};
template <int G>
struct GroupPoint {  // a point of at most 2 G coordinates in a group of G lanes
  double x0, x1;
  uint64_t D;
  __device__ inline uint64_t size() const { return D; }
  __device__ inline double operator()(uint64_t i) const {  // i the same in every lane
    const int src = (lane_id() & ~(G - 1)) + static_cast<int>(i >> 1);
    const double own = (i & 1) ? x1 : x0;
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(own));
    const uint32_t lo = static_cast<uint32_t>(__shfl(static_cast<int>(b & 0xffffffffu), src, 64));
    const uint32_t hi = static_cast<uint32_t>(__shfl(static_cast<int>(b >> 32), src, 64));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
  }
  template <typename F>
  __device__ inline double sum(F g) const {
    const uint64_t e0 = 2 * static_cast<uint64_t>(lane_id() & (G - 1));
    double acc = 0.0;
    if (e0 < D) acc = acc + g(x0, e0);
    if (e0 + 1 < D) acc = acc + g(x1, e0 + 1);
    if constexpr (G > 1)
      butterfly_levels<G / 2>([&](auto off) { acc = acc + lane_xor<decltype(off)::value>(acc); });
    return acc;
  }
};

// f(x) for the point held by the wave; all lanes return the same bits.
template <int OBJ, int CHUNKS>
__device__ inline double wave_objective(const double (&xv)[CHUNKS][2], uint64_t D) {
  using O = Objective<OBJ>;
  if constexpr (O::kWhole) return O::whole(WavePoint<CHUNKS>{xv, D}, D);
  const int lane = lane_id();
  const uint64_t nt = O::n_terms(D);
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    double xn = 0.0;
    if (O::kChain) {
      // x[e0+2]: lane+1's first element (DPP wave shift), or lane 0 of the next chunk for lane 63
      const double same = lane_down1(xv[c][0]);
      double next = 0.0;
      if (c + 1 < CHUNKS) next = lane_first(xv[c + 1][0]);
      xn = (lane == 63) ? next : same;
    }
    if (e0 < nt) acc = acc + O::term(xv[c][0], xv[c][1]);
    if (e0 + 1 < nt) acc = acc + O::term(xv[c][1], xn);
  }
  return O::finish(wave_sum(acc), D);
}


// The lane's share of wave_objective (its terms, added in order) before the butterfly, for callers
// that reduce several points at once (wave_sum4); term objectives only.
template <int OBJ>
__device__ inline double wave_objective_partial(double x0, double x1, uint64_t D) {
  using O = Objective<OBJ>;
  const int lane = lane_id();
  const uint64_t nt = O::n_terms(D), e0 = 2 * static_cast<uint64_t>(lane);
  double xn = 0.0;
  if (O::kChain) {
    const double same = lane_down1(x0);
    xn = (lane == 63) ? 0.0 : same;
  }
  double acc = 0.0;
  if (e0 < nt) acc = acc + O::term(x0, x1);
  if (e0 + 1 < nt) acc = acc + O::term(x1, xn);
  return acc;
}
// Four wave_sum butterflies at once. The levels 32 and 16 halve the set of values a lane carries
// (it keeps the values its lane bits select and hands the others to its partner), the levels 8 .. 1
// run on the one that is left: 7 exchanges and additions instead of 24, every addition the same
// own + partner pair as in wave_sum — lanes 16 g .. 16 g + 15 end with the total of value g, the
// bits wave_sum(value g) leaves in every lane.
__device__ inline double wave_sum4(double a, double b, double c, double d) {
  const int lane = lane_id();
  const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0;
  const double y0 = (b5 ? c : a) + lane_xor<32>(b5 ? a : c);
  const double y1 = (b5 ? d : b) + lane_xor<32>(b5 ? b : d);
  double v = (b4 ? y1 : y0) + lane_xor<16>(b4 ? y0 : y1);
  v = v + lane_xor<8>(v);
  v = v + lane_xor<4>(v);
  v = v + lane_xor<2>(v);
  v = v + lane_xor<1>(v);
  return v;
}

// ---- reference-order ("sequential") sums. The reference adds in index order (its objective
// functors, math::dot / norm, the matrix-vector loops: plain left-to-right `acc += ...` loops);
// the kernels' default is the lane tree above, which differs from it in the last bits. Engines
// that differentiate an objective numerically divide differences of such sums by 12 eps or
// 600 eps^2, so the last bit of a sum is worth 1e-8 .. 1e-6 of the result after a few iterations.
// Their reference-order mode (NLSG_BFGS_REFERENCE_ORDER, NLSG_LM_REFERENCE_ORDER) takes every sum
// in the reference's order instead — the terms are still computed side by side, one per lane
// slot, only their addition is serial — and reproduces the reference's own runs bit for bit
// (oracle order 0, pinned to tests/golden). It is a parity mode: a sum costs n dependent additions.
//
// sum over e < n of t_e, e ascending; t[c][k] is element 128 c + 2 lane + k. All lanes return it.
template <int CHUNKS>
__device__ inline double wave_sum_seq(const double (&t)[CHUNKS][2], uint64_t n) {
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t base = 128ull * static_cast<uint64_t>(c);
    if (base < n) {  // wave-uniform
      const int m = n - base >= 128 ? 128 : static_cast<int>(n - base);
      for (int l = 0; 2 * l < m; l++) {
        acc = acc + lane_broadcast(t[c][0], l);
        if (2 * l + 1 < m) acc = acc + lane_broadcast(t[c][1], l);
      }
    }
  }
  return acc;
}
// f(x) with the objective's terms added in index order (oracle_objective.c orc_objective_seq)
template <int OBJ, int CHUNKS>
__device__ inline double wave_objective_seq(const double (&xv)[CHUNKS][2], uint64_t D) {
  using O = Objective<OBJ>;
  if constexpr (O::kWhole) return O::whole(WavePoint<CHUNKS>{xv, D}, D);  // (rejected at engine creation)
  const int lane = lane_id();
  const uint64_t nt = O::n_terms(D);
  double t[CHUNKS][2];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    double xn = 0.0;
    if (O::kChain) {
      const double same = lane_down1(xv[c][0]);
      double next = 0.0;
      if (c + 1 < CHUNKS) next = lane_first(xv[c + 1][0]);
      xn = (lane == 63) ? next : same;
    }
    t[c][0] = O::term(xv[c][0], xv[c][1]);
    t[c][1] = O::term(xv[c][1], xn);
  }
  return O::finish(wave_sum_seq<CHUNKS>(t, nt), D);
}
// wave_sum_seq through a buffer of the wave's own (LDS; 128 CHUNKS doubles): the terms are stored once
// — the lane layout IS index order — and every lane walks them at a wave-uniform address, a read
// that does not depend on the chain, so the additions follow each other at the adder's latency
// instead of behind two v_readlane each (measured: 80 -> ~10 cycles per term). Same additions in
// the same order: the same bits.
// sum of f(buf[0 .. m)) in index order, one chain. The reads of the NEXT eight values are issued before
// the additions of the current eight (which only wait for each other), so a term costs the adder's
// latency, not an LDS round trip per block. Reads up to 7 doubles past m (inside the caller's
// allocation; never added).
template <typename F>
__device__ inline double serial_chain_lds(const double *buf, int m, double acc, F f) {
  double a[8], b[8];
#pragma unroll
  for (int u = 0; u < 8; u++) a[u] = buf[u];
  int e = 0;
  for (; e + 16 <= m; e += 16) {
#pragma unroll
    for (int u = 0; u < 8; u++) b[u] = buf[e + 8 + u];
#pragma unroll
    for (int u = 0; u < 8; u++) acc = acc + f(a[u]);
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = buf[e + 16 + u];
#pragma unroll
    for (int u = 0; u < 8; u++) acc = acc + f(b[u]);
  }
  if (e + 8 <= m) {
#pragma unroll
    for (int u = 0; u < 8; u++) b[u] = buf[e + 8 + u];
#pragma unroll
    for (int u = 0; u < 8; u++) acc = acc + f(a[u]);
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = b[u];
    e += 8;
  }
#pragma unroll
  for (int u = 0; u < 8; u++)
    if (e + u < m) acc = acc + f(a[u]);  // (wave-uniform)
  return acc;
}
__device__ inline double serial_sum_lds(const double *buf, int m, double acc = 0.0) {
  return serial_chain_lds(buf, m, acc, [](double v) { return v; });
}
template <int CHUNKS>
__device__ inline double wave_sum_seq_buf(const double (&t)[CHUNKS][2], uint64_t n, double *buf) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++)
    *reinterpret_cast<double2 *>(buf + 128 * c + 2 * lane) = make_double2(t[c][0], t[c][1]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double acc = serial_sum_lds(buf, static_cast<int>(n));
  __builtin_amdgcn_wave_barrier();  // (the buffer's next stores come after these reads)
  return acc;
}
// wave_objective_seq with its serial sum through such a buffer (term objectives)
template <int OBJ, int CHUNKS>
__device__ inline double wave_objective_seq_buf(const double (&xv)[CHUNKS][2], uint64_t D, double *buf) {
  using O = Objective<OBJ>;
  if constexpr (O::kWhole) {
    return O::whole(WavePoint<CHUNKS>{xv, D}, D);  // (engines reject reference order for whole-vector bodies)
  } else {
    const int lane = lane_id();
    double t[CHUNKS][2];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      double xn = 0.0;
      if (O::kChain) {
        const double same = lane_down1(xv[c][0]);
        double next = 0.0;
        if (c + 1 < CHUNKS) next = lane_first(xv[c + 1][0]);
        xn = (lane == 63) ? next : same;
      }
      t[c][0] = O::term(xv[c][0], xv[c][1]);
      t[c][1] = O::term(xv[c][1], xn);
    }
    return O::finish(wave_sum_seq_buf<CHUNKS>(t, O::n_terms(D), buf), D);
  }
}
// the same for a point held by a group of G lanes (group_objective below): lane g of the group
// holds terms 2g and 2g+1; they are added in index order by walking the group's lanes
template <int OBJ, int G>
__device__ inline double group_objective_seq(double x0, double x1, uint64_t D) {
  using O = Objective<OBJ>;
  if constexpr (O::kWhole) return O::whole(GroupPoint<G>{x0, x1, D}, D);
  const int base = lane_id() & ~(G - 1);
  const uint64_t nt = O::n_terms(D);
  double xn = 0.0;
  if (O::kChain) xn = lane_down1(x0);
  const double t0 = O::term(x0, x1), t1 = O::term(x1, xn);
  auto from = [&](double v, int src) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = static_cast<uint32_t>(__shfl(static_cast<int>(b & 0xffffffffu), src, 64));
    const uint32_t hi = static_cast<uint32_t>(__shfl(static_cast<int>(b >> 32), src, 64));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
  };
  double acc = 0.0;
  for (int l = 0; 2 * static_cast<uint64_t>(l) < nt; l++) {  // nt <= 2 G: the point fits the group
    acc = acc + from(t0, base + l);
    if (2 * static_cast<uint64_t>(l) + 1 < nt) acc = acc + from(t1, base + l);
  }
  return O::finish(acc, D);
}

// The chunk loop of wave_objective, resumable: a row longer than the registers hold (D > 1024) is
// taken in segments of CHUNKS chunks; `acc` carries the lane's partial from segment to segment
// (ascending — the order of the whole-row loop), `e_base` is the segment's first element and
// `next_first` the element after its last one (x[e_base + 128 CHUNKS]; chain objectives only).
// objective_finish() closes the sum. Same bits as wave_objective on a hypothetical longer row.
template <int OBJ, int CHUNKS>
__device__ inline void objective_accumulate(double &acc, const double (&xv)[CHUNKS][2], uint64_t e_base,
                                            uint64_t D, double next_first) {
  using O = Objective<OBJ>;
  const int lane = lane_id();
  const uint64_t nt = O::n_terms(D);
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = e_base + static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    double xn = 0.0;
    if (O::kChain) {
      const double same = lane_down1(xv[c][0]);
      double next = next_first;
      if (c + 1 < CHUNKS) next = lane_first(xv[c + 1][0]);
      xn = (lane == 63) ? next : same;
    }
    if (e0 < nt) acc = acc + O::term(xv[c][0], xv[c][1]);
    if (e0 + 1 < nt) acc = acc + O::term(xv[c][1], xn);
  }
}
template <int OBJ>
__device__ inline double objective_finish(double acc, uint64_t D) {
  return Objective<OBJ>::finish(wave_sum(acc), D);
}

// The same for a point of at most 2 G coordinates held by a GROUP of G lanes (G a power of two;
// lane g of the group holds x[2g], x[2g+1]): a wave then evaluates 64 / G points at once. It is
// wave_objective restricted to the group — the same per-lane partial and the levels G/2 .. 1 of
// the same butterfly; the levels it drops only ever add the zeros of unused lanes, so the value
// has the bits of the full-wave tree. All lanes of the group return it.
template <int OBJ, int G>
__device__ inline double group_objective(double x0, double x1, uint64_t D) {
  using O = Objective<OBJ>;
  if constexpr (O::kWhole) return O::whole(GroupPoint<G>{x0, x1, D}, D);
  const uint64_t e0 = 2 * static_cast<uint64_t>(lane_id() & (G - 1));
  const uint64_t nt = O::n_terms(D);
  double xn = 0.0;
  if (O::kChain) xn = lane_down1(x0);  // x[e0+2]; only read where e0 + 2 < D, inside the group
  double acc = 0.0;
  if (e0 < nt) acc = acc + O::term(x0, x1);
  if (e0 + 1 < nt) acc = acc + O::term(x1, xn);
  if constexpr (G > 1)
    butterfly_levels<G / 2>([&](auto off) { acc = acc + lane_xor<decltype(off)::value>(acc); });
  return O::finish(acc, D);
}


// ---------------------------------------------------------------------------
// pieces shared by the population engines (DE, PSO)
// ---------------------------------------------------------------------------
constexpr int kTile = 1024;     // reduction tile
constexpr int kRecHeader = 5;   // exchange record header: min, argmin, sum, M2, valid

// Per tile of kTile scores: block-tree sum, minimum and first index of it,
// block-tree sum of squared deviations (second pass).
struct TilePartial {
  double sum;
  double minv;
  uint64_t mini;  // shard-local index (~0 if the tile holds only NaN)
  double m2;
};

// ---- single-launch reductions over tiles: every tile block publishes its partial with
// write-through stores, drains them and takes a ticket; the block whose ticket is the last one
// reads all partials and finishes (DE head, PSO head). Relaxed agent-scope accesses bypass the
// non-coherent levels, and the ticket's read-modify-write orders the blocks where it matters; no
// release / acquire fence (a release would write back every dirty L2 line other blocks hold).
__device__ inline void sc1_store(double *ptr, double v) {
  __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double sc1_load(const double *ptr) {
  return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0, after the block's partial stores: true in the block that arrived last of `count`
__device__ inline bool take_ticket(uint32_t *ticket, uint32_t count) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t != count - 1) return false;
  __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// ---- row access ------------------------------------------------------------
// Branch-free and select-free on purpose: a load guarded by a runtime condition
// makes hipcc branch around it and drain vmcnt before the next one, and a select
// on the loaded value forces the wait to the load site; both serialise the row
// gathers. Lanes past the end of the row read 16 bytes of zeros (`zero`) instead,
// so all loads of an agent are issued back to back and waited for at first use.
// VEC = rows are 16-byte aligned (D even).
template <int CHUNKS, bool VEC>
__device__ inline void load_row(const double *__restrict__ row, uint64_t D,
                                const double *__restrict__ zero, double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (VEC) {
      const double *src = (e0 < D) ? row + e0 : zero;  // D even: e0 + 1 < D as well
      const double2 t = *reinterpret_cast<const double2 *>(src);
      v[c][0] = t.x;
      v[c][1] = t.y;
    } else {
      v[c][0] = *((e0 < D) ? row + e0 : zero);
      v[c][1] = *((e0 + 1 < D) ? row + e0 + 1 : zero);
    }
  }
}
template <int CHUNKS, bool VEC>
__device__ inline void store_row(double *__restrict__ row, uint64_t D,
                                 const double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (VEC) {
      if (e0 < D) *reinterpret_cast<double2 *>(row + e0) = make_double2(v[c][0], v[c][1]);
    } else {
      if (e0 < D) row[e0] = v[c][0];
      if (e0 + 1 < D) row[e0 + 1] = v[c][1];
    }
  }
}

// The same for rows that are touched once per pass and by their owner only (a particle's
// position, a trial row nobody reads before the next generation): nontemporal, streamed past the
// caches' retention so that what IS re-read (donor rows, the swarm best) keeps its place.
typedef double nlsg_v2d __attribute__((ext_vector_type(2)));
template <int CHUNKS, bool VEC>
__device__ inline void load_row_stream(const double *__restrict__ row, uint64_t D,
                                       const double *__restrict__ zero, double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (VEC) {
      const nlsg_v2d t = __builtin_nontemporal_load(reinterpret_cast<const nlsg_v2d *>((e0 < D) ? row + e0 : zero));
      v[c][0] = t.x;
      v[c][1] = t.y;
    } else {
      v[c][0] = __builtin_nontemporal_load((e0 < D) ? row + e0 : zero);
      v[c][1] = __builtin_nontemporal_load((e0 + 1 < D) ? row + e0 + 1 : zero);
    }
  }
}
template <int CHUNKS, bool VEC>
__device__ inline void store_row_stream(double *__restrict__ row, uint64_t D,
                                        const double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (VEC) {
      nlsg_v2d t;
      t.x = v[c][0];
      t.y = v[c][1];
      if (e0 < D) __builtin_nontemporal_store(t, reinterpret_cast<nlsg_v2d *>(row + e0));
    } else {
      if (e0 < D) __builtin_nontemporal_store(v[c][0], row + e0);
      if (e0 + 1 < D) __builtin_nontemporal_store(v[c][1], row + e0 + 1);
    }
  }
}

// a segment of kSeg chunks of a long row (D > 1024): elements e_base + 128 c + 2 lane + k
constexpr int kSeg = 8;
template <bool VEC, bool STREAM = false>
__device__ inline void load_segment(const double *__restrict__ row, uint64_t e_base, uint64_t D,
                                    const double *__restrict__ zero, double (&v)[kSeg][2]) {
  const uint64_t left = e_base < D ? D - e_base : 0;
  if (STREAM)
    load_row_stream<kSeg, VEC>(row + e_base, left, zero, v);
  else
    load_row<kSeg, VEC>(row + e_base, left, zero, v);
}
template <bool VEC, bool STREAM = false>
__device__ inline void store_segment(double *__restrict__ row, uint64_t e_base, uint64_t D,
                                     const double (&v)[kSeg][2]) {
  const uint64_t left = e_base < D ? D - e_base : 0;
  if (STREAM)
    store_row_stream<kSeg, VEC>(row + e_base, left, v);
  else
    store_row<kSeg, VEC>(row + e_base, left, v);
}

// lower value wins; equal values keep the lower index; NaN never wins
__device__ inline void argmin_combine(double &v, uint64_t &i, double ov, uint64_t oi) {
  if (ov < v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

// higher value wins; equal values keep the lower index; NaN never wins
__device__ inline void argmax_combine(double &v, uint64_t &i, double ov, uint64_t oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

// wave- then block-level argmin; thread 0 returns the block result
__device__ inline void block_argmin_256(double &bv, uint64_t &bi, double *mv, uint64_t *mi) {
  butterfly_levels<32>([&](auto off) {
    const double ov = lane_xor<decltype(off)::value>(bv);
    const uint64_t oi = lane_xor<decltype(off)::value>(bi);
    argmin_combine(bv, bi, ov, oi);
  });
  const int wid = static_cast<int>(threadIdx.x) >> 6;
  __syncthreads();
  if (lane_id() == 0) {
    mv[wid] = bv;
    mi[wid] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < 4; w++) argmin_combine(bv, bi, mv[w], mi[w]);
}

// Per-shard summary; consumed by the finaliser directly (one GPU) or exchanged
// between ranks (record).
struct ShardLocal {
  double sum;     // tiled sum of the shard's scores (eps > 0 only)
  double mean;    // sum / shard_n
  double minv;    // shard minimum (incumbent keeps ties)
  uint64_t mini;  // GLOBAL index
  double m2;      // tiled sum of squared deviations from `mean`
  double valid;   // 1.0 when `mini` is owned by this shard
};

}  // namespace nlsg
