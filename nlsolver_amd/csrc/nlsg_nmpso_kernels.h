// nlsolver_amd/csrc/nlsg_nmpso_kernels.h — gfx950 kernel of the batched Nelder-Mead / PSO hybrid
// (SURVEY.md §8f N4).
//
// Replaces (reference file:line): NelderMeadPSO::solve nlsolver.h:3623-3685, init_solver_state
// 3686-3738, apply_simplex 3739-3822, apply_pso 3823-3866, update_centroid 3867-3884, shrink
// 3885-3902, simplex_std_err 3903-3918, the minimize / maximize wrappers 3583-3620, with
// simplex_transform 1986-2007 and max_abs_vec 1894-1904 — for `batch` independent instances.
//
// One persistent 256-thread workgroup per instance. The 3n + 1 particles (n <= 128) and their
// never-changing velocities (H3 below) live in global memory — 0.4 MB per instance at n = 128,
// L2-resident while the instance runs — their values and the best-to-worst order in LDS. Per
// iteration: a rank sort of the values (every thread counts the particles that precede its own:
// no barrier ladder, ties keep their current order), the stop tests, the simplex step on the
// best n + 1 particles (a chain of data-dependent decisions: thread j owns coordinate j, wave 0
// evaluates the trial points), then the PSO move of the other 2n particles: one PAIR per group of
// lanes (the second particle of the first pair reads the first one's new position, H4), several
// pairs per wave pass when n < 128, the objective evaluated by the same lanes.
//
// Reference behaviour kept literally (oracle_nmpso.c lists the evidence): H1 the last simplex
// particle keeps x; H2 the no-change counter compares with the first particle's INITIAL value;
// H3 velocities are never written back; H4 the "better of the pair" is the pair's second
// particle (the first pair: its first); H5 (bounded overloads) the velocity clamp uses the
// coordinate's bounds. Draws are keyed by (seed, instance, iteration, particle rank, slot);
// oracle_nmpso.c's orc_nmpso_sync executes the same run on the CPU.
#pragma once

#include "nlsg_common.h"

namespace nlsg {

constexpr int kHybThreads = 1024;  // the largest workgroup
// threads of an instance's workgroup: one per particle (the rank sort's unit of work), in whole
// waves (n = 128: seven), at most kHybThreads — small instances then run as many small workgroups,
// large ones have enough waves to cover the row loads of the move
__host__ __device__ inline int hyb_block_threads(uint64_t n) {
  const uint64_t waves = (3 * n + 1 + 63) / 64;
  return 64 * static_cast<int>(waves > kHybThreads / 64 ? kHybThreads / 64 : waves);
}
constexpr int kHybMaxN = 128;
constexpr int kHybMaxParticles = 3 * kHybMaxN + 1;

struct HybProblem {
  double f;
  uint64_t iter, fcalls;
};

struct HybParams {
  double *x;                    // [batch][n] in: start, out: best particle
  double *pos, *vel;            // [batch][3n+1][n]
  const double *upper, *lower;  // [n], bounded overloads only
  HybProblem *prob;             // [batch]
  const double *zero;
  uint64_t batch, n, max_iter, no_change_iter, seed, inst_lo;
  double alpha, gamma, rho, sigma, inertia, cog, soc, eps, fmul;
  int32_t bounded, pad;
};

struct HybShared {
  double val[kHybMaxParticles + 1];
  uint64_t key[kHybMaxParticles + 3];  // the values in their current order as ordered integers
  uint32_t order[2][kHybMaxParticles + 1];
  double centroid[kHybMaxN], tr[kHybMaxN], te[kHybMaxN], tc[kHybMaxN];
  double up[kHybMaxN], lo[kHybMaxN];
  double ref_score, trial_score, best_val0;
  uint64_t iter, fcalls, no_change;
  int stop, cur;  // cur: which order[] buffer is current
};

// objective of the point at `pt` (n <= 128 doubles, LDS or global), one wave; all lanes get it
template <int OBJ>
__device__ inline double hyb_wave_f(const double *pt, uint64_t n, double fmul) {
  const int lane = lane_id();
  double xv[1][2];
  xv[0][0] = (2u * lane < n) ? pt[2 * lane] : 0.0;
  xv[0][1] = (2u * lane + 1 < n) ? pt[2 * lane + 1] : 0.0;
  return fmul * wave_objective<OBJ, 1>(xv, n);
}

// The value order as an integer order: a < b (NaN last, -0 = +0) <=> hyb_key(a) < hyb_key(b)
__device__ inline uint64_t hyb_key(double v) {
  if (v != v) return ~0ull;
  const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v + 0.0));  // -0 -> +0
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// stable rank sort of the particles by value: order[cur] -> order[cur ^ 1]. The values are first
// laid out in their current order as ordered integers, so the counting loop reads one LDS word
// per step that every lane shares (a broadcast) and compares integers, four steps per trip.
template <typename SH>
__device__ inline void hyb_sort(SH &sh, uint32_t total) {
  const uint32_t *src = sh.order[sh.cur];
  uint32_t *dst = sh.order[sh.cur ^ 1];
  for (uint32_t q = threadIdx.x; q < total + 3; q += blockDim.x)
    sh.key[q] = q < total ? hyb_key(sh.val[src[q]]) : ~0ull;  // the pads sort last, behind q
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
    const uint64_t v = sh.key[q];
    uint32_t rank = 0;
    for (uint32_t r = 0; r < total; r += 4) {
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint64_t w = sh.key[r + u];
        rank += (w < v || (w == v && r + u < q)) ? 1u : 0u;
      }
    }
    dst[rank] = src[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) sh.cur ^= 1;
  __syncthreads();
}

// simplex_std_err (3903-3918) over the best `count` particles, by one wave (lane l adds the
// elements l, l + 64, ... in order, then the xor butterfly; two passes)
template <typename SH>
__device__ inline double hyb_std_err_wave(const SH &sh, uint32_t count) {
  const uint32_t *ord = sh.order[sh.cur];
  const int lane = lane_id();
  double acc = 0.0;
  for (uint32_t i = lane; i < count; i += 64) acc = acc + sh.val[ord[i]];
  const double mean = wave_sum(acc) / static_cast<double>(count);
  acc = 0.0;
  for (uint32_t i = lane; i < count; i += 64) {
    const double d = sh.val[ord[i]] - mean;
    acc = acc + d * d;
  }
  return sqrt(wave_sum(acc) / static_cast<double>(count - 1));
}

// ---- group-packed vector work: a particle of n coordinates fills n / 2 lanes, so a wave handles
// 64 / G particles at once, one per group of G lanes (G = 4 .. 64 for n <= 8 .. 128; lane g of a
// group holds coordinates 2g, 2g + 1). group_objective gives each the bits of the full-wave tree.

// values of `count` particles (row_of(i) = particle id of the i-th), 64 / G per wave pass
template <int OBJ, int G, typename SH, typename RowOf>
__device__ inline void hyb_eval_rows(const HybParams &p, SH &sh, const double *pos,
                                     uint32_t n, uint32_t count, RowOf row_of) {
  constexpr int P = 64 / G;
  const int lane = lane_id(), g = lane & (G - 1), gi = lane / G;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const uint32_t j0 = 2 * g, j1 = 2 * g + 1;
  for (uint32_t i0 = wid * P; i0 < count; i0 += (blockDim.x >> 6) * P) {
    const uint32_t i = i0 + gi;
    const bool live = i < count;
    const uint32_t id = row_of(live ? i : 0);
    const double *row = pos + id * n;
    const double x0 = j0 < n ? row[j0] : 0.0, x1 = j1 < n ? row[j1] : 0.0;
    const double f = p.fmul * group_objective<OBJ, G>(x0, x1, n);
    if (live && g == 0) sh.val[id] = f;
  }
}

// apply_pso (3823-3866): the 2n particles behind the simplex, one PAIR per group and pass
template <int OBJ, int G, typename SH>
__device__ inline void hyb_pso_move(const HybParams &p, SH &sh, double *pos,
                                    const double *vel, uint32_t n, uint32_t ns, uint64_t kc) {
  constexpr int P = 64 / G;
  const int lane = lane_id(), g = lane & (G - 1), gi = lane / G;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const uint32_t *ord = sh.order[sh.cur];
  const double *best = pos + ord[0] * n;
  const uint64_t kit = ctr_key(kc, sh.iter + 1);
  const uint32_t j0 = 2 * g, j1 = 2 * g + 1;
  const bool in0 = j0 < n, in1 = j1 < n;
  const double bb[2] = {in0 ? best[j0] : 0.0, in1 ? best[j1] : 0.0};
  const double lo[2] = {in0 ? sh.lo[j0] : 0.0, in1 ? sh.lo[j1] : 0.0};
  const double up[2] = {in0 ? sh.up[j0] : 0.0, in1 ? sh.up[j1] : 0.0};
  for (uint32_t m0 = wid * P; m0 < n; m0 += (blockDim.x >> 6) * P) {
    const bool live = m0 + gi < n;
    const uint32_t m = live ? m0 + gi : 0;  // idle groups shadow pair 0 and store nothing
    const uint32_t id_a = ord[ns + 2 * m], id_b = ord[ns + 2 * m + 1];
    double *ra = pos + id_a * n, *rb = pos + id_b * n;
    const double *va = vel + id_a * n, *vb = vel + id_b * n;
    double a[2] = {in0 ? ra[j0] : 0.0, in1 ? ra[j1] : 0.0};
    double b[2] = {in0 ? rb[j0] : 0.0, in1 ? rb[j1] : 0.0};
    const double wa[2] = {in0 ? va[j0] : 0.0, in1 ? va[j1] : 0.0};
    const double wb[2] = {in0 ? vb[j0] : 0.0, in1 ? vb[j1] : 0.0};
    // draws 2j, 2j+1 of coordinate j = 2g + k: mix64(kp + G64 (2j + 1 [+ 1])), 2j + 1 = 4g + 2k + 1
    const uint64_t goff = kGolden * (4 * static_cast<uint64_t>(g) + 1);
    // first of the pair: its "pairwise best" is itself (pair 0) or its partner (H4)
    {
      const uint64_t kp_lane = ctr_key(kit, 2 * m) + goff;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const double r_p = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k)));
        const double r_g = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k + 1)));
        const double pair = m == 0 ? a[k] : b[k];
        double temp = (p.inertia * wa[k]) + p.cog * r_p * (pair - a[k]) +
                      p.soc * r_g * (bb[k] - a[k]);
        if (p.bounded) temp = temp < lo[k] ? lo[k] : (up[k] < temp ? up[k] : temp);
        a[k] = a[k] + temp;
      }
    }
    // second of the pair: the first one's NEW position (pair 0) or itself
    {
      const uint64_t kp_lane = ctr_key(kit, 2 * m + 1) + goff;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const double r_p = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k)));
        const double r_g = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k + 1)));
        const double pair = m == 0 ? a[k] : b[k];
        double temp = (p.inertia * wb[k]) + p.cog * r_p * (pair - b[k]) +
                      p.soc * r_g * (bb[k] - b[k]);
        if (p.bounded) temp = temp < lo[k] ? lo[k] : (up[k] < temp ? up[k] : temp);
        b[k] = b[k] + temp;
      }
    }
    const double fa = p.fmul * group_objective<OBJ, G>(in0 ? a[0] : 0.0, in1 ? a[1] : 0.0, n);
    const double fb = p.fmul * group_objective<OBJ, G>(in0 ? b[0] : 0.0, in1 ? b[1] : 0.0, n);
    if (live) {
      if (in0) ra[j0] = a[0], rb[j0] = b[0];
      if (in1) ra[j1] = a[1], rb[j1] = b[1];
      if (g == 0) {
        sh.val[id_a] = fa;
        sh.val[id_b] = fb;
      }
    }
  }
}


// ---- more than 128 coordinates (the reference has no limit): the same kernel body over a view of
// DYNAMIC shared memory sized by n (values, sort keys, two orders: 3n + 1 entries each; six
// n-vectors) — 121 KiB at n = 1024 — a 1024-thread workgroup, and one particle (or one pair)
// per wave pass with the point in CHUNKS x 128 registers per lane. wave_objective gives the same
// bits for every CHUNKS that covers n, so the oracle is the one of the packed kernels.
constexpr int kHybWideMaxN = 1024;
struct HybScalars {
  double ref_score, trial_score, best_val0;
  uint64_t iter, fcalls, no_change;
  int stop, cur;
};
struct HybView {
  double *val;
  uint64_t *key;
  uint32_t *order[2];
  double *centroid, *tr, *te, *tc, *up, *lo;
  double &ref_score, &trial_score, &best_val0;
  uint64_t &iter, &fcalls, &no_change;
  int &stop, &cur;
};
__host__ __device__ inline size_t hyb_view_bytes(uint64_t n) {
  const uint64_t total = 3 * n + 1;
  return sizeof(HybScalars) + (total + 1) * 8 + (total + 3) * 8 + 2 * (total + 2) * 4 + 6 * n * 8;
}
__device__ inline HybView hyb_make_view(unsigned char *base, uint32_t n) {
  const uint32_t total = 3 * n + 1;
  HybScalars *sc = reinterpret_cast<HybScalars *>(base);
  double *val = reinterpret_cast<double *>(base + sizeof(HybScalars));
  uint64_t *key = reinterpret_cast<uint64_t *>(val + total + 1);
  double *vec = reinterpret_cast<double *>(key + total + 3);
  uint32_t *ord = reinterpret_cast<uint32_t *>(vec + 6 * n);
  return HybView{val, key, {ord, ord + total + 2}, vec, vec + n, vec + 2 * n, vec + 3 * n,
                 vec + 4 * n, vec + 5 * n, sc->ref_score, sc->trial_score, sc->best_val0,
                 sc->iter, sc->fcalls, sc->no_change, sc->stop, sc->cur};
}

template <int OBJ, int CHUNKS>
__device__ inline double hyb_wave_f_wide(const double *pt, uint64_t n, double fmul) {
  const int lane = lane_id();
  double xv[CHUNKS][2];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint32_t j0 = 128 * c + 2 * lane;
    xv[c][0] = j0 < n ? pt[j0] : 0.0;
    xv[c][1] = j0 + 1 < n ? pt[j0 + 1] : 0.0;
  }
  return fmul * wave_objective<OBJ, CHUNKS>(xv, n);
}

template <int OBJ, int CHUNKS, typename SH, typename RowOf>
__device__ inline void hyb_eval_rows_wide(const HybParams &p, SH &sh, const double *pos,
                                          uint32_t n, uint32_t count, RowOf row_of) {
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  for (uint32_t i = wid; i < count; i += (blockDim.x >> 6)) {
    const uint32_t id = row_of(i);
    const double f = hyb_wave_f_wide<OBJ, CHUNKS>(pos + static_cast<uint64_t>(id) * n, n, p.fmul);
    if (lane == 0) sh.val[id] = f;
  }
}

// apply_pso (3823-3866), one pair per wave pass; the two new points stay in registers for their
// evaluation, everything else streams through per 128-coordinate chunk
template <int OBJ, int CHUNKS, typename SH>
__device__ inline void hyb_pso_move_wide(const HybParams &p, SH &sh, double *pos,
                                         const double *vel, uint32_t n, uint32_t ns, uint64_t kc) {
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const uint32_t *ord = sh.order[sh.cur];
  const double *best = pos + static_cast<uint64_t>(ord[0]) * n;
  const uint64_t kit = ctr_key(kc, sh.iter + 1);
  for (uint32_t m = wid; m < n; m += (blockDim.x >> 6)) {
    const uint32_t id_a = ord[ns + 2 * m], id_b = ord[ns + 2 * m + 1];
    double *ra = pos + static_cast<uint64_t>(id_a) * n, *rb = pos + static_cast<uint64_t>(id_b) * n;
    const double *va = vel + static_cast<uint64_t>(id_a) * n, *vb = vel + static_cast<uint64_t>(id_b) * n;
    const uint64_t ka = ctr_key(kit, 2 * m), kb = ctr_key(kit, 2 * m + 1);
    double a[CHUNKS][2], b[CHUNKS][2];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t j = 128 * c + 2 * lane + k;
        const bool in = j < n;
        const uint32_t jj = in ? j : 0;
        const double bbj = best[jj], loj = sh.lo[jj], upj = sh.up[jj];
        double av = ra[jj], bv = rb[jj];
        const double wa = va[jj], wb = vb[jj];
        // draws 2j, 2j + 1 of the particle's key: mix64(key + G64 (2j + 1 [+ 1]))
        const uint64_t off = kGolden * (2 * static_cast<uint64_t>(j) + 1);
        {  // first of the pair: its "pairwise best" is itself (pair 0) or its partner (H4)
          const double r_p = u01(mix64(ka + off)), r_g = u01(mix64(ka + off + kGolden));
          const double pair = m == 0 ? av : bv;
          double temp = (p.inertia * wa) + p.cog * r_p * (pair - av) + p.soc * r_g * (bbj - av);
          if (p.bounded) temp = temp < loj ? loj : (upj < temp ? upj : temp);
          av = av + temp;
        }
        {  // second of the pair: the first one's NEW position (pair 0) or itself
          const double r_p = u01(mix64(kb + off)), r_g = u01(mix64(kb + off + kGolden));
          const double pair = m == 0 ? av : bv;
          double temp = (p.inertia * wb) + p.cog * r_p * (pair - bv) + p.soc * r_g * (bbj - bv);
          if (p.bounded) temp = temp < loj ? loj : (upj < temp ? upj : temp);
          bv = bv + temp;
        }
        a[c][k] = in ? av : 0.0;
        b[c][k] = in ? bv : 0.0;
        if (in) ra[j] = av, rb[j] = bv;
      }
    }
    const double fa = p.fmul * wave_objective<OBJ, CHUNKS>(a, n);
    const double fb = p.fmul * wave_objective<OBJ, CHUNKS>(b, n);
    if (lane == 0) {
      sh.val[id_a] = fa;
      sh.val[id_b] = fb;
    }
  }
}

template <int OBJ, int WIDE>
__device__ inline double hyb_trial_f(const double *pt, uint64_t n, double fmul) {
  if constexpr (WIDE == 0)
    return hyb_wave_f<OBJ>(pt, n, fmul);
  else
    return hyb_wave_f_wide<OBJ, WIDE>(pt, n, fmul);
}

// lanes per particle: the smallest power of two >= n / 2, at least 4
#define NLSG_HYB_GROUPS(n, CALL) \
  do {                           \
    if ((n) <= 8) {              \
      CALL(4);                   \
    } else if ((n) <= 16) {      \
      CALL(8);                   \
    } else if ((n) <= 32) {      \
      CALL(16);                  \
    } else if ((n) <= 64) {      \
      CALL(32);                  \
    } else {                     \
      CALL(64);                  \
    }                            \
  } while (0)

// WIDE = 0: the packed kernels above (n <= 128); else the point's chunks per lane (n <= 128 WIDE)
template <int OBJ, int WIDE, typename SH>
__device__ inline void nmpso_run(const HybParams &p, SH &sh) {
  const uint64_t inst = blockIdx.x;
  const uint32_t n = static_cast<uint32_t>(p.n), ns = n + 1, total = 3 * n + 1;
  const int t = threadIdx.x, lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  double *pos = p.pos + inst * total * static_cast<uint64_t>(n), *vel = p.vel + inst * total * static_cast<uint64_t>(n);
  const double *x0 = p.x + inst * n;
  const uint64_t kc = ctr_key(p.seed, p.inst_lo + inst);

  // ---- bounds (3587-3593 for the unbounded overloads) and init_solver_state (3686-3738)
  if (t < static_cast<int>(n)) {
    if (p.bounded) {
      sh.up[t] = p.upper[t];
      sh.lo[t] = p.lower[t];
    } else {
      const double temp = fabs(2.5 * x0[t]);
      sh.lo[t] = -temp;
      sh.up[t] = temp;
    }
  }
  if (wid == 0) {  // max_abs_vec (1894-1904): the maximum is order-independent
    double m = 0.0;
    for (uint32_t j = lane; j < n; j += 64) m = fmax(m, fabs(x0[j]));
    butterfly_levels<32>([&](auto off) { m = fmax(m, lane_xor<decltype(off)::value>(m)); });
    if (lane == 0) {
      const double a = m < 1.0 ? 1.0 : m;
      sh.ref_score = a < 10 ? a : 10;  // scale, parked here until the first iteration
      sh.iter = 0;
      sh.no_change = 0;
      sh.cur = 0;
      sh.stop = 0;
    }
  }
  __syncthreads();
  {
    const double scale = sh.ref_score;
    const double nn = static_cast<double>(n);
    for (uint32_t e = t; e < ns * n; e += blockDim.x) {
      const uint32_t i = e / n, j = e % n;
      double v = x0[j];
      if (i == 0) v = x0[j] + ((1.0 - sqrt(nn + 1.0)) / nn * scale);
      else if (i == j) v = x0[j] + scale;  // i == n has no such element (H1)
      pos[e] = v;
      vel[e] = 0.0;
    }
    const uint64_t kinit = ctr_key(kc, 0);
    for (uint32_t r = wid; r < 2 * n; r += (blockDim.x >> 6)) {  // PSO particles, one per wave pass
      const uint64_t kp_lane = ctr_key(kinit, r) + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
#pragma unroll
      for (int c = 0; c < (WIDE ? WIDE : 1); c++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint32_t j = 128 * c + 2 * lane + k;  // draws 2j, 2j + 1 of the particle's key
          if (j < n) {
            const double lo = sh.lo[j], up = sh.up[j];
            const double temp = fabs(up - lo);
            const double u1 = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(256 * c + 2 * k)));
            const double u2 = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(256 * c + 2 * k + 1)));
            pos[static_cast<uint64_t>(ns + r) * n + j] = lo + ((up - lo) * u1);
            vel[static_cast<uint64_t>(ns + r) * n + j] = -temp + (u2 * temp);
          }
        }
      }
    }
  }
  __syncthreads();  // block-scope visibility of the rows written above
  if constexpr (WIDE != 0) {
    hyb_eval_rows_wide<OBJ, WIDE>(p, sh, pos, n, total, [](uint32_t i) { return i; });
  } else {
#define HYB_EVAL_ALL(G) hyb_eval_rows<OBJ, G>(p, sh, pos, n, total, [](uint32_t i) { return i; })
    NLSG_HYB_GROUPS(n, HYB_EVAL_ALL);
#undef HYB_EVAL_ALL
  }
  for (uint32_t i = t; i < total; i += blockDim.x) sh.order[0][i] = i;
  __syncthreads();
  if (t == 0) {
    sh.best_val0 = sh.val[0];  // read once, never updated (3658, H2)
    sh.fcalls = total;
  }

  for (;;) {
    hyb_sort(sh, total);
    // ---- stop tests (3660-3676)
    if (wid == 0) {
      const uint32_t *ord = sh.order[sh.cur];
      const double se = hyb_std_err_wave(sh, ns);
      if (lane == 0) {
        const bool same = sh.best_val0 == sh.val[ord[0]];
        sh.no_change = same ? sh.no_change + 1 : 0;
        sh.stop = (sh.iter >= p.max_iter || sh.no_change >= p.no_change_iter || se < p.eps) ? 1 : 0;
      }
    }
    __syncthreads();
    if (sh.stop) break;
    // ---- apply_simplex (3739-3822)
    {
      const uint32_t *ord = sh.order[sh.cur];
      const uint32_t worst = ord[ns - 1], second = ord[ns - 2];
      const double best_score = sh.val[ord[0]];
      double *wrow = pos + worst * n;
      if (t < static_cast<int>(n)) {  // update_centroid (3867-3884): particles in sorted order
        double c = 0.0;
        uint32_t i = 0;
        for (; i + 16 <= ns - 1; i += 16) {  // sixteen loads in flight, added in the reference's order
          double v[16];
#pragma unroll
          for (int u = 0; u < 16; u++) v[u] = pos[ord[i + u] * n + t];
#pragma unroll
          for (int u = 0; u < 16; u++) c = c + v[u];
        }
        for (; i + 8 <= ns - 1; i += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = pos[ord[i + u] * n + t];
#pragma unroll
          for (int u = 0; u < 8; u++) c = c + v[u];
        }
        for (; i < ns - 1; i++) c = c + pos[ord[i] * n + t];
        c = c / static_cast<double>(ns - 1);
        sh.centroid[t] = c;
        double r = c + p.alpha * (c - wrow[t]);  // reflect
        if (p.bounded) r = r < sh.lo[t] ? sh.lo[t] : (sh.up[t] < r ? sh.up[t] : r);
        sh.tr[t] = r;
      }
      __syncthreads();
      if (wid == 0) {
        const double f = hyb_trial_f<OBJ, WIDE>(sh.tr, n, p.fmul);
        if (lane == 0) sh.ref_score = f;
      }
      __syncthreads();
      const double ref_score = sh.ref_score;
      uint64_t calls = 1;
      if (ref_score >= best_score && ref_score < sh.val[second]) {
        if (t < static_cast<int>(n)) wrow[t] = sh.tr[t];
        if (t == 0) sh.val[worst] = ref_score;
      } else if (ref_score < best_score) {  // expand
        if (t < static_cast<int>(n)) {
          const double c = sh.centroid[t];
          double e = c + p.gamma * (sh.tr[t] - c);
          if (p.bounded) e = e < sh.lo[t] ? sh.lo[t] : (sh.up[t] < e ? sh.up[t] : e);
          sh.te[t] = e;
        }
        __syncthreads();
        if (wid == 0) {
          const double f = hyb_trial_f<OBJ, WIDE>(sh.te, n, p.fmul);
          if (lane == 0) sh.trial_score = f;
        }
        __syncthreads();
        const double exp_score = sh.trial_score;
        calls++;
        if (t < static_cast<int>(n)) wrow[t] = exp_score < ref_score ? sh.te[t] : sh.tr[t];
        if (t == 0) sh.val[worst] = exp_score < ref_score ? exp_score : ref_score;
      } else {  // contract outside (the reflected point) or inside (the worst point)
        const double worst_score = sh.val[worst];
        if (t < static_cast<int>(n)) {
          const double c = sh.centroid[t];
          const double from = ref_score < worst_score ? sh.tr[t] : wrow[t];
          double v = c + p.rho * (from - c);
          if (p.bounded) v = v < sh.lo[t] ? sh.lo[t] : (sh.up[t] < v ? sh.up[t] : v);
          sh.tc[t] = v;
        }
        __syncthreads();
        if (wid == 0) {
          const double f = hyb_trial_f<OBJ, WIDE>(sh.tc, n, p.fmul);
          if (lane == 0) sh.trial_score = f;
        }
        __syncthreads();
        const double cont_score = sh.trial_score;
        calls++;
        if (cont_score < (worst_score < ref_score ? worst_score : ref_score)) {
          if (t < static_cast<int>(n)) wrow[t] = sh.tc[t];
          if (t == 0) sh.val[worst] = cont_score;
        } else {  // shrink towards the best particle (3885-3902), rescore, re-sort
          const double *best = pos + ord[0] * n;
          for (uint32_t e = t; e < (ns - 1) * n; e += blockDim.x) {
            const uint32_t i = 1 + e / n, j = e % n;
            double *cur = pos + ord[i] * n;
            cur[j] = best[j] + p.sigma * (cur[j] - best[j]);
          }
          __syncthreads();
          if constexpr (WIDE != 0) {
            hyb_eval_rows_wide<OBJ, WIDE>(p, sh, pos, n, ns - 1,
                                          [ord](uint32_t i) { return ord[1 + i]; });
          } else {
#define HYB_EVAL_SHRUNK(G) \
  hyb_eval_rows<OBJ, G>(p, sh, pos, n, ns - 1, [ord](uint32_t i) { return ord[1 + i]; })
            NLSG_HYB_GROUPS(n, HYB_EVAL_SHRUNK);
#undef HYB_EVAL_SHRUNK
          }
          calls += ns - 1;
          __syncthreads();
          hyb_sort(sh, total);
        }
      }
      if (t == 0) sh.fcalls += calls;
      __syncthreads();
    }
    // ---- apply_pso (3823-3866)
    if constexpr (WIDE != 0) {
      hyb_pso_move_wide<OBJ, WIDE>(p, sh, pos, vel, n, ns, kc);
    } else {
#define HYB_MOVE(G) hyb_pso_move<OBJ, G>(p, sh, pos, vel, n, ns, kc)
      NLSG_HYB_GROUPS(n, HYB_MOVE);
#undef HYB_MOVE
    }
    __syncthreads();  // every wave has read sh.iter (the iteration's key) before it moves on
    if (t == 0) {
      sh.fcalls += 2 * n;
      sh.iter += 1;
    }
  }
  // ---- x = particle_positions[current_order[0]] (3671-3675)
  {
    const uint32_t bid = sh.order[sh.cur][0];
    if (t < static_cast<int>(n)) p.x[inst * n + t] = pos[bid * n + t];
    if (t == 0) {
      p.prob[inst].f = sh.val[bid];
      p.prob[inst].iter = sh.iter;
      p.prob[inst].fcalls = sh.fcalls;
    }
  }
}

template <int OBJ>
__global__ __launch_bounds__(kHybThreads) void nmpso_solve_kernel(HybParams p) {
  __shared__ HybShared sh;
  nmpso_run<OBJ, 0>(p, sh);
}

// 128 < n <= 128 CHUNKS: dynamic shared memory of hyb_view_bytes(n); a thread per coordinate, so
// 512 threads up to n = 512 (256 registers each: the pair of a wave's move stays in registers)
// and 1024 beyond
__host__ __device__ constexpr int hyb_wide_threads(int chunks) { return chunks <= 4 ? 512 : 1024; }
template <int OBJ, int CHUNKS>
__global__ __launch_bounds__(hyb_wide_threads(CHUNKS)) void nmpso_solve_wide_kernel(HybParams p) {
  extern __shared__ __align__(16) unsigned char hyb_smem[];
  HybView sh = hyb_make_view(hyb_smem, static_cast<uint32_t>(p.n));
  nmpso_run<OBJ, CHUNKS>(p, sh);
}

}  // namespace nlsg
