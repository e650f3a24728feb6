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
// evaluates the trial points), then the PSO move of the other 2n particles, one PAIR per wave
// pass (the second particle of the first pair reads the first one's new position, H4) with the
// objective evaluated by the same wave.
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

constexpr int kHybThreads = 256;
constexpr int kHybWaves = kHybThreads / 64;
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
  uint32_t order[2][kHybMaxParticles + 1];
  double centroid[kHybMaxN], tr[kHybMaxN], te[kHybMaxN], tc[kHybMaxN];
  double up[kHybMaxN], lo[kHybMaxN];
  double ref_score, trial_score, best_val0;
  uint64_t iter, fcalls, no_change;
  int stop, cur;  // cur: which order[] buffer is current
};

// value order of the sort: NaN last
__device__ inline bool hyb_less(double a, double b) { return (a < b) || (b != b && a == a); }

// objective of the point at `pt` (n <= 128 doubles, LDS or global), one wave; all lanes get it
template <int OBJ>
__device__ inline double hyb_wave_f(const double *pt, uint64_t n, double fmul) {
  const int lane = lane_id();
  double xv[1][2];
  xv[0][0] = (2u * lane < n) ? pt[2 * lane] : 0.0;
  xv[0][1] = (2u * lane + 1 < n) ? pt[2 * lane + 1] : 0.0;
  return fmul * wave_objective<OBJ, 1>(xv, n);
}

// stable rank sort of the particles by value: order[cur] -> order[cur ^ 1]
__device__ inline void hyb_sort(HybShared &sh, uint32_t total) {
  const uint32_t *src = sh.order[sh.cur];
  uint32_t *dst = sh.order[sh.cur ^ 1];
  for (uint32_t q = threadIdx.x; q < total; q += kHybThreads) {
    const uint32_t id = src[q];
    const double v = sh.val[id];
    uint32_t rank = 0;
    for (uint32_t r = 0; r < total; r++) {
      const double w = sh.val[src[r]];
      rank += (hyb_less(w, v) || (!hyb_less(v, w) && r < q)) ? 1u : 0u;
    }
    dst[rank] = id;
  }
  __syncthreads();
  if (threadIdx.x == 0) sh.cur ^= 1;
  __syncthreads();
}

// simplex_std_err (3903-3918) over the best `count` particles, by one wave (lane l adds the
// elements l, l + 64, ... in order, then the xor butterfly; two passes)
__device__ inline double hyb_std_err_wave(const HybShared &sh, uint32_t count) {
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

template <int OBJ>
__global__ __launch_bounds__(kHybThreads) void nmpso_solve_kernel(HybParams p) {
  __shared__ HybShared sh;
  const uint64_t inst = blockIdx.x;
  const uint32_t n = static_cast<uint32_t>(p.n), ns = n + 1, total = 3 * n + 1;
  const int t = threadIdx.x, lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  double *pos = p.pos + inst * total * n, *vel = p.vel + inst * total * n;
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
    for (uint32_t e = t; e < ns * n; e += kHybThreads) {
      const uint32_t i = e / n, j = e % n;
      double v = x0[j];
      if (i == 0) v = x0[j] + ((1.0 - sqrt(nn + 1.0)) / nn * scale);
      else if (i == j) v = x0[j] + scale;  // i == n has no such element (H1)
      pos[e] = v;
      vel[e] = 0.0;
    }
    const uint64_t kinit = ctr_key(kc, 0);
    for (uint32_t r = wid; r < 2 * n; r += kHybWaves) {  // PSO particles, one per wave pass
      const uint64_t kp_lane = ctr_key(kinit, r) + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t j = 2 * lane + k;
        if (j < n) {
          const double lo = sh.lo[j], up = sh.up[j];
          const double temp = fabs(up - lo);
          const double u1 = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k)));
          const double u2 = u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k + 1)));
          pos[(ns + r) * n + j] = lo + ((up - lo) * u1);
          vel[(ns + r) * n + j] = -temp + (u2 * temp);
        }
      }
    }
  }
  __syncthreads();  // block-scope visibility of the rows written above
  for (uint32_t i = wid; i < total; i += kHybWaves) {
    const double f = hyb_wave_f<OBJ>(pos + i * n, n, p.fmul);
    if (lane == 0) sh.val[i] = f;
  }
  for (uint32_t i = t; i < total; i += kHybThreads) sh.order[0][i] = i;
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
        for (uint32_t i = 0; i < ns - 1; i++) c = c + pos[ord[i] * n + t];
        c = c / static_cast<double>(ns - 1);
        sh.centroid[t] = c;
        double r = c + p.alpha * (c - wrow[t]);  // reflect
        if (p.bounded) r = r < sh.lo[t] ? sh.lo[t] : (sh.up[t] < r ? sh.up[t] : r);
        sh.tr[t] = r;
      }
      __syncthreads();
      if (wid == 0) {
        const double f = hyb_wave_f<OBJ>(sh.tr, n, p.fmul);
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
          const double f = hyb_wave_f<OBJ>(sh.te, n, p.fmul);
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
          const double f = hyb_wave_f<OBJ>(sh.tc, n, p.fmul);
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
          for (uint32_t e = t; e < (ns - 1) * n; e += kHybThreads) {
            const uint32_t i = 1 + e / n, j = e % n;
            double *cur = pos + ord[i] * n;
            cur[j] = best[j] + p.sigma * (cur[j] - best[j]);
          }
          __syncthreads();
          for (uint32_t i = 1 + wid; i < ns; i += kHybWaves) {
            const double f = hyb_wave_f<OBJ>(pos + ord[i] * n, n, p.fmul);
            if (lane == 0) sh.val[ord[i]] = f;
          }
          calls += ns - 1;
          __syncthreads();
          hyb_sort(sh, total);
        }
      }
      if (t == 0) sh.fcalls += calls;
      __syncthreads();
    }
    // ---- apply_pso (3823-3866): the 2n particles behind the simplex, one pair per wave pass
    {
      const uint32_t *ord = sh.order[sh.cur];
      const double *best = pos + ord[0] * n;
      const uint64_t kit = ctr_key(kc, sh.iter + 1);
      const uint32_t j0 = 2 * lane, j1 = 2 * lane + 1;
      const double b0 = j0 < n ? best[j0] : 0.0, b1 = j1 < n ? best[j1] : 0.0;
      const double lo0 = j0 < n ? sh.lo[j0] : 0.0, lo1 = j1 < n ? sh.lo[j1] : 0.0;
      const double up0 = j0 < n ? sh.up[j0] : 0.0, up1 = j1 < n ? sh.up[j1] : 0.0;
      for (uint32_t m = wid; m < n; m += kHybWaves) {
        const uint32_t id_a = ord[ns + 2 * m], id_b = ord[ns + 2 * m + 1];
        double *ra = pos + id_a * n, *rb = pos + id_b * n;
        const double *va = vel + id_a * n, *vb = vel + id_b * n;
        double a[2], b[2], wa[2], wb[2];
        a[0] = j0 < n ? ra[j0] : 0.0;
        a[1] = j1 < n ? ra[j1] : 0.0;
        b[0] = j0 < n ? rb[j0] : 0.0;
        b[1] = j1 < n ? rb[j1] : 0.0;
        wa[0] = j0 < n ? va[j0] : 0.0;
        wa[1] = j1 < n ? va[j1] : 0.0;
        wb[0] = j0 < n ? vb[j0] : 0.0;
        wb[1] = j1 < n ? vb[j1] : 0.0;
        const double bb[2] = {b0, b1}, lo[2] = {lo0, lo1}, up[2] = {up0, up1};
        // first of the pair: its "pairwise best" is itself (pair 0) or its partner (H4)
        {
          const uint64_t kp_lane = ctr_key(kit, 2 * m) + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
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
          const uint64_t kp_lane =
              ctr_key(kit, 2 * m + 1) + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
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
        double xa[1][2] = {{j0 < n ? a[0] : 0.0, j1 < n ? a[1] : 0.0}};
        double xb[1][2] = {{j0 < n ? b[0] : 0.0, j1 < n ? b[1] : 0.0}};
        const double fa = p.fmul * wave_objective<OBJ, 1>(xa, n);
        const double fb = p.fmul * wave_objective<OBJ, 1>(xb, n);
        if (j0 < n) ra[j0] = a[0], rb[j0] = b[0];
        if (j1 < n) ra[j1] = a[1], rb[j1] = b[1];
        if (lane == 0) {
          sh.val[id_a] = fa;
          sh.val[id_b] = fb;
        }
      }
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

}  // namespace nlsg
