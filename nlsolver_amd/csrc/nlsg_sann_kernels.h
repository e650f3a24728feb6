// nlsolver_amd/csrc/nlsg_sann_kernels.h — gfx950 kernel of the batched simulated-annealing engine
// ("batched SANN chains").
//
// Replaces (reference file:line): SANN::solve nlsolver.h:2777-2814 with rnorm 2479-2485, for
// `batch` independent chains. One wave per chain, the chain's three points (current p, trial
// ptry, best x) in registers in the lane layout of the other engines (element 128 c + 2 l + k
// in lane l); a chain never touches memory between its first load and its last store, so the
// kernel is bound by the fp64 VALU (two counter draws, log, cos, sqrt per coordinate and step:
// the arithmetic of the Accelerated-PSO move) and the objective's reduction latency.
//
// The reference draws from one sequential generator; here every draw is keyed by (seed, chain,
// step, slot) like in the population engines: kc = key(seed, chain), inner step
// s = iter * (temperature_iter - 1) + (j - 1) has ks = key(kc, s); coordinate e takes draws 2e
// (log) and 2e + 1 (cos) of ks, the acceptance test draw 2 D. oracle_sann.c's orc_sann_sync
// executes the same chain on the CPU.
#pragma once

#include "nlsg_common.h"
#include "nlsg_math.h"

namespace nlsg {

struct SannProblem {
  double best;  // f_multiplier * f at x (:2781, 2809)
  uint64_t iter, fcalls;
};

struct SannParams {
  double *x;          // [batch][D] in: start, out: best point (:2808)
  double *p;          // [batch][D] current point of the chain between launches
  double *trial;      // [batch][D] trial point, D > 1024 only (sann_anneal_long_kernel)
  SannProblem *prob;  // [batch]
  const double *zero;
  uint64_t batch, D, seed, chain_lo;
  uint64_t inner;     // trial points per temperature: temperature_iter - 1 (0 if that is 0), :2795
  double temp_max, fmul;
};

// The acceptance test of a worse point (:2805), u < exp(-difference / t). A difference beyond
// 710 t puts the exponential's argument below -708, where det_exp returns 0 and no uniform is below
// it: the draw, the division and the exponential are then skipped — the same decision, and the
// usual case once a chain is cold (draws are keyed, so skipping one changes nothing else).
// `sure_reject` must be wave-uniform where it is branched on.
__device__ inline bool sann_sure_reject(double difference, double t) {
  return t > 0.0 && difference > 710.0 * t;
}

// outer iterations [iter_begin, iter_end) of every chain; iter_begin == 0 also scores the start
template <int OBJ, int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void sann_anneal_kernel(SannParams p, uint64_t iter_begin,
                                                          uint64_t iter_end) {
  __shared__ double rn_tab[kRnormTabDoubles];  // det_rnorm's logarithm table
  rnorm_table_to_lds(rn_tab);
  const uint64_t chain = static_cast<uint64_t>(blockIdx.x) * 4 +
                         __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (chain >= p.batch) return;
  const int lane = lane_id();
  const uint64_t D = p.D;
  constexpr double e_minus_1 = 1.7182818;  // :2780
  double xb[CHUNKS][2], pc[CHUNKS][2], pt[CHUNKS][2];
  load_row<CHUNKS, VEC>(p.x + chain * D, D, p.zero, xb);
  double best;
  uint64_t fcalls;
  if (iter_begin == 0) {  // :2781-2785
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) pc[c][0] = xb[c][0], pc[c][1] = xb[c][1];
    best = p.fmul * wave_objective<OBJ, CHUNKS>(xb, D);
    fcalls = 1;
  } else {
    load_row<CHUNKS, VEC>(p.p + chain * D, D, p.zero, pc);
    best = p.prob[chain].best;
    fcalls = p.prob[chain].fcalls;
  }
  const double scale = 1.0 / p.temp_max;
  const uint64_t kc = ctr_key(p.seed, p.chain_lo + chain);
  const uint64_t inner = p.inner;
  for (uint64_t iter = iter_begin; iter < iter_end; iter++) {
    // temperature annealing schedule (:2793-2794)
    const double t = p.temp_max / det_log(static_cast<double>(iter) + e_minus_1);
    const double current_scale = t * scale;
    for (uint64_t j = 0; j < inner; j++) {
      const uint64_t ks = ctr_key(kc, iter * inner + j);
      // draw 2e (+1) of element e = 128 c + 2 lane + k: mix64(ks + G (2e + 1 [+ 1])), with
      // 2e + 1 = (4 lane + 1) + (256 c + 2 k): one 64-bit multiply per step and lane
      const uint64_t ks_lane = ks + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
#pragma unroll
      for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint64_t z1 = mix64(ks_lane + kGolden * static_cast<uint64_t>(256 * c + 2 * k));  // one draw (slot 2e) per normal variate
          // rnorm (:2479-2485): sqrt(-2 log u1) * cos(2 pi_ u2), pi_ = 3.141593
          const double rn = det_rnorm(z1, rn_tab);
          pt[c][k] = pc[c][k] + current_scale * rn;  // :2800
        }
      }
      if (D != 128u * CHUNKS) {  // lanes past the point's end hold zeros: one wave-uniform branch per
        asm volatile("");        // trial point instead of a select pair per element (as in pso_move_kernel)
#pragma unroll
        for (int c = 0; c < CHUNKS; c++)
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
            pt[c][k] = (e < D) ? pt[c][k] : 0.0;
          }
      }
      const double current_val = p.fmul * wave_objective<OBJ, CHUNKS>(pt, D);
      fcalls++;
      const double difference = current_val - best;  // against the best so far (:2804)
      bool accept = difference <= 0.0;
      if (!accept && !sann_sure_reject(difference, t))  // wave-uniform
        accept = u01(ctr_key(ks, 2 * D)) < det_exp(-difference / t);  // :2805
      if (accept) {  // wave-uniform
        const bool better = current_val <= best;
#pragma unroll
        for (int c = 0; c < CHUNKS; c++)
#pragma unroll
          for (int k = 0; k < 2; k++) {
            pc[c][k] = pt[c][k];
            xb[c][k] = better ? pt[c][k] : xb[c][k];
          }
        best = better ? current_val : best;
      }
    }
  }
  store_row<CHUNKS, VEC>(p.x + chain * D, D, xb);
  store_row<CHUNKS, VEC>(p.p + chain * D, D, pc);
  if (lane == 0) {
    p.prob[chain].best = best;
    p.prob[chain].iter = iter_end;
    p.prob[chain].fcalls = fcalls;
  }
}

// Chains longer than a wave's registers hold (D > 1024; the reference has no limit): the three
// points live in memory (x: best, p: current, trial), a trial is built and scored segment by
// segment of 1024 coordinates (objective_accumulate: the whole-row order) and copied over the
// current / best point when it is accepted. Same draws, same arithmetic as the register-resident
// kernel.
template <int OBJ, bool VEC>
__global__ __launch_bounds__(256) void sann_anneal_long_kernel(SannParams p, uint64_t iter_begin,
                                                               uint64_t iter_end) {
  __shared__ double rn_tab[kRnormTabDoubles];  // det_rnorm's logarithm table
  rnorm_table_to_lds(rn_tab);
  const uint64_t chain = static_cast<uint64_t>(blockIdx.x) * 4 +
                         __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (chain >= p.batch) return;
  const int lane = lane_id();
  const uint64_t D = p.D;
  constexpr double e_minus_1 = 1.7182818;  // :2780
  double *xb = p.x + chain * D, *pc = p.p + chain * D, *pt = p.trial + chain * D;
  auto copy_row = [&](double *dst, const double *src) {
    for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
      double v[kSeg][2];
      load_segment<VEC>(src, e_base, D, p.zero, v);
      store_segment<VEC>(dst, e_base, D, v);
    }
  };
  double best;
  uint64_t fcalls;
  if (iter_begin == 0) {  // :2781-2785
    double acc = 0.0;
    for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
      double v[kSeg][2];
      load_segment<VEC>(xb, e_base, D, p.zero, v);
      store_segment<VEC>(pc, e_base, D, v);
      const uint64_t en = e_base + 128 * kSeg;
      objective_accumulate<OBJ, kSeg>(acc, v, e_base, D, en < D ? xb[en] : 0.0);
    }
    best = p.fmul * objective_finish<OBJ>(acc, D);
    fcalls = 1;
  } else {
    best = p.prob[chain].best;
    fcalls = p.prob[chain].fcalls;
  }
  const double scale = 1.0 / p.temp_max;
  const uint64_t kc = ctr_key(p.seed, p.chain_lo + chain);
  const uint64_t inner = p.inner;
  for (uint64_t iter = iter_begin; iter < iter_end; iter++) {
    const double t = p.temp_max / det_log(static_cast<double>(iter) + e_minus_1);  // :2793-2794
    const double current_scale = t * scale;
    for (uint64_t j = 0; j < inner; j++) {
      const uint64_t ks = ctr_key(kc, iter * inner + j);
      const uint64_t ks_lane = ks + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
      auto trial_at = [&](uint64_t e) {  // one coordinate of the trial point, the same in every lane
        if (e >= D) return 0.0;
        const uint64_t z1 = ctr_key(ks, 2 * e);
        const double rn = det_rnorm(z1, rn_tab);
        return pc[e] + current_scale * rn;
      };
      double acc = 0.0;
      for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
        double vc[kSeg][2], vt[kSeg][2];
        load_segment<VEC>(pc, e_base, D, p.zero, vc);
        const uint64_t kseg = ks_lane + kGolden * (2 * e_base);
#pragma unroll
        for (int c = 0; c < kSeg; c++)
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const uint64_t e = e_base + static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
            const uint64_t z1 = mix64(kseg + kGolden * static_cast<uint64_t>(256 * c + 2 * k));
            const double rn = det_rnorm(z1, rn_tab);  // rnorm, :2479-2485
            vt[c][k] = (e < D) ? vc[c][k] + current_scale * rn : 0.0;  // :2800
          }
        store_segment<VEC>(pt, e_base, D, vt);
        objective_accumulate<OBJ, kSeg>(acc, vt, e_base, D, trial_at(e_base + 128 * kSeg));
      }
      const double current_val = p.fmul * objective_finish<OBJ>(acc, D);
      fcalls++;
      const double difference = current_val - best;  // against the best so far (:2804)
      bool accept = difference <= 0.0;
      if (!accept && !sann_sure_reject(difference, t))  // wave-uniform
        accept = u01(ctr_key(ks, 2 * D)) < det_exp(-difference / t);  // :2805
      if (accept) {  // wave-uniform
        copy_row(pc, pt);
        if (current_val <= best) {
          copy_row(xb, pt);
          best = current_val;
        }
      }
    }
  }
  if (lane == 0) {
    p.prob[chain].best = best;
    p.prob[chain].iter = iter_end;
    p.prob[chain].fcalls = fcalls;
  }
}

// The same for chains of at most 64 coordinates: a chain fills D / 2 lanes, so a wave anneals
// 64 / G chains at once, one per group of G lanes (G = 4, 8, 16, 32 for D <= 8, 16, 32, 64; lane g
// of a group holds coordinates 2g, 2g + 1). Acceptance is per group (selects instead of the
// wave-uniform branch); group_objective gives every trial the bits of the full-wave tree, so a
// chain's history does not depend on how it was packed.
template <int OBJ, int G>
__global__ __launch_bounds__(256) void sann_anneal_groups_kernel(SannParams p, uint64_t iter_begin,
                                                                 uint64_t iter_end) {
  __shared__ double rn_tab[kRnormTabDoubles];  // det_rnorm's logarithm table
  rnorm_table_to_lds(rn_tab);
  constexpr int P = 64 / G;
  const int lane = lane_id(), g = lane & (G - 1), gi = lane / G;
  const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * 4 +
                        __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (wave * P >= p.batch) return;
  const bool live = wave * P + gi < p.batch;
  const uint64_t chain = live ? wave * P + gi : wave * P;  // idle groups shadow a live chain
  const uint64_t D = p.D;
  constexpr double e_minus_1 = 1.7182818;  // :2780
  const uint32_t j0 = 2 * g, j1 = 2 * g + 1;
  const bool in0 = j0 < D, in1 = j1 < D;
  double *xrow = p.x + chain * D, *prow = p.p + chain * D;
  double xb[2] = {in0 ? xrow[j0] : 0.0, in1 ? xrow[j1] : 0.0};
  double pc[2], best;
  uint64_t fcalls;
  if (iter_begin == 0) {  // :2781-2785
    pc[0] = xb[0];
    pc[1] = xb[1];
    best = p.fmul * group_objective<OBJ, G>(xb[0], xb[1], D);
    fcalls = 1;
  } else {
    pc[0] = in0 ? prow[j0] : 0.0;
    pc[1] = in1 ? prow[j1] : 0.0;
    best = p.prob[chain].best;
    fcalls = p.prob[chain].fcalls;
  }
  const double scale = 1.0 / p.temp_max;
  const uint64_t kc = ctr_key(p.seed, p.chain_lo + chain);
  const uint64_t inner = p.inner;
  const uint64_t goff = kGolden * (4 * static_cast<uint64_t>(g) + 1);
  for (uint64_t iter = iter_begin; iter < iter_end; iter++) {
    const double t = p.temp_max / det_log(static_cast<double>(iter) + e_minus_1);
    const double current_scale = t * scale;
    for (uint64_t j = 0; j < inner; j++) {
      const uint64_t ks = ctr_key(kc, iter * inner + j);
      const uint64_t ks_lane = ks + goff;
      double pt[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t z1 = mix64(ks_lane + kGolden * static_cast<uint64_t>(2 * k));  // one draw (slot 2e) per normal variate
        const double rn = det_rnorm(z1, rn_tab);
        pt[k] = ((k ? in1 : in0)) ? pc[k] + current_scale * rn : 0.0;
      }
      const double current_val = p.fmul * group_objective<OBJ, G>(pt[0], pt[1], D);
      fcalls++;
      const double difference = current_val - best;
      bool accept = difference <= 0.0;
      // (per group; the exponential is skipped when no group of the wave needs it)
      if (__ballot(!accept && !sann_sure_reject(difference, t)) != 0ull)
        accept = accept || (u01(ctr_key(ks, 2 * D)) < det_exp(-difference / t));
      const bool better = accept && current_val <= best;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        pc[k] = accept ? pt[k] : pc[k];
        xb[k] = better ? pt[k] : xb[k];
      }
      best = better ? current_val : best;
    }
  }
  if (live) {
    if (in0) xrow[j0] = xb[0], prow[j0] = pc[0];
    if (in1) xrow[j1] = xb[1], prow[j1] = pc[1];
    if (g == 0) {
      p.prob[chain].best = best;
      p.prob[chain].iter = iter_end;
      p.prob[chain].fcalls = fcalls;
    }
  }
}

}  // namespace nlsg
