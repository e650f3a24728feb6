// nlsolver_amd/csrc/nlsg_pso_kernels.h — gfx950 kernels of the PSO engine.
//
// Replaces (nlsolver.h): init_solver_state 2626-2657, update_velocities 2658-2677,
// update_positions 2678-2700, threshold_positions 2701-2715, update_best_positions
// 2716-2741 and the stop tests of solve 2599-2605.
//
// Layout: positions (and, Vanilla, velocities + personal-best positions) row-major
// [shard_n][D] fp64, updated IN PLACE by the wave that owns the particle (particles
// only interact through the swarm-best vector). One wave64 per particle; lane l
// holds elements c*128 + 2l, +1 of each 128-element chunk (D <= 64: several particles per wave,
// one per group of lanes, pso_move_groups_kernel). The head of an iteration is one launch
// (pso_scan_head_kernel) on one GPU with eps <= 0, else scan / local / finalize kernels.
#pragma once

#include "nlsg_common.h"
#include "nlsg_math.h"

namespace nlsg {

struct PsoState {
  double gbest_val;        // swarm_best_value (+inf sentinel)
  uint64_t gbest_idx;      // global particle index of the swarm best
  uint64_t iter;           // completed position updates
  uint64_t val_no_change;
  uint64_t fevals;
  double std_err;
  int32_t done;
  int32_t pending;         // a move ran since the last head (iter++ due)
};

struct PsoParams {
  double *pos, *vel, *pbest_pos;  // [shard_n][D]; vel / pbest_pos only for Vanilla
  double *pbest_val, *cur_val;    // [shard_n]
  double *gbest_x;                // [D]
  const double *lower, *upper;    // [D]
  const double *inertia_tab;      // pow(inertia, k), k < tab_len (Accelerated, :2613)
  PsoState *state;
  TilePartial *part;
  uint32_t *ticket;               // arrival counter of pso_scan_head_kernel's blocks
  const double *zero;
  uint64_t tab_len;
  uint32_t ntiles;
  int32_t tab_fixed;              // the table ends at a fixed point of pow (0, 1, inf): later k repeat it
  uint64_t n, D, shard_lo, shard_n;
  double inertia, cog, soc, eps, fmul;
  uint64_t max_iter, best_val_no_change, seed;
  int32_t type, bounded;
};

// inertia = pow(init_inertia, iter) (:2613) as the host libm computes it: from the table, whose
// last entry repeats for ever when it is a fixed point of the sequence. Only a schedule that is
// still moving after the table's 2^22 entries (|inertia| within 2e-4 of 1, or negative) falls
// back to the device's pow, which is not bit-identical to glibc's.
__device__ inline double pso_inertia_at(const PsoParams &p, uint64_t iter) {
  if (iter < p.tab_len) return p.inertia_tab[iter];
  return p.tab_fixed ? p.inertia_tab[p.tab_len - 1] : pow(p.inertia, static_cast<double>(iter));
}

__global__ void pso_reset_state_kernel(PsoParams p) {
  PsoState *s = p.state;
  s->gbest_val = __builtin_inf();
  s->gbest_idx = 0;
  s->iter = 0;
  s->val_no_change = 0;
  s->fevals = 0;
  s->std_err = __builtin_nan("");
  s->done = 0;
  s->pending = 0;
}

// init_solver_state (2626-2657) + the evaluations of the first
// update_best_positions (2595): pos = lo + (hi - lo) * u; Vanilla: vel = -w + u*w.
template <int OBJ, int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void pso_init_kernel(PsoParams p) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (i >= p.shard_n) return;
  const int lane = lane_id();
  const uint64_t kp = ctr_key(ctr_key(p.seed, 0), p.shard_lo + i);
  double lo[CHUNKS][2], hi[CHUNKS][2], xv[CHUNKS][2], vv[CHUNKS][2];
  load_row<CHUNKS, VEC>(p.lower, p.D, p.zero, lo);
  load_row<CHUNKS, VEC>(p.upper, p.D, p.zero, hi);
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
      const double temp = fabs(hi[c][k] - lo[c][k]);  // :2645
      xv[c][k] = lo[c][k] + ((hi[c][k] - lo[c][k]) * u01(ctr_key(kp, 2 * e)));
      vv[c][k] = -temp + (u01(ctr_key(kp, 2 * e + 1)) * temp);
      if (e >= p.D) xv[c][k] = 0.0;
    }
  }
  store_row<CHUNKS, VEC>(p.pos + i * p.D, p.D, xv);
  if (p.type == NLSG_PSO_VANILLA) {
    store_row<CHUNKS, VEC>(p.vel + i * p.D, p.D, vv);
    store_row<CHUNKS, VEC>(p.pbest_pos + i * p.D, p.D, xv);  // :2652
  }
  const double f = p.fmul * wave_objective<OBJ, CHUNKS>(xv, p.D);
  if (lane == 0) {
    p.cur_val[i] = f;
    p.pbest_val[i] = f;  // +inf sentinel: the first value always wins
  }
}

// One position update + evaluation (2606-2621 without the best bookkeeping).
template <int OBJ, int CHUNKS, bool VEC, int TYPE>
__global__ __launch_bounds__(256) void pso_move_kernel(PsoParams p, int timing, uint64_t iter_ovr) {
  const PsoState *__restrict__ st = p.state;
  if (!timing && st->done) return;
  // the logarithm table of det_rnorm (Accelerated only; a workgroup-wide step, so before any
  // wave leaves)
  __shared__ double rn_tab[TYPE == NLSG_PSO_ACCELERATED ? kRnormTabDoubles : 1];
  if (TYPE == NLSG_PSO_ACCELERATED) rnorm_table_to_lds(rn_tab);
  // A wave takes particles i0, i0 + (waves in the grid), ...: the rows every particle reads
  // (swarm best, bounds), the iteration's key and inertia, and the table above are fetched once
  // per wave instead of once per particle (they were 3/5 of the kernel's L2 reads).
  const uint64_t nwaves = static_cast<uint64_t>(gridDim.x) * 4;
  const int lane = lane_id();
  const uint64_t D = p.D;
  const uint64_t iter = timing ? iter_ovr : st->iter;
  const uint64_t kit = ctr_key(p.seed, iter + 1);
  double gb[CHUNKS][2], lo[CHUNKS][2], hi[CHUNKS][2];
  load_row<CHUNKS, VEC>(p.gbest_x, D, p.zero, gb);
  // bounds are only needed when thresholding; unbounded runs read the zero pad
  load_row<CHUNKS, VEC>(p.lower, p.bounded ? D : 0, p.zero, lo);
  load_row<CHUNKS, VEC>(p.upper, p.bounded ? D : 0, p.zero, hi);
  double inertia = p.inertia;
  if (TYPE == NLSG_PSO_ACCELERATED)  // :2613 inertia = pow(init_inertia, iter)
    inertia = pso_inertia_at(p, iter);
  const uint64_t lane_off = kGolden * (4 * static_cast<uint64_t>(lane) + 1);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 4 +
                    __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
       i < p.shard_n; i += nwaves) {
  const uint64_t kp = ctr_key(kit, p.shard_lo + i);
  double xv[CHUNKS][2], vv[CHUNKS][2], pb[CHUNKS][2];
  double *row = p.pos + i * D;
  load_row_stream<CHUNKS, VEC>(row, D, p.zero, xv);
  if (TYPE == NLSG_PSO_VANILLA) {
    load_row_stream<CHUNKS, VEC>(p.vel + i * D, D, p.zero, vv);
    load_row_stream<CHUNKS, VEC>(p.pbest_pos + i * D, D, p.zero, pb);
  }
  const double old_pbest = p.pbest_val[i];

  // draws 2e and 2e+1 of element e = 128 c + 2 lane + k: ctr_key(kp, j) = mix64(kp + G (j + 1))
  // with j + 1 = (4 lane + 1) + (256 c + 2 k [+ 1]) -- one 64-bit multiply per wave, the rest
  // are compile-time constants
  const uint64_t kp_lane = kp + lane_off;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      // element e = 128 c + 2 lane + k. Accelerated: one draw (slot 2e) feeds both uniforms of the
      // element's normal variate;
      // Vanilla: r_p and r_g are draws 2e and 2e + 1
      const uint64_t z1 = mix64(kp_lane + kGolden * static_cast<uint64_t>(256 * c + 2 * k));
      const double u1 = u01(z1);
      const double u2 = TYPE == NLSG_PSO_ACCELERATED
                            ? u01_low32(z1)
                            : u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(256 * c + 2 * k + 1)));
      double pnew;
      if (TYPE == NLSG_PSO_ACCELERATED) {
        // rnorm (2479-2485): sqrt(-2 log u1) * cos(2 pi_ u2), pi_ = 3.141593
        const double rn = det_rnorm(z1, rn_tab);
        pnew = inertia * rn + (1 - p.cog) * xv[c][k] + p.soc * gb[c][k];  // :2693-2697
      } else {
        // intended Vanilla update (B7 repaired): pbest[j] - pos, gbest[j] - pos
        vv[c][k] = (inertia * vv[c][k]) + p.cog * u1 * (pb[c][k] - xv[c][k]) +
                   p.soc * u2 * (gb[c][k] - xv[c][k]);
        pnew = xv[c][k] + vv[c][k];  // :2683
      }
      xv[c][k] = pnew;
    }
  }
  // thresholds (:2701-2715) and the zeros past the row's end, each behind ONE wave-uniform branch
  // per particle: written per element the compiler turned both into compare-and-select pairs that
  // every element paid for (10 of the 132 vector instructions per variate) although an unbounded
  // run never thresholds and a row that fills its chunks has no lanes past its end. The empty asm
  // keeps the blocks from being if-converted back.
  if (p.bounded) {
    asm volatile("");
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        double pnew = xv[c][k];
        pnew = pnew < lo[c][k] ? lo[c][k] : pnew;
        pnew = pnew > hi[c][k] ? hi[c][k] : pnew;
        xv[c][k] = pnew;
      }
  }
  if (D != 128u * CHUNKS) {
    asm volatile("");
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        xv[c][k] = (e < D) ? xv[c][k] : 0.0;
      }
  }
  const double f = p.fmul * wave_objective<OBJ, CHUNKS>(xv, D);
  store_row_stream<CHUNKS, VEC>(row, D, xv);
  if (TYPE == NLSG_PSO_VANILLA) store_row_stream<CHUNKS, VEC>(p.vel + i * D, D, vv);
  const bool better = f < old_pbest;  // :2733-2735
  if (TYPE == NLSG_PSO_VANILLA && better) store_row_stream<CHUNKS, VEC>(p.pbest_pos + i * D, D, xv);
  if (lane == 0) {
    p.cur_val[i] = f;
    if (better) p.pbest_val[i] = f;
  }
  }  // particles of this wave
}

// The same move for particles of at most 64 coordinates: 64 / G particles per wave, one per group
// of G lanes (G = 4, 8, 16, 32 for D <= 8, 16, 32, 64; lane g of a group holds coordinates 2g,
// 2g + 1). The particle's key lives in vector registers (it differs from group to group) and
// group_objective scores the new position with the bits of the full-wave tree, so a swarm's
// history does not depend on the packing.
template <int OBJ, int G, int TYPE>
__global__ __launch_bounds__(256) void pso_move_groups_kernel(PsoParams p, int timing,
                                                              uint64_t iter_ovr) {
  constexpr int P = 64 / G;
  const PsoState *__restrict__ st = p.state;
  if (!timing && st->done) return;
  // the logarithm table of det_rnorm (Accelerated only; a workgroup-wide step, so before any
  // wave leaves)
  __shared__ double rn_tab[TYPE == NLSG_PSO_ACCELERATED ? kRnormTabDoubles : 1];
  if (TYPE == NLSG_PSO_ACCELERATED) rnorm_table_to_lds(rn_tab);
  const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * 4 +
                        __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (wave * P >= p.shard_n) return;
  const int lane = lane_id(), g = lane & (G - 1), gi = lane / G;
  const bool live = wave * P + gi < p.shard_n;
  const uint64_t i = live ? wave * P + gi : wave * P;  // idle groups shadow a live particle
  const uint64_t D = p.D;
  const uint64_t iter = timing ? iter_ovr : st->iter;
  const uint64_t kit = first64(ctr_key(on_valu(p.seed), iter + 1));
  const uint64_t kp = ctr_key(kit, p.shard_lo + i);
  const uint32_t j0 = 2 * g, j1 = 2 * g + 1;
  const bool in[2] = {j0 < D, j1 < D};
  const uint32_t d32 = static_cast<uint32_t>(D);
  const uint64_t off = static_cast<uint64_t>(static_cast<uint32_t>(i)) * d32;
  double *row = p.pos + off;
  auto load2 = [&](const double *rp, bool on, double (&v)[2]) {
    v[0] = (on && in[0]) ? rp[j0] : 0.0;
    v[1] = (on && in[1]) ? rp[j1] : 0.0;
  };
  double xv[2], gb[2], lo[2], hi[2], vv[2], pb[2];
  load2(row, true, xv);
  load2(p.gbest_x, true, gb);
  load2(p.lower, p.bounded != 0, lo);
  load2(p.upper, p.bounded != 0, hi);
  load2(p.vel + off, TYPE == NLSG_PSO_VANILLA, vv);
  load2(p.pbest_pos + off, TYPE == NLSG_PSO_VANILLA, pb);
  const double old_pbest = p.pbest_val[i];
  double inertia = p.inertia;
  if (TYPE == NLSG_PSO_ACCELERATED)  // :2613 inertia = pow(init_inertia, iter)
    inertia = pso_inertia_at(p, iter);
  // draws 2e and 2e+1 of element e = 2g + k: mix64(kp + G64 (2e + 1 [+ 1])), 2e + 1 = 4g + 2k + 1
  const uint64_t kp_lane = kp + kGolden * (4 * static_cast<uint64_t>(g) + 1);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint64_t z1 = mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k));
    const double u1 = u01(z1);
    const double u2 = TYPE == NLSG_PSO_ACCELERATED
                          ? u01_low32(z1)
                          : u01(mix64(kp_lane + kGolden * static_cast<uint64_t>(2 * k + 1)));
    double pnew;
    if (TYPE == NLSG_PSO_ACCELERATED) {
      const double rn = det_rnorm(z1, rn_tab);  // rnorm, :2479-2485
      pnew = inertia * rn + (1 - p.cog) * xv[k] + p.soc * gb[k];  // :2693-2697
    } else {
      vv[k] = (inertia * vv[k]) + p.cog * u1 * (pb[k] - xv[k]) + p.soc * u2 * (gb[k] - xv[k]);
      pnew = xv[k] + vv[k];  // :2683
    }
    if (p.bounded) {  // :2701-2715
      pnew = pnew < lo[k] ? lo[k] : pnew;
      pnew = pnew > hi[k] ? hi[k] : pnew;
    }
    xv[k] = in[k] ? pnew : 0.0;
  }
  const double f = p.fmul * group_objective<OBJ, G>(xv[0], xv[1], D);
  const bool better = f < old_pbest;  // :2733-2735
  if (live) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t j = 2 * g + k;
      if (in[k]) {
        row[j] = xv[k];
        if (TYPE == NLSG_PSO_VANILLA) {
          p.vel[off + j] = vv[k];
          if (better) p.pbest_pos[off + j] = xv[k];
        }
      }
    }
    if (g == 0) {
      p.cur_val[i] = f;
      if (better) p.pbest_val[i] = f;
    }
  }
}

// ---- particles longer than a wave's registers hold (D > 1024; the reference has no limit) --------
// One wave per particle, rows taken in segments of 1024 coordinates, updated in place; the
// objective accumulates from segment to segment in the whole-row order. Vanilla: the personal-best
// row is copied in a second pass once the new value is known to be better.
template <int OBJ, bool VEC>
__global__ __launch_bounds__(256) void pso_init_long_kernel(PsoParams p) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (i >= p.shard_n) return;
  const int lane = lane_id();
  const uint64_t D = p.D;
  const uint64_t kp = ctr_key(ctr_key(p.seed, 0), p.shard_lo + i);
  auto pos_at = [&](uint64_t e) {  // :2645-2648
    return e < D ? p.lower[e] + ((p.upper[e] - p.lower[e]) * u01(ctr_key(kp, 2 * e))) : 0.0;
  };
  double acc = 0.0;
  for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
    double lo[kSeg][2], hi[kSeg][2], xv[kSeg][2], vv[kSeg][2];
    load_segment<VEC>(p.lower, e_base, D, p.zero, lo);
    load_segment<VEC>(p.upper, e_base, D, p.zero, hi);
#pragma unroll
    for (int c = 0; c < kSeg; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = e_base + static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        const double temp = fabs(hi[c][k] - lo[c][k]);
        xv[c][k] = lo[c][k] + ((hi[c][k] - lo[c][k]) * u01(ctr_key(kp, 2 * e)));
        vv[c][k] = -temp + (u01(ctr_key(kp, 2 * e + 1)) * temp);
        if (e >= D) xv[c][k] = 0.0;
      }
    store_segment<VEC>(p.pos + i * D, e_base, D, xv);
    if (p.type == NLSG_PSO_VANILLA) {
      store_segment<VEC>(p.vel + i * D, e_base, D, vv);
      store_segment<VEC>(p.pbest_pos + i * D, e_base, D, xv);
    }
    objective_accumulate<OBJ, kSeg>(acc, xv, e_base, D, pos_at(e_base + 128 * kSeg));
  }
  const double f = p.fmul * objective_finish<OBJ>(acc, D);
  if (lane == 0) {
    p.cur_val[i] = f;
    p.pbest_val[i] = f;
  }
}

template <int OBJ, bool VEC, int TYPE>
__global__ __launch_bounds__(256) void pso_move_long_kernel(PsoParams p, int timing, uint64_t iter_ovr) {
  const PsoState *__restrict__ st = p.state;
  if (!timing && st->done) return;
  // the logarithm table of det_rnorm (Accelerated only; a workgroup-wide step, so before any
  // wave leaves)
  __shared__ double rn_tab[TYPE == NLSG_PSO_ACCELERATED ? kRnormTabDoubles : 1];
  if (TYPE == NLSG_PSO_ACCELERATED) rnorm_table_to_lds(rn_tab);
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (i >= p.shard_n) return;
  const int lane = lane_id();
  const uint64_t D = p.D;
  const uint64_t iter = timing ? iter_ovr : st->iter;
  const uint64_t kp = ctr_key(ctr_key(p.seed, iter + 1), p.shard_lo + i);
  double *row = p.pos + i * D, *vrow = p.vel + i * D, *brow = p.pbest_pos + i * D;
  const double old_pbest = p.pbest_val[i];
  double inertia = p.inertia;
  if (TYPE == NLSG_PSO_ACCELERATED) inertia = pso_inertia_at(p, iter);
  // one coordinate of the new position, the same arithmetic as the lane code below (wave-uniform)
  auto new_at = [&](uint64_t e) {
    if (e >= D) return 0.0;
    const uint64_t z1 = ctr_key(kp, 2 * e);
    const double u1 = u01(z1);
    double pnew;
    if (TYPE == NLSG_PSO_ACCELERATED) {
      const double rn = det_rnorm(z1, rn_tab);
      pnew = inertia * rn + (1 - p.cog) * row[e] + p.soc * p.gbest_x[e];
    } else {
      const double u2 = u01(ctr_key(kp, 2 * e + 1));
      const double v = (inertia * vrow[e]) + p.cog * u1 * (brow[e] - row[e]) + p.soc * u2 * (p.gbest_x[e] - row[e]);
      pnew = row[e] + v;
    }
    if (p.bounded) {
      pnew = pnew < p.lower[e] ? p.lower[e] : pnew;
      pnew = pnew > p.upper[e] ? p.upper[e] : pnew;
    }
    return pnew;
  };
  const uint64_t kp_lane = kp + kGolden * (4 * static_cast<uint64_t>(lane) + 1);
  double acc = 0.0;
  for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
    // the first coordinate of the NEXT segment, from its old position: before this segment's store
    const double next_first = new_at(e_base + 128 * kSeg);
    double xv[kSeg][2], gb[kSeg][2], lo[kSeg][2], hi[kSeg][2], vv[kSeg][2], pb[kSeg][2];
    load_segment<VEC, true>(row, e_base, D, p.zero, xv);
    load_segment<VEC>(p.gbest_x, e_base, D, p.zero, gb);
    load_segment<VEC>(p.lower, e_base, p.bounded ? D : 0, p.zero, lo);
    load_segment<VEC>(p.upper, e_base, p.bounded ? D : 0, p.zero, hi);
    if (TYPE == NLSG_PSO_VANILLA) {
      load_segment<VEC, true>(vrow, e_base, D, p.zero, vv);
      load_segment<VEC, true>(brow, e_base, D, p.zero, pb);
    }
    const uint64_t kseg = kp_lane + kGolden * (2 * e_base);  // draws 2e, 2e + 1 of element e
#pragma unroll
    for (int c = 0; c < kSeg; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = e_base + static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        const uint64_t z1 = mix64(kseg + kGolden * static_cast<uint64_t>(256 * c + 2 * k));
        const double u1 = u01(z1);
        const double u2 = TYPE == NLSG_PSO_ACCELERATED
                              ? u01_low32(z1)
                              : u01(mix64(kseg + kGolden * static_cast<uint64_t>(256 * c + 2 * k + 1)));
        double pnew;
        if (TYPE == NLSG_PSO_ACCELERATED) {
          const double rn = det_rnorm(z1, rn_tab);
          pnew = inertia * rn + (1 - p.cog) * xv[c][k] + p.soc * gb[c][k];
        } else {
          vv[c][k] = (inertia * vv[c][k]) + p.cog * u1 * (pb[c][k] - xv[c][k]) +
                     p.soc * u2 * (gb[c][k] - xv[c][k]);
          pnew = xv[c][k] + vv[c][k];
        }
        if (p.bounded) {
          pnew = pnew < lo[c][k] ? lo[c][k] : pnew;
          pnew = pnew > hi[c][k] ? hi[c][k] : pnew;
        }
        xv[c][k] = (e < D) ? pnew : 0.0;
      }
    store_segment<VEC, true>(row, e_base, D, xv);
    if (TYPE == NLSG_PSO_VANILLA) store_segment<VEC, true>(vrow, e_base, D, vv);
    objective_accumulate<OBJ, kSeg>(acc, xv, e_base, D, next_first);
  }
  const double f = p.fmul * objective_finish<OBJ>(acc, D);
  const bool better = f < old_pbest;  // :2733-2735
  if (TYPE == NLSG_PSO_VANILLA && better)
    for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
      double xv[kSeg][2];
      load_segment<VEC>(row, e_base, D, p.zero, xv);
      store_segment<VEC, true>(brow, e_base, D, xv);
    }
  if (lane == 0) {
    p.cur_val[i] = f;
    if (better) p.pbest_val[i] = f;
  }
}

// First level of update_best_positions' scan (min / first argmin of the last
// evaluation) and of std_err(particle_best_values)'s first pass.
__global__ __launch_bounds__(256) void pso_scan_partial_kernel(PsoParams p) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  if (p.state->done) return;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  double acc = 0.0;
  double bv = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < p.shard_n; i += 256) {
    acc = acc + p.pbest_val[i];
    argmin_combine(bv, bi, p.cur_val[i], i);
  }
  const double total = block_tree_256(acc, red);
  block_argmin_256(bv, bi, mv, mi);
  if (threadIdx.x == 0) {
    p.part[blockIdx.x].sum = total;
    p.part[blockIdx.x].minv = bv;
    p.part[blockIdx.x].mini = bi;
  }
}

__global__ __launch_bounds__(256) void pso_var_partial_kernel(PsoParams p, const double *mean_ptr) {
  __shared__ double red[4];
  if (p.state->done) return;
  const double mean = *mean_ptr;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  double acc = 0.0;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < p.shard_n; i += 256) {
    const double d = p.pbest_val[i] - mean;
    acc = acc + d * d;
  }
  const double total = block_tree_256(acc, red);
  if (threadIdx.x == 0) p.part[blockIdx.x].m2 = total;
}

__device__ inline void pso_apply_pending(PsoState *st) {
  if (st->pending) {
    st->iter += 1;
    st->pending = 0;
  }
}

// shard minimum of the last evaluation (first occurrence); thread 0 gets the result
__device__ inline void pso_shard_best(const PsoParams &p, double *mv, uint64_t *mi, double &bv,
                                      uint64_t &bi) {
  bv = __builtin_inf();
  bi = ~0ull;
  for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256)
    argmin_combine(bv, bi, p.part[j].minv, p.part[j].mini);
  block_argmin_256(bv, bi, mv, mi);
}

// update_best_positions' bookkeeping (2727-2740) + stop tests (2599-2600); thread 0
__device__ inline bool pso_finish_turn(PsoState *st, const PsoParams &p, bool have, double bv,
                                       uint64_t gi, double se) {
  const bool update_happened = have && bv < st->gbest_val;  // strict '<' (:2724)
  if (update_happened) {
    st->gbest_val = bv;
    st->gbest_idx = gi;
  }
  st->fevals += p.n;  // :2734
  st->val_no_change = update_happened ? 0 : st->val_no_change + 1;  // :2740, B9 repaired
  st->std_err = se;
  if (st->iter >= p.max_iter || st->val_no_change >= p.best_val_no_change ||
      (p.eps > 0 && se < p.eps)) {
    st->done = 1;
  } else {
    st->pending = 1;
  }
  return update_happened;
}

// One GPU, eps <= 0: whole head of a turn in one single-block launch.
__global__ __launch_bounds__(256) void pso_head_kernel(PsoParams p) {
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  __shared__ uint64_t s_row;
  __shared__ int s_copy;
  PsoState *st = p.state;
  if (st->done) return;
  if (threadIdx.x == 0) pso_apply_pending(st);
  __syncthreads();
  double bv;
  uint64_t bi;
  pso_shard_best(p, mv, mi, bv, bi);
  if (threadIdx.x == 0) {
    const bool have = bi != ~0ull;
    const bool upd = pso_finish_turn(st, p, have, bv, p.shard_lo + bi, __builtin_nan(""));
    s_row = bi;
    s_copy = upd ? 1 : 0;
  }
  __syncthreads();
  if (!s_copy) return;
  const double *row = p.pos + s_row * p.D;  // swarm_best_position = positions[best], :2737
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.gbest_x[d] = row[d];
}

// One GPU, eps <= 0: the tile scan and the head in ONE launch (a dependent launch costs ~5 us on
// this platform): every tile block publishes its minimum, the block that arrives last finishes
// the turn (same arithmetic as pso_scan_partial_kernel + pso_head_kernel: min / first index).
__global__ __launch_bounds__(256) void pso_scan_head_kernel(PsoParams p) {
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  __shared__ uint64_t s_row;
  __shared__ int s_copy, s_last;
  PsoState *st = p.state;
  if (st->done) return;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  double bv = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < p.shard_n; i += 256)
    argmin_combine(bv, bi, p.cur_val[i], i);
  block_argmin_256(bv, bi, mv, mi);
  if (threadIdx.x == 0) {
    sc1_store(&p.part[blockIdx.x].minv, bv);
    __hip_atomic_store(&p.part[blockIdx.x].mini, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = take_ticket(p.ticket, p.ntiles) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) pso_apply_pending(st);
  bv = __builtin_inf();
  bi = ~0ull;
  for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256)
    argmin_combine(bv, bi, sc1_load(&p.part[j].minv),
                   __hip_atomic_load(&p.part[j].mini, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  block_argmin_256(bv, bi, mv, mi);
  if (threadIdx.x == 0) {
    const bool have = bi != ~0ull;
    const bool upd = pso_finish_turn(st, p, have, bv, p.shard_lo + bi, __builtin_nan(""));
    s_row = bi;
    s_copy = upd ? 1 : 0;
  }
  __syncthreads();
  if (!s_copy) return;
  const double *row = p.pos + s_row * p.D;  // swarm_best_position = positions[best], :2737
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.gbest_x[d] = row[d];
}

// ---- eps > 0 and the sharded path ----------------------------------------------
__global__ __launch_bounds__(256) void pso_local_kernel(PsoParams p, ShardLocal *loc, double *rec) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  __shared__ uint64_t s_bi;
  PsoState *st = p.state;
  if (st->done) return;
  if (threadIdx.x == 0) pso_apply_pending(st);
  __syncthreads();
  double bv;
  uint64_t bi;
  pso_shard_best(p, mv, mi, bv, bi);
  double total = 0.0;
  if (p.eps > 0) {
    double acc = 0.0;
    for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256) acc = acc + p.part[j].sum;
    total = block_tree_256(acc, red);
  }
  if (threadIdx.x == 0) {
    const bool valid = bi != ~0ull;
    loc->sum = total;
    loc->mean = total / static_cast<double>(p.shard_n);
    loc->minv = bv;
    loc->mini = valid ? p.shard_lo + bi : 0;
    loc->m2 = 0.0;
    loc->valid = valid ? 1.0 : 0.0;
    if (rec != nullptr) {
      rec[0] = bv;
      rec[1] = __longlong_as_double(static_cast<long long>(loc->mini));
      rec[2] = total;
      rec[3] = 0.0;
      rec[4] = loc->valid;
    }
    s_bi = bi;
  }
  if (rec == nullptr) return;
  __syncthreads();
  const bool valid = s_bi != ~0ull;
  const double *row = p.pos + (valid ? s_bi : 0) * p.D;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) rec[kRecHeader + d] = valid ? row[d] : 0.0;
}

__global__ __launch_bounds__(256) void pso_var_local_kernel(PsoParams p, ShardLocal *loc) {
  __shared__ double red[4];
  if (p.state->done) return;
  double acc = 0.0;
  for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256) acc = acc + p.part[j].m2;
  const double total = block_tree_256(acc, red);
  if (threadIdx.x == 0) loc->m2 = total;
}

__global__ __launch_bounds__(256) void pso_pack_record_kernel(PsoParams p, const ShardLocal *loc,
                                                            double *rec) {
  if (p.state->done) return;
  const bool valid = loc->valid == 1.0;
  if (threadIdx.x == 0) {
    rec[0] = loc->minv;
    rec[1] = __longlong_as_double(static_cast<long long>(loc->mini));
    rec[2] = loc->sum;
    rec[3] = loc->m2;
    rec[4] = loc->valid;
  }
  const double *row = p.pos + (valid ? (loc->mini - p.shard_lo) : 0) * p.D;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) rec[kRecHeader + d] = valid ? row[d] : 0.0;
}

__global__ __launch_bounds__(256) void pso_finalize_kernel(PsoParams p, const double *recs,
                                                         int32_t world, uint64_t rec_stride) {
  __shared__ int s_win, s_copy;
  PsoState *st = p.state;
  if (st->done) return;
  if (threadIdx.x == 0) {
    int win = -1;
    double bv = __builtin_inf();
    uint64_t bi = 0;
    for (int r = 0; r < world; r++) {
      const double *rec = recs + static_cast<uint64_t>(r) * rec_stride;
      if (rec[4] != 1.0) continue;
      const uint64_t i = static_cast<uint64_t>(__double_as_longlong(rec[1]));
      if (win < 0 || rec[0] < bv || (rec[0] == bv && i < bi)) {  // first occurrence wins
        bv = rec[0];
        bi = i;
        win = r;
      }
    }
    double se = __builtin_nan("");
    if (p.eps > 0) {  // std_err(particle_best_values), :2601; shards merged in rank order
      const double n_r = static_cast<double>(p.shard_n);
      double tot = 0.0;
      for (int r = 0; r < world; r++) tot = tot + recs[static_cast<uint64_t>(r) * rec_stride + 2];
      const double gmean = tot / static_cast<double>(p.n);
      double m2 = 0.0;
      for (int r = 0; r < world; r++) {
        const double *rec = recs + static_cast<uint64_t>(r) * rec_stride;
        double term = rec[3];
        if (world > 1) {
          const double dm = rec[2] / n_r - gmean;
          term = term + n_r * (dm * dm);
        }
        m2 = m2 + term;
      }
      se = sqrt(m2 / static_cast<double>(p.n - 1));
    }
    const bool upd = pso_finish_turn(st, p, win >= 0, bv, bi, se);
    s_win = win;
    s_copy = upd ? 1 : 0;
  }
  __syncthreads();
  if (!s_copy) return;
  const double *src = recs + static_cast<uint64_t>(s_win) * rec_stride + kRecHeader;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.gbest_x[d] = src[d];
}

__global__ void pso_settle_kernel(PsoParams p) { pso_apply_pending(p.state); }

}  // namespace nlsg
