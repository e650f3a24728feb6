// nlsolver_amd/csrc/nlsg_de_kernels.h — gfx950 kernels of the DE engine.
//
// Replaces the loops of DE::solve (nlsolver.h:2414-2476):
//   de_init_kernel        init_agents + initial scoring      (2315-2323, 2423-2425)
//   de_scan_partial/final best scan, no-change counter, stop (2429-2447) + std_err (2037-2052)
//   de_generation_kernel  generate_indices + propose_new_agent + f() + selection (2449-2472)
//
// Data layout in HBM: population row-major [shard_n][D] fp64, two buffers
// (synchronous generation: donors are read from `cur`, survivors written to
// `nxt`); scores [shard_n] fp64 updated in place by the owning wave.
// Mapping: one wave64 per agent; lane l holds elements c*128 + 2l, +1 of each
// 128-element chunk c, so every wave-level load/store is one contiguous 1 KiB
// burst (global_load_dwordx4 / global_store_dwordx4) when D is even.
#pragma once

#include "nlsg_common.h"

namespace nlsg {

constexpr int kDeMaxTries = 64;      // bounded donor rejection loop
constexpr int kTile = 1024;          // reduction tile
constexpr int kTraceWords = 5;       // r1, r2, r3, jrand, accept

// Device-resident solver state (one per engine).
struct DeState {
  uint64_t best_id;        // global index of the incumbent best
  double best_f;           // its score
  uint64_t iter;           // completed generations
  uint64_t val_no_change;  // nlsolver.h:2439
  uint64_t fcalls;
  double std_err;
  int32_t done;
  int32_t parity;          // population buffer holding the current generation
  int32_t pending;         // a generation ran since the last scan (iter++ due)
  int32_t pad;
};

struct DeParams {
  double *buf[2];      // population ping-pong
  double *scores;      // [shard_n]
  double *best_x;      // [D] row of the incumbent best (valid after a scan)
  uint64_t *trace;     // [shard_n*5] or nullptr
  DeState *state;
  uint64_t pop, D, shard_lo, shard_n;
  double CR, F, eps, fmul;
  uint64_t max_iter, best_val_no_change, seed;
  int32_t strategy;
  int32_t vec;         // rows are 16-byte aligned (D even)
};

// ---- row access ------------------------------------------------------------
template <int CHUNKS>
__device__ inline void load_row(const double *__restrict__ row, uint64_t D, int vec,
                                double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (vec) {
      double2 t = make_double2(0.0, 0.0);
      if (e0 < D) t = *reinterpret_cast<const double2 *>(row + e0);
      v[c][0] = t.x;
      v[c][1] = t.y;
    } else {
      v[c][0] = (e0 < D) ? row[e0] : 0.0;
      v[c][1] = (e0 + 1 < D) ? row[e0 + 1] : 0.0;
    }
  }
}
template <int CHUNKS>
__device__ inline void store_row(double *__restrict__ row, uint64_t D, int vec,
                                 const double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (vec) {
      if (e0 < D) *reinterpret_cast<double2 *>(row + e0) = make_double2(v[c][0], v[c][1]);
    } else {
      if (e0 < D) row[e0] = v[c][0];
      if (e0 + 1 < D) row[e0 + 1] = v[c][1];
    }
  }
}

// ---- generation 0 ----------------------------------------------------------
template <int OBJ, int CHUNKS>
__global__ __launch_bounds__(256) void de_init_kernel(DeParams p, const double *__restrict__ x0) {
  const uint64_t a = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (a >= p.shard_n) return;
  const int lane = lane_id();
  const uint64_t ka = ctr_key(ctr_key(p.seed, 0), p.shard_lo + a);
  double xv[CHUNKS][2];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint64_t e = e0 + k;
      // generate_sequence, nlsolver.h:2309: (u - 0.5) * offset[i]
      xv[c][k] = (e < p.D) ? (u01(ctr_key(ka, e)) - 0.5) * x0[e] : 0.0;
    }
  }
  store_row<CHUNKS>(p.buf[0] + a * p.D, p.D, p.vec, xv);
  const double f = p.fmul * wave_objective<OBJ, CHUNKS>(xv, p.D);  // :2423-2425
  if (lane == 0) p.scores[a] = f;
}

__global__ void de_reset_state_kernel(DeParams p) {
  DeState *s = p.state;
  s->best_id = 0;  // "best_id = 0" (:2428), a GLOBAL index on every rank
  s->best_f = 0.0;
  s->iter = 0;
  s->val_no_change = 0;
  s->fcalls = p.pop;
  s->std_err = __builtin_nan("");
  s->done = 0;
  s->parity = 0;
  s->pending = 0;
}

// ---- one generation ----------------------------------------------------------
template <int OBJ, int CHUNKS>
__global__ __launch_bounds__(256) void de_generation_kernel(DeParams p, int par_override,
                                                          uint64_t gen_override) {
  // par_override >= 0: timing mode (nlsg_de_time_generation_kernel) — buffer
  // parity and generation number come from the host instead of the state.
  const DeState *__restrict__ st = p.state;
  if (par_override < 0 && st->done) return;  // a stop test fired: the turn is a no-op
  const uint64_t a = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (a >= p.shard_n) return;
  const int lane = lane_id();
  const int par = par_override >= 0 ? par_override : st->parity;
  const uint64_t generation = par_override >= 0 ? gen_override : st->iter + 1;
  const double *__restrict__ cur = p.buf[par];
  double *__restrict__ nxt = p.buf[par ^ 1];
  const uint64_t D = p.D;
  const uint64_t ga = p.shard_lo + a;  // global agent id keys the RNG
  const uint64_t ka = ctr_key(ctr_key(p.seed, generation), ga);

  // generate_indices (nlsolver.h:2331-2355): three distinct donors != fixed,
  // by rejection, drawn inside this engine's shard. Wave-uniform (scalar) code.
  const uint64_t fixed = (p.strategy == NLSG_DE_RANDOM) ? ga : st->best_id;  // :2451-2457
  uint64_t r[3];
  int have = 0;
  for (int k = 0; k < kDeMaxTries && have < 3; k++) {
    const uint64_t cand = p.shard_lo + clamp_index(u01(ctr_key(ka, D + 1 + k)), p.shard_n);
    bool used = (cand == fixed);
    for (int j = 0; j < 3; j++) used |= (j < have && r[j] == cand);
    if (!used) {
      if (have == 0) r[0] = cand;
      else if (have == 1) r[1] = cand;
      else r[2] = cand;
      have++;
    }
  }
  for (uint64_t cand = p.shard_lo; have < 3; cand++) {  // fallback: lowest unused
    bool used = (cand == fixed);
    for (int j = 0; j < 3; j++) used |= (j < have && r[j] == cand);
    if (!used) {
      if (have == 0) r[0] = cand;
      else if (have == 1) r[1] = cand;
      else r[2] = cand;
      have++;
    }
  }
  const uint64_t jrand = clamp_index(u01(ctr_key(ka, D)), D);  // :2364

  // rows: own (selection survivor), base (non-crossed coordinates), 3 donors
  double own[CHUNKS][2], d1[CHUNKS][2], d2[CHUNKS][2], d3[CHUNKS][2];
  load_row<CHUNKS>(cur + a * D, D, p.vec, own);
  load_row<CHUNKS>(cur + (r[0] - p.shard_lo) * D, D, p.vec, d1);
  load_row<CHUNKS>(cur + (r[1] - p.shard_lo) * D, D, p.vec, d2);
  load_row<CHUNKS>(cur + (r[2] - p.shard_lo) * D, D, p.vec, d3);
  const double old_score = p.scores[a];

  // propose_new_agent (nlsolver.h:2357-2375)
  double trial[CHUNKS][2];
  if (p.strategy == NLSG_DE_RANDOM) {
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        const double u = u01(ctr_key(ka, e));
        const double mut = d1[c][k] + p.F * (d2[c][k] - d3[c][k]);
        trial[c][k] = (u < p.CR || e == jrand) ? mut : own[c][k];
      }
    }
  } else {
    double base[CHUNKS][2];
    load_row<CHUNKS>(p.best_x, D, p.vec, base);  // row of best_id (L2-resident)
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        const double u = u01(ctr_key(ka, e));
        const double mut = d1[c][k] + p.F * (d2[c][k] - d3[c][k]);
        trial[c][k] = (u < p.CR || e == jrand) ? mut : base[c][k];
      }
    }
  }
  // (elements >= D are 0 in every loaded row, hence 0 in the trial as well)
  const double score = p.fmul * wave_objective<OBJ, CHUNKS>(trial, D);  // :2463
  const bool accept = score < old_score;                                 // :2466 (NaN -> keep)
  double *out = nxt + a * D;
  if (accept) {
    store_row<CHUNKS>(out, D, p.vec, trial);
    if (lane == 0) p.scores[a] = score;
  } else {
    store_row<CHUNKS>(out, D, p.vec, own);
  }
  if (p.trace != nullptr && lane == 0) {
    uint64_t *t = p.trace + a * kTraceWords;
    t[0] = r[0];
    t[1] = r[1];
    t[2] = r[2];
    t[3] = jrand;
    t[4] = accept ? 1u : 0u;
  }
}

// ---- best scan + std_err + stop tests ------------------------------------------
// Per tile of 1024 scores: block-tree sum, min value and first index of it.
struct TilePartial {
  double sum;
  double minv;
  uint64_t mini;  // shard-local index
  double m2;      // sum of squared deviations from the mean (second pass)
};

__device__ inline void argmin_combine(double &v, uint64_t &i, double ov, uint64_t oi) {
  // lower value wins; equal values keep the lower index; NaN never wins
  if (ov < v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

__global__ __launch_bounds__(256) void de_scan_partial_kernel(DeParams p, TilePartial *part) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  const DeState *st = p.state;
  if (st->done) return;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  const uint64_t n = p.shard_n;
  double acc = 0.0;
  double bv = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < n; i += 256) {
    const double s = p.scores[i];
    acc = acc + s;
    argmin_combine(bv, bi, s, i);
  }
  const double total = block_tree_256(acc, red);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(bv, off, 64);
    const uint64_t oi = __shfl_xor(bi, off, 64);
    argmin_combine(bv, bi, ov, oi);
  }
  const int wid = static_cast<int>(threadIdx.x) >> 6;
  if (lane_id() == 0) {
    mv[wid] = bv;
    mi[wid] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) argmin_combine(bv, bi, mv[w], mi[w]);
    part[blockIdx.x].sum = total;
    part[blockIdx.x].minv = bv;
    part[blockIdx.x].mini = bi;
  }
}

// Second pass of std_err (nlsolver.h:2046-2049) with the mean of pass one.
__global__ __launch_bounds__(256) void de_var_partial_kernel(DeParams p, TilePartial *part,
                                                           const double *mean_ptr) {
  __shared__ double red[4];
  const DeState *st = p.state;
  if (st->done) return;
  const double mean = *mean_ptr;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  double acc = 0.0;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < p.shard_n; i += 256) {
    const double d = p.scores[i] - mean;
    acc = acc + d * d;
  }
  const double total = block_tree_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x].m2 = total;
}

// Local (per-shard) summary produced by the scan; consumed by the finaliser
// directly (one GPU) or exchanged between ranks (record).
struct DeLocal {
  double sum;      // tiled sum of the shard's scores
  double mean;     // sum / shard_n
  double minv;     // shard minimum
  uint64_t mini;   // GLOBAL index of its first occurrence (incumbent keeps ties)
  double m2;       // tiled sum of squared deviations from `mean`
};

// Single block. phase 0: applies the pending iter++ of the previous generation,
// reduces the tile partials to (sum, min, argmin) and the local mean.
__global__ __launch_bounds__(256) void de_scan_local_kernel(DeParams p, const TilePartial *part,
                                                          uint32_t ntiles, DeLocal *loc) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  DeState *st = p.state;
  if (st->done) return;
  double acc = 0.0;
  double bv = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint32_t j = threadIdx.x; j < ntiles; j += 256) {
    acc = acc + part[j].sum;
    argmin_combine(bv, bi, part[j].minv, part[j].mini);
  }
  const double total = block_tree_256(acc, red);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(bv, off, 64);
    const uint64_t oi = __shfl_xor(bi, off, 64);
    argmin_combine(bv, bi, ov, oi);
  }
  const int wid = static_cast<int>(threadIdx.x) >> 6;
  if (lane_id() == 0) {
    mv[wid] = bv;
    mi[wid] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) argmin_combine(bv, bi, mv[w], mi[w]);
    if (st->pending) {  // the generation enqueued before this scan has run
      st->iter += 1;
      st->fcalls += p.pop;
      st->parity ^= 1;
      st->pending = 0;
    }
    // incumbent keeps ties (strict '<' scan starting from best_id, :2432-2437)
    uint64_t gi = (bi == ~0ull) ? st->best_id : p.shard_lo + bi;
    const uint64_t inc = st->best_id;
    if (inc >= p.shard_lo && inc < p.shard_lo + p.shard_n) {
      const double inc_score = p.scores[inc - p.shard_lo];
      if (!(bv < inc_score)) {  // nobody is strictly better than the incumbent
        gi = inc;
        bv = inc_score;
      }
    }
    loc->sum = total;
    loc->mean = total / static_cast<double>(p.shard_n);  // :2044
    loc->minv = bv;
    loc->mini = gi;
    loc->m2 = 0.0;
  }
}

__global__ __launch_bounds__(256) void de_var_local_kernel(DeParams p, const TilePartial *part,
                                                         uint32_t ntiles, DeLocal *loc) {
  __shared__ double red[4];
  if (p.state->done) return;
  double acc = 0.0;
  for (uint32_t j = threadIdx.x; j < ntiles; j += 256) acc = acc + part[j].m2;
  const double total = block_tree_256(acc, red);
  if (threadIdx.x == 0) loc->m2 = total;
}

// Record exchanged between ranks (kRecHeader + D doubles):
//   [minv, mini(bits), sum, m2, valid, x_best[0..D)]
// `valid` is 1 when the record's row belongs to the sending shard (it is 0 only
// for a shard whose scores are all NaN and that does not own the incumbent).
constexpr int kRecHeader = 5;

__global__ __launch_bounds__(256) void de_pack_record_kernel(DeParams p, const DeLocal *loc,
                                                           double *rec) {
  const DeState *st = p.state;
  if (st->done) return;
  const uint64_t gi = loc->mini;
  const bool mine = gi >= p.shard_lo && gi < p.shard_lo + p.shard_n;
  if (threadIdx.x == 0) {
    rec[0] = loc->minv;
    rec[1] = __longlong_as_double(static_cast<long long>(gi));
    rec[2] = loc->sum;
    rec[3] = loc->m2;
    rec[4] = mine ? 1.0 : 0.0;
  }
  const double *row = p.buf[st->parity] + (mine ? (gi - p.shard_lo) : 0) * p.D;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) rec[kRecHeader + d] = mine ? row[d] : 0.0;
}

// Finaliser: picks the global best among `world` records (world == 1: the
// local record), applies the no-change counter and the stop tests
// (nlsolver.h:2439-2447), refreshes best_x. Single block.
__global__ __launch_bounds__(256) void de_finalize_kernel(DeParams p, const double *recs,
                                                        int32_t world, uint64_t rec_stride) {
  __shared__ int s_win;
  DeState *st = p.state;
  if (st->done) return;
  if (threadIdx.x == 0) {
    const uint64_t inc = st->best_id;
    int win = -1;
    double bv = __builtin_inf();
    uint64_t bi = inc;
    for (int r = 0; r < world; r++) {
      const double *rec = recs + static_cast<uint64_t>(r) * rec_stride;
      if (rec[4] != 1.0) continue;
      const double v = rec[0];
      const uint64_t i = static_cast<uint64_t>(__double_as_longlong(rec[1]));
      // lower value wins; on ties the incumbent wins, then the lower index
      const bool better =
          win < 0 || v < bv || (v == bv && bi != inc && (i == inc || i < bi));
      if (better) {
        bv = v;
        bi = i;
        win = r;
      }
    }
    // not_updated <=> best_id did not move: the strict '<' scan (:2431-2437) can
    // never return to the incumbent once it has left it.
    const bool not_updated = (bi == inc);
    st->val_no_change = not_updated ? st->val_no_change + 1 : 0;  // :2439
    st->best_id = bi;
    if (win >= 0) st->best_f = bv;
    // std_err over the global scores (only when it can decide: eps > 0)
    double se = __builtin_nan("");
    if (p.eps > 0) {
      // merge per-shard (n, sum, M2) in rank order; one shard == the two-pass
      // formula of nlsolver.h:2037-2052
      const double n_r = static_cast<double>(p.shard_n);
      double tot = 0.0;
      for (int r = 0; r < world; r++) tot = tot + recs[static_cast<uint64_t>(r) * rec_stride + 2];
      const double gmean = tot / static_cast<double>(p.pop);
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
      se = sqrt(m2 / static_cast<double>(p.pop - 1));  // :2050-2051
    }
    st->std_err = se;
    if (st->iter >= p.max_iter || st->val_no_change >= p.best_val_no_change ||
        (p.eps > 0 && se < p.eps)) {  // :2441-2443
      st->done = 1;
    } else {
      st->pending = 1;  // the generation enqueued right after this kernel will run
    }
    s_win = win;
  }
  __syncthreads();
  if (s_win < 0) return;  // no valid record: keep best_x
  const double *src = recs + static_cast<uint64_t>(s_win) * rec_stride + kRecHeader;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.best_x[d] = src[d];
}

// Applies a pending iter++ without scanning (used before reading the state).
__global__ void de_settle_kernel(DeParams p) {
  DeState *st = p.state;
  if (st->pending) {
    st->iter += 1;
    st->fcalls += p.pop;
    st->parity ^= 1;
    st->pending = 0;
  }
}

}  // namespace nlsg
