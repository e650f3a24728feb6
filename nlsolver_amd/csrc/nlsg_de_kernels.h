// nlsolver_amd/csrc/nlsg_de_kernels.h — gfx950 kernels of the DE engine.
//
// Replaces the loops of DE::solve (nlsolver.h:2414-2476):
//   de_init_kernel        init_agents + initial scoring      (2315-2323, 2423-2425)
//   de_generation_kernel  generate_indices + propose_new_agent + f() + selection (2449-2472)
//   de_scan_partial_kernel + de_head_kernel
//                         best scan, no-change counter, stop tests (2429-2447)
//   de_var_* kernels      second pass of std_err (2037-2052), only when eps > 0 can decide
//
// Data layout in HBM: population row-major [shard_n][D] fp64, two buffers
// (synchronous generation: donors are read from `cur`, survivors written to
// `nxt`); scores [shard_n] fp64 updated in place by the owning wave.
// Mapping: one wave64 per agent; lane l holds elements c*128 + 2l, +1 of each
// 128-element chunk c, so every wave-level load/store is one contiguous 1 KiB
// burst (global_load_dwordx4 / global_store_dwordx4) when D is even. One agent
// per wave and 4 agents per 256-thread block, dispatched dynamically: measured
// 10-13 % faster at pop = 2^20 than persistent grid-stride waves (same device,
// same run), which lose to load imbalance between CUs (DESIGN.md §DE kernel).
#pragma once

#include "nlsg_common.h"

namespace nlsg {

constexpr int kDeMaxTries = 64;       // bounded donor rejection loop
constexpr int kTraceWords = 5;        // r1, r2, r3, jrand, accept

// Device-resident solver state (one per engine).
struct DeState {
  uint64_t best_id;        // global index of the incumbent best
  double best_f;           // its score
  uint64_t iter;           // completed generations
  uint64_t val_no_change;  // nlsolver.h:2439
  uint64_t fcalls;
  double std_err;
  int32_t done;
  int32_t parity;          // population / score buffer holding the current generation
  int32_t pad[2];
};

struct DeParams {
  double *buf[2];      // population ping-pong
  double *scores[2];   // [shard_n] each, ping-pong with the population: a generation reads
                       // scores[src] and writes scores[src^1], so it never destroys the
                       // state it started from (it may run speculatively, see nlsg_de.hip)
  double *best_x;      // [D] row of the incumbent best (valid after a scan)
  uint64_t *trace;     // [shard_n*5] or nullptr
  DeState *state;
  TilePartial *part;   // [ntiles] written by de_scan_partial_kernel
  const double *zero;  // 16 bytes of zeros: source for lanes past the row end
  uint32_t ntiles;
  uint32_t pad0;
  uint64_t pop, D, shard_lo, shard_n;
  double CR, F, eps, fmul;
  uint64_t max_iter, best_val_no_change, seed;
  int32_t strategy;
  int32_t vec;         // rows are 16-byte aligned (D even)
};

// ---- generation 0 ----------------------------------------------------------
template <int OBJ, int CHUNKS, bool VEC>
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
  store_row<CHUNKS, VEC>(p.buf[0] + a * p.D, p.D, xv);
  const double f = p.fmul * wave_objective<OBJ, CHUNKS>(xv, p.D);  // :2423-2425
  if (lane == 0) p.scores[0][a] = f;
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
}

// ---- one generation ----------------------------------------------------------
// Everything a wave needs about one agent before it can build the trial: donor indices
// (wave-uniform), the rows and the old score. All loads of a fetch are issued back to back.
template <int CHUNKS>
struct DeAgent {
  uint64_t a, ka, r0, r1, r2, jrand;
  double own[CHUNKS][2], keep[CHUNKS][2], d1[CHUNKS][2], d2[CHUNKS][2], d3[CHUNKS][2];
  double old_score;
};

template <int CHUNKS, bool VEC>
__device__ inline void de_fetch_agent(const DeParams &p, const double *__restrict__ cur, int par,
                                      uint64_t kg, uint64_t best_id, uint64_t a, bool valid,
                                      DeAgent<CHUNKS> &c) {
  // !valid (second agent of a wave past the end of the shard): the loads are still issued but
  // every lane reads the 16 zero bytes, and the agent is not processed
  const uint64_t D = valid ? p.D : 0;
  const uint64_t ga = p.shard_lo + a;  // the global agent id keys the RNG
  const uint64_t ka = ctr_key(kg, ga);
  // generate_indices (nlsolver.h:2331-2355): three distinct donors != fixed,
  // by rejection, drawn inside this engine's shard. Wave-uniform (scalar) code.
  const uint64_t fixed = (p.strategy == NLSG_DE_RANDOM) ? ga : best_id;  // :2451-2457
  uint64_t r0 = ~0ull, r1 = ~0ull, r2 = ~0ull;
  int have = 0;
  for (int k = 0; k < kDeMaxTries && have < 3; k++) {
    const uint64_t cand = p.shard_lo + clamp_index(u01(ctr_key(ka, p.D + 1 + k)), p.shard_n);
    const bool used = (cand == fixed) || (have > 0 && cand == r0) || (have > 1 && cand == r1);
    if (!used) {
      if (have == 0) r0 = cand;
      else if (have == 1) r1 = cand;
      else r2 = cand;
      have++;
    }
  }
  for (uint64_t cand = p.shard_lo; have < 3; cand++) {  // fallback: lowest unused
    const bool used = (cand == fixed) || (have > 0 && cand == r0) || (have > 1 && cand == r1);
    if (!used) {
      if (have == 0) r0 = cand;
      else if (have == 1) r1 = cand;
      else r2 = cand;
      have++;
    }
  }
  c.a = a;
  c.ka = ka;
  c.r0 = r0;
  c.r1 = r1;
  c.r2 = r2;
  c.jrand = clamp_index(u01(ctr_key(ka, p.D)), p.D);  // :2364
  // rows: own (selection survivor; non-crossed coordinates for strategy random), 3 donors,
  // and for strategy best the row of best_id (L2-resident; strategy random reads the zero
  // pad instead so that the instruction stream does not depend on the strategy)
  load_row<CHUNKS, VEC>(cur + a * p.D, D, p.zero, c.own);
  load_row<CHUNKS, VEC>(cur + (r0 - p.shard_lo) * p.D, D, p.zero, c.d1);
  load_row<CHUNKS, VEC>(cur + (r1 - p.shard_lo) * p.D, D, p.zero, c.d2);
  load_row<CHUNKS, VEC>(cur + (r2 - p.shard_lo) * p.D, D, p.zero, c.d3);
  load_row<CHUNKS, VEC>(p.best_x, p.strategy == NLSG_DE_RANDOM ? 0 : D, p.zero, c.keep);
  c.old_score = *(valid ? p.scores[par] + a : p.zero);
}

template <int OBJ, int CHUNKS, bool VEC>
__device__ inline void de_process_agent(const DeParams &p, double *__restrict__ nxt, int par,
                                        const DeAgent<CHUNKS> &c) {
  const int lane = lane_id();
  const uint64_t D = p.D;
  const bool rnd = p.strategy == NLSG_DE_RANDOM;
  // propose_new_agent (nlsolver.h:2357-2375)
  double trial[CHUNKS][2];
#pragma unroll
  for (int ch = 0; ch < CHUNKS; ch++) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint64_t e = static_cast<uint64_t>(ch) * 128 + 2 * static_cast<uint64_t>(lane) + k;
      const double u = u01(ctr_key(c.ka, e));
      const double mut = c.d1[ch][k] + p.F * (c.d2[ch][k] - c.d3[ch][k]);
      trial[ch][k] = (u < p.CR || e == c.jrand) ? mut : (rnd ? c.own[ch][k] : c.keep[ch][k]);
    }
  }
  // (elements >= D are 0 in every loaded row, hence 0 in the trial as well)
  const double score = p.fmul * wave_objective<OBJ, CHUNKS>(trial, D);  // :2463
  const bool accept = score < c.old_score;                               // :2466 (NaN -> keep)
  double *out = nxt + c.a * D;
  if (accept) {
    store_row<CHUNKS, VEC>(out, D, trial);
  } else {
    store_row<CHUNKS, VEC>(out, D, c.own);
  }
  if (lane == 0) p.scores[par ^ 1][c.a] = accept ? score : c.old_score;
  if (p.trace != nullptr && lane == 0) {
    uint64_t *t = p.trace + c.a * kTraceWords;
    t[0] = c.r0;
    t[1] = c.r1;
    t[2] = c.r2;
    t[3] = c.jrand;
    t[4] = accept ? 1u : 0u;
  }
}

// One agent per wave. (Two agents per wave — ten gathers in flight — measured -7 % kernel time at
// pop = 65536 but +9 % at pop = 2^20 and only -2 % per turn; not kept.)
template <int OBJ, int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void de_generation_kernel(DeParams p, int par, uint64_t generation,
                                                          int ignore_done) {
  // `generation` (k+1) and the source buffer `par` (k & 1) come from the host: the k-th
  // turn's head may still be running when this kernel starts (speculative launch for
  // strategy random); the device state is only consulted for the stop flag, which is
  // final for every head older than that one.
  const DeState *__restrict__ st = p.state;
  if (!ignore_done && st->done) return;  // a stop test fired: the turn is a no-op
  const uint64_t a0 = static_cast<uint64_t>(blockIdx.x) * 4 +
                      __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (a0 >= p.shard_n) return;
  const double *__restrict__ cur = p.buf[par];
  double *__restrict__ nxt = p.buf[par ^ 1];
  const uint64_t kg = ctr_key(p.seed, generation);
  const uint64_t best_id = st->best_id;
  DeAgent<CHUNKS> A;
  de_fetch_agent<CHUNKS, VEC>(p, cur, par, kg, best_id, a0, true, A);
  de_process_agent<OBJ, CHUNKS, VEC>(p, nxt, par, A);
}

// ---- best scan + stop tests -------------------------------------------------
// Head number k looks at the population after k generations: buffer k & 1. It records
// that position in the state; a head that fires a stop test freezes the state there.
__device__ inline void head_position(DeState *st, const DeParams &p, uint64_t k) {
  st->iter = k;
  st->fcalls = p.pop * (k + 1);
  st->parity = static_cast<int32_t>(k & 1);
}

// First level of the best scan (and of std_err's first pass): one block per
// tile of kTile scores.
__global__ __launch_bounds__(256) void de_scan_partial_kernel(DeParams p, int par) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  if (p.state->done) return;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  double acc = 0.0;
  double bv = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < p.shard_n; i += 256) {
    const double sc = p.scores[par][i];
    acc = acc + sc;
    argmin_combine(bv, bi, sc, i);
  }
  const double total = block_tree_256(acc, red);
  block_argmin_256(bv, bi, mv, mi);
  if (threadIdx.x == 0) {
    p.part[blockIdx.x].sum = total;
    p.part[blockIdx.x].minv = bv;
    p.part[blockIdx.x].mini = bi;
  }
}

// Shard minimum with the reference's tie rule (strict '<' scan starting from the
// incumbent, nlsolver.h:2432-2437): the incumbent survives when nobody in the
// shard is strictly better. Returns (score, GLOBAL index, owned) in thread 0.
__device__ inline void shard_best(const DeParams &p, const DeState *st, int par, double *mv,
                                  uint64_t *mi, double &bv, uint64_t &gi, bool &mine) {
  bv = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256)
    argmin_combine(bv, bi, p.part[j].minv, p.part[j].mini);
  block_argmin_256(bv, bi, mv, mi);
  gi = 0;
  mine = false;
  if (threadIdx.x == 0) {
    const uint64_t inc = st->best_id;
    gi = (bi == ~0ull) ? inc : p.shard_lo + bi;
    if (inc >= p.shard_lo && inc < p.shard_lo + p.shard_n) {
      const double inc_score = p.scores[par][inc - p.shard_lo];
      if (!(bv < inc_score)) {
        gi = inc;
        bv = inc_score;
      }
    }
    mine = gi >= p.shard_lo && gi < p.shard_lo + p.shard_n;
  }
}

// Counters and stop tests shared by the two finalisers (thread 0 only).
__device__ inline void finish_turn(DeState *st, const DeParams &p, uint64_t bi, double bv,
                                   bool have_best, double se) {
  // not_updated <=> best_id did not move: the strict '<' scan (:2431-2437) can
  // never return to the incumbent once it has left it.
  const bool not_updated = (bi == st->best_id);
  st->val_no_change = not_updated ? st->val_no_change + 1 : 0;  // :2439
  st->best_id = bi;
  if (have_best) st->best_f = bv;
  st->std_err = se;
  if (st->iter >= p.max_iter || st->val_no_change >= p.best_val_no_change ||
      (p.eps > 0 && se < p.eps)) {  // :2441-2443
    st->done = 1;
  }
}

// One-GPU, eps <= 0 (std_err cannot decide): the whole head of a turn in one
// single-block launch.
__global__ __launch_bounds__(256) void de_head_kernel(DeParams p, uint64_t k) {
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  __shared__ uint64_t s_row;
  __shared__ int s_have;
  __shared__ int s_par;
  DeState *st = p.state;
  if (st->done) return;
  const int par = static_cast<int>(k & 1);
  if (threadIdx.x == 0) head_position(st, p, k);
  __syncthreads();
  double bv;
  uint64_t gi;
  bool mine;
  shard_best(p, st, par, mv, mi, bv, gi, mine);
  if (threadIdx.x == 0) {
    s_par = par;
    finish_turn(st, p, gi, bv, mine, __builtin_nan(""));
    s_row = gi - p.shard_lo;
    s_have = mine ? 1 : 0;
  }
  __syncthreads();
  if (!s_have) return;
  const double *row = p.buf[s_par] + s_row * p.D;  // x = agents[best_id], :2444
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.best_x[d] = row[d];
}

// ---- std_err (eps > 0) and the sharded path ------------------------------------
// Second pass of std_err (nlsolver.h:2046-2049) with the mean of pass one.
__global__ __launch_bounds__(256) void de_var_partial_kernel(DeParams p, const double *mean_ptr,
                                                           int par) {
  __shared__ double red[4];
  if (p.state->done) return;
  const double mean = *mean_ptr;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kTile;
  double acc = 0.0;
  for (uint64_t i = base + threadIdx.x; i < base + kTile && i < p.shard_n; i += 256) {
    const double d = p.scores[par][i] - mean;
    acc = acc + d * d;
  }
  const double total = block_tree_256(acc, red);
  if (threadIdx.x == 0) p.part[blockIdx.x].m2 = total;
}

// `rec` != nullptr (only when eps <= 0, i.e. no second std_err pass is needed):
// the exchange record is packed by this launch as well.
__global__ __launch_bounds__(256) void de_local_kernel(DeParams p, ShardLocal *loc, double *rec,
                                                     uint64_t k) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  __shared__ uint64_t s_gi;
  __shared__ int s_mine, s_par;
  DeState *st = p.state;
  if (st->done) return;
  const int par = static_cast<int>(k & 1);
  if (threadIdx.x == 0) head_position(st, p, k);
  __syncthreads();
  double bv;
  uint64_t gi;
  bool mine;
  shard_best(p, st, par, mv, mi, bv, gi, mine);
  double total = 0.0;
  if (p.eps > 0) {
    double acc = 0.0;
    for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256) acc = acc + p.part[j].sum;
    total = block_tree_256(acc, red);
  }
  if (threadIdx.x == 0) {
    loc->sum = total;
    loc->mean = total / static_cast<double>(p.shard_n);  // :2044
    loc->minv = bv;
    loc->mini = gi;
    loc->m2 = 0.0;
    loc->valid = mine ? 1.0 : 0.0;
    if (rec != nullptr) {
      rec[0] = bv;
      rec[1] = __longlong_as_double(static_cast<long long>(gi));
      rec[2] = total;
      rec[3] = 0.0;
      rec[4] = mine ? 1.0 : 0.0;
    }
    s_gi = gi;
    s_mine = mine ? 1 : 0;
    s_par = par;
  }
  if (rec == nullptr) return;
  __syncthreads();
  const double *row = p.buf[s_par] + (s_mine ? (s_gi - p.shard_lo) : 0) * p.D;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256)
    rec[kRecHeader + d] = s_mine ? row[d] : 0.0;
}

__global__ __launch_bounds__(256) void de_var_local_kernel(DeParams p, ShardLocal *loc) {
  __shared__ double red[4];
  if (p.state->done) return;
  double acc = 0.0;
  for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256) acc = acc + p.part[j].m2;
  const double total = block_tree_256(acc, red);
  if (threadIdx.x == 0) loc->m2 = total;
}

// Record exchanged between ranks (kRecHeader + D doubles):
//   [minv, mini(bits), sum, m2, valid, x_best[0..D)]
// `valid` is 1 when the record's row belongs to the sending shard (it is 0 only
// for a shard whose scores are all NaN and that does not own the incumbent).

__global__ __launch_bounds__(256) void de_pack_record_kernel(DeParams p, const ShardLocal *loc,
                                                           double *rec, int par) {
  const DeState *st = p.state;
  if (st->done) return;
  const uint64_t gi = loc->mini;
  const bool mine = loc->valid == 1.0;
  if (threadIdx.x == 0) {
    rec[0] = loc->minv;
    rec[1] = __longlong_as_double(static_cast<long long>(gi));
    rec[2] = loc->sum;
    rec[3] = loc->m2;
    rec[4] = loc->valid;
  }
  const double *row = p.buf[par] + (mine ? (gi - p.shard_lo) : 0) * p.D;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) rec[kRecHeader + d] = mine ? row[d] : 0.0;
}

// Finaliser over `world` records (world == 1: the local record): global best
// (lower value; on ties the incumbent, then the lower global index), counters,
// stop tests (nlsolver.h:2439-2447), best_x. Single block.
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
      const bool better =
          win < 0 || v < bv || (v == bv && bi != inc && (i == inc || i < bi));
      if (better) {
        bv = v;
        bi = i;
        win = r;
      }
    }
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
    finish_turn(st, p, bi, bv, win >= 0, se);
    s_win = win;
  }
  __syncthreads();
  if (s_win < 0) return;  // no valid record: keep best_x
  const double *src = recs + static_cast<uint64_t>(s_win) * rec_stride + kRecHeader;
  for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.best_x[d] = src[d];
}

// Before the host reads the state: unless a stop test fired, the engine stands after the
// k generations it has launched (the last head only saw k-1 of them).
__global__ void de_settle_kernel(DeParams p, uint64_t k) {
  if (!p.state->done) head_position(p.state, p, k);
}

}  // namespace nlsg
