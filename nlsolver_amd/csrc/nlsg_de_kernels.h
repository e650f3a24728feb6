// nlsolver_amd/csrc/nlsg_de_kernels.h — gfx950 kernels of the DE engine.
//
// Replaces the loops of DE::solve (nlsolver.h:2414-2476):
//   de_init_kernel        init_agents + initial scoring      (2315-2323, 2423-2425)
//   de_generation_kernel  generate_indices + propose_new_agent + f() + selection (2449-2472)
//   de_scan_head_kernel   best scan, std_err (2037-2052) when eps > 0 can decide, no-change
//                         counter, stop tests (2429-2447) -- or the shard's exchange record
//   de_turn_kernel        head k and generation k+1 in one launch (strategy random)
//   de_generation_groups_kernel / de_turn_groups_kernel   the same for D <= 64: several agents
//                         per wave, one per group of lanes
//   de_finalize_kernel    the head's decisions from the records of all shards
//
// Data layout in HBM: population row-major [shard_n][D] fp64, two buffers
// (synchronous generation: donors are read from `cur`, survivors written to
// `nxt`); scores [shard_n] fp64 updated in place by the owning wave.
// Mapping (D > 64): one wave64 per agent; lane l holds elements c*128 + 2l, +1 of each
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
  uint32_t *ticket;    // arrival counter of de_scan_head_kernel's blocks (zero between launches)
  const double *zero;  // 16 bytes of zeros: source for lanes past the row end
  uint32_t ntiles;
  uint32_t stream;     // the new generation's rows are stored nontemporally: they are read again a
                       // whole generation later, the caches are better spent on the rows being
                       // gathered (NLSG_DE_STREAM=0 is the A/B switch)
  uint64_t pop, D, shard_lo, shard_n;
  double CR, F, eps, fmul;
  uint64_t max_iter, best_val_no_change, seed;
  int32_t strategy;
  int32_t vec;         // rows are 16-byte aligned (D even)
  // set by the host (nlsg_de.hip): wave-uniform values every wave of a launch would recompute
  uint64_t gen_key;    // ctr_key(seed, generation) of the generation being launched
  uint64_t cr_thresh;  // u01(z) < CR  <=>  z < cr_thresh (u01 is monotone in z): the crossover
  int32_t cr_all;      // test on the draw's bits; cr_all: CR > 1, every draw passes
  int32_t pad2;
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
  // the agent's key and its wave-uniform draws are computed on the vector unit (see on_valu):
  // lane L takes draw D + L of the agent's stream — lane 0 the crossover's jrand (:2364), lane
  // 1 + k donor candidate k — so the donor loop below only picks lanes
  const uint64_t ka = ctr_key(kg, ga);  // wave-uniform: on the scalar unit (the vector unit is
                                         // the busier one here: 68 % against 35 % at pop = 65 536)
  const int lane = lane_id();
  // generate_index (:2325-2329), size_t(u * max): shard sizes and D fit 32 bits, so the
  // conversion is the one-instruction 32-bit one (same truncation)
  const uint32_t lim = static_cast<uint32_t>(lane == 0 ? p.D : p.shard_n);
  const uint32_t idx = static_cast<uint32_t>(u01(ctr_key(ka, p.D + static_cast<uint64_t>(lane))) *
                                             static_cast<double>(lim));
  const uint64_t drawn = idx >= lim ? lim - 1 : idx;  // the u == 1.0 corner clamped (B10)
  // generate_indices (nlsolver.h:2331-2355): three distinct donors != fixed,
  // by rejection, drawn inside this engine's shard. Wave-uniform (scalar) code.
  const uint64_t fixed = (p.strategy == NLSG_DE_RANDOM) ? ga : best_id;  // :2451-2457
  // Candidate k is the draw of lane k + 1: the first three lanes (in order) whose candidate
  // differs from `fixed` and from the picks before it are found with three wave-wide compares
  // and find-first-set — the rejection loop as a handful of instructions instead of a scalar
  // loop of branches (this kernel was bound by the CU's scalar issue at pop = 65 536: ~400 scalar
  // instructions per wave against ~320 vector ones). Same candidates in the same order.
  const uint64_t cand_v = p.shard_lo + drawn;
  constexpr uint64_t top = 1ull << 63;
  const uint64_t m0 = __ballot(cand_v != fixed) & ~1ull;  // lane 0 holds jrand
  const int i0 = __builtin_ctzll(m0 | top);
  uint64_t r0 = readlane64(cand_v, i0);
  const uint64_t m1 = m0 & __ballot(cand_v != r0) & ((~0ull << i0) << 1);
  const int i1 = __builtin_ctzll(m1 | top);
  uint64_t r1 = readlane64(cand_v, i1);
  const uint64_t m2 = m1 & __ballot(cand_v != r1) & ((~0ull << i1) << 1);
  const int i2 = __builtin_ctzll(m2 | top);
  uint64_t r2 = readlane64(cand_v, i2);
  if (m0 == 0 || m1 == 0 || m2 == 0) {  // fewer than three among 63 candidates (tiny shards):
                                         // the literal loop, from the start
    r0 = r1 = r2 = ~0ull;
    int have = 0;
    for (int k = 0; k < kDeMaxTries && have < 3; k++) {
      // (candidate 63 has no lane: after 61 rejections it is drawn the slow way)
      const uint64_t cand =
          p.shard_lo + (k < 63 ? readlane64(drawn, k + 1)
                               : clamp_index(u01(ctr_key(ka, p.D + 1 + k)), p.shard_n));
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
  }
  c.a = a;
  c.ka = ka;
  c.r0 = r0;
  c.r1 = r1;
  c.r2 = r2;
  c.jrand = readlane64(drawn, 0);  // :2364
  // rows: own (selection survivor; non-crossed coordinates for strategy random), 3 donors,
  // and for strategy best the row of best_id (L2-resident; strategy random reads the zero
  // pad instead so that the instruction stream does not depend on the strategy)
  // (shard-local indices and D fit 32 bits: one scalar multiply pair per row offset)
  const uint32_t d32 = static_cast<uint32_t>(p.D);
  auto row = [&](uint64_t local) { return cur + static_cast<uint64_t>(static_cast<uint32_t>(local)) * d32; };
  load_row<CHUNKS, VEC>(row(a), D, p.zero, c.own);
  load_row<CHUNKS, VEC>(row(r0 - p.shard_lo), D, p.zero, c.d1);
  load_row<CHUNKS, VEC>(row(r1 - p.shard_lo), D, p.zero, c.d2);
  load_row<CHUNKS, VEC>(row(r2 - p.shard_lo), D, p.zero, c.d3);
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
  // ctr_key(ka, e) = mix64(ka + G (e + 1)), e + 1 = (2 lane + 1) + (128 ch + k): one 64-bit
  // multiply per lane, the rest are compile-time constants
  const uint64_t ka_lane = c.ka + kGolden * (2 * static_cast<uint64_t>(lane) + 1);
#pragma unroll
  for (int ch = 0; ch < CHUNKS; ch++) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint64_t e = static_cast<uint64_t>(ch) * 128 + 2 * static_cast<uint64_t>(lane) + k;
      // u01(z) < CR decided on the draw's bits (cr_thresh: the smallest z whose uniform is >= CR)
      const uint64_t z = mix64(ka_lane + kGolden * static_cast<uint64_t>(128 * ch + k));
      const bool cross = z < p.cr_thresh || p.cr_all;
      const double mut = c.d1[ch][k] + p.F * (c.d2[ch][k] - c.d3[ch][k]);
      trial[ch][k] = (cross || e == c.jrand) ? mut : (rnd ? c.own[ch][k] : c.keep[ch][k]);
    }
  }
  // (elements >= D are 0 in every loaded row, hence 0 in the trial as well)
  const double score = p.fmul * wave_objective<OBJ, CHUNKS>(trial, D);  // :2463
  const bool accept = score < c.old_score;                               // :2466 (NaN -> keep)
  double *out = nxt + c.a * D;
  if (p.stream) {  // wave-uniform
    if (accept) {
      store_row_stream<CHUNKS, VEC>(out, D, trial);
    } else {
      store_row_stream<CHUNKS, VEC>(out, D, c.own);
    }
  } else if (accept) {
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

// ---- agents of at most 64 coordinates: 64 / G agents per wave, one per group of G lanes (G = 4, 8,
// 16, 32 for D <= 8, 16, 32, 64; lane g of a group holds coordinates 2g, 2g + 1). One agent per
// wave leaves 64 - D / 2 lanes idle and costs the same 38 us per generation at D = 16 as at
// D = 64. Everything that is wave-uniform above is group-uniform here and lives in vector
// registers: the agent's key, its draws (lane g of a group takes draw D + round * G + g of the
// agent's stream: draw D is jrand, draw D + 1 + k donor candidate k) and the donor picks, which
// walk the candidates in order with selects instead of branches. group_objective gives the trial
// the bits of the full-wave tree, so a population's history does not depend on the packing.
template <int OBJ, int G>
__device__ inline void de_generation_groups_block(const DeParams &p, int par, uint64_t generation,
                                                  int ignore_done, uint64_t block) {
  constexpr int P = 64 / G;
  const DeState *__restrict__ st = p.state;
  if (!ignore_done && st->done) return;
  const uint64_t wave = block * 4 + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (wave * P >= p.shard_n) return;
  const int lane = lane_id(), g = lane & (G - 1), gi = lane / G;
  const bool live = wave * P + gi < p.shard_n;
  const uint64_t a = live ? wave * P + gi : wave * P;  // idle groups shadow a live agent
  const uint64_t D = p.D;
  const double *__restrict__ cur = p.buf[par];
  double *__restrict__ nxt = p.buf[par ^ 1];
  const bool rnd = p.strategy == NLSG_DE_RANDOM;
  const uint64_t kg = first64(ctr_key(on_valu(p.seed), generation));
  const uint64_t ga = p.shard_lo + a;
  const uint64_t ka = ctr_key(kg, ga);
  const uint64_t fixed = rnd ? ga : st->best_id;  // :2451-2457
  // draws D + round * G + g; cross-lane reads stay inside the group
  const int base = gi * G;
  uint64_t drawn = ctr_key(ka, D + static_cast<uint64_t>(g));
  const uint64_t jrand = clamp_index(u01(__shfl(drawn, base, 64)), D);  // :2364
  // generate_indices (nlsolver.h:2331-2355): three distinct donors != fixed, by rejection
  uint64_t r0 = ~0ull, r1 = ~0ull, r2 = ~0ull;
  int have = 0;
  for (int k = 0; k < kDeMaxTries; k++) {
    if (__ballot(have < 3) == 0ull) break;
    const int pos = k + 1;  // candidate k is draw D + pos: round pos / G, lane pos % G
    if (pos >= G && (pos & (G - 1)) == 0)
      drawn = ctr_key(ka, D + static_cast<uint64_t>(pos + g));
    const uint64_t cand =
        p.shard_lo + clamp_index(u01(__shfl(drawn, base + (pos & (G - 1)), 64)), p.shard_n);
    const bool used = (cand == fixed) || (have > 0 && cand == r0) || (have > 1 && cand == r1);
    const bool take = !used && have < 3;
    r0 = (take && have == 0) ? cand : r0;
    r1 = (take && have == 1) ? cand : r1;
    r2 = (take && have == 2) ? cand : r2;
    have += take ? 1 : 0;
  }
  for (uint64_t cand = p.shard_lo; __ballot(have < 3) != 0ull; cand++) {  // fallback: lowest unused
    const bool used = (cand == fixed) || (have > 0 && cand == r0) || (have > 1 && cand == r1);
    const bool take = !used && have < 3;
    r0 = (take && have == 0) ? cand : r0;
    r1 = (take && have == 1) ? cand : r1;
    r2 = (take && have == 2) ? cand : r2;
    have += take ? 1 : 0;
  }
  // rows: own, three donors, and for strategy best the row of best_id
  const uint32_t j0 = 2 * g, j1 = 2 * g + 1;
  const bool in0 = j0 < D, in1 = j1 < D;
  const uint32_t d32 = static_cast<uint32_t>(D);
  auto row = [&](uint64_t local) { return cur + static_cast<uint64_t>(static_cast<uint32_t>(local)) * d32; };
  auto load2 = [&](const double *rp, double (&v)[2]) {
    v[0] = in0 ? rp[j0] : 0.0;
    v[1] = in1 ? rp[j1] : 0.0;
  };
  double own[2], d1[2], d2[2], d3[2], keep[2] = {0.0, 0.0};
  load2(row(a), own);
  load2(row(r0 - p.shard_lo), d1);
  load2(row(r1 - p.shard_lo), d2);
  load2(row(r2 - p.shard_lo), d3);
  if (!rnd) load2(p.best_x, keep);
  const double old_score = p.scores[par][a];
  // propose_new_agent (nlsolver.h:2357-2375)
  double trial[2];
  const uint64_t ka_lane = ka + kGolden * (2 * static_cast<uint64_t>(g) + 1);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint64_t e = 2 * static_cast<uint64_t>(g) + k;
    const uint64_t zc = mix64(ka_lane + kGolden * static_cast<uint64_t>(k));
    const double mut = d1[k] + p.F * (d2[k] - d3[k]);
    // u01(zc) < CR on the draw's bits (DeParams.cr_thresh)
    const double t = (zc < p.cr_thresh || p.cr_all || e == jrand) ? mut : (rnd ? own[k] : keep[k]);
    trial[k] = (k ? in1 : in0) ? t : 0.0;
  }
  const double score = p.fmul * group_objective<OBJ, G>(trial[0], trial[1], D);  // :2463
  const bool accept = score < old_score;                                         // :2466
  if (live) {
    double *out = nxt + static_cast<uint64_t>(static_cast<uint32_t>(a)) * d32;
    if (in0) out[j0] = accept ? trial[0] : own[0];
    if (in1) out[j1] = accept ? trial[1] : own[1];
    if (g == 0) {
      p.scores[par ^ 1][a] = accept ? score : old_score;
      if (p.trace != nullptr) {
        uint64_t *t = p.trace + a * kTraceWords;
        t[0] = r0;
        t[1] = r1;
        t[2] = r2;
        t[3] = jrand;
        t[4] = accept ? 1u : 0u;
      }
    }
  }
}

template <int OBJ, int G>
__global__ __launch_bounds__(256) void de_generation_groups_kernel(DeParams p, int par,
                                                                   uint64_t generation,
                                                                   int ignore_done) {
  de_generation_groups_block<OBJ, G>(p, par, generation, ignore_done, blockIdx.x);
}

// One agent per wave. (Two agents per wave — ten gathers in flight — measured -7 % kernel time at
// pop = 65536 but +9 % at pop = 2^20 and only -2 % per turn; not kept.)
template <int OBJ, int CHUNKS, bool VEC>
__device__ inline void de_generation_block(const DeParams &p, int par, uint64_t generation,
                                           int ignore_done, uint64_t block) {
  // `generation` (k+1) and the source buffer `par` (k & 1) come from the host: the k-th
  // turn's head may still be running when this kernel starts (speculative launch for
  // strategy random); the device state is only consulted for the stop flag, which is
  // final for every head older than that one.
  const DeState *__restrict__ st = p.state;
  if (!ignore_done && st->done) return;  // a stop test fired: the turn is a no-op
  const uint64_t a0 = block * 4 + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (a0 >= p.shard_n) return;
  const double *__restrict__ cur = p.buf[par];
  double *__restrict__ nxt = p.buf[par ^ 1];
  const uint64_t kg = p.gen_key;  // = ctr_key(p.seed, generation), from the host
  const uint64_t best_id = st->best_id;
  DeAgent<CHUNKS> A;
  de_fetch_agent<CHUNKS, VEC>(p, cur, par, kg, best_id, a0, true, A);
  de_process_agent<OBJ, CHUNKS, VEC>(p, nxt, par, A);
}

template <int OBJ, int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void de_generation_kernel(DeParams p, int par, uint64_t generation,
                                                          int ignore_done) {
  de_generation_block<OBJ, CHUNKS, VEC>(p, par, generation, ignore_done, blockIdx.x);
}

// ---- best scan + stop tests -------------------------------------------------
// Head number k looks at the population after k generations: buffer k & 1. It records
// that position in the state; a head that fires a stop test freezes the state there.
__device__ inline void head_position(DeState *st, const DeParams &p, uint64_t k) {
  st->iter = k;
  st->fcalls = p.pop * (k + 1);
  st->parity = static_cast<int32_t>(k & 1);
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

// ---- the head of a turn ---------------------------------------------------------
// Separate scan / finisher launches cost a dependent dispatch each (~5 us; 10 us of a 59 us
// turn at pop 65536 for eps <= 0, five launches and ~25 us with a two-pass std_err), so a head
// is ONE launch: one block per tile of kTile scores reduces it -- minimum and first index and,
// when std_err can decide (eps > 0), the tile's sum and its M2 about the TILE mean, scores held
// in registers -- publishes that with write-through stores, drains them and takes a ticket;
// the block that takes the last ticket reads every partial with cache-bypassing loads and
// finishes: best with the incumbent rule, std_err, counters, stop tests, best_x -- or, on a
// shard, the exchange record
//   [minv, mini(bits), sum, M2 about the shard mean, valid, x_best[0..D)]   (kRecHeader + D)
// (`valid` is 0 only for a shard whose scores are all NaN and that does not own the incumbent).
// std_err (nlsolver.h:2037-2052) in one pass: tiles are merged the way shards are,
//   total = tree_t(sum_t), mean = total / n, M2 = tree_t(M2_t + n_t (sum_t / n_t - mean)^2)
// (a population of one tile is exactly the two-pass formula). min / first-index are order-
// independent and the sums keep their fixed trees (thread-sequential partials, wave butterfly,
// ((w0+w1)+w2)+w3 per tile, the same tree over tiles), so which block ends up last does not
// matter. No release / acquire fence: a release would write back every dirty line that
// generation blocks of the same launch hold in the XCD's L2.
// thread 0, after the block's partial stores: true in the block that arrived last
__device__ inline bool take_ticket(const DeParams &p) { return take_ticket(p.ticket, p.ntiles); }

// `rec` == nullptr: one GPU, the launch finishes the turn; else the shard's exchange record
__device__ inline void de_scan_head_block(const DeParams &p, uint64_t k, uint32_t tile, double *rec) {
  __shared__ double red[4];
  __shared__ double mv[4];
  __shared__ uint64_t mi[4];
  __shared__ uint64_t s_row;
  __shared__ int s_have;
  __shared__ int s_last;
  DeState *st = p.state;
  if (st->done) return;  // only written by a last block, after every block got here
  const int par = static_cast<int>(k & 1);
  const bool need_se = p.eps > 0;
  const double *__restrict__ sc = p.scores[par];
  const uint64_t base = static_cast<uint64_t>(tile) * kTile;
  const uint64_t tile_n = (p.shard_n - base) < kTile ? (p.shard_n - base) : kTile;
  double v[kTile / 256];
  double acc = 0.0;
  double bv = __builtin_inf();
  uint64_t bi = ~0ull;
#pragma unroll
  for (int q = 0; q < kTile / 256; q++) {
    const uint64_t i = threadIdx.x + 256u * q;
    v[q] = sc[base + (i < tile_n ? i : 0)];  // clamped, masked below
  }
#pragma unroll
  for (int q = 0; q < kTile / 256; q++) {
    const uint64_t i = threadIdx.x + 256u * q;
    if (i < tile_n) {
      acc = acc + v[q];
      argmin_combine(bv, bi, v[q], base + i);
    }
  }
  double tile_sum = 0.0, tile_m2 = 0.0;
  if (need_se) {
    tile_sum = block_tree_256(acc, red);
    const double tile_mean = tile_sum / static_cast<double>(tile_n);
    acc = 0.0;
#pragma unroll
    for (int q = 0; q < kTile / 256; q++) {
      const double d = v[q] - tile_mean;
      if (threadIdx.x + 256u * q < tile_n) acc = acc + d * d;  // :2046-2049
    }
    tile_m2 = block_tree_256(acc, red);
  }
  block_argmin_256(bv, bi, mv, mi);
  if (threadIdx.x == 0) {
    sc1_store(&p.part[tile].minv, bv);
    __hip_atomic_store(&p.part[tile].mini, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (need_se) {
      sc1_store(&p.part[tile].sum, tile_sum);
      sc1_store(&p.part[tile].m2, tile_m2);
    }
    s_last = take_ticket(p) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) head_position(st, p, k);
  bv = __builtin_inf();
  bi = ~0ull;
  for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256)
    argmin_combine(bv, bi, sc1_load(&p.part[j].minv),
                   __hip_atomic_load(&p.part[j].mini, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  block_argmin_256(bv, bi, mv, mi);
  double total = 0.0, m2 = 0.0;
  if (need_se) {
    acc = 0.0;
    for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256) acc = acc + sc1_load(&p.part[j].sum);
    total = block_tree_256(acc, red);
    const double mean = total / static_cast<double>(p.shard_n);  // :2044
    acc = 0.0;
    for (uint32_t j = threadIdx.x; j < p.ntiles; j += 256) {
      const uint64_t jb = static_cast<uint64_t>(j) * kTile;
      const double nj = static_cast<double>((p.shard_n - jb) < kTile ? (p.shard_n - jb) : kTile);
      const double dm = sc1_load(&p.part[j].sum) / nj - mean;
      acc = acc + (sc1_load(&p.part[j].m2) + nj * (dm * dm));
    }
    m2 = block_tree_256(acc, red);
  }
  if (threadIdx.x == 0) {
    // Shard minimum with the reference's tie rule (strict '<' scan starting from the
    // incumbent, nlsolver.h:2432-2437): the incumbent survives when nobody in the shard is
    // strictly better.
    const uint64_t inc = st->best_id;
    uint64_t gi = (bi == ~0ull) ? inc : p.shard_lo + bi;
    if (inc >= p.shard_lo && inc < p.shard_lo + p.shard_n) {
      const double inc_score = sc[inc - p.shard_lo];
      if (!(bv < inc_score)) {
        gi = inc;
        bv = inc_score;
      }
    }
    const bool mine = gi >= p.shard_lo && gi < p.shard_lo + p.shard_n;
    if (rec == nullptr) {
      finish_turn(st, p, gi, bv, mine,
                  need_se ? sqrt(m2 / static_cast<double>(p.pop - 1))  // :2050-2051
                          : __builtin_nan(""));
    } else {
      rec[0] = bv;
      rec[1] = __longlong_as_double(static_cast<long long>(gi));
      rec[2] = total;
      rec[3] = m2;
      rec[4] = mine ? 1.0 : 0.0;
    }
    s_row = gi - p.shard_lo;
    s_have = mine ? 1 : 0;
  }
  __syncthreads();
  const bool have = s_have != 0;
  const double *row = p.buf[par] + (have ? s_row : 0) * p.D;  // x = agents[best_id], :2444
  if (rec != nullptr) {
    for (uint64_t d = threadIdx.x; d < p.D; d += 256) rec[kRecHeader + d] = have ? row[d] : 0.0;
  } else if (have) {
    for (uint64_t d = threadIdx.x; d < p.D; d += 256) p.best_x[d] = row[d];
  }
}

__global__ __launch_bounds__(256) void de_scan_head_kernel(DeParams p, uint64_t k, double *rec) {
  de_scan_head_block(p, k, blockIdx.x, rec);
}

// One launch per turn for strategy random (one GPU): the first ntiles blocks run
// head k (scan of population k, stop tests), the others build generation k+1 from population
// k at the same time. Random donors do not depend on the head's result; if head k fires a
// stop test, generation k+1 went to the other buffers and is never adopted (exactly the
// speculative turn of the sharded path), so results equal the serial order head -> generation.
template <int OBJ, int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void de_turn_kernel(DeParams p, int par, uint64_t generation) {
  if (blockIdx.x < p.ntiles) {
    de_scan_head_block(p, generation - 1, blockIdx.x, nullptr);
    return;
  }
  de_generation_block<OBJ, CHUNKS, VEC>(p, par, generation, 0, blockIdx.x - p.ntiles);
}

// the same turn with the packed generation (agents of at most 64 coordinates, several per wave)
template <int OBJ, int G>
__global__ __launch_bounds__(256) void de_turn_groups_kernel(DeParams p, int par, uint64_t generation) {
  if (blockIdx.x < p.ntiles) {
    de_scan_head_block(p, generation - 1, blockIdx.x, nullptr);
    return;
  }
  de_generation_groups_block<OBJ, G>(p, par, generation, 0, blockIdx.x - p.ntiles);
}

// ---- the sharded path ----------------------------------------------------------
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

// ---- rows longer than a wave's registers hold (D > 1024; the reference has no limit) -----------
// One wave per agent, the rows taken in segments of 1024 coordinates. A generation makes two
// passes: the first builds the trial segment by segment, scores it (objective_accumulate: the
// whole-row summation order) and stores it; the second, only for a rejected trial, copies the
// survivor's row over it. Same draws (the element's index keys them), same arithmetic, same bits
// as the register-resident kernels would give — the oracle restates neither layout.
template <int OBJ, bool VEC>
__global__ __launch_bounds__(256) void de_init_long_kernel(DeParams p, const double *__restrict__ x0) {
  const uint64_t a = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (a >= p.shard_n) return;
  const int lane = lane_id();
  const uint64_t D = p.D;
  const uint64_t ka = ctr_key(ctr_key(p.seed, 0), p.shard_lo + a);
  auto element = [&](uint64_t e) {  // generate_sequence, nlsolver.h:2309
    return (e < D) ? (u01(ctr_key(ka, e)) - 0.5) * x0[e] : 0.0;
  };
  double acc = 0.0;
  for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
    double xv[kSeg][2];
#pragma unroll
    for (int c = 0; c < kSeg; c++)
#pragma unroll
      for (int k = 0; k < 2; k++)
        xv[c][k] = element(e_base + static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k);
    store_segment<VEC>(p.buf[0] + a * D, e_base, D, xv);
    objective_accumulate<OBJ, kSeg>(acc, xv, e_base, D, element(e_base + 128 * kSeg));
  }
  const double f = p.fmul * objective_finish<OBJ>(acc, D);  // :2423-2425
  if (lane == 0) p.scores[0][a] = f;
}

template <int OBJ, bool VEC>
__global__ __launch_bounds__(256) void de_generation_long_kernel(DeParams p, int par, uint64_t generation,
                                                               int ignore_done) {
  const DeState *__restrict__ st = p.state;
  if (!ignore_done && st->done) return;
  const uint64_t a = static_cast<uint64_t>(blockIdx.x) * 4 +
                     __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (a >= p.shard_n) return;
  const int lane = lane_id();
  const uint64_t D = p.D;
  const double *__restrict__ cur = p.buf[par];
  double *__restrict__ nxt = p.buf[par ^ 1];
  const uint64_t kg = first64(ctr_key(on_valu(p.seed), generation));
  // donors and the forced dimension exactly as de_fetch_agent draws them
  const uint64_t ga = p.shard_lo + a;
  const uint64_t ka = first64(ctr_key(kg, on_valu(ga)));
  const uint64_t drawn = clamp_index(u01(ctr_key(ka, D + static_cast<uint64_t>(lane))),
                                     lane == 0 ? D : p.shard_n);
  const uint64_t best_id = st->best_id;
  const bool rnd = p.strategy == NLSG_DE_RANDOM;
  const uint64_t fixed = rnd ? ga : best_id;
  uint64_t r0 = ~0ull, r1 = ~0ull, r2 = ~0ull;
  int have = 0;
  for (int k = 0; k < kDeMaxTries && have < 3; k++) {
    const uint64_t cand = p.shard_lo + (k < 63 ? readlane64(drawn, k + 1)
                                               : clamp_index(u01(ctr_key(ka, D + 1 + k)), p.shard_n));
    const bool used = (cand == fixed) || (have > 0 && cand == r0) || (have > 1 && cand == r1);
    if (!used) {
      if (have == 0) r0 = cand;
      else if (have == 1) r1 = cand;
      else r2 = cand;
      have++;
    }
  }
  for (uint64_t cand = p.shard_lo; have < 3; cand++) {
    const bool used = (cand == fixed) || (have > 0 && cand == r0) || (have > 1 && cand == r1);
    if (!used) {
      if (have == 0) r0 = cand;
      else if (have == 1) r1 = cand;
      else r2 = cand;
      have++;
    }
  }
  const uint64_t jrand = readlane64(drawn, 0);
  const double *own = cur + a * D, *d1 = cur + (r0 - p.shard_lo) * D, *d2 = cur + (r1 - p.shard_lo) * D,
               *d3 = cur + (r2 - p.shard_lo) * D;
  const double *keep = rnd ? own : p.best_x;  // non-crossed coordinates (:2369-2372)
  auto trial_at = [&](uint64_t e) {  // one coordinate of the trial, the same in every lane
    if (e >= D) return 0.0;
    const uint64_t zc = ctr_key(ka, e);
    const double mut = d1[e] + p.F * (d2[e] - d3[e]);
    return (zc < p.cr_thresh || p.cr_all || e == jrand) ? mut : keep[e];
  };
  double *out = nxt + a * D;
  double acc = 0.0;
  const uint64_t ka_lane = ka + kGolden * (2 * static_cast<uint64_t>(lane) + 1);
  for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
    double v1[kSeg][2], v2[kSeg][2], v3[kSeg][2], vk[kSeg][2], trial[kSeg][2];
    load_segment<VEC>(d1, e_base, D, p.zero, v1);
    load_segment<VEC>(d2, e_base, D, p.zero, v2);
    load_segment<VEC>(d3, e_base, D, p.zero, v3);
    load_segment<VEC>(keep, e_base, D, p.zero, vk);
    const uint64_t kseg = ka_lane + kGolden * e_base;
#pragma unroll
    for (int ch = 0; ch < kSeg; ch++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = e_base + static_cast<uint64_t>(ch) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        const uint64_t zc = mix64(kseg + kGolden * static_cast<uint64_t>(128 * ch + k));
        const double mut = v1[ch][k] + p.F * (v2[ch][k] - v3[ch][k]);
        trial[ch][k] = (zc < p.cr_thresh || p.cr_all || e == jrand) ? mut : vk[ch][k];
      }
    store_segment<VEC, true>(out, e_base, D, trial);
    objective_accumulate<OBJ, kSeg>(acc, trial, e_base, D, trial_at(e_base + 128 * kSeg));
  }
  const double score = p.fmul * objective_finish<OBJ>(acc, D);  // :2463
  const double old_score = p.scores[par][a];
  const bool accept = score < old_score;  // :2466 (NaN -> keep)
  if (!accept)
    for (uint64_t e_base = 0; e_base < D; e_base += 128 * kSeg) {
      double vo[kSeg][2];
      load_segment<VEC>(own, e_base, D, p.zero, vo);
      store_segment<VEC, true>(out, e_base, D, vo);
    }
  if (lane == 0) p.scores[par ^ 1][a] = accept ? score : old_score;
  if (p.trace != nullptr && lane == 0) {
    uint64_t *t = p.trace + a * kTraceWords;
    t[0] = r0;
    t[1] = r1;
    t[2] = r2;
    t[3] = jrand;
    t[4] = accept ? 1u : 0u;
  }
}

// Before the host reads the state: unless a stop test fired, the engine stands after the
// k generations it has launched (the last head only saw k-1 of them).
__global__ void de_settle_kernel(DeParams p, uint64_t k) {
  if (!p.state->done) head_position(p.state, p, k);
}

}  // namespace nlsg
