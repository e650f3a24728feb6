// nlsolver_amd/csrc/nlsg_tinyqr_kernels.h — batched linear least squares by Givens QR:
// tinyqr::lm (tinyqr.h:461-470) = back_solve(qr_decomposition(X)) on `batch` independent n x p
// systems, n >= p, p <= 64 (the square damped system of the LM engine is the
// special case n = p that nlsg_lm_kernels.h solves inside its iteration).
//
// The reference eliminates column by column, rows bottom-up (qr_impl, tinyqr.h:253-283): rotation
// (j, i) mixes rows i-1 and i to annihilate R[i][j], j = 0 .. p-1, i = n-1 .. j+1. Rotations of
// different columns commute as soon as they touch different row pairs, and (j, i) depends only
// on (j, i+1) and (j-1, i-1), so rotation (j, i) can run at STEP (n-1-i) + 2j: the eliminations
// of column j form a chain that climbs one row per step, two rows behind chain j-1. Every element
// sees exactly the sequence of updates of the serial loop — same (a, b) -> (c, s), same element
// updates, hence the same bits — while up to p rotations run side by side.
//
// At step k the chains occupy rows n-2-k .. n-1-k+2(p-1): a WINDOW of 2p+1 rows that slides up by
// one row per step. Rows below it are finished (annihilated in all p columns — lm() never reads
// them), rows above it have not been touched. So a system of ANY height needs 2p+4 rows of LDS:
// a ring indexed by row mod RING, fed one row per step from global memory (two steps ahead, the
// value waiting in a register in between).
//
// Arithmetic = oracle_lm.c order 1 (tinyqr_lm_corotated), as in the LM engine's solver:
//  * no Q: lm() uses Q only through Q^T y (back_solve, :437-459), so y is rotated along with R
//    as column p;
//  * an element update is one rounded product and one fused multiply-add,
//    lower' = fma(c, lower, s * upper), upper' = fma(c, upper, (-s) * lower);
//  * Givens pair as givens_rotation (:86-97) with r * r for pow(r, 2);
//  * back-substitution sums from j = p-1 down to i+1, lm()'s cleanup (|v| < tol -> 0, :278-282)
//    applied to the entries that are read.
// One workgroup per system. Wave 0 computes the step's Givens pairs (one chain per lane) while
// waves 1 and 2 move the entering row; then all waves apply the rotations (chain j on wave
// j mod W — one chain per (wave, slot) for the whole run —, one column per lane). A chain's
// rotation at step k+1 reads the row its rotation at step k wrote as `lower'`, and nothing else
// reads that row's columns right of the pivot in between, so the wave keeps it in a register
// (`carry`): per rotation one row read, one written. Lanes without a column work on a padding
// column (index p + 1) instead of being masked off. Two barriers per step; bound by instruction
// issue and the Givens pair's ~45 dependent fp64 instructions, not roofline-graded.
#pragma once
#include "nlsg_common.h"
#include "nlsg_math.h"

namespace nlsg {

constexpr int kTqrThreads = 512;
constexpr int kTqrMaxP = 64;

struct TqrParams {
  const double *X;  // [batch][p][n]: each system column-major n x p, as tinyqr takes it
  const double *y;  // [batch][n]
  double *beta;     // [batch][p]
  uint64_t batch, n;
  uint32_t p, ring, stride;  // ring rows (2p + 4), doubles per ring row ((p + 2) | 1: odd, >= p + 2)
  double tol;
};

__host__ __device__ inline uint32_t tqr_ring_rows(uint32_t p) { return 2 * p + 4; }
__host__ __device__ inline uint32_t tqr_stride(uint32_t p) { return (p + 2) | 1u; }  // columns 0 .. p-1, y, padding
__host__ __device__ inline size_t tqr_lds_bytes(uint32_t p) {
  return (static_cast<size_t>(tqr_ring_rows(p)) * tqr_stride(p) + 2 * kTqrMaxP) * sizeof(double);
}

// THREADS / 64 waves serve systems of up to MAXP columns: (512, 64), (256, 32), (128, 8) — a small
// system gets a small workgroup, so that many of them share a CU (the host side picks by p)
template <int THREADS, int MAXP = kTqrMaxP>
__global__ __launch_bounds__(THREADS) void tinyqr_lm_kernel(TqrParams q) {
  constexpr int W = THREADS / 64;
  static_assert(W >= 2 && MAXP <= kTqrMaxP, "wave 1 loads the entering row");
  extern __shared__ __align__(16) double tqr_smem[];
  const int t = threadIdx.x, lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int p = static_cast<int>(q.p), S = static_cast<int>(q.stride), RING = static_cast<int>(q.ring);
  const int n = static_cast<int>(q.n);  // (the host side admits n < 2^30)
  double *ring = tqr_smem;                                              // [RING][S]
  double2 *cs = reinterpret_cast<double2 *>(tqr_smem + RING * S);       // [64]
  const uint64_t sys = blockIdx.x;
  const double *X = q.X + sys * q.n * q.p, *y = q.y + sys * q.n;
  const int nch = n - 1 < p ? n - 1 : p;  // chains: columns that have a row below the diagonal
  const int last = nch > 0 ? nch - 1 + n - 2 : -1;  // last step

  // element `c` of row r (c < p: X, c == p: y) — what the loader lanes fetch
  auto fetch = [&](int r, int c) -> double {
    if (r < 0) return 0.0;
    return c < p ? X[static_cast<uint64_t>(c) * q.n + static_cast<uint64_t>(r)] : y[r];
  };
  auto slot_of = [&](int r) -> int { return r % RING; };

  // prologue: rows n-1 and n-2 (step 0), row n-3 waits in a register for step 1
  const bool loader = (wid == 1 && lane <= p) || (wid == 2 && lane == 0 && p == 64);
  const int lcol = wid == 1 ? lane : 64;  // wave 1: columns 0 .. min(p, 63); wave 2 lane 0: column 64
  double pending = 0.0;
  if (loader) {
    ring[slot_of(n - 1) * S + lcol] = fetch(n - 1, lcol);
    if (n >= 2) ring[slot_of(n - 2) * S + lcol] = fetch(n - 2, lcol);
    pending = fetch(n - 3, lcol);
  }
  // the apply waves' chains: slot u of wave `wid` follows chain j = wid + u W
  constexpr int SL = (MAXP + W - 1) / W;
  int col[SL];
  double carry[SL];
  // wave-uniform per chain: the ring offsets of the rotation's rows i-1 and i, followed from step to
  // step while it runs. EVENT DRIVEN (as the LM engine's step, nlsg_lm_kernels.h): a wave's chains
  // start at steps 2 j and end at steps j + n - 2, both in slot order, so two compares per step find
  // "a chain of mine starts now" / "ended in the previous step"; a slot's hot path is a bit test of
  // `amask` (75 scalar instructions per wave and step went into per-slot window arithmetic before:
  // profiles/r04/tinyqr_issue_summary.json, the CU's one scalar unit 44 % busy).
  int om[SL], oi[SL];
#pragma unroll
  for (int u = 0; u < SL; u++) {
    const int j = wid + u * W;
    const int c = j + 1 + lane;
    col[u] = c <= p ? c : p + 1;
    carry[u] = 0.0;
    om[u] = oi[u] = 0;
  }
  constexpr int kNever = 0x40000000;
  uint32_t amask = 0;
  int js = wid, je = wid;  // next chain of mine to start / to be finished
  int next_start = js < nch ? 2 * js : kNever, next_fin = je < nch ? je + n - 1 : kNever;
  uint32_t us = 0, ue = 0;  // their slots
  auto on_slot = [&](uint32_t sl, auto &&f) {  // f(int_c<sl>) for a wave-uniform sl: static register indices
    if constexpr (SL > 0) if (sl == 0) f(int_c<0>{});
    if constexpr (SL > 1) if (sl == 1) f(int_c<1>{});
    if constexpr (SL > 2) if (sl == 2) f(int_c<2>{});
    if constexpr (SL > 3) if (sl == 3) f(int_c<3>{});
    if constexpr (SL > 4) if (sl == 4) f(int_c<4>{});
    if constexpr (SL > 5) if (sl == 5) f(int_c<5>{});
    if constexpr (SL > 6) if (sl == 6) f(int_c<6>{});
    if constexpr (SL > 7) if (sl == 7) f(int_c<7>{});
  };
  static_assert(SL <= 8, "on_slot covers eight slots");
  const int om_start = n >= 2 ? slot_of(n - 2) * S : 0, oi_start = slot_of(n - 1) * S;
  auto events = [&](int k) {
    if (k == next_fin) {  // chain je ended in the previous step: its carried row is row je of R, final
      on_slot(ue, [&](auto c) {
        constexpr int sl = decltype(c)::value;
        ring[je * S + col[sl]] = carry[sl];
        amask &= ~(1u << sl);
      });
      je += W;
      ue++;
      next_fin = je < nch ? je + n - 1 : kNever;
    }
    if (k == next_start) {  // chain js starts at the bottom: rows n-2, n-1
      on_slot(us, [&](auto c) {
        constexpr int sl = decltype(c)::value;
        om[sl] = om_start;
        oi[sl] = oi_start;
        carry[sl] = ring[oi_start + col[sl]];
        amask |= 1u << sl;
      });
      js += W;
      us++;
      next_start = js < nch ? 2 * js : kNever;
    }
  };
  __syncthreads();

  int s0 = slot_of(n - 1);  // slot of row i0 = n-1-k, chain 0's lower row at step k
  int sin = slot_of(n >= 3 ? n - 3 : 0);  // slot of the row entering for step k+1: n-3-k
  for (int k = 0; k <= last; k++) {
    // ---- phase A: Givens pairs of this step (wave 0), the entering row (waves 1, 2)
    if (wid == 0) {
      const int j = lane;
      const bool act = j < nch && k >= 2 * j && k <= j + n - 2;
      if (act) {
        int si = s0 + 2 * j;  // row i = i0 + 2j
        si -= si >= RING ? RING : 0;
        const int sm = si == 0 ? RING - 1 : si - 1;  // row i-1
        const double a = ring[sm * S + j], b = ring[si * S + j];
        // givens_rotation (tinyqr.h:86-97): both branches are r = small / large,
        // t = 1 / sqrt(r^2 + 1), {t, t r} — selected, not branched
        const bool swap = fabs(b) > fabs(a);
        const double r = (swap ? a : b) / (swap ? b : a);
        const double tt = div_unscaled(1.0, sqrt_unscaled(r * r + 1.0));  // r^2 + 1 in [1, 2] or NaN
        const double tr = tt * r;
        const double c = swap ? tr : tt, s = swap ? tt : tr;
        cs[j] = make_double2(c, s);
        ring[sm * S + j] = __builtin_fma(c, a, s * b);  // the pivot column's lower' (upper' is annihilated)
      }
    } else if (loader && k + 1 <= last) {
      // row n-3-k enters for step k+1; the one after it is requested now and waits in `pending`
      if (n - 3 - k >= 0) ring[sin * S + lcol] = pending;
      pending = fetch(n - 4 - k, lcol);
    }
    __syncthreads();
    // ---- phase B: rotate_matrix (tinyqr.h:126-139) on the columns right of each chain's pivot;
    // one pass per chain (the SIMD's other waves cover the LDS round trip)
    events(k);
#pragma unroll
    for (int u = 0; u < SL; u++) {
      if (amask >> u & 1u) {  // wave-uniform
        const double2 g = cs[wid + u * W];
        const double t1 = ring[om[u] + col[u]];
        const double c = g.x, s = g.y, t2 = carry[u];
        const double lo = __builtin_fma(c, t1, s * t2);
        ring[oi[u] + col[u]] = __builtin_fma(c, t2, (-s) * t1);
        carry[u] = lo;
        oi[u] = om[u];
        om[u] = om[u] == 0 ? (RING - 1) * S : om[u] - S;
      }
    }
    s0 = s0 == 0 ? RING - 1 : s0 - 1;
    sin = sin == 0 ? RING - 1 : sin - 1;
    __syncthreads();
  }

  events(last + 1);  // the last chain's finished row
  __syncthreads();
  // back_solve (tinyqr.h:437-459) on R beta = w; rows 0 .. p-1 sit in ring slots 0 .. p-1
  if (t < 64) {
    const double tol = q.tol;
    const double w = t < p ? ring[t * S + p] : 0.0;
    double temp = 0.0, u = 0.0;
    for (int j0 = ((p - 1) | 3); j0 >= 0; j0 -= 4) {
      double h[4], d[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = min(j0 - e, p - 1);
        h[e] = ring[min(t, j) * S + j];
        d[e] = ring[j * S + j];
        h[e] = fabs(h[e]) < tol ? 0.0 : h[e];
        d[e] = fabs(d[e]) < tol ? 0.0 : d[e];
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = j0 - e;
        if (j < p) {  // wave-uniform
          if (t == j) u = (w - temp) / d[e];
          const double uj = lane_broadcast(u, j);
          if (t < j) temp += h[e] * uj;
        }
      }
    }
    if (t < p) q.beta[sys * q.p + t] = u;
  }
}

// ---------------------------------------------------------------------------------------------
// Reference-order mode (a PARITY mode, not the fast path): tinyqr::qr_decomposition and lm exactly
// as the reference computes them (tinyqr.h:253-310, 437-470) — rotations in the serial order
// (column by column, rows bottom-up), Q FORMED (every rotation applied to the n x n identity
// too), an element update as two products and an add (rotate_matrix, :126-139: no fused
// multiply-add), back_solve's sums in index order — so that Q, R and beta are the reference's own
// bits (tests/golden/tinyqr.json holds them). One WAVE per system; lane l owns columns l, l + 64,
// ... of the joint matrix [R | Q^T-in-progress] (n rows of p + n columns, a global workspace), so
// a column's history stays within one lane and rows never need exchanging: per rotation the lane
// that owns column j hands (a, b) to the others through v_readlane, everyone rotates its own
// columns of rows i-1 and i. The row a chain pushes upwards stays in registers (`carry`).
// n + p <= 64 * MAXCH columns; any p (the wavefront kernel above stops at 64).
struct TqrRefParams {
  const double *X;  // [batch][p][n] column-major systems
  const double *y;  // [batch][n] or nullptr
  double *work;     // [systems of this launch][n][p + n]
  double *Q;        // [batch][p][n] (thin Q as the reference returns it: Q[i * n + j]) or nullptr
  double *R;        // [batch][p][p] (R[j * p + i] = R(i, j), cleaned) or nullptr
  double *beta;     // [batch][p] or nullptr (needs y; p <= 64)
  uint64_t n, p, sys0;
  double tol;
};

template <int MAXCH>
__global__ __launch_bounds__(64) void tinyqr_reference_kernel(TqrRefParams q) {
  extern __shared__ __align__(16) double tqr_ref_smem[];  // beta: R as it is returned (p * p), Q^T y, beta
  const int lane = lane_id();
  const uint64_t sys = q.sys0 + blockIdx.x;
  const int n = static_cast<int>(q.n), p = static_cast<int>(q.p), W = n + p;
  double *M = q.work + static_cast<uint64_t>(blockIdx.x) * q.n * static_cast<uint64_t>(W);
  const double *X = q.X + sys * q.n * q.p;
  auto row = [&](int r) { return M + static_cast<uint64_t>(r) * W; };
  // R starts as X transposed (:298-303), Q as the identity (:295)
  for (int r = 0; r < n; r++) {
#pragma unroll
    for (int m = 0; m < MAXCH; m++) {
      const int c = lane + 64 * m;
      if (c < W) row(r)[c] = c < p ? X[static_cast<uint64_t>(c) * q.n + r] : (c - p == r ? 1.0 : 0.0);
    }
  }
  double carry[MAXCH], t1[MAXCH];
  for (int j = 0; j < p && j < n - 1; j++) {  // qr_impl, :257-272
    const int jm = j >> 6, jl = j & 63;       // (wave-uniform)
#pragma unroll
    for (int m = 0; m < MAXCH; m++) carry[m] = lane + 64 * m < W ? row(n - 1)[lane + 64 * m] : 0.0;
    for (int i = n - 1; i > j; --i) {
#pragma unroll
      for (int m = 0; m < MAXCH; m++) t1[m] = lane + 64 * m < W ? row(i - 1)[lane + 64 * m] : 0.0;
      double a = 0.0, b = 0.0;  // R[i-1][j], R[i][j]
#pragma unroll
      for (int m = 0; m < MAXCH; m++)
        if (m == jm) {
          a = lane_broadcast(t1[m], jl);
          b = lane_broadcast(carry[m], jl);
        }
      // givens_rotation (:86-97); pow(r, 2) is r * r exactly
      double c, s;
      if (fabs(b) > fabs(a)) {
        const double r = a / b;
        const double sv = 1.0 / __builtin_sqrt(r * r + 1.0);
        c = sv * r;
        s = sv;
      } else {
        const double r = b / a;
        const double cv = 1.0 / __builtin_sqrt(r * r + 1.0);
        c = cv;
        s = cv * r;
      }
      // rotate_matrix on R's p columns and on Q's n columns (:126-139, 269-270): lower = row i-1
#pragma unroll
      for (int m = 0; m < MAXCH; m++) {
        const int col = lane + 64 * m;
        if (col < W) {
          const double lo = c * t1[m] + s * carry[m];
          const double up = -s * t1[m] + c * carry[m];
          row(i)[col] = up;
          carry[m] = lo;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MAXCH; m++)
      if (lane + 64 * m < W) row(j)[lane + 64 * m] = carry[m];
  }
  // outputs: every lane reads back its own columns only
  const bool want_beta = q.beta && q.y;  // (p <= 64: checked by the host side)
  double *Rl = tqr_ref_smem;
#pragma unroll
  for (int m = 0; m < MAXCH; m++) {
    const int col = lane + 64 * m;
    if (col < p) {  // column `col` of R: cleanup (:278-282), transposed leading block (:305-307)
      for (int b = 0; b < p; b++) {
        double v = row(b)[col];
        v = fabs(v) < q.tol ? 0.0 : v;
        if (q.R) q.R[sys * q.p * q.p + static_cast<uint64_t>(col) * p + b] = v;
        if (want_beta) Rl[col * p + b] = v;
      }
    } else if (col < W && q.Q) {  // column col - p of Q's stored rows
      for (int i = 0; i < p; i++) q.Q[sys * q.n * q.p + static_cast<uint64_t>(i) * n + (col - p)] = row(i)[col];
    }
  }
  if (!want_beta) return;
  // back_solve (:437-459): Q^T y coefficient by coefficient in index order — lane i sums its own
  // row of Q, whose elements other lanes wrote: read past the vector L1 (agent-scope loads) after
  // a fence — then the triangular solve, the sums in increasing j, by one lane
  __threadfence();
  double *qty = tqr_ref_smem + p * p, *res = qty + p;
  const double *y = q.y + sys * q.n;
  if (lane < p) {
    double ytmp = 0.0;
    const double *qr = row(lane) + p;
    for (int jj = 0; jj < n; jj++)
      ytmp += __hip_atomic_load(qr + jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * y[jj];
    qty[lane] = ytmp;
  }
  __syncthreads();
  if (lane == 0) {
    for (int i = p; i-- > 0;) {
      double temp = 0.0;
      for (int jj = i + 1; jj < p; ++jj) temp += Rl[jj * p + i] * res[jj];
      res[i] = (qty[i] - temp) / Rl[i * p + i];
    }
  }
  __syncthreads();
  if (lane < p) q.beta[sys * q.p + lane] = res[lane];
}

}  // namespace nlsg
