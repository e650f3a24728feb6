// nlsolver_amd/csrc/nlsg_lm_kernels.h — gfx950 kernel of the batched LM engine.
//
// Replaces (reference file:line): LevenbergMarquardt::solve nlsolver.h:3465-3544 with
// Gauss-Newton functors (f = sum r^2, g = 2 J^T r, H = 2 J^T J), math::cholesky /
// forwardsolve_inplace / backsolve_inplace_t / is_diagonal / get_update_with_hessian
// nlsolver.h:251-330.
//
// nlsolver.h:251-330, and tinyqr::lm (tinyqr.h:253-310, 437-470) as the alternative solver.
//
// All problems advance in lock step, one launch per iteration (see lm_iter_kernel below). Per
// evaluation the design matrix A (m x 64 fp64, 256 KiB at m = 512) is streamed ONCE from HBM,
// sixteen rows at a time by the problem's wave:
//   global -> registers (1 KiB coalesced per wave instruction) -> z = A theta by a 32-lane
//   butterfly -> tanh, residual, weight -> scaled Jacobian rows -> LDS (row stride 80 doubles:
//   conflict-free ds_read_b64 for the MFMA operand pattern) -> J^T J on the fp64 matrix
//   cores (v_mfma_f64_16x16x4_f64, the ten lower 16 x 16 tiles), J^T r on the VALU from the
//   same operands.
// fp64 MFMA is a k-ordered fma chain (verified on gfx950), so H = 2 * fma-chain over the
// rows in order; the CPU restatement (oracle_lm.c, order = 1) mirrors every sum.
// The damped system is solved in LDS: one-wave Cholesky (same per-element arithmetic as the
// reference's row order) with column-sweep substitutions, or wavefront Givens QR.
#pragma once

#include "nlsg_common.h"
#include "nlsg_math.h"

namespace nlsg {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d_nt __attribute__((ext_vector_type(2)));

constexpr int kLmN = 64;        // parameters are padded to 64 columns
constexpr int kLmJStride = 80;  // LDS row stride of the Jacobian block (doubles)
constexpr int kLmQrThreads = 512;  // the QR step's workgroup: each wave follows 32 / 8 column chains
constexpr int kLmTri = 33 * 64;  // packed lower triangle of a 64 x 64 matrix, every row starting at
                                 // an even offset (lm_tri_row): 2112 doubles exactly
// offset of row r of the packed triangle: rows are padded to an even length, so that a row's
// pairs of columns (2m, 2m + 1) are 16-byte aligned and the Cholesky panel reads them as one
// 128-bit LDS load (half as many LDS instructions in its inner loop)
__host__ __device__ constexpr int lm_tri_row(int r) { return 2 * ((r + 1) / 2) * ((r + 2) / 2); }
static_assert(lm_tri_row(64) == kLmTri && lm_tri_row(1) == 2 && lm_tri_row(3) == 8, "row offsets");

struct LmProblem {
  double f, lambda, prev;
  uint64_t iter, fcalls;
  int32_t done;
  int32_t upper;  // finite-difference model: an upper off-diagonal of H exceeds eps * 1e12; the one-pass
                  // evaluations past 64 parameters: 2 | (some off-diagonal of H exceeds it) (LmParams::verdict)
};

struct LmParams {
  const double *A;   // [row group][batch][16][64] (zero padded), see nlsg_lm_set_data
  const double *y;   // [row group][batch][16]
  double *theta;     // [batch][64]
  LmProblem *prob;   // [batch]
  const double *zero;
  uint64_t batch, m, n, max_iter;
  uint64_t nstep;    // ceil(m / 16): row groups of the device layout of A and y
  double lambda0, up, down, f_delta;
  double *Hg;        // [batch][kLmTri] lower triangle of 2 J^T J, packed by rows
  double *gg;        // [batch][64] 2 J^T r
  double eps_h;      // finite-difference model: step of fin_diff_h, pow(DBL_EPSILON, 1/4)
  int32_t fd;        // 1: the functors are the reference's defaults on a built-in objective
  int32_t verdict;   // 1: the evaluation kernel leaves is_diagonal's verdict on H in LmProblem::upper (2 | any)
  // n > 64 (the lm_wide_* kernels below; theta and gg are then [batch][n]):
  double *Hw;        // [batch][n][n] the Hessian as evaluated, row-major, both triangles
  const double *Aw;  // [batch][m][n] design matrices in the caller's layout
  const double *yw;  // [batch][m]
  double *rw;        // [batch][2][m] residuals r_i and weights 1 - tanh^2 of the current evaluation
};

// The damped matrix as the Cholesky solve sees it: the packed lower triangle (the solve only
// reads H[i][j] with j <= i: Cholesky, both substitutions, and is_diagonal by symmetry).
struct LmRowsTri {
  double *base;
  __device__ double &operator()(int i, int j) const { return base[lm_tri_row(i) + j]; }
  __device__ double2 pair(int i, int j) const {  // columns j (even), j + 1
    return *reinterpret_cast<const double2 *>(base + lm_tri_row(i) + j);
  }
};
// The same for an LDS image that ends with row n - 1: the solve's masked reads (rows >= n,
// columns past the diagonal) are folded back inside it.
struct LmRowsTriShort {
  double *base;
  int last;  // n - 1
  __device__ double &operator()(int i, int j) const {
    return base[lm_tri_row(min(i, last)) + min(j, last)];
  }
  __device__ double2 pair(int i, int j) const {  // columns j (even), j + 1, both <= last
    return *reinterpret_cast<const double2 *>(base + lm_tri_row(min(i, last)) + j);
  }
};

// a * b + c: one fused multiply-add (the kernels' arithmetic, oracle order 1), or — REF, the
// reference's literal arithmetic (oracle order 0) — a rounded product, then the addition
template <bool REF>
__device__ inline double lm_mad(double a, double b, double c) {
  if constexpr (REF) return c + a * b;
  return __builtin_fma(a, b, c);
}
// REF: NLSG_LM_CHOLESKY_REFERENCE_ORDER — math::cholesky / forwardsolve_inplace /
// backsolve_inplace_t exactly as the reference rounds them (separate multiply and add; the
// back-substitution's sums from j = i+1 up to n-1, which makes it a serial sweep).
template <typename Rows, bool REF = false>
__device__ inline void lm_solve_cholesky_wave(Rows H, const double *g, double *upd, int n,
                                              bool off_upper = false) {
  const int t = lane_id();
  const bool row = t < n;
  // is_diagonal (:295-307): any off-diagonal above eps * 1e12 (positive values only). The
  // Gauss-Newton matrix is bitwise symmetric (fma chains of commuting products), so the lower
  // triangle decides; a finite-difference Hessian is not, and its evaluation hands over the
  // verdict on the upper triangle (`off_upper`). Reads past the lane's own row end stay inside
  // the buffer and are masked.
  bool off = off_upper;
  for (int j0 = 0; j0 < n; j0 += 8) {
    double h[8];
#pragma unroll
    for (int c = 0; c < 8; c++) h[c] = H(t, min(j0 + c, 63));
#pragma unroll
    for (int c = 0; c < 8; c++) off |= (j0 + c < t) & row & (h[c] > 2.220446049250313e-16 * 1e12);
  }
  const double gt = g[t];
  if (__ballot(off) == 0ull) {
    if (row) upd[t] = gt / H(t, t);
    return;
  }
  // cholesky (:251-269): every element's sum runs over k in order, each accumulate-multiply as
  // one fused multiply-add (the inner loops are bound by exactly these instructions; the oracle's
  // order-1 solve does the same, the reference's separate multiply and add differ in the last
  // bits). Columns are taken four at
  // a time so one LDS read of L[t][k] feeds four sums; the other factor L[j][k] is the same
  // read's value in lane j, broadcast through a scalar register (no second LDS read).
  for (int j0 = 0; j0 < n; j0 += 4) {
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    const bool act = row && t >= j0;
    // rows above the panel's sixteen-row group are finished: whole groups of sixteen lanes sit the
    // sums out. (A partly enabled wave issues its fp64 instructions cheaper — the SIMD skips the
    // sixteen-lane passes without an enabled lane: measured, 0.170 -> 0.157 ms for the step.)
    if (t >= (j0 & ~15))
    for (int k = 0; k < j0; k += 4) {  // j0 is a multiple of 4
      // (rows start at even offsets: a pair of columns is one 128-bit LDS read)
      const double2 xa = H.pair(t, k), xb = H.pair(t, k + 2);  // lanes t < j0: in-buffer, unused
      const double x[4] = {xa.x, xa.y, xb.x, xb.y};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        // L[j0+c][k..k+3]: a wave-uniform LDS address (broadcast reads on the otherwise idle LDS
        // port instead of scalar lane reads on the VALU port)
        const double2 la = H.pair(min(j0 + c, 63), k), lb = H.pair(min(j0 + c, 63), k + 2);
        const double l[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
        for (int q = 0; q < 4; q++) s4[c] = lm_mad<REF>(x[q], l[q], s4[c]);
      }
    }
    double hd[4];
#pragma unroll
    for (int c = 0; c < 4; c++) hd[c] = H(t, min(j0 + c, 63));
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int j = j0 + c;
      if (j < n) {  // wave-uniform
        // diagonal first (lane j: sum = sum_k L[j][k]^2), then the column below it. Root AND
        // reciprocal are taken by lane j alone — one enabled lane: one sixteen-lane pass per
        // instruction instead of four — and the reciprocal travels through scalar registers
        double v = 0.0, rinv = 0.0;
        if (t == j) {
          v = sqrt(hd[c] - s4[c]);
          rinv = 1.0 / v;
        }
        const double ri = lane_broadcast(rinv, j);
        if (act && t > j) v = (ri * (hd[c] - s4[c]));
        if (act && t >= j) H(t, j) = v;
        // the panel's later columns continue their sums with k = j
#pragma unroll
        for (int c2 = c + 1; c2 < 4; c2++)
          s4[c2] = lm_mad<REF>(v, lane_broadcast(v, min(j0 + c2, 63)), s4[c2]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const double dg = H(t, t);
  // forwardsolve_inplace (:282-294): column sweep, each row's sum grows in j order
  double sum = 0.0, u = 0.0;
  for (int j0 = 0; j0 < n; j0 += 4) {
    double h[4];
#pragma unroll
    for (int c = 0; c < 4; c++) h[c] = H(t, min(j0 + c, 63));
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int j = j0 + c;
      if (j < n) {
        if (t == j) u = (gt - sum) / dg;
        const double uj = lane_broadcast(u, j);
        if (row && t > j) sum = lm_mad<REF>(h[c], uj, sum);
      }
    }
  }
  if constexpr (REF) {
    // backsolve_inplace_t (:270-281) literally: component i from the sum over j = i+1 .. n-1 in
    // that order — the terms are ready side by side (lane j holds L[j][i] b[j]), their addition
    // walks the lanes
    for (int i = n - 1; i >= 0; i--) {
      const double term = H(max(t, i), i) * u;
      double acc = 0.0;
      for (int j = i + 1; j < n; j++) acc = acc + lane_broadcast(term, j);
      if (t == i) u = (u - acc) / dg;
    }
    if (row) upd[t] = u;
    return;
  }
  // backsolve_inplace_t (:270-281) with the inner sums taken from j = n-1 down to i+1
  sum = 0.0;
  for (int j0 = ((n - 1) | 3); j0 >= 0; j0 -= 4) {  // j0, j0-1, j0-2, j0-3
    double h[4];
#pragma unroll
    for (int c = 0; c < 4; c++) h[c] = H(j0 - c, min(t, j0 - c));  // row j, column t (t < j)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int j = j0 - c;
      if (j < n) {
        if (t == j) u = (u - sum) / dg;
        const double uj = lane_broadcast(u, j);
        if (row && t < j) sum = __builtin_fma(h[c], uj, sum);
      }
    }
  }
  if (row) upd[t] = u;
}

// tinyqr::lm on the damped matrix (tinyqr.h:253-310, 437-470): Givens QR in the reference's
// rotations (column j, rows bottom-up; same (a, b) -> (c, s), same element updates), executed as
// wavefronts: rotation (j, i) runs at step (n-1-i) + 2j, all rotations of a step touch disjoint
// row pairs, so every element sees exactly the sequence of updates the serial loop applies.
//
// What is kept, and where (one workgroup per problem):
//  * no Q. tinyqr::lm only ever uses Q through Q^T y (back_solve, :437-459), so the right-hand
//    side is rotated along with R as its column 64: w = G_k ... G_1 g. Mathematically the same
//    vector; the roundings differ from "accumulate Q, then multiply" (oracle order 1 mirrors the
//    co-rotation; the reference stays within the rounding tolerance test). Halves the LDS image
//    (33 KiB: four workgroups per CU) and more than halves the rotation work.
//  * an element update is one rounded product and one fused multiply-add,
//    lower' = fma(c, lower, s * upper), upper' = fma(c, upper, (-s) * lower) (oracle order 1
//    mirrors; tinyqr's two products and an add differ in the last bit).
//  * the eliminations of one column form a CHAIN that climbs one row per step: rotation (j, i)
//    writes row i-1, and the only rotation that reads that row next is (j, i-1), one step later.
//    The wave that owns chain j keeps that row in registers (`carry`) between steps: per
//    rotation one row is read from LDS and one written instead of two and two. Chains are dealt
//    to the seven APPLY waves by j mod 7; at most 32 are active at a time, so a wave follows at
//    most five of them (`slot`; chains j and j + 35 share one, never at the same time), each with
//    its own carry register, LDS offset and live-lane mask kept from step to step.
//  * the eighth wave computes the Givens pairs, one chain per lane, ONE STEP AHEAD of the apply
//    waves: the pair of rotation (j, i) needs a = R[i-1][j] and b = R[i][j]. b is the pivot
//    element of chain j's carried row, a the upper output of chain j-1's previous rotation in
//    chain j-1's first off-pivot column; the Givens lane of a chain tracks exactly those two
//    elements of its carried row (`bp`, `bq`, same arithmetic as the apply lanes, same bits) and
//    hands `a` to its neighbour lane. Everything it reads from LDS was written two steps back, so
//    step k's rotations and step k+1's Givens pairs run side by side: one barrier per step, and
//    the ~40 dependent fp64 instructions of a Givens pair are off the apply waves' path.
// Columns left of a chain's pivot hold annihilated remnants that nothing reads; they are skipped.
constexpr int kLmQrStride = 65;  // doubles per row: 64 columns + the right-hand side; odd: the
                                 // Givens lanes' gather (row and column both vary) is conflict-free
struct LmQrShared {              // LDS of the QR step (one workgroup per problem)
  double R[64 * kLmQrStride];    // working R (row-major), column 64 = co-rotated right-hand side
  double scr[64 + kLmQrStride];  // where the apply waves' lanes without a column load and store
  double2 cs[2][33];             // Givens pairs (c, s) by step parity and chain j mod 32; [32]: where
                                 // the Givens lanes without a rotation put theirs
  double upd[64];
};
static_assert(__builtin_offsetof(LmQrShared, scr) == 64 * kLmQrStride * sizeof(double), "scr follows R");

// PROBE (scripts/ubench/qr_phases.hip only): 1 = the apply waves keep their barriers and skip their
// rotations, 2 = the Givens wave keeps its barriers and skips its phase — what each side of a
// step costs when the other is absent. The product instantiates PROBE = 0.
template <int THREADS, int PROBE = 0>
__device__ inline void lm_solve_qr(LmQrShared &qs, int n) {
  constexpr int W = THREADS / 64, AW = W - 1, SLOTS = (32 + AW - 1) / AW, SPAN = AW * SLOTS;
  constexpr int S = kLmQrStride;
  static_assert(SPAN >= 32, "chains that share a slot must never be active together");
  const int t = threadIdx.x, lane = lane_id();
  // (Rotating the Givens role over the workgroup's waves with the workgroup index — in case the
  // four Givens waves of a CU's four workgroups shared a SIMD — changed nothing: 0.812 vs 0.815 ms.)
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int last = 2 * n - 4;  // last wavefront step
  // rotation of chain j at step k: rows i-1, i with i = n-1-(k-2j); exists for 2j <= k <= j+n-2

  if (wid == AW) {
    // ---- the Givens wave: lane L follows chain L, then chain L + 32. Its phase is one long
    // chain of dependent fp64 instructions (two divisions and a square root, IEEE-correct: ~45 of
    // them, ~1000 cycles with the LDS round trip and the barrier) and it is the workgroup's
    // critical path (raising its issue priority over the apply waves changed nothing: measured).
    // Around that chain the loop is kept short (112 instructions per step; 163 when the step window
    // and the tracked elements' addresses were rebuilt from (k, j) every step and the two halves sat
    // behind divergent branches): the window is one unsigned compare against per-lane constants,
    // the addresses are followed in bytes, and every lane computes a pair — lanes without a rotation
    // store theirs to a spare slot. Worth 1 % alone (off-chain instructions issue in the chain's
    // shadow), 4 % between the evaluation launches of a real iteration.
    char *const Rb = reinterpret_cast<char *>(qs.R);
    constexpr int S8 = S * 8;
    constexpr int kNever = 0x40000000;
    int j = lane;                          // current chain (lanes >= 32 never have one)
    auto has_chain = [&](int jj) { return lane < 32 && jj <= n - 2; };
    int kstart = has_chain(j) ? 2 * j : kNever;  // first step of my chain (kNever: no chain)
    int kspan = has_chain(j) ? n - 2 - j : 0;    // last step - first step (compared unsigned: never negative)
    int bst = ((n - 1) * S + (j < 63 ? j : 62)) * 8;  // R[n-1][j], R[n-1][j+1]: where my chain starts
    int t1off = 0;                         // R[i-1][j+1] of my chain's rotation at the step just applied
    int t1start = ((n - 2) * S + j + 1) * 8;  // ... at its first step (rows n-2, n-1)
    int a0off = (n - 2 > 0 ? n - 2 : 0) * S8;  // chain 0's a for the pair of step k+1 = R[n-3-k][0], k = -1
    double bp = 0.0, bq = 0.0, cp = 0.0, sp = 0.0;
    bool had = false;                      // my chain had a rotation at the step just applied
    int dprev = 0;                         // ... which was its step number dprev
    const int nowhere = (64 * S + 2 * lane) * 8;  // LmQrShared::scr, a pair per lane
    int pend = nowhere;                    // byte offset of R[j][j] of a finished chain still to be stored
    double pend_p = 0.0, pend_q = 0.0;
    for (int k = -1; k <= last + 1; k++) {  // one phase past the last step: the last chain's store
      if (PROBE != 2 && lane < 32) {  // (chains live in lanes 0 .. 31: half the sixteen-lane passes per instruction)
      // Every LDS read of the phase is issued here, before anything waits: the phase is a chain
      // of dependent fp64 instructions behind ONE read round trip (what a read returns is only
      // used where the comments below say so; the addresses are always inside R).
      const double t1q = *reinterpret_cast<const double *>(Rb + t1off);       // (1): row i-1 of step k, column j+1
      const double b_start = *reinterpret_cast<const double *>(Rb + bst);     // (2) if my chain starts: row n-1
      const double bq_start = *reinterpret_cast<const double *>(Rb + bst + 8);
      const double a_col0 = *reinterpret_cast<const double *>(Rb + a0off);    // (2) chain 0: a = R[i-1][0], i = n-2-k
      a0off = a0off >= S8 ? a0off - S8 : 0;
      // R[j][j], R[j][j+1] of the chain that ended in the previous phase — one phase late: the
      // apply lane of column j+1 read the old R[j][j+1] then (no chain ended: a scratch pair)
      *reinterpret_cast<double *>(Rb + pend) = pend_p;
      *reinterpret_cast<double *>(Rb + pend + 8) = pend_q;
      pend = nowhere;
      // (1) step k's rotation of my chain, on the two tracked columns j and j+1; its inputs
      //     were written by step k-1's rotations (complete: barrier)
      // (Computed by every lane, rotation or not: what a lane without one produces is never read —
      // chain j+1 has a rotation at step k+1 only if chain j had one at step k, and a chain
      // reloads its tracked elements when it starts — and the loop stays free of divergent
      // branches, whose merges cost more instructions than the arithmetic they would skip.)
      const double upq = __builtin_fma(cp, bq, (-sp) * t1q);  // = R[i][j+1] after the step: chain j+1's next a
      bq = __builtin_fma(cp, t1q, sp * bq);
      if (had && dprev == kspan) {  // the chain ended with this step: row j of R is final in these columns
        pend = (j * S + j) * 8;
        pend_p = bp;
        pend_q = bq;
        j += 32;
        kstart = has_chain(j) ? 2 * j : kNever;
        kspan = has_chain(j) ? n - 2 - j : 0;
        bst = ((n - 1) * S + (j < 63 ? j : 62)) * 8;
        t1start += 32 * 8;
      }
      // chain j's `a` is chain j-1's upq: lane L from lane L-1 (a DPP wave shift), lane 0 from lane 31
      double a_in = lane_up1(upq);
      const double a_wrap = lane_broadcast(upq, 31);
      if (lane == 0) a_in = a_wrap;
      // (2) the Givens pair of my chain's rotation at step k+1 (givens_rotation, tinyqr.h:86-97)
      const int d = k + 1 - kstart;
      had = static_cast<uint32_t>(d) <= static_cast<uint32_t>(kspan);
      dprev = d;
      const bool starts = d == 0;  // the chain starts: rows n-2, n-1
      const double b = starts ? b_start : bp;
      bq = starts ? bq_start : bq;
      const int t1next = starts ? t1start : t1off - S8;
      t1off = had ? t1next : t1off;
      const double a = j == 0 ? a_col0 : a_in;
      // both branches of the reference are r = small / large, t = 1 / sqrt(r^2 + 1), {t, t r}:
      // one division, one square root, one reciprocal — selected, not branched
      const bool swap = fabs(b) > fabs(a);
      const double r = (swap ? a : b) / (swap ? b : a);
      // r^2 + 1 lies in [1, 2] (or is NaN): the square root and the reciprocal without the
      // compiler's operand scaling and fix-up steps — nine instructions off the chain that is
      // the workgroup's critical path, same correctly rounded values
      const double tt = div_unscaled(1.0, sqrt_unscaled(r * r + 1.0));
      const double tr = tt * r;
      cp = swap ? tr : tt;
      sp = swap ? tt : tr;
      bp = __builtin_fma(cp, a, sp * b);  // the chain's new pivot element
      qs.cs[(k + 1) & 1][had ? (j & 31) : 32] = make_double2(cp, sp);
      }
      __syncthreads();
    }
  } else {
    // ---- an apply wave: rotate_matrix on rows i-1, i (tinyqr.h:126-139), one chain per slot.
    // WHAT BOUNDS THIS SIDE (round 4, scripts/ubench/qr_phases.hip + rocprofv3 SQ counters): not the
    // vector unit (40 % busy) and not the Givens wave (its side alone takes 56 us per round of four
    // problems on a CU whatever the residency) but the CU's ONE scalar unit: the apply waves of four
    // resident problems issued 281 scalar instructions per problem and step (13 per rotation, 4 per
    // idle slot: window compares, start / end tests, lane-mask copies), 78 % of what the unit can
    // issue, and the step stretched with every workgroup added (54.6 -> 103.6 us per round from one
    // to four problems per CU). So the per-slot bookkeeping is EVENT DRIVEN now:
    //  * a wave's chains j = wid, wid + AW, ... start at steps 2j — an arithmetic sequence — and
    //    end at steps j + n - 2 — another: two compares per step find "a chain of mine starts now"
    //    and "a chain of mine ended in the previous step"; the (rare) bodies open the chain in its
    //    slot (lane addresses, the carried row n-1 loaded right there) or store the finished row j
    //    (one step late: nobody reads it before the back-substitution) and flip the slot's bit in
    //    a wave-uniform mask;
    //  * the hot path of a slot is a bit test, two LDS reads, four fp64 instructions, one LDS
    //    write and one address decrement — no window arithmetic, no start / end tests.
    // A slot's per-step state is kept in vector registers: `off` the lane's LDS byte offset of
    // R[i-1][col] (minus one row per step), lanes without a column are pointed at a per-lane
    // scratch pair instead of being masked off; the step parity (which pair buffer) is a template
    // argument of an unrolled-by-two loop. Same rotations in the same order on the same bits.
    int off[SLOTS], dec[SLOTS], csb[SLOTS], fin[SLOTS];  // per lane: byte offsets into R / a pair buffer
    double carry[SLOTS];
    const int scratch = (64 * S + lane) * 8;  // LmQrShared::scr: [lane] and [lane + S]
    char *const Rb = reinterpret_cast<char *>(qs.R);
    constexpr int kNever = 0x40000000;
    uint32_t amask = 0;                       // bit sl: slot sl has a rotation at this step
    int js = wid, je = wid;                   // next chain of mine to start / to be finished
    int next_start = js <= n - 2 ? 2 * js : kNever;      // its first step
    int next_fin = je <= n - 2 ? je + n - 1 : kNever;    // the step after its last one
    uint32_t ss = 0, se = 0;                  // their slots (chain number mod SLOTS)
    auto on_slot = [&](uint32_t sl, auto &&f) {  // f(int_c<sl>) for a wave-uniform sl: static register indices
      if constexpr (SLOTS > 0) if (sl == 0) f(int_c<0>{});
      if constexpr (SLOTS > 1) if (sl == 1) f(int_c<1>{});
      if constexpr (SLOTS > 2) if (sl == 2) f(int_c<2>{});
      if constexpr (SLOTS > 3) if (sl == 3) f(int_c<3>{});
      if constexpr (SLOTS > 4) if (sl == 4) f(int_c<4>{});
      if constexpr (SLOTS > 5) if (sl == 5) f(int_c<5>{});
      if constexpr (SLOTS > 6) if (sl == 6) f(int_c<6>{});
      if constexpr (SLOTS > 7) if (sl == 7) f(int_c<7>{});
    };
    static_assert(SLOTS <= 8, "on_slot covers eight slots");
#pragma unroll
    for (int sl = 0; sl < SLOTS; sl++) {
      off[sl] = scratch;
      dec[sl] = 0;
      csb[sl] = 0;
      fin[sl] = scratch;
      carry[sl] = 0.0;
    }
    auto events = [&](int k) {
      if (k == next_fin) {  // chain je ended in the previous step: row je of R is final (its columns
                            // je, je+1 come from the Givens lane)
        on_slot(se, [&](auto c) {
          constexpr int sl = decltype(c)::value;
          *reinterpret_cast<double *>(Rb + fin[sl]) = carry[sl];
          amask &= ~(1u << sl);
        });
        je += AW;
        se = se + 1 == SLOTS ? 0 : se + 1;
        next_fin = je <= n - 2 ? je + n - 1 : kNever;
      }
      if (k == next_start) {  // chain js starts: rows n-2, n-1
        on_slot(ss, [&](auto c) {
          constexpr int sl = decltype(c)::value;
          const int col = js + 1 + lane;
          const bool live = col < n || col == 64;
          off[sl] = live ? ((n - 2) * S + col) * 8 : scratch;  // R[i-1][col] at the first step (i = n-1)
          dec[sl] = live ? S * 8 : 0;
          fin[sl] = live && lane != 0 ? (js * S + col) * 8 : scratch;  // where the finished row goes
          int cb = (js & 31) * 16;
          asm volatile("" : "+v"(cb));  // (a wave-uniform value, kept in a vector register on purpose)
          csb[sl] = cb;
          carry[sl] = *reinterpret_cast<const double *>(Rb + off[sl] + S * 8);  // row n-1
          amask |= 1u << sl;
        });
        js += AW;
        ss = ss + 1 == SLOTS ? 0 : ss + 1;
        next_start = js <= n - 2 ? 2 * js : kNever;
      }
    };
    __syncthreads();  // the prologue phase of the Givens wave (k = -1)
    auto step = [&](auto parity, int k) {
      events(k);
      const char *csk = reinterpret_cast<const char *>(qs.cs[decltype(parity)::value]);
#pragma unroll
      for (int sl = 0; sl < SLOTS; sl++) {
        if (PROBE != 1 && (amask >> sl & 1u)) {  // wave-uniform
          const double2 g = *reinterpret_cast<const double2 *>(csk + csb[sl]);
          const double t1 = *reinterpret_cast<const double *>(Rb + off[sl]);
          const double c = g.x, sv = g.y, t2 = carry[sl];
          const double lo = __builtin_fma(c, t1, sv * t2);
          *reinterpret_cast<double *>(Rb + off[sl] + S * 8) = __builtin_fma(c, t2, (-sv) * t1);
          carry[sl] = lo;
          off[sl] -= dec[sl];
        }
      }
      __syncthreads();
    };
    int k = 0;
    for (; k + 1 <= last; k += 2) {
      step(int_c<0>{}, k);
      step(int_c<1>{}, k + 1);
    }
    if (k <= last) {
      step(int_c<0>{}, k);
      k++;
    }
    events(k);  // k = last + 1: the last chain's finished row
    __syncthreads();  // the Givens wave's last phase
  }
  // back_solve (tinyqr.h:437-459) on R x = w, with lm()'s cleanup (tol = 1e-12, :278-282, 465)
  // applied to the entries it reads; inner sums taken from j = n-1 down to i+1 (oracle order 1)
  if (t < 64) {
    const double w = t < n ? qs.R[t * S + 64] : 0.0;
    double temp = 0.0, u = 0.0;
    for (int j0 = ((n - 1) | 3); j0 >= 0; j0 -= 4) {
      double h[4], d[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = min(j0 - q, n - 1);
        h[q] = qs.R[min(t, j) * S + j];
        d[q] = qs.R[j * S + j];
        h[q] = fabs(h[q]) < 1e-12 ? 0.0 : h[q];
        d[q] = fabs(d[q]) < 1e-12 ? 0.0 : d[q];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = j0 - q;
        if (j < n) {  // wave-uniform
          if (t == j) u = (w - temp) / d[q];
          const double uj = lane_broadcast(u, j);
          if (t < j) temp += h[q] * uj;
        }
      }
    }
    if (t < n) qs.upd[t] = u;
  }
  __syncthreads();
}

// ---- Cholesky solver: all problems advance in lock step, ONE WAVE per problem, one launch per
// iteration (lm_iter_kernel = step k, then evaluation k + 1). Running both in one launch lets a
// wave's latency-bound solve sit beside the other waves' MFMA phases instead of in a kernel of
// its own, and saves a launch; H and g still travel through global memory between launches.
// lm_eval_wave         f, g, H at theta. The wave streams its m x 64
//                      block of A sixteen rows at a time (next sixteen in flight meanwhile),
//                      scales them into a wave-private LDS tile and feeds all ten lower
//                      16 x 16 tiles of J^T J from it. No workgroup barrier anywhere: the
//                      waves of a SIMD drift apart, so one wave's load / tanh phase is covered
//                      by the others' MFMA phases (fp64 MFMA and fp64 VALU share the DP pipe,
//                      the bound is their sum).
// lm_step_wave         stop tests, damping, Cholesky solve, theta update
// Reductions: rows -> k-steps of 4, f partials per (row/16 % 4, row parity), g partials per
// row % 4 -- the order oracle_lm.c's order-1 evaluation mirrors.
struct LmWaveShared {  // view of the wave's LDS during the evaluation
  double *J;           // [16][kLmJStride]
  double *r;           // [16]
};

// the iteration's bookkeeping after an evaluation (one lane)
__device__ inline void lm_publish_state(const LmParams &p, LmProblem *pr, int first, double f) {
  if (first) {  // g, H, f at x0 (:3513-3516)
    pr->prev = 0.0;
    pr->f = f;
    pr->lambda = p.lambda0;
    pr->iter = 0;
    pr->fcalls = 1;
    pr->done = 0;
  } else {  // :3535-3542
    const double prev = pr->f;
    pr->prev = prev;
    pr->f = f;
    pr->fcalls += 1;
    pr->iter += 1;
    pr->lambda = f < prev ? pr->lambda / p.down : pr->lambda * p.up;
  }
}

// theta_lds: the parameters as the step left them in LDS (nullptr: read them from global)
__device__ inline void lm_eval_wave(const LmParams &p, int first, uint64_t pid, LmWaveShared sh,
                                    const double *theta_lds) {
  LmProblem *pr = p.prob + pid;
  const int lane = threadIdx.x;
  const int half = lane >> 5, lp = lane & 31, kk = lane >> 4, cc = lane & 15;
  const double *theta = theta_lds ? theta_lds : p.theta + pid * kLmN;
  const double th0 = theta[2 * lp], th1 = theta[2 * lp + 1];
  // device layout of A: [row group s][problem][16 rows][64], zero padded past m; y alike
  const double *Ap = p.A + pid * (16 * kLmN) + 2 * lp;
  const double *yp = p.y + pid * 16 + 2 * (lp >> 2) + half;  // the row whose z this lane ends up with
  const uint64_t stride = p.batch * (16 * kLmN), ystride = p.batch * 16;
  v4d acc[10];
#pragma unroll
  for (int i = 0; i < 10; i++) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  double gacc[4] = {0.0, 0.0, 0.0, 0.0}, facc[4] = {0.0, 0.0, 0.0, 0.0};
  const uint64_t nstep = p.nstep;
  double2 a[8];
  double ysel;
  auto fetch = [&](uint64_t s) {
#pragma unroll
    for (int k = 0; k < 8; k++) {  // the design matrix is read once per evaluation: streamed (nt)
      const v2d_nt v = __builtin_nontemporal_load(
          reinterpret_cast<const v2d_nt *>(Ap + s * stride + (2 * k + half) * kLmN));
      a[k] = make_double2(v.x, v.y);
    }
    ysel = yp[s * ystride];
  };
  auto step = [&](uint64_t s, double &fw) {
    // z = A theta for 16 rows (8 per half): a reduce-scatter over the half's 32 lanes instead of
    // eight full butterflies — at the levels 16, 8, 4 a lane keeps half of its rows (by bit 4, 3,
    // 2 of its index) and hands the other half to its partner, then the last row standing takes
    // the levels 2 and 1. Every sum pairs the same lanes in the same order as the butterfly did
    // (own + partner's, commutative), so z has the same bits; 9 exchanges instead of 40.
    double z[8];
#pragma unroll
    for (int k = 0; k < 8; k++) z[k] = __builtin_fma(a[k].y, th1, a[k].x * th0);
    const bool b4 = (lp & 16) != 0, b3 = (lp & 8) != 0, b2 = (lp & 4) != 0;
    double y4[4], y2[2];
#pragma unroll
    for (int q = 0; q < 4; q++)
      y4[q] = (b4 ? z[q + 4] : z[q]) + lane_xor<16>(b4 ? z[q] : z[q + 4]);
#pragma unroll
    for (int q = 0; q < 2; q++)
      y2[q] = (b3 ? y4[q + 2] : y4[q]) + lane_xor<8>(b3 ? y4[q] : y4[q + 2]);
    double zsel = (b2 ? y2[1] : y2[0]) + lane_xor<4>(b2 ? y2[0] : y2[1]);
    zsel = zsel + lane_xor<2>(zsel);
    zsel = zsel + lane_xor<1>(zsel);
    // tanh / residual / weight once per row: the lanes of a half with the same lp >> 2 hold row
    // k = lp >> 2 (rows 2k + half of the group)
    const double tsel = det_tanh(zsel);
    const double rsel = ysel - tsel;
    const double wsel = 1 - tsel * tsel;
    if ((lp & 3) == 0) sh.r[2 * (lp >> 2) + half] = rsel;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const double r = __shfl(rsel, 32 * half + 4 * k, 64);
      const double wgt = __shfl(wsel, 32 * half + 4 * k, 64);
      fw = __builtin_fma(r, r, fw);
      double2 jv;
      jv.x = -(wgt * a[k].x);
      jv.y = -(wgt * a[k].y);
      *reinterpret_cast<double2 *>(&sh.J[(2 * k + half) * kLmJStride + 2 * lp]) = jv;
    }
    // the next sixteen rows travel from HBM while these are on the matrix cores
    if (s + 1 < nstep) fetch(s + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- J^T J on the matrix cores, J^T r on the VALU (same operands); for
    // v_mfma_f64_16x16x4 the A operand of column block b (A[i][k] = J[k][16b+i]) and its B
    // operand (B[k][j] = J[k][16b+j]) are the same register
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double *row = &sh.J[(4 * q + kk) * kLmJStride];
      const double rv = sh.r[4 * q + kk];
      double op[4];
#pragma unroll
      for (int b = 0; b < 4; b++) op[b] = row[16 * b + cc];
#pragma unroll
      for (int b = 0; b < 4; b++) gacc[b] = __builtin_fma(op[b], rv, gacc[b]);
#pragma unroll
      for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int cb = 0; cb <= rb; cb++)
          acc[rb * (rb + 1) / 2 + cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(
              op[rb], op[cb], acc[rb * (rb + 1) / 2 + cb], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  fetch(0);
  for (uint64_t s0 = 0; s0 < nstep; s0 += 4) {
    step(s0, facc[0]);
    if (s0 + 1 < nstep) step(s0 + 1, facc[1]);
    if (s0 + 2 < nstep) step(s0 + 2, facc[2]);
    if (s0 + 3 < nstep) step(s0 + 3, facc[3]);
  }
  // ---- publish the lower triangle of H = 2 J^T J, g = 2 J^T r, f
#pragma unroll
  for (int rb = 0; rb < 4; rb++)
#pragma unroll
    for (int cb = 0; cb <= rb; cb++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int row = 16 * rb + kk + 4 * rg, col = 16 * cb + cc;
        if (col <= row)
          p.Hg[pid * kLmTri + lm_tri_row(row) + col] = 2 * acc[rb * (rb + 1) / 2 + cb][rg];
      }
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const double g0 = __shfl(gacc[b], cc, 64), g1 = __shfl(gacc[b], cc + 16, 64);
    const double g2 = __shfl(gacc[b], cc + 32, 64), g3 = __shfl(gacc[b], cc + 48, 64);
    if (kk == 0) p.gg[pid * kLmN + 16 * b + cc] = 2 * (((g0 + g1) + g2) + g3);
  }
  double f = 0.0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    f = f + __shfl(facc[w], 0, 64);
    f = f + __shfl(facc[w], 32, 64);
  }
  if (lane == 0) lm_publish_state(p, pr, first, f);
}

struct LmStepShared {  // view of the wave's LDS during the step
  double *tri;         // [kLmTri]
  double *g, *upd;     // [64] each; upd ends up holding the new parameters
};

// false: a stop test fired (the problem is done). FD = the finite-difference model: the LDS image
// of the triangle ends with row n - 1 (`chunks` x 64 doubles), and is_diagonal also needs the
// evaluation's verdict on the upper triangle.
template <bool FD, bool REF = false>
__device__ inline bool lm_step_wave(const LmParams &p, uint64_t pid, LmStepShared sh, int chunks = 33) {
  LmProblem *pr = p.prob + pid;
  const int t = threadIdx.x, n = static_cast<int>(p.n);
  const double prev = pr->prev, cur = pr->f;
  if (pr->iter >= p.max_iter || fabs(prev - cur) < p.f_delta || isnan(prev)) {  // :3520-3527
    if (t == 0) pr->done = 1;
    return false;
  }
  const double *src = p.Hg + pid * kLmTri;
  if constexpr (FD) {
    for (int q = 0; q < chunks; q++) sh.tri[64 * q + t] = src[64 * q + t];
    sh.g[t] = p.gg[pid * kLmN + t];
  } else {  // all loads in flight before the first LDS write (the pad holds stale, unused values)
    double h[33];
#pragma unroll
    for (int q = 0; q < 33; q++) h[q] = src[64 * q + t];
    const double gv = p.gg[pid * kLmN + t];
#pragma unroll
    for (int q = 0; q < 33; q++) sh.tri[64 * q + t] = h[q];
    sh.g[t] = gv;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (t < n) sh.tri[lm_tri_row(t) + t] += pr->lambda;  // :3529-3531
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if constexpr (FD)
    lm_solve_cholesky_wave<LmRowsTriShort, REF>(LmRowsTriShort{sh.tri, n - 1}, sh.g, sh.upd, n,
                                                pr->upper != 0);
  else
    lm_solve_cholesky_wave(LmRowsTri{sh.tri}, sh.g, sh.upd, n);
  const double th = p.theta[pid * kLmN + t];
  const double tn = t < n ? th - sh.upd[t] : th;  // :3534
  p.theta[pid * kLmN + t] = tn;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  sh.upd[t] = tn;  // handed to the evaluation of the same launch
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return true;
}

// One launch per iteration: step k (skipped on the first launch), then evaluation k + 1.
// LDS of the wave: the step's packed triangle | g | upd, overlaid by the evaluation's Jacobian
// tile | r (the new parameters wait in `upd`, which the tile does not reach).
// `with_step` = 0: evaluation only (the first launch, and every launch of the QR pipeline, whose
// step is lm_qr_step_kernel).
__global__ __launch_bounds__(64, 2) void lm_iter_kernel(LmParams p, int first, int with_step) {
  __shared__ __align__(16) double smem[kLmTri + 128];
  static_assert(16 * kLmJStride + 16 <= kLmTri + 64, "the Jacobian tile must not reach upd");
  const uint64_t pid = blockIdx.x;
  const double *theta_lds = nullptr;
  if (!first) {
    if (p.prob[pid].done) return;
    if (with_step) {
      if (!lm_step_wave<false>(p, pid, LmStepShared{smem, smem + kLmTri, smem + kLmTri + 64})) return;
      theta_lds = smem + kLmTri + 64;
    }
  }
  lm_eval_wave(p, first, pid, LmWaveShared{smem, smem + 16 * kLmJStride}, theta_lds);
}

// ---- The reference's default functors on a built-in objective: when the
// caller gives LevenbergMarquardt no Grad / Hess, solve() builds them from finite differences
// (nlsolver.h:3494-3511 -> fin_diff :1385-1413 and fin_diff_h :1446-1515, both at accuracy 1):
// f (1 probe), the gradient (4 n probes), the Hessian (16 n^2 probes) per evaluation, one wave
// per problem.
//
// A point of n <= 64 coordinates fills at most 32 lanes, so the wave evaluates P = 64 / G
// probe points at once, each in a group of G lanes (G = 4, 8, 16 or 32 for n <= 8, 16, 32, 64;
// lane g of a group holds x[2g], x[2g+1]). group_objective is wave_objective restricted to a
// group: the same per-lane partial and the levels G/2 .. 1 of the same butterfly; the levels
// it drops only ever add the zeros of the unused lanes, so every probe value has the bits the
// full-wave tree (oracle order 1) gives.
//
// fin_diff_h moves x[i] and x[j] through a fixed sequence of += / -= steps between its sixteen
// probes of an entry. The sequence is replayed literally on two wave-uniform scalars (one when
// i == j, where both names are the same element); group k % P captures the pair at probe k,
// 16 / P passes evaluate all sixteen, and the weighted sums are taken in the reference's order
// from the broadcast values. The result is not symmetric; Cholesky and the substitutions read
// the lower triangle only (:251-294), is_diagonal reads both (:295-307): the lower triangle goes
// to Hg, the upper one is reduced to its verdict.
// REF (NLSG_LM_CHOLESKY_REFERENCE_ORDER): every probe sums its objective in index order
// (group_objective_seq), which makes gradient and Hessian the reference's own bit for bit.
template <int OBJ, int G, bool REF = false>
__device__ inline void lm_fd_eval_groups(const LmParams &p, int first, uint64_t pid,
                                         const double *theta_lds) {
  auto objective = [](double a, double b, uint64_t D) {
    if constexpr (REF) return group_objective_seq<OBJ, G>(a, b, D);
    else return group_objective<OBJ, G>(a, b, D);
  };
  constexpr int P = 64 / G;                  // probe points per pass
  constexpr int HP = P >= 16 ? 1 : 16 / P;   // passes per Hessian entry
  LmProblem *pr = p.prob + pid;
  const int lane = threadIdx.x, n = static_cast<int>(p.n);
  const int g = lane & (G - 1), gi = lane / G;
  const double *theta = theta_lds ? theta_lds : p.theta + pid * kLmN;
  const double x0 = theta[2 * g], x1 = theta[2 * g + 1];  // zero past n
  const double f = lane_broadcast(objective(x0, x1, n), 0);
  {  // fin_diff<1>: coeff {1,-8,8,-1}, coeff2 {-2,-1,1,2}, eps = DBL_EPSILON * 10e7; probe
     // q = 4 d + s of the 4 n probes runs in group q % P of pass q / P
    constexpr double eps = 2.220446049250313e-16 * 10e7;
    constexpr double dd_val = 12 * eps;
    double gl = 0.0, acc = 0.0;
    for (int q0 = 0; q0 < 4 * n; q0 += P) {
      const int q = q0 + gi, d = q >> 2, sq = q & 3;
      const double c2 = sq == 0 ? -2.0 : sq == 1 ? -1.0 : sq == 2 ? 1.0 : 2.0;
      const double xp0 = d == 2 * g ? x0 + c2 * eps : x0;
      const double xp1 = d == 2 * g + 1 ? x1 + c2 * eps : x1;
      const double fv = objective(xp0, xp1, n);
#pragma unroll
      for (int u = 0; u < P; u++) {
        const int qu = q0 + u, su = qu & 3;  // wave-uniform
        if (qu < 4 * n) {
          const double fq = lane_broadcast(fv, u * G);
          const double c = su == 0 ? 1.0 : su == 1 ? -8.0 : su == 2 ? 8.0 : -1.0;
          acc = (su == 0 ? 0.0 : acc) + c * fq;
          if (su == 3 && lane == (qu >> 2)) gl = acc / dd_val;
        }
      }
    }
    p.gg[pid * kLmN + lane] = gl;
  }
  // fin_diff_h<1>
  const double e1 = p.eps_h, e2 = 2 * e1, e3 = 3 * e1, e4 = 4 * e1;
  const double denom = (600.0 * e1 * e1);
  bool upper = false;
  for (int i = 0; i < n; i++) {
    const double ti = lane_broadcast((i & 1) ? x1 : x0, i >> 1);
    const bool mi0 = 2 * g == i, mi1 = 2 * g + 1 == i;
    double hrow = 0.0;
    for (int j = 0; j < n; j++) {
      const bool mj0 = 2 * g == j, mj1 = 2 * g + 1 == j;
      const bool same = i == j;
      double xi = ti, xj = lane_broadcast((j & 1) ? x1 : x0, j >> 1);
      double ci[HP], cj[HP];  // the pair this lane's group probes in each pass
#pragma unroll
      for (int h = 0; h < HP; h++) ci[h] = cj[h] = 0.0;
      auto add_i = [&](double d) { xi = xi + d; xj = same ? xi : xj; };
      auto sub_i = [&](double d) { xi = xi - d; xj = same ? xi : xj; };
      auto add_j = [&](double d) { xj = xj + d; xi = same ? xj : xi; };
      auto sub_j = [&](double d) { xj = xj - d; xi = same ? xj : xi; };
      auto at = [&](auto k) {  // probe k is taken here
        constexpr int K = decltype(k)::value;
        const bool mine = gi == K % P;
        ci[K / P] = mine ? xi : ci[K / P];
        cj[K / P] = mine ? xj : cj[K / P];
      };
      add_i(e1); sub_j(e2); at(int_c<0>{});
      add_i(e1); add_j(e1); at(int_c<1>{});
      sub_i(e4); add_j(e2); at(int_c<2>{});
      add_i(e1); add_j(e1); at(int_c<3>{});
      sub_j(e4); at(int_c<4>{});
      sub_i(e1); add_j(e1); at(int_c<5>{});
      add_i(e3); add_j(e3); at(int_c<6>{});
      add_i(e1); sub_j(e1); at(int_c<7>{});
      sub_j(e3); at(int_c<8>{});
      sub_i(e4); add_j(e4); at(int_c<9>{});
      sub_j(e4); at(int_c<10>{});
      add_i(e4); add_j(e4); at(int_c<11>{});
      sub_i(e3); sub_j(e3); at(int_c<12>{});
      add_i(e2); add_j(e2); at(int_c<13>{});
      sub_j(e2); at(int_c<14>{});
      sub_i(e2); add_j(e2); at(int_c<15>{});
      double fv[HP];
#pragma unroll
      for (int h = 0; h < HP; h++) {
        const double xp0 = mj0 ? cj[h] : mi0 ? ci[h] : x0;
        const double xp1 = mj1 ? cj[h] : mi1 ? ci[h] : x1;
        fv[h] = objective(xp0, xp1, n);
      }
      auto probe = [&](auto k) {
        constexpr int K = decltype(k)::value;
        return lane_broadcast(fv[K / P], (K % P) * G);
      };
      double result = 0.0, temp = 0.0;
      temp = temp + probe(int_c<0>{});
      temp = temp + probe(int_c<1>{});
      temp = temp + probe(int_c<2>{});
      temp = temp + probe(int_c<3>{});
      result = result - 63 * temp;
      temp = 0.0;
      temp = temp + probe(int_c<4>{});
      temp = temp + probe(int_c<5>{});
      temp = temp + probe(int_c<6>{});
      temp = temp + probe(int_c<7>{});
      result = result + 63 * temp;
      temp = 0.0;
      temp = temp + probe(int_c<8>{});
      temp = temp + probe(int_c<9>{});
      temp = temp - probe(int_c<10>{});
      temp = temp - probe(int_c<11>{});
      result = result + 44 * temp;
      temp = 0.0;
      temp = temp + probe(int_c<12>{});
      temp = temp + probe(int_c<13>{});
      temp = temp - probe(int_c<14>{});
      temp = temp - probe(int_c<15>{});
      result = result + 74 * temp;
      const double hij = result / denom;
      hrow = lane == j ? hij : hrow;
    }
    if (lane <= i) p.Hg[pid * kLmTri + lm_tri_row(i) + lane] = hrow;
    upper |= lane > i && lane < n && hrow > 2.220446049250313e-16 * 1e12;
  }
  const bool any_upper = __ballot(upper) != 0ull;
  if (lane == 0) {
    pr->upper = any_upper ? 1 : 0;
    lm_publish_state(p, pr, first, f);
  }
}

// The same evaluation in REFERENCE ORDER (NLSG_LM_CHOLESKY_REFERENCE_ORDER), a probe per LANE. Every
// probe is the objective at the base point with one or two coordinates moved, its terms added in
// index order; only the terms that contain a moved coordinate differ from the base point's. So the
// base terms t_e are computed once and sit in LDS, and a lane walks them at a wave-uniform address,
// substituting its own few modified terms where they fall:
//  * fin_diff: lane d = coordinate d, its four probes side by side; they start from the base point's
//    prefix sum (one serial chain per gradient, captured on the way) and add the tail;
//  * fin_diff_h: row i at a time, lane j = entry (i, j), its sixteen probes as two batches of eight
//    chains; the moved pair comes from the literal += / -= replay (per lane: i == j aliases the
//    two), the up to four modified terms (e = i-1, i, j-1, j for a chain objective) are computed
//    up front, and the walk over e picks per lane between them and the base term.
// ~30 instructions per probe instead of ~n / G * 3 + 60 for a serial sum by a group of lanes: the
// reference-order evaluation is as fast as the tree-order one at n = 16 and faster past it, and has
// the reference's bits (tests/golden/lm_fd.json; oracle order 0).
// xs: 65 doubles (x, then a zero), ts: 64 doubles.
template <int OBJ>
__device__ inline void lm_fd_eval_lanes(const LmParams &p, int first, uint64_t pid, const double *theta,
                                        double *xs, double *ts) {
  using O = Objective<OBJ>;
  LmProblem *pr = p.prob + pid;
  const int lane = threadIdx.x, n = static_cast<int>(p.n), nt = static_cast<int>(O::n_terms(p.n));
  const double xl = theta[lane];  // zero past n
  xs[lane] = xl;
  if (lane == 0) xs[64] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double xnext = xs[lane + 1], xprev = xs[lane > 0 ? lane - 1 : 0];
  ts[lane] = O::term(xl, xnext);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // f, and on the way the prefix sum each coordinate's probes start from
  constexpr int off = O::kChain ? 2 : 1;
  double run = 0.0, start = 0.0;
  for (int e = 0; e < nt; e++) {
    run = run + ts[e];
    start = e == lane - off ? run : start;
  }
  const double f = O::finish(run, p.n);
  {  // fin_diff<1>: coeff {1,-8,8,-1}, coeff2 {-2,-1,1,2}, eps = DBL_EPSILON * 10e7
    constexpr double eps = 2.220446049250313e-16 * 10e7;
    constexpr double coeff[4] = {1, -8, 8, -1}, coeff2[4] = {-2, -1, 1, 2};
    constexpr double dd_val = 12 * eps;
    const int d = lane;
    double acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double xq = xl + coeff2[q] * eps;
      acc[q] = start;
      if constexpr (O::kChain) {
        const double m0 = acc[q] + O::term(xprev, xq);
        acc[q] = d >= 1 ? m0 : acc[q];
        const double m1 = acc[q] + O::term(xq, xnext);
        acc[q] = d < nt ? m1 : acc[q];
      } else {
        acc[q] = acc[q] + O::term(xq, 0.0);
      }
    }
    for (int e = 1; e < nt; e++) {
      const double te = ts[e];
      if (e > d) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = acc[q] + te;
      }
    }
    double ga = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) ga = ga + coeff[q] * O::finish(acc[q], p.n);
    p.gg[pid * kLmN + lane] = d < n ? ga / dd_val : 0.0;
  }
  // fin_diff_h<1>
  const double e1 = p.eps_h, e2 = 2 * e1, e3 = 3 * e1, e4 = 4 * e1;
  const double denom = (600.0 * e1 * e1);
  const int j = lane;
  const double xj0 = xl, xjm = xprev, xjp = xnext;
  bool upper = false;
  for (int i = 0; i < n; i++) {
    const double xi0 = xs[i], xim = xs[i > 0 ? i - 1 : 0], xip = xs[i + 1];
    const bool same = i == j;
    double fv[16];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      // the pairs of this half's eight probes: the literal sequence, captured where a probe is taken
      double ci[8], cj[8];
      {
        double xi = xi0, xj = xj0;
        auto add_i = [&](double d) { xi = xi + d; xj = same ? xi : xj; };
        auto sub_i = [&](double d) { xi = xi - d; xj = same ? xi : xj; };
        auto add_j = [&](double d) { xj = xj + d; xi = same ? xj : xi; };
        auto sub_j = [&](double d) { xj = xj - d; xi = same ? xj : xi; };
        auto at = [&](auto k) {
          constexpr int K = decltype(k)::value;
          if constexpr (K / 8 == 0) { if (half == 0) { ci[K % 8] = xi; cj[K % 8] = xj; } }
          else { if (half == 1) { ci[K % 8] = xi; cj[K % 8] = xj; } }
        };
        add_i(e1); sub_j(e2); at(int_c<0>{});
        add_i(e1); add_j(e1); at(int_c<1>{});
        sub_i(e4); add_j(e2); at(int_c<2>{});
        add_i(e1); add_j(e1); at(int_c<3>{});
        sub_j(e4); at(int_c<4>{});
        sub_i(e1); add_j(e1); at(int_c<5>{});
        add_i(e3); add_j(e3); at(int_c<6>{});
        add_i(e1); sub_j(e1); at(int_c<7>{});
        sub_j(e3); at(int_c<8>{});
        sub_i(e4); add_j(e4); at(int_c<9>{});
        sub_j(e4); at(int_c<10>{});
        add_i(e4); add_j(e4); at(int_c<11>{});
        sub_i(e3); sub_j(e3); at(int_c<12>{});
        add_i(e2); add_j(e2); at(int_c<13>{});
        sub_j(e2); at(int_c<14>{});
        sub_i(e2); add_j(e2); at(int_c<15>{});
      }
      // the modified terms: x_j -> cj takes precedence over x_i -> ci (they are equal when i == j)
      double A[8], B[8], C[8], D[8];  // terms i-1, i, j-1, j
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const double vi = same ? cj[k] : ci[k];                    // x_i of the probe
        if constexpr (O::kChain) {
          const double vim = i - 1 == j ? cj[k] : xim;             // x_{i-1}
          const double vip = i + 1 == j ? cj[k] : xip;             // x_{i+1}
          const double vjm = j - 1 == i ? ci[k] : xjm;             // x_{j-1}
          const double vjp = j + 1 == i ? ci[k] : xjp;             // x_{j+1}
          A[k] = O::term(vim, vi);
          B[k] = O::term(vi, vip);
          C[k] = O::term(vjm, cj[k]);
          D[k] = O::term(cj[k], vjp);
        } else {
          A[k] = C[k] = 0.0;
          B[k] = O::term(vi, 0.0);
          D[k] = O::term(cj[k], 0.0);
        }
      }
      double acc[8];
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] = 0.0;
      for (int e = 0; e < nt; e++) {
        const double te = ts[e];
        const bool at_j = e == j, at_jm = O::kChain && e == j - 1;
        if (O::kChain && e == i - 1) {  // (wave-uniform branches)
#pragma unroll
          for (int k = 0; k < 8; k++) acc[k] = acc[k] + (at_j ? D[k] : at_jm ? C[k] : A[k]);
        } else if (e == i) {
#pragma unroll
          for (int k = 0; k < 8; k++) acc[k] = acc[k] + (at_j ? D[k] : at_jm ? C[k] : B[k]);
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) acc[k] = acc[k] + (at_j ? D[k] : at_jm ? C[k] : te);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) fv[8 * half + k] = O::finish(acc[k], p.n);
    }
    double result = 0.0, temp = 0.0;
    temp = temp + fv[0];
    temp = temp + fv[1];
    temp = temp + fv[2];
    temp = temp + fv[3];
    result = result - 63 * temp;
    temp = 0.0;
    temp = temp + fv[4];
    temp = temp + fv[5];
    temp = temp + fv[6];
    temp = temp + fv[7];
    result = result + 63 * temp;
    temp = 0.0;
    temp = temp + fv[8];
    temp = temp + fv[9];
    temp = temp - fv[10];
    temp = temp - fv[11];
    result = result + 44 * temp;
    temp = 0.0;
    temp = temp + fv[12];
    temp = temp + fv[13];
    temp = temp - fv[14];
    temp = temp - fv[15];
    result = result + 74 * temp;
    const double hij = result / denom;
    if (lane <= i) p.Hg[pid * kLmTri + lm_tri_row(i) + lane] = hij;
    upper |= lane > i && lane < n && hij > 2.220446049250313e-16 * 1e12;
  }
  const bool any_upper = __ballot(upper) != 0ull;
  if (lane == 0) {
    pr->upper = any_upper ? 1 : 0;
    lm_publish_state(p, pr, first, f);
  }
}

// LDS of the wave: `chunks` x 64 doubles of the triangle (rows 0 .. n-1) | g | upd
__host__ __device__ inline int lm_fd_chunks(uint64_t n) {
  return (lm_tri_row(static_cast<int>(n)) + 63) / 64;
}
template <int OBJ, bool REF = false>
__global__ __launch_bounds__(64) void lm_fd_iter_kernel(LmParams p, int first) {
  extern __shared__ __align__(16) double lm_fd_smem[];
  const uint64_t pid = blockIdx.x;
  const double *theta_lds = nullptr;
  if (!first) {
    if (p.prob[pid].done) return;
    const int chunks = lm_fd_chunks(p.n);
    double *g = lm_fd_smem + 64 * chunks;
    if (!lm_step_wave<true, REF>(p, pid, LmStepShared{lm_fd_smem, g, g + 64}, chunks)) return;
    theta_lds = g + 64;
  }
  if constexpr (REF && !Objective<OBJ>::kWhole) {
    // (the triangle's LDS image is dead during the evaluation; its first 64 doubles and the 65 past
    // g | upd hold the base terms and the point)
    double *xs = lm_fd_smem + 64 * lm_fd_chunks(p.n) + 128;
    lm_fd_eval_lanes<OBJ>(p, first, pid, theta_lds ? theta_lds : p.theta + pid * kLmN, xs, lm_fd_smem);
    return;
  }
  if (p.n <= 8)
    lm_fd_eval_groups<OBJ, 4, REF>(p, first, pid, theta_lds);
  else if (p.n <= 16)
    lm_fd_eval_groups<OBJ, 8, REF>(p, first, pid, theta_lds);
  else if (p.n <= 32)
    lm_fd_eval_groups<OBJ, 16, REF>(p, first, pid, theta_lds);
  else
    lm_fd_eval_groups<OBJ, 32, REF>(p, first, pid, theta_lds);
}

// ---- QR solver (tinyqr::lm on the damped matrix): the step as a kernel of its own, one
// workgroup per problem (R and the co-rotated right-hand side live in LDS, 33 KiB), between two
// evaluation launches. Stop tests, damping, QR solve, theta update.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void lm_qr_step_kernel(LmParams p) {
  extern __shared__ __align__(16) unsigned char lm_smem[];
  LmQrShared &qs = *reinterpret_cast<LmQrShared *>(lm_smem);
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (pr->done) return;
  const int t = threadIdx.x, n = static_cast<int>(p.n);
  const double prev = pr->prev, cur = pr->f;
  if (pr->iter >= p.max_iter || fabs(prev - cur) < p.f_delta || isnan(prev)) {  // :3520-3527
    __syncthreads();  // every thread has read `done` before it flips
    if (t == 0) pr->done = 1;
    return;
  }
  const double *tri = p.Hg + pid * kLmTri;
  const double lambda = pr->lambda;
  for (int e = t; e < 64 * 64; e += THREADS) {  // the full symmetric matrix from its lower triangle
    const int i = e >> 6, j = e & 63;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    const double v = tri[lm_tri_row(hi) + lo];
    qs.R[i * kLmQrStride + j] = (i == j && i < n) ? v + lambda : v;  // :3529-3531
  }
  if (t < 64) qs.R[t * kLmQrStride + 64] = p.gg[pid * kLmN + t];
  __syncthreads();
  lm_solve_qr<THREADS>(qs, n);
  if (t < n) p.theta[pid * kLmN + t] = p.theta[pid * kLmN + t] - qs.upd[t];  // :3534
}

// host layout -> device layout of A and y (see nlsg_lm_set_data), for the problems [b0, b0 + nb):
// a_raw holds those problems' matrices only ([nb][m][n]), y_raw all of y ([batch][m])
__global__ void lm_repack_kernel(LmParams p, const double *a_raw, const double *y_raw, double *A,
                                 double *y, uint64_t b0, uint64_t nb) {
  const uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= p.nstep * nb * 16 * kLmN) return;
  const uint64_t c = q % kLmN, r = (q / kLmN) % 16, bl = (q / (16 * kLmN)) % nb;
  const uint64_t s = q / (16 * kLmN * nb), i = 16 * s + r, b = b0 + bl;
  A[((s * p.batch + b) * 16 + r) * kLmN + c] = (i < p.m && c < p.n) ? a_raw[(bl * p.m + i) * p.n + c] : 0.0;
  if (c == 0) y[(s * p.batch + b) * 16 + r] = i < p.m ? y_raw[b * p.m + i] : 0.0;
}

// ---- n > 64 (the reference has no limit, nlsolver.h:3428-3545): one WORKGROUP per problem, the
// Hessian as a full n x n matrix in global memory (L2-resident: 128 KiB at n = 128), step and
// evaluation as launches of their own. Same arithmetic rules as the one-wave kernels above, so
// oracle_lm.c's order-1 restatement covers both: every element of the Cholesky factor is its
// k-ordered fma chain (:251-269), the substitutions are column sweeps (forward sums in j order,
// backward sums from j = n-1 down), the finite-difference probes are full lane-tree evaluations,
// 2 J^T J is one fma chain over the rows per element. Built for generality, not for a roofline:
// n <= 1024 (a thread follows at most four matrix rows; a wave holds a probe point).
constexpr int kLmWideThreads = 256;
constexpr int kLmWideMaxN = 1024;

// REF (NLSG_LM_CHOLESKY_REFERENCE_ORDER past 64 parameters, round 4): a product and an add instead of
// a fused multiply-add, and backsolve_inplace_t's sums in increasing j — the reference's own bits.
template <bool REF = false>
__global__ __launch_bounds__(kLmWideThreads) void lm_wide_step_kernel(LmParams p) {
  __shared__ double piv[kLmWideMaxN];  // the pivot row of a column step / the solution vector
  __shared__ double terms[REF ? kLmWideMaxN : 1];  // REF: the products of one backsolve row
  __shared__ double diag;
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (pr->done) return;
  const int t = threadIdx.x, n = static_cast<int>(p.n);
  constexpr int T = kLmWideThreads, R = kLmWideMaxN / kLmWideThreads;
  const double prev = pr->prev, cur = pr->f;
  if (pr->iter >= p.max_iter || fabs(prev - cur) < p.f_delta || isnan(prev)) {  // :3520-3527
    __syncthreads();  // every thread has read `done` before it flips
    if (t == 0) pr->done = 1;
    return;
  }
  double *H = p.Hw + pid * p.n * p.n;
  const double *g = p.gg + pid * p.n;
  double *th = p.theta + pid * p.n;
  const double lambda = pr->lambda;
  for (int i = t; i < n; i += T) H[static_cast<uint64_t>(i) * n + i] += lambda;  // :3529-3531
  __syncthreads();
  // is_diagonal (:295-307): any off-diagonal of either triangle above eps * 1e12
  bool off = false;
  for (int i = 0; i < n; i++)
    for (int j = t; j < n; j += T)
      off |= (i != j) && H[static_cast<uint64_t>(i) * n + j] > 2.220446049250313e-16 * 1e12;
  if (!__syncthreads_or(off)) {  // :310-318
    for (int i = t; i < n; i += T) th[i] = th[i] - g[i] / H[static_cast<uint64_t>(i) * n + i];
    return;
  }
  // cholesky (:251-269), column by column: L[i][j] = 1 / L[j][j] * (A[i][j] - sum_k<j L[i][k] L[j][k]),
  // each sum one fma chain in k order (the diagonal's is the same chain on its own row).
  // (A transposed second copy of L in the upper triangle, so that a thread's walk along its row
  // is coalesced across threads, was measured SLOWER — 4.5 -> 4.9 ms at n = 256: the row walk
  // reuses each of its cache lines for sixteen consecutive k, the column walk none.)
  double s[R];
  for (int j = 0; j < n; j++) {
    for (int k = t; k < j; k += T) piv[k] = H[static_cast<uint64_t>(j) * n + k];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int i = t + T * q;
      double sum = 0.0;
      if (i >= j && i < n) {
        const double *row = H + static_cast<uint64_t>(i) * n;
        for (int k = 0; k < j; k++) sum = lm_mad<REF>(row[k], piv[k], sum);
        if (i == j) diag = sqrt(row[j] - sum);
      }
      s[q] = sum;
    }
    __syncthreads();
    const double d = diag;
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int i = t + T * q;
      if (i >= j && i < n) {
        double *e = H + static_cast<uint64_t>(i) * n + j;
        *e = i == j ? d : (1.0 / d * (*e - s[q]));
      }
    }
    __syncthreads();
  }
  // forwardsolve_inplace (:282-294): column sweep, each row's sum grows in j order
#pragma unroll
  for (int q = 0; q < R; q++) s[q] = 0.0;
  for (int j = 0; j < n; j++) {
    if (t == (j & (T - 1))) {
      double sj = 0.0;
#pragma unroll
      for (int q = 0; q < R; q++) sj = (j / T == q) ? s[q] : sj;
      piv[j] = (g[j] - sj) / H[static_cast<uint64_t>(j) * n + j];
    }
    __syncthreads();
    const double uj = piv[j];
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int i = t + T * q;
      if (i > j && i < n) s[q] = lm_mad<REF>(H[static_cast<uint64_t>(i) * n + j], uj, s[q]);
    }
  }
  if constexpr (REF) {
    // backsolve_inplace_t (:270-281) literally: component i from the sum over j = i+1 .. n-1 in
    // that order. The products are made side by side (thread j: U[j][i] b[j]), one thread adds them.
    for (int i = n - 1; i >= 0; i--) {
      for (int j = i + 1 + t; j < n; j += T) terms[j] = H[static_cast<uint64_t>(j) * n + i] * piv[j];
      __syncthreads();
      if (t == 0) {
        double sum = 0.0;
        for (int j = i + 1; j < n; j++) sum = sum + terms[j];
        piv[i] = (piv[i] - sum) / H[static_cast<uint64_t>(i) * n + i];
      }
      __syncthreads();
    }
    for (int i = t; i < n; i += T) th[i] = th[i] - piv[i];  // :3534
    return;
  }
  // backsolve_inplace_t (:270-281), the inner sums taken from j = n-1 down to i+1
#pragma unroll
  for (int q = 0; q < R; q++) s[q] = 0.0;
  for (int j = n - 1; j >= 0; j--) {
    if (t == (j & (T - 1))) {
      double sj = 0.0;
#pragma unroll
      for (int q = 0; q < R; q++) sj = (j / T == q) ? s[q] : sj;
      piv[j] = (piv[j] - sj) / H[static_cast<uint64_t>(j) * n + j];
    }
    __syncthreads();
    const double bj = piv[j];
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int i = t + T * q;
      if (i < j) s[q] = __builtin_fma(H[static_cast<uint64_t>(j) * n + i], bj, s[q]);
    }
  }
  __syncthreads();
  for (int i = t; i < n; i += T) th[i] = th[i] - piv[i];  // :3534
}

// ---- The step for 64 < n <= 1024 (the default) as a BLOCKED left-looking Cholesky with the panel sums on the
// matrix cores. Element (i, j) of the factor is 1 / L_jj (A_ij - sum_{k<j} L_ik L_jk) with the sum
// one k-ordered fma chain starting at zero (cholesky, :251-269, order 1 of the oracle) — which is
// what v_mfma_f64_16x16x4_f64 computes for a 16 x 16 tile of (i, j) at once when it is fed
// k = 0, 1, 2, ... in order. So, per panel J of sixteen columns:
//   P1  S_IJ = L[I rows][0 .. 16J) L[J rows][0 .. 16J)^T for every row block I >= J, one tile per
//       wave at a time (4J MFMAs each), into LDS; wave 0 takes the diagonal tile and then factors
//       the 16 x 16 diagonal block (a lane per row, the other rows' entries by v_readlane) while
//       the other waves are on the matrix cores; it publishes the block and the 1 / L_jj;
//   P2  thread t = row t below the block continues its sixteen chains through the panel's own
//       columns (k = 16J .. 16J + c - 1, from the published block) and writes its sixteen entries.
// Two barriers per panel (32 at n = 256) where lm_wide_step_kernel has three per column (768), and
// the n^3 / 3 flops run at MFMA rate instead of from uncoalesced row walks.
// L overwrites the lower triangle of H in a PERMUTED layout: inside every full 16-column block,
// column c sits at position 4 (c mod 4) + c / 4, so that the four values a lane feeds to four
// consecutive MFMAs (k = g, g + 4, g + 8, g + 12 for lane group g) are 32 contiguous bytes and a
// tile's operand load touches sixteen full cache lines. H is scratch of the iteration (the next
// evaluation rewrites it), so the layout never leaves this kernel.
// The substitutions are blocked the same way: the wave that owns a block's rows solves its
// sixteen unknowns in sequence (lane broadcasts), publishes them, and after ONE barrier every
// thread extends its running sum by the sixteen products, in the order-1 oracle's order (forward
// sums grow with j, backward sums run from j = n-1 down).
__host__ __device__ constexpr int lm_wchol_threads(uint64_t n) { return n <= 256 ? 256 : n <= 512 ? 512 : 1024; }
__host__ __device__ constexpr size_t lm_wchol_lds_bytes(int threads) {
  return (static_cast<size_t>(threads) * 17 + 16 * 16 + 16 + 2 * 16) * sizeof(double);
}
__device__ inline int lm_wchol_pos(int c) { return ((c & 3) << 2) | (c >> 2); }

#ifdef NLSG_WCHOL_PROFILE
#define WCHOL_T(k) do { const long long now_ = __builtin_readcyclecounter(); prof_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define WCHOL_T(k) do { } while (0)
#endif
template <int THREADS, bool EVEN>  // EVEN: n is even, rows start on 16-byte boundaries (128-bit loads)
__global__ __launch_bounds__(THREADS, 4) void lm_wide_chol_step_kernel(LmParams p) {
#ifdef NLSG_WCHOL_PROFILE
  long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  extern __shared__ __align__(16) double lm_wchol_smem[];
  constexpr int W = THREADS / 64;
  double *S = lm_wchol_smem;                  // [THREADS][17]: the panel sums, row t for thread t
  double *ld = S + THREADS * 17;              // [16][16]: the factored diagonal block
  double *rinv = ld + 256;                    // [16]: 1 / L_jj
  double *xbuf = rinv + 16;                   // [2][16]: a block's unknowns in the substitutions
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (pr->done) return;
  const int t = threadIdx.x, lane = t & 63, n = static_cast<int>(p.n);
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const double prev = pr->prev, cur = pr->f;
  if (pr->iter >= p.max_iter || fabs(prev - cur) < p.f_delta || isnan(prev)) {  // :3520-3527
    __syncthreads();  // every thread has read `done` before it flips
    if (t == 0) pr->done = 1;
    return;
  }
  double *H = p.Hw + pid * p.n * p.n;
  const double *g = p.gg + pid * p.n;
  double *th = p.theta + pid * p.n;
  const double lambda = pr->lambda;
  const int tr = t < n ? t : n - 1;  // threads past n alias the last row for their (unused) reads
  for (int i = t; i < n; i += THREADS) H[static_cast<uint64_t>(i) * n + i] += lambda;  // :3529-3531
#pragma unroll
  for (int c = 0; c < 16; c++) S[t * 17 + c] = 0.0;  // the first panel's sums
  __syncthreads();
  // is_diagonal (:295-307): the one-pass evaluations leave their verdict on the matrix they publish
  // (the damping only touches the diagonal); otherwise column t of eight rows in flight
  bool off = false;
  const int verdict = p.verdict ? pr->upper : 0;  // wave-uniform
  if (verdict & 2) {
    off = (verdict & 1) != 0;
  } else {
    for (int i0 = 0; i0 < n; i0 += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = H[static_cast<uint64_t>(i0 + q < n ? i0 + q : n - 1) * n + tr];
#pragma unroll
      for (int q = 0; q < 8; q++) off |= (i0 + q != t) && (i0 + q < n) && (t < n) && v[q] > 2.220446049250313e-16 * 1e12;
    }
  }
  if (!__syncthreads_or(off)) {  // :310-318
    if (t < n) th[t] = th[t] - g[t] / H[static_cast<uint64_t>(t) * n + t];
    return;
  }
  const int NB = (n + 15) >> 4;
  WCHOL_T(0);
  auto load4 = [&](const double *src, double (&v)[4]) {
    if constexpr (EVEN) {
      const double2 a = *reinterpret_cast<const double2 *>(src), b = *reinterpret_cast<const double2 *>(src + 2);
      v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = src[q];
    }
  };
  const int cc = lane & 15, kk = lane >> 4;
  double *Hrow = H + static_cast<uint64_t>(tr) * n;
  // ---- cholesky (:251-269)
  for (int J = 0; J < NB; J++) {
    const int j0 = 16 * J, nc = n - j0 < 16 ? n - j0 : 16;
    const bool full = nc == 16;
    if (J > 0) {
      // P1: wave 0 the diagonal tile, waves 1 .. W-1 the tiles below it
      const double *pb = H + static_cast<uint64_t>(j0 + cc < n ? j0 + cc : n - 1) * n + 4 * kk;
      for (int I = wid == 0 ? J : J + wid; I < NB; I += wid == 0 ? NB : W - 1) {
        const int ra = 16 * I + cc;
        const double *pa = H + static_cast<uint64_t>(ra < n ? ra : n - 1) * n + 4 * kk;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        // four column blocks per batch of loads (the entries come from HBM or the far cache: the
        // other waves of the SIMD cover the wait)
        int kb = 0;
        for (; kb + 4 <= J; kb += 4) {
          double a[4][4], b[4][4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            load4(pa + 16 * (kb + u), a[u]);
            load4(pb + 16 * (kb + u), b[u]);
          }
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][s], b[u][s], acc, 0, 0, 0);
        }
        if (kb < J) {
          double a[3][4], b[3][4];
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const int k2 = kb + u < J ? kb + u : J - 1;
            load4(pa + 16 * k2, a[u]);
            load4(pb + 16 * k2, b[u]);
          }
#pragma unroll
          for (int u = 0; u < 3; u++)
            if (kb + u < J) {
#pragma unroll
              for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][s], b[u][s], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) S[(16 * I + kk + 4 * rg) * 17 + cc] = acc[rg];
      }
    }
    WCHOL_T(1);
    if (wid == 0) {
      // the diagonal block: lane a = row j0 + a; L[c][e] of another row comes from lane c
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int a = lane < nc ? lane : nc - 1;
      const double *hr = H + static_cast<uint64_t>(j0 + a) * n + j0;
      // column by column; a finished column extends the sums of all later ones at once (each sum
      // still grows in k order), so only the square root and the reciprocal are a serial chain
      double mine[16], sv[16], hv[16];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        hv[c] = hr[c < nc ? c : 0];
        sv[c] = S[(j0 + a) * 17 + c];
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nc) {  // uniform
          const double x = hv[c] - sv[c];
          const double d = sqrt(lane_broadcast(x, c));
          const double ri = 1.0 / d;
          mine[c] = lane == c ? d : (ri * x);
          if (lane == 0) rinv[c] = ri;
#pragma unroll
          for (int e = c + 1; e < 16; e++) sv[e] = __builtin_fma(mine[c], lane_broadcast(mine[c], e), sv[e]);
        }
      }
      if (lane < nc) {
        double *wr = H + static_cast<uint64_t>(j0 + lane) * n + j0;
#pragma unroll
        for (int c = 0; c < 16; c++)
          if (c <= lane) {
            ld[lane * 16 + c] = mine[c];
            wr[full ? lm_wchol_pos(c) : c] = mine[c];
          }
      }
    }
    WCHOL_T(2);
    __syncthreads();
    WCHOL_T(3);
    // P2: the rows below the block (then the block is full)
    if (t >= j0 + 16 && t < n) {
      double h[16], mine[16];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        double v[4];
        load4(Hrow + j0 + 4 * q, v);
#pragma unroll
        for (int e = 0; e < 4; e++) h[4 * q + e] = v[e];
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        double sv = S[t * 17 + c];
#pragma unroll
        for (int e = 0; e < c; e++) sv = __builtin_fma(mine[e], ld[c * 16 + e], sv);
        mine[c] = (rinv[c] * (h[c] - sv));
        __builtin_amdgcn_sched_barrier(0);  // (keeps the 136 block entries from being loaded up front)
      }
      // position 4 g + s holds column 4 s + g
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        if constexpr (EVEN) {
          *reinterpret_cast<double2 *>(Hrow + j0 + 4 * gq) = make_double2(mine[gq], mine[4 + gq]);
          *reinterpret_cast<double2 *>(Hrow + j0 + 4 * gq + 2) = make_double2(mine[8 + gq], mine[12 + gq]);
        } else {
#pragma unroll
          for (int s = 0; s < 4; s++) Hrow[j0 + 4 * gq + s] = mine[4 * s + gq];
        }
      }
    }
    WCHOL_T(4);
    __syncthreads();
    WCHOL_T(5);
  }
  // position of column j in its row
  auto col = [&](int j) { return (j | 15) < n ? (j & ~15) + lm_wchol_pos(j & 15) : j; };
  // ---- forwardsolve_inplace (:282-294)
  double sum = 0.0, own = 0.0;  // own: u_t, then the step's component t
  const double gt = g[tr];
  for (int J = 0; J < NB; J++) {
    const int j0 = 16 * J, nc = n - j0 < 16 ? n - j0 : 16;
    double *xb = xbuf + 16 * (J & 1);
    double lrow[16];  // the thread's entries in this block's columns
#pragma unroll
    for (int c = 0; c < 16; c++) lrow[c] = (c < nc && t >= j0) ? Hrow[col(j0 + c)] : 1.0;
    if (wid == (j0 >> 6)) {  // the wave that holds the block's rows: lanes b .. b + 15
      const int b = j0 & 63;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nc) {
          // lane b + c: its sum is complete
          const double uc = lane_broadcast((gt - sum) / lrow[c], b + c);
          if (lane == b + c) own = uc;
          if (lane > b + c && lane < b + 16) sum = __builtin_fma(lrow[c], uc, sum);
          if (lane == 0) xb[c] = uc;
        }
      }
    }
    __syncthreads();  // (the next block's unknowns go to the other buffer)
    if (t >= j0 + 16 && t < n) {
#pragma unroll
      for (int c = 0; c < 16; c++) sum = __builtin_fma(lrow[c], xb[c], sum);
    }
  }
  WCHOL_T(6);
  // ---- backsolve_inplace_t (:270-281), the inner sums taken from j = n-1 down to i+1
  sum = 0.0;
  const double ut = own;
  for (int J = NB - 1; J >= 0; J--) {
    const int j0 = 16 * J, nc = n - j0 < 16 ? n - j0 : 16;
    double *xb = xbuf + 16 * (J & 1);
    if (wid == (j0 >> 6)) {
      const int b = j0 & 63;
      const int a = lane - b;  // column inside the block for lanes b .. b + 15
      const bool in = a >= 0 && a < nc;
      // L[j0 + c][j0 + a]: row j0 + c across the lanes (one cache line)
      double lc[16];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int c2 = c < nc ? c : nc - 1;
        lc[c] = in ? H[static_cast<uint64_t>(j0 + c2) * n + col(j0 + (a <= c2 ? a : c2))] : 1.0;
      }
#pragma unroll
      for (int c = 15; c >= 0; c--) {
        if (c < nc) {
          const double lca = lc[c];
          const double xc = lane_broadcast((ut - sum) / lca, b + c);  // lane b + c divides by L_cc
          if (a == c) own = xc;
          if (in && a < c) sum = __builtin_fma(lca, xc, sum);
          if (lane == 0) xb[c] = xc;
        }
      }
    }
    double lt[16];  // column t of the block's rows, requested before the wait
    if (t < j0) {   // (t < n)
      const int ct = col(t);
#pragma unroll
      for (int c = 0; c < 16; c++) lt[c] = H[static_cast<uint64_t>(j0 + (c < nc ? c : nc - 1)) * n + ct];
    }
    __syncthreads();
    if (t < j0) {
#pragma unroll
      for (int c = 15; c >= 0; c--)
        if (c < nc) sum = __builtin_fma(lt[c], xb[c], sum);
    }
  }
  if (t < n) th[t] = th[t] - own;  // :3534
  WCHOL_T(7);
#ifdef NLSG_WCHOL_PROFILE
  if (pid == 5 && lane == 0 && (wid == 0 || wid == 1 || wid == W - 1))
    printf("wchol wave %d: setup %lld | P1 %lld diag %lld wait %lld P2 %lld wait %lld | fwd %lld bwd %lld\n", wid,
           prof_[0], prof_[1], prof_[2], prof_[3], prof_[4], prof_[5], prof_[6], prof_[7]);
#endif
}
#undef WCHOL_T

// The default functors (fin_diff, fin_diff_h; see lm_fd_eval_groups above) for n > 64: sixteen
// (n > 512: eight) waves per problem, a probe point per wave (CHUNKS x 128 coordinates in registers), gradient
// coordinates and Hessian entries dealt to the waves round robin. Every probe is a full
// wave_objective evaluation: the bits of the oracle's tree.
// the ONE place that maps n to the CHUNKS instantiation (engine creation, the run-time compiler's
// template arguments and every launch take it from here)
__host__ __device__ constexpr int lm_wide_chunks(uint64_t n) { return n <= 128 ? 1 : n <= 256 ? 2 : n <= 512 ? 4 : 8; }
__host__ __device__ constexpr int lm_wide_fd_threads(int chunks) {
  return chunks >= 8 ? 512 : 1024;  // the 1024-coordinate point needs more than 128 registers
}
template <int OBJ, int CHUNKS>  // (reference order: lm_wide_fd_lanes_kernel below)
__global__ __launch_bounds__(lm_wide_fd_threads(CHUNKS)) void lm_wide_fd_eval_kernel(LmParams p, int first) {
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (!first && pr->done) return;
  // gridDim.y workgroups share a problem (a single start would otherwise keep one CU busy): the
  // gradient coordinates and Hessian entries are dealt over all their waves
  const int W = (lm_wide_fd_threads(CHUNKS) / 64) * static_cast<int>(gridDim.y);
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6) +
                (lm_wide_fd_threads(CHUNKS) / 64) * static_cast<int>(blockIdx.y);
  const uint64_t n = p.n;
  const double *th = p.theta + pid * n;
  double xv[CHUNKS][2];
  load_row<CHUNKS, false>(th, n, p.zero, xv);
  // the point with coordinate i set to vi and coordinate j set to vj (j wins when i == j: the
  // reference's x[i] and x[j] are then the same element)
  auto eval2 = [&](uint64_t i, double vi, uint64_t j, double vj) {
    double xp[CHUNKS][2];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        xp[c][k] = e == j ? vj : e == i ? vi : xv[c][k];
      }
    return wave_objective<OBJ, CHUNKS>(xp, n);
  };
  {  // fin_diff<1> (:1385-1413)
    constexpr double eps = 2.220446049250313e-16 * 10e7;
    constexpr double dd_val = 12 * eps;
    for (uint64_t d = w; d < n; d += W) {
      const double xd = th[d];
      double acc = 0.0;
      acc = acc + 1.0 * eval2(d, xd + -2.0 * eps, d, xd + -2.0 * eps);
      acc = acc + -8.0 * eval2(d, xd + -1.0 * eps, d, xd + -1.0 * eps);
      acc = acc + 8.0 * eval2(d, xd + 1.0 * eps, d, xd + 1.0 * eps);
      acc = acc + -1.0 * eval2(d, xd + 2.0 * eps, d, xd + 2.0 * eps);
      if (lane == 0) p.gg[pid * n + d] = acc / dd_val;
    }
  }
  // fin_diff_h<1> (:1446-1515): the sixteen probes of an entry, x[i] and x[j] walked through the
  // reference's += / -= steps (one element when i == j)
  const double e1 = p.eps_h, e2 = 2 * e1, e3 = 3 * e1, e4 = 4 * e1;
  const double denom = (600.0 * e1 * e1);
  double *H = p.Hw + pid * n * n;
  for (uint64_t en = w; en < n * n; en += W) {
    const uint64_t i = en / n, j = en - i * n;
    const bool same = i == j;
    double xi = th[i], xj = th[j];
    auto add_i = [&](double d) { xi = xi + d; xj = same ? xi : xj; };
    auto sub_i = [&](double d) { xi = xi - d; xj = same ? xi : xj; };
    auto add_j = [&](double d) { xj = xj + d; xi = same ? xj : xi; };
    auto sub_j = [&](double d) { xj = xj - d; xi = same ? xj : xi; };
    auto f = [&]() { return eval2(i, xi, j, xj); };
    double result = 0.0, temp = 0.0;
    add_i(e1); sub_j(e2); temp = temp + f();
    add_i(e1); add_j(e1); temp = temp + f();
    sub_i(e4); add_j(e2); temp = temp + f();
    add_i(e1); add_j(e1); temp = temp + f();
    result = result - 63 * temp;
    temp = 0.0;
    sub_j(e4); temp = temp + f();
    sub_i(e1); add_j(e1); temp = temp + f();
    add_i(e3); add_j(e3); temp = temp + f();
    add_i(e1); sub_j(e1); temp = temp + f();
    result = result + 63 * temp;
    temp = 0.0;
    sub_j(e3); temp = temp + f();
    sub_i(e4); add_j(e4); temp = temp + f();
    sub_j(e4); temp = temp - f();
    add_i(e4); add_j(e4); temp = temp - f();
    result = result + 44 * temp;
    temp = 0.0;
    sub_i(e3); sub_j(e3); temp = temp + f();
    add_i(e2); add_j(e2); temp = temp + f();
    sub_j(e2); temp = temp - f();
    sub_i(e2); add_j(e2); temp = temp - f();
    result = result + 74 * temp;
    if (lane == 0) H[en] = result / denom;
  }
  if (w == 0) {
    const double fx = wave_objective<OBJ, CHUNKS>(xv, n);
    if (lane == 0) lm_publish_state(p, pr, first, fx);
  }
}

// The default functors for n > 64 in REFERENCE ORDER: lm_fd_eval_lanes' probe per lane, a workgroup
// (or gridDim.y of them) per problem. The block keeps the point, the base terms t_e and their prefix
// sums S_e in LDS (one serial chain, by wave 0); the waves deal out units of work:
//   gradient   64 coordinates, lane = coordinate: start S_{d-off}, modified terms, tail;
//   Hessian    row i x 64 columns, lane = entry (i, j): sixteen probes as two batches of eight
//              chains that start from S just before the first term i or the unit's columns can
//              touch, walk the base terms at a uniform address and substitute the lane's up to four
//              modified terms where they fall — only inside the two short zones around i and around
//              the unit's columns does a step cost selects, elsewhere it is eight additions.
// Every probe has the bits of wave_objective_seq at its point (tests: the reference's runs at
// n = 100 and 130, oracle order 0 up to n = 257). smem: xs[n + 2] | ts[n] | S[n].
__host__ __device__ constexpr size_t lm_wide_fd_lanes_lds_bytes(uint64_t n) { return (3 * n + 2) * sizeof(double); }
template <int OBJ>
__global__ __launch_bounds__(256) void lm_wide_fd_lanes_kernel(LmParams p, int first) {
  using O = Objective<OBJ>;
  extern __shared__ __align__(16) double lm_wfl_smem[];
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (!first && pr->done) return;
  const int n = static_cast<int>(p.n), nt = static_cast<int>(O::n_terms(p.n));
  const int t = threadIdx.x, lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int W = 4 * static_cast<int>(gridDim.y), w = wid + 4 * static_cast<int>(blockIdx.y);
  double *xs = lm_wfl_smem, *ts = xs + n + 2, *S = ts + n;
  const double *th = p.theta + pid * p.n;
  for (int i = t; i < n + 2; i += 256) xs[i] = i < n ? th[i] : 0.0;
  __syncthreads();
  for (int e = t; e < nt; e += 256) ts[e] = O::term(xs[e], xs[e + 1]);
  __syncthreads();
  if (wid == 0) {
    double run = 0.0;
#pragma unroll 4
    for (int e = 0; e < nt; e++) {
      run = run + ts[e];
      if (lane == 0) S[e] = run;
    }
  }
  __syncthreads();
  constexpr int off = O::kChain ? 2 : 1, back = O::kChain ? 1 : 0;
  const int nb = (n + 63) >> 6;
  // fin_diff<1> (:1385-1413)
  for (int db = w; db < nb; db += W) {
    constexpr double eps = 2.220446049250313e-16 * 10e7;
    constexpr double coeff[4] = {1, -8, 8, -1}, coeff2[4] = {-2, -1, 1, 2};
    constexpr double dd_val = 12 * eps;
    const int d0 = 64 * db, d = d0 + lane, dc = d < n ? d : n - 1;
    const double xd = xs[dc], xm = xs[dc > 0 ? dc - 1 : 0], xp = xs[dc + 1];
    const double a = dc - off >= 0 ? S[dc - off] : 0.0;
    double acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double xq = xd + coeff2[q] * eps;
      acc[q] = a;
      if constexpr (O::kChain) {
        const double m0 = acc[q] + O::term(xm, xq);
        acc[q] = d >= 1 ? m0 : acc[q];
        const double m1 = acc[q] + O::term(xq, xp);
        acc[q] = d < nt ? m1 : acc[q];
      } else {
        acc[q] = acc[q] + O::term(xq, 0.0);
      }
    }
    int e = d0 + 1;
    const int e_win = d0 + 64 < nt ? d0 + 64 : nt;
    for (; e < e_win; e++) {
      const double te = ts[e];
      if (e > d) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = acc[q] + te;
      }
    }
#pragma unroll 4
    for (; e < nt; e++) {
      const double te = ts[e];
#pragma unroll
      for (int q = 0; q < 4; q++) acc[q] = acc[q] + te;
    }
    double ga = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) ga = ga + coeff[q] * O::finish(acc[q], p.n);
    if (d < n) p.gg[pid * p.n + d] = ga / dd_val;
  }
  // fin_diff_h<1> (:1446-1515)
  const double e1 = p.eps_h, e2 = 2 * e1, e3 = 3 * e1, e4 = 4 * e1;
  const double denom = (600.0 * e1 * e1);
  double *H = p.Hw + pid * p.n * p.n;
  for (int u = w; u < n * nb; u += W) {
    const int i = u / nb, j0 = 64 * (u - i * nb), j = j0 + lane, jc = j < n ? j : n - 1;
    const double xi0 = xs[i], xim = xs[i > 0 ? i - 1 : 0], xip = xs[i + 1];
    const double xj0 = xs[jc], xjm = xs[jc > 0 ? jc - 1 : 0], xjp = xs[jc + 1];
    const bool same = i == j;
    const int lo = i < j0 ? i : j0;
    const int e_first = lo - back > 0 ? lo - back : 0;   // the first term a moved coordinate can touch
    const double base = e_first >= 1 ? S[e_first - 1] : 0.0;
    const int jz_lo = j0 - back, jz_hi = j0 + 63;         // the zone of the unit's own columns
    double fv[16];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      double ci[8], cj[8];
      {
        double xi = xi0, xj = xj0;
        auto add_i = [&](double d) { xi = xi + d; xj = same ? xi : xj; };
        auto sub_i = [&](double d) { xi = xi - d; xj = same ? xi : xj; };
        auto add_j = [&](double d) { xj = xj + d; xi = same ? xj : xi; };
        auto sub_j = [&](double d) { xj = xj - d; xi = same ? xj : xi; };
        auto at = [&](auto k) {
          constexpr int K = decltype(k)::value;
          if constexpr (K / 8 == 0) { if (half == 0) { ci[K % 8] = xi; cj[K % 8] = xj; } }
          else { if (half == 1) { ci[K % 8] = xi; cj[K % 8] = xj; } }
        };
        add_i(e1); sub_j(e2); at(int_c<0>{});
        add_i(e1); add_j(e1); at(int_c<1>{});
        sub_i(e4); add_j(e2); at(int_c<2>{});
        add_i(e1); add_j(e1); at(int_c<3>{});
        sub_j(e4); at(int_c<4>{});
        sub_i(e1); add_j(e1); at(int_c<5>{});
        add_i(e3); add_j(e3); at(int_c<6>{});
        add_i(e1); sub_j(e1); at(int_c<7>{});
        sub_j(e3); at(int_c<8>{});
        sub_i(e4); add_j(e4); at(int_c<9>{});
        sub_j(e4); at(int_c<10>{});
        add_i(e4); add_j(e4); at(int_c<11>{});
        sub_i(e3); sub_j(e3); at(int_c<12>{});
        add_i(e2); add_j(e2); at(int_c<13>{});
        sub_j(e2); at(int_c<14>{});
        sub_i(e2); add_j(e2); at(int_c<15>{});
      }
      double A[8], B[8], C[8], D[8];  // terms i-1, i, j-1, j (x_j -> cj takes precedence over x_i -> ci)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const double vi = same ? cj[k] : ci[k];
        if constexpr (O::kChain) {
          const double vim = i - 1 == j ? cj[k] : xim;
          const double vip = i + 1 == j ? cj[k] : xip;
          const double vjm = j - 1 == i ? ci[k] : xjm;
          const double vjp = j + 1 == i ? ci[k] : xjp;
          A[k] = O::term(vim, vi);
          B[k] = O::term(vi, vip);
          C[k] = O::term(vjm, cj[k]);
          D[k] = O::term(cj[k], vjp);
        } else {
          A[k] = C[k] = 0.0;
          B[k] = O::term(vi, 0.0);
          D[k] = O::term(cj[k], 0.0);
        }
      }
      double acc[8];
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] = base;
      for (int e = e_first; e < nt; e++) {
        const double te = ts[e];
        const bool zi = e == i || (O::kChain && e == i - 1), zj = e >= jz_lo && e <= jz_hi;  // wave-uniform
        if (!zi && !zj) {
#pragma unroll
          for (int k = 0; k < 8; k++) acc[k] = acc[k] + te;
        } else {
          const bool at_j = e == j, at_jm = O::kChain && e == j - 1;
          if (O::kChain && e == i - 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = acc[k] + (at_j ? D[k] : at_jm ? C[k] : A[k]);
          } else if (e == i) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = acc[k] + (at_j ? D[k] : at_jm ? C[k] : B[k]);
          } else {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = acc[k] + (at_j ? D[k] : at_jm ? C[k] : te);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) fv[8 * half + k] = O::finish(acc[k], p.n);
    }
    double result = 0.0, temp = 0.0;
    temp = temp + fv[0];
    temp = temp + fv[1];
    temp = temp + fv[2];
    temp = temp + fv[3];
    result = result - 63 * temp;
    temp = 0.0;
    temp = temp + fv[4];
    temp = temp + fv[5];
    temp = temp + fv[6];
    temp = temp + fv[7];
    result = result + 63 * temp;
    temp = 0.0;
    temp = temp + fv[8];
    temp = temp + fv[9];
    temp = temp - fv[10];
    temp = temp - fv[11];
    result = result + 44 * temp;
    temp = 0.0;
    temp = temp + fv[12];
    temp = temp + fv[13];
    temp = temp - fv[14];
    temp = temp - fv[15];
    result = result + 74 * temp;
    if (j < n) H[static_cast<uint64_t>(i) * p.n + j] = result / denom;
  }
  if (w == 0 && t == 0) lm_publish_state(p, pr, first, O::finish(nt > 0 ? S[nt - 1] : 0.0, p.n));
}

// Gauss-Newton functors of the tanh regression for n > 64: f = sum r^2, g = 2 J^T r, H = 2 J^T J
// with J = -diag(1 - tanh^2(A theta)) A, in the order-1 sums of oracle_lm.c: z_i by 32 column
// pairs per 64 columns (the pairs of later 64-column blocks continue each lane's fma chain), then
// the xor butterfly 16..1; f in eight fma chains by (row / 16 mod 4, row parity); g_j in four by
// row mod 4; H_jk one fma chain over the rows. A thread owns an 8 x 8 block of a 128 x 128
// super-block of H; the scaled Jacobian rows of the two column ranges pass through LDS sixteen
// rows at a time; the upper triangle is the mirror image (products commute: bitwise symmetric).
__global__ __launch_bounds__(kLmWideThreads) void lm_wide_tanh_eval_kernel(LmParams p, int first) {
  __shared__ double Jj[16][128 + 8], Jk[16][128 + 8];
  __shared__ double part[8];
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (!first && pr->done) return;
  constexpr int T = kLmWideThreads;
  const int t = threadIdx.x, lane = lane_id(), w = t >> 6;
  const uint64_t n = p.n, m = p.m;
  const double *A = p.Aw + pid * m * n, *y = p.yw + pid * m, *th = p.theta + pid * n;
  double *r = p.rw + pid * 2 * m, *wt = r + m;
  // residuals and weights, a wave per row
  const int lp = lane & 31;
  for (uint64_t i = w; i < m; i += T / 64) {
    const double *row = A + i * n;
    double acc = 0.0;
    for (uint64_t c0 = 0; c0 < n; c0 += 64) {
      const uint64_t e0 = c0 + 2 * lp, e1 = e0 + 1;
      const double a0 = e0 < n ? row[e0] : 0.0, b0 = e0 < n ? th[e0] : 0.0;
      const double a1 = e1 < n ? row[e1] : 0.0, b1 = e1 < n ? th[e1] : 0.0;
      acc = c0 == 0 ? __builtin_fma(a1, b1, a0 * b0) : __builtin_fma(a1, b1, __builtin_fma(a0, b0, acc));
    }
    acc = acc + lane_xor<16>(acc);
    acc = acc + lane_xor<8>(acc);
    acc = acc + lane_xor<4>(acc);
    acc = acc + lane_xor<2>(acc);
    acc = acc + lane_xor<1>(acc);
    const double tz = det_tanh(acc);
    if (lane == 0) {
      r[i] = y[i] - tz;
      wt[i] = 1 - tz * tz;
    }
  }
  __syncthreads();
  if (t < 8) {  // f: eight chains, idx = 2 * ((i % 64) / 16) + i % 2
    double acc = 0.0;
    for (uint64_t i = 0; i < m; i++) {
      const uint64_t rb = i & 63;
      if (2 * (rb >> 4) + (rb & 1) == static_cast<uint64_t>(t)) acc = __builtin_fma(r[i], r[i], acc);
    }
    part[t] = acc;
  }
  for (uint64_t j = t; j < n; j += T) {  // g_j: four chains by i mod 4, then ((a0 + a1) + a2) + a3
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (uint64_t i = 0; i < m; i++) {
      const double Jij = -(wt[i] * A[i * n + j]);
      const double v = __builtin_fma(Jij, r[i], a[i & 3]);
      a[0] = (i & 3) == 0 ? v : a[0];
      a[1] = (i & 3) == 1 ? v : a[1];
      a[2] = (i & 3) == 2 ? v : a[2];
      a[3] = (i & 3) == 3 ? v : a[3];
    }
    p.gg[pid * n + j] = 2 * (((a[0] + a[1]) + a[2]) + a[3]);
  }
  // H by super-blocks of the lower triangle
  double *H = p.Hw + pid * n * n;
  const int tj = t >> 4, tk = t & 15;
  const uint64_t SB = (n + 127) / 128;
  for (uint64_t bj = 0; bj < SB; bj++)
    for (uint64_t bk = 0; bk <= bj; bk++) {
      double acc[8][8];
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = 0.0;
      for (uint64_t i0 = 0; i0 < m; i0 += 16) {
        __syncthreads();  // the previous group's tiles are consumed
        for (int e = t; e < 16 * 128; e += T) {
          const int rr = e >> 7, cc = e & 127;
          const uint64_t i = i0 + rr, cj = bj * 128 + cc, ck = bk * 128 + cc;
          const double wi = i < m ? wt[i] : 0.0;
          Jj[rr][cc] = (i < m && cj < n) ? -(wi * A[i * n + cj]) : 0.0;
          Jk[rr][cc] = (i < m && ck < n) ? -(wi * A[i * n + ck]) : 0.0;
        }
        __syncthreads();
        for (int rr = 0; rr < 16; rr++) {  // rows in order: each element's one fma chain
          double vj[8], vk[8];
#pragma unroll
          for (int a = 0; a < 8; a++) vj[a] = Jj[rr][tj * 8 + a];
#pragma unroll
          for (int b = 0; b < 8; b++) vk[b] = Jk[rr][tk * 8 + b];
#pragma unroll
          for (int a = 0; a < 8; a++)
#pragma unroll
            for (int b = 0; b < 8; b++) acc[a][b] = __builtin_fma(vj[a], vk[b], acc[a][b]);
        }
      }
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const uint64_t j = bj * 128 + tj * 8 + a, k = bk * 128 + tk * 8 + b;
          if (j < n && k < n) {
            const double v = 2 * acc[a][b];
            H[j * n + k] = v;
            H[k * n + j] = v;
          }
        }
    }
  __syncthreads();
  if (t == 0) {
    double f = 0.0;
    for (int k = 0; k < 8; k++) f = f + part[k];
    lm_publish_state(p, pr, first, f);
  }
}

// ---- The step for 64 < n <= 128 with the damped matrix in LDS (NLSG_LM_WIDE_CHOL=0; the blocked
// kernel above replaced it as the default: 0.18 against 0.38 ms at n = 128). lm_wide_step_kernel keeps H in
// global memory and walks it row-wise from every thread (uncoalesced, three barriers per column):
// at n = 128 it took 0.72 ms for 1024 problems, 2.4 x the evaluation beside it. Here the packed
// lower triangle (66.5 KB: two workgroups per CU) is loaded once, column-coalesced, and thread t
// owns row t, as lane t does in lm_solve_cholesky_wave — the same left-looking four-column
// panels (one read of the own row feeds four sums, the other factor is a broadcast read). What a
// wave does with lane broadcasts is done here by REPLICATION: the panel's four rows publish their
// running sums, and every thread factors the 4 x 4 diagonal block (and, in the substitutions,
// solves the four unknowns) itself — the same arithmetic on the same inputs, hence the same bits
// — so a panel costs two barriers (one in the substitutions), not two per column. Every sum is
// the k-ordered fma chain of the order-1 oracle; back-substitution sums run from j = n-1 down.
// (Four threads per row — each accumulating one column's panel sum, the sums meeting in LDS, the
// rest replicated — was built and measured: 0.54 ms against 0.39. The replicated block factorisation
// and substitutions then run on eight waves instead of two: more instructions in total on SIMDs
// that two workgroups of lone waves already keep half busy. Not kept.)
constexpr int kLmW128Tri = lm_tri_row(128);
struct LmWide128StepShared {
  double tri[kLmW128Tri];
  double g[128], u[128];
  double pubs[2][4][4], pubh[4][4];
};

__global__ __launch_bounds__(128) void lm_wide128_step_kernel(LmParams p) {
  extern __shared__ __align__(16) unsigned char lm_w128_smem[];  // 69 KB: past the static limit
  LmWide128StepShared &sh = *reinterpret_cast<LmWide128StepShared *>(lm_w128_smem);
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (pr->done) return;
  const int t = threadIdx.x, n = static_cast<int>(p.n);
  const double prev = pr->prev, cur = pr->f;
  if (pr->iter >= p.max_iter || fabs(prev - cur) < p.f_delta || isnan(prev)) {  // :3520-3527
    __syncthreads();  // every thread has read `done` before it flips
    if (t == 0) pr->done = 1;
    return;
  }
  const double *Hg = p.Hw + pid * p.n * p.n;
  double *th = p.theta + pid * p.n;
  const double lambda = pr->lambda;
  const bool row = t < n;
  const int tt = row ? t : n - 1;  // threads past n alias the last row for their (unused) reads
  // column t of every row: coalesced; the lower triangle goes to LDS with the damping on its
  // diagonal (:3529-3531), both triangles decide is_diagonal (:295-307)
  bool off = false;
  for (int i0 = 0; i0 < n; i0 += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = (i0 + q < n) ? Hg[static_cast<uint64_t>(i0 + q) * n + tt] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int i = i0 + q;
      if (i < n && row) {
        off |= (i != t) && v[q] > 2.220446049250313e-16 * 1e12;
        if (t <= i) sh.tri[lm_tri_row(i) + t] = i == t ? v[q] + lambda : v[q];
      }
    }
  }
  const double gt = p.gg[pid * p.n + tt];
  sh.g[t] = gt;
  if (!__syncthreads_or(off)) {  // :310-318
    if (row) th[t] = th[t] - gt / sh.tri[lm_tri_row(t) + t];
    return;
  }
  auto L = [&](int i, int j) -> double & { return sh.tri[lm_tri_row(i) + j]; };
  auto pair = [&](int i, int j) { return *reinterpret_cast<const double2 *>(&sh.tri[lm_tri_row(i) + j]); };
  // ---- cholesky (:251-269)
  for (int j0 = 0; j0 < n; j0 += 4) {
    const int nc = n - j0 < 4 ? n - j0 : 4;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < j0; k += 4) {  // (rows start at even offsets: a column pair is one 128-bit read)
      const double2 xa = pair(tt, k), xb = pair(tt, k + 2);  // rows above the panel: in-buffer, unused
      const double x[4] = {xa.x, xa.y, xb.x, xb.y};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int jc = j0 + c < n ? j0 + c : n - 1;
        const double2 la = pair(jc, k), lb = pair(jc, k + 2);  // wave-uniform address: broadcast
        const double l[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
        for (int q = 0; q < 4; q++) s4[c] = __builtin_fma(x[q], l[q], s4[c]);
      }
    }
    double hd[4];
#pragma unroll
    for (int c = 0; c < 4; c++) hd[c] = L(tt, min(j0 + c, tt));
    if (row && t >= j0 && t < j0 + 4) {
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (c <= t - j0) {
          sh.pubs[0][t - j0][c] = s4[c];
          sh.pubh[t - j0][c] = hd[c];
        }
    }
    __syncthreads();
    // the panel's 4 x 4 diagonal block, factored by every thread from the published sums
    double l[4][4], d[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c < nc) {  // uniform
        double sd = sh.pubs[0][c][c];
#pragma unroll
        for (int e = 0; e < c; e++) sd = __builtin_fma(l[c][e], l[c][e], sd);
        d[c] = sqrt(sh.pubh[c][c] - sd);
        l[c][c] = d[c];
#pragma unroll
        for (int a = c + 1; a < 4; a++) {
          if (a < nc) {
            double sa = sh.pubs[0][a][c];
#pragma unroll
            for (int e = 0; e < c; e++) sa = __builtin_fma(l[a][e], l[c][e], sa);
            l[a][c] = (1.0 / d[c] * (sh.pubh[a][c] - sa));
          }
        }
      }
    }
    // the thread's own row below the block: column c continues its sum with k = j0 .. j0 + c - 1
    if (row && t >= j0) {
      double mine[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (c < nc && t >= j0 + c) {
          double sv = s4[c];
#pragma unroll
          for (int e = 0; e < c; e++) sv = __builtin_fma(mine[e], l[c][e], sv);
          mine[c] = t == j0 + c ? sqrt(hd[c] - sv) : (1.0 / d[c] * (hd[c] - sv));
          L(t, j0 + c) = mine[c];
        }
      }
    }
    __syncthreads();
  }
  // ---- forwardsolve_inplace (:282-294): each row's sum grows in j order
  const double dg = L(tt, tt);
  double sum = 0.0;
  int buf = 0;
  for (int j0 = 0; j0 < n; j0 += 4, buf ^= 1) {
    const int nc = n - j0 < 4 ? n - j0 : 4;
    if (row && t >= j0 && t < j0 + 4) sh.pubs[buf][t - j0][0] = sum;
    __syncthreads();
    double uu[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c < nc) {
        double sc = sh.pubs[buf][c][0];
#pragma unroll
        for (int e = 0; e < c; e++) sc = __builtin_fma(L(j0 + c, j0 + e), uu[e], sc);
        uu[c] = (sh.g[j0 + c] - sc) / L(j0 + c, j0 + c);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c < nc) {
        if (t == j0 + c) sh.u[t] = uu[c];
        if (row && t > j0 + c) sum = __builtin_fma(L(t, j0 + c), uu[c], sum);
      }
    }
  }
  (void)dg;
  __syncthreads();  // u complete; the exchange buffers are free again
  // ---- backsolve_inplace_t (:270-281), the inner sums taken from j = n-1 down to i+1
  sum = 0.0;
  double mine_b = 0.0;
  buf = 0;
  for (int j0 = ((n - 1) >> 2) << 2; j0 >= 0; j0 -= 4, buf ^= 1) {
    const int nc = n - j0 < 4 ? n - j0 : 4;
    if (row && t >= j0 && t < j0 + 4) sh.pubs[buf][t - j0][0] = sum;
    __syncthreads();
    double bb[4];
#pragma unroll
    for (int c = 3; c >= 0; c--) {
      if (c < nc) {
        double sc = sh.pubs[buf][c][0];
#pragma unroll
        for (int e = 3; e > c; e--)
          if (e < nc) sc = __builtin_fma(L(j0 + e, j0 + c), bb[e], sc);
        bb[c] = (sh.u[j0 + c] - sc) / L(j0 + c, j0 + c);
      }
    }
#pragma unroll
    for (int c = 3; c >= 0; c--) {
      if (c < nc) {
        if (t == j0 + c) mine_b = bb[c];
        if (row && t < j0 + c) sum = __builtin_fma(L(j0 + c, t), bb[c], sum);
      }
    }
  }
  if (row) th[t] = th[t] - mine_b;  // :3534
}

// ---- The same functors for 64 < n <= 128 on the matrix cores: ONE pass over A. A workgroup of
// four waves per problem; per sixteen rows: global -> registers (the group after next is in flight)
// -> z_i = A_i theta (a half-wave per row: the 32-lane chains and butterfly of the order-1 oracle)
// -> tanh, residual, weight -> scaled Jacobian rows into a double-buffered LDS tile (row stride 144
// doubles: conflict-free ds_read_b64 for the MFMA operand pattern, as stride 80 is at n = 64) ->
// 2 J^T J on v_mfma_f64_16x16x4_f64. An fp64 MFMA is a k-ordered fma chain, so every H_jk is the
// oracle's one fma chain over the rows; H is bitwise symmetric, so only the 36 lower 16 x 16 tiles
// of the 8 x 8 are accumulated — nine per wave: wave w takes tile row 7 - w (8 - w tiles) and tile
// row w (w + 1 tiles), whose operands are a subset of the first's, and for a diagonal block A and
// B operand are the same register: 8 - w LDS reads feed 9 MFMAs per k-step. g_j (four chains by
// row mod 4) rides on the operands of column blocks 6 - 2w and 7 - 2w of wave w, f (eight chains
// by row / 16 mod 4 and row parity) on wave 1. One barrier per sixteen rows. Same bits as lm_wide_tanh_eval_kernel (same oracle).
constexpr int kLmW128Stride = 144;
struct LmWide128Shared {
  double J[2][16 * kLmW128Stride];
  double r[2][16];
};

// ---- The kernel as it runs: EIGHT waves per problem. (The four-wave form described above was the
// first build — 0.43 of the fp64 MFMA peak against 0.475 for eight waves, profiles/r03 — and was
// removed in round 4 together with its switch.) With four waves a SIMD holds two of them and each carries nine
// MFMAs plus ~60 vector instructions of staging per k-step group; eight waves halve a wave's
// staging (one row per half-wave and group instead of two), need ~110 registers instead of 196
// (four waves per SIMD) and so leave the matrix pipe fewer gaps. Tiles: 36 = 4 x 5 + 4 x 4, waves
// w and w + 4 (which share a SIMD) together nine:
//   w0 (7,0..4)  w1 (6,0..4)  w2 (5,0..4)  w3 (4,0..4)
//   w4 (7,5)(7,6)(7,7)(6,5)   w5 (6,6)(5,5)(3,3)(2,2)   w6 (3,0)(3,1)(3,2)(2,0)   w7 (2,1)(1,0)(1,1)(0,0)
// g: wave w < 4 takes column block 7 - w (its long row's own operand), w5 block 3, w6 block 2, w7
// blocks 1 and 0; f on wave 4. Same chains, same bits.
struct LmW128Tiles {
  int nt;         // tiles
  int tr[5], tc[5];  // (row block, column block)
  int nop;        // distinct operands
  int ops[6];     // their column blocks
  int ng;         // g column blocks (taken from the wave's operands)
  int gb[2];
};
__host__ __device__ constexpr LmW128Tiles lm_w128_tiles(int w) {
  switch (w) {
    case 0: return {5, {7, 7, 7, 7, 7}, {0, 1, 2, 3, 4}, 6, {7, 0, 1, 2, 3, 4}, 1, {7, 0}};
    case 1: return {5, {6, 6, 6, 6, 6}, {0, 1, 2, 3, 4}, 6, {6, 0, 1, 2, 3, 4}, 1, {6, 0}};
    case 2: return {5, {5, 5, 5, 5, 5}, {0, 1, 2, 3, 4}, 6, {5, 0, 1, 2, 3, 4}, 1, {5, 0}};
    case 3: return {5, {4, 4, 4, 4, 4}, {0, 1, 2, 3, 4}, 5, {4, 0, 1, 2, 3, 0}, 1, {4, 0}};
    case 4: return {4, {7, 7, 7, 6, 0}, {5, 6, 7, 5, 0}, 3, {7, 5, 6, 0, 0, 0}, 0, {0, 0}};
    case 5: return {4, {6, 5, 3, 2, 0}, {6, 5, 3, 2, 0}, 4, {6, 5, 3, 2, 0, 0}, 1, {3, 0}};
    case 6: return {4, {3, 3, 3, 2, 0}, {0, 1, 2, 0, 0}, 4, {3, 0, 1, 2, 0, 0}, 1, {2, 0}};
    default: return {4, {2, 1, 1, 0, 0}, {1, 0, 1, 0, 0}, 3, {2, 1, 0, 0, 0, 0}, 2, {1, 0}};
  }
}
// the slot of column block b among the wave's operands
__host__ __device__ constexpr int lm_w128_slot(const LmW128Tiles &t, int b) {
  for (int i = 0; i < t.nop; i++)
    if (t.ops[i] == b) return i;
  return 0;
}

// returns: some off-diagonal entry this wave published exceeds is_diagonal's threshold (:295-307)
template <int W>
__device__ inline bool lm_wide128_run8(const LmParams &p, int first, uint64_t pid, LmWide128Shared &sh,
                                       bool vec) {
  constexpr LmW128Tiles T = lm_w128_tiles(W);
  constexpr int S = kLmW128Stride;
  LmProblem *pr = p.prob + pid;
  const int lane = lane_id();
  const int half = lane >> 5, lp = lane & 31, kk = lane >> 4, cc = lane & 15;
  const uint64_t n = p.n, m = p.m;
  const double *A = p.Aw + pid * m * n, *y = p.yw + pid * m, *th = p.theta + pid * n;
  const uint64_t e0 = 2 * static_cast<uint64_t>(lp), e1 = e0 + 1, e2 = 64 + e0, e3 = e2 + 1;
  const double t0 = e0 < n ? th[e0] : 0.0, t1 = e1 < n ? th[e1] : 0.0;
  const double t2 = e2 < n ? th[e2] : 0.0, t3 = e3 < n ? th[e3] : 0.0;
  const uint64_t nstep = (m + 15) / 16;
  const int rr = 2 * W + half;  // the row of a group this lane's half stages
  double a[4], yv;
  auto fetch = [&](uint64_t s) {
    const uint64_t i = 16 * s + rr;
    const bool in = i < m;
    const double *row = A + (in ? i : 0) * n;
    if (vec) {  // n even: rows start 16-byte aligned; the matrix is read once: streamed (nt)
      const v2d_nt u = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt *>(row + (e0 < n ? e0 : 0)));
      const v2d_nt v = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt *>(row + (e2 < n ? e2 : 0)));
      a[0] = (in && e0 < n) ? u.x : 0.0;
      a[1] = (in && e0 < n) ? u.y : 0.0;
      a[2] = (in && e2 < n) ? v.x : 0.0;
      a[3] = (in && e2 < n) ? v.y : 0.0;
    } else {
      a[0] = (in && e0 < n) ? __builtin_nontemporal_load(row + (e0 < n ? e0 : 0)) : 0.0;
      a[1] = (in && e1 < n) ? __builtin_nontemporal_load(row + (e1 < n ? e1 : 0)) : 0.0;
      a[2] = (in && e2 < n) ? __builtin_nontemporal_load(row + (e2 < n ? e2 : 0)) : 0.0;
      a[3] = (in && e3 < n) ? __builtin_nontemporal_load(row + (e3 < n ? e3 : 0)) : 0.0;
    }
    yv = y[in ? i : 0];
  };
  auto stage = [&](uint64_t s, int buf) {
    double z = __builtin_fma(a[3], t3, __builtin_fma(a[2], t2, __builtin_fma(a[1], t1, a[0] * t0)));
    butterfly_levels<16>([&](auto off) { z = z + lane_xor<decltype(off)::value>(z); });
    const bool in = 16 * s + rr < m;
    const double tz = det_tanh(z);
    const double res = in ? yv - tz : 0.0;
    const double wgt = 1 - tz * tz;
    double *row = &sh.J[buf][rr * S];
    double2 lo, hi;
    lo.x = -(wgt * a[0]);
    lo.y = -(wgt * a[1]);
    hi.x = -(wgt * a[2]);
    hi.y = -(wgt * a[3]);
    *reinterpret_cast<double2 *>(row + e0) = lo;
    *reinterpret_cast<double2 *>(row + e2) = hi;
    if (lp == 0) sh.r[buf][rr] = res;
  };
  v4d acc[T.nt];
#pragma unroll
  for (int c = 0; c < T.nt; c++) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
  double gacc[2] = {0.0, 0.0};
  double facc[4] = {0.0, 0.0, 0.0, 0.0};

  fetch(0);
  stage(0, 0);
  if (nstep > 1) fetch(1);
  __syncthreads();
  for (uint64_t s = 0; s < nstep; s++) {
    const int buf = static_cast<int>(s & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const double *row = &sh.J[buf][(4 * ks + kk) * S];
      double op[T.nop];
#pragma unroll
      for (int b = 0; b < T.nop; b++) op[b] = row[16 * T.ops[b] + cc];
      if constexpr (T.ng > 0) {  // g: chain kk of column 16 b + cc takes the rows = kk (mod 4) in order
        const double rv = sh.r[buf][4 * ks + kk];
#pragma unroll
        for (int q = 0; q < T.ng; q++) gacc[q] = __builtin_fma(op[lm_w128_slot(T, T.gb[q])], rv, gacc[q]);
      }
#pragma unroll
      for (int c = 0; c < T.nt; c++)
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[lm_w128_slot(T, T.tr[c])], op[lm_w128_slot(T, T.tc[c])],
                                                      acc[c], 0, 0, 0);
    }
    if constexpr (W == 4) {  // f: chain (s mod 4, parity) takes its rows of the group in order
      const int sm = static_cast<int>(s & 3);
      double fw = sm == 0 ? facc[0] : sm == 1 ? facc[1] : sm == 2 ? facc[2] : facc[3];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const double rv = sh.r[buf][2 * k + half];
        fw = __builtin_fma(rv, rv, fw);
      }
      facc[0] = sm == 0 ? fw : facc[0];
      facc[1] = sm == 1 ? fw : facc[1];
      facc[2] = sm == 2 ? fw : facc[2];
      facc[3] = sm == 3 ? fw : facc[3];
    }
    if (s + 1 < nstep) {
      stage(s + 1, buf ^ 1);
      if (s + 2 < nstep) fetch(s + 2);
    }
    __syncthreads();
  }
  // ---- publish H = 2 J^T J (both triangles), g = 2 J^T r, f
  double *H = p.Hw + pid * n * n;
  bool offd = false;
#pragma unroll
  for (int c = 0; c < T.nt; c++) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const uint64_t row = 16 * T.tr[c] + kk + 4 * rg, col = 16 * T.tc[c] + cc;
      if (row < n && col < n) {
        const double v = 2 * acc[c][rg];
        H[row * n + col] = v;
        if (T.tr[c] != T.tc[c]) H[col * n + row] = v;
        offd |= row != col && v > 2.220446049250313e-16 * 1e12;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < T.ng; q++) {
    const double g0 = __shfl(gacc[q], cc, 64), g1 = __shfl(gacc[q], cc + 16, 64);
    const double g2 = __shfl(gacc[q], cc + 32, 64), g3 = __shfl(gacc[q], cc + 48, 64);
    const uint64_t col = 16 * T.gb[q] + cc;
    if (kk == 0 && col < n) p.gg[pid * n + col] = 2 * (((g0 + g1) + g2) + g3);
  }
  if constexpr (W == 4) {
    double f = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      f = f + __shfl(facc[w], 0, 64);
      f = f + __shfl(facc[w], 32, 64);
    }
    if (lane == 0) lm_publish_state(p, pr, first, f);
  }
  return offd;
}

__global__ __launch_bounds__(512, 4) void lm_wide128x8_tanh_eval_kernel(LmParams p, int first) {
  __shared__ __align__(16) LmWide128Shared sh;
  const uint64_t pid = blockIdx.x;
  if (!first && p.prob[pid].done) return;
  const bool vec = (p.n & 1) == 0;
  bool offd;
  switch (__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6)) {
    case 0: offd = lm_wide128_run8<0>(p, first, pid, sh, vec); break;
    case 1: offd = lm_wide128_run8<1>(p, first, pid, sh, vec); break;
    case 2: offd = lm_wide128_run8<2>(p, first, pid, sh, vec); break;
    case 3: offd = lm_wide128_run8<3>(p, first, pid, sh, vec); break;
    case 4: offd = lm_wide128_run8<4>(p, first, pid, sh, vec); break;
    case 5: offd = lm_wide128_run8<5>(p, first, pid, sh, vec); break;
    case 6: offd = lm_wide128_run8<6>(p, first, pid, sh, vec); break;
    default: offd = lm_wide128_run8<7>(p, first, pid, sh, vec); break;
  }
  // is_diagonal's verdict on the matrix just published, for the step (it saves the step a pass over H)
  const int any = __syncthreads_or(offd);
  if (threadIdx.x == 0) p.prob[pid].upper = 2 | (any ? 1 : 0);
}

// ---- 128 < n <= 256 in ONE pass over A too: the 136 lower tiles of the sixteen column blocks fit
// the accumulators of eight waves, seventeen each — wave w carries the tile rows 15 - w (16 - w
// tiles) and w (w + 1 tiles), whose operands are the column blocks 0 .. 15 - w of the staged rows.
// Otherwise lm_wide128x8_tanh_eval_kernel: a half-wave computes one row's z per group (its lanes'
// fma chains over their column pair of the four 64-column blocks, then the butterfly 16 .. 1 — the
// chains of lm_wide_mfma_tanh_eval_kernel's first phase, which this kernel replaces at these
// sizes), scales the row into a double-buffered LDS tile (stride 272 doubles: conflict-free
// ds_read_b64 for the MFMA operand pattern), g rides on the wave's two row-block operands, f on
// wave 4. A is read once; 1.42 -> 0.95 ms at n = 256, m = 512, batch 1024 (0.49 of the fp64 MFMA peak).
constexpr int kLmW256Stride = 272;
struct LmWide256Shared {  // 70 KB: dynamic LDS
  double J[2][16 * kLmW256Stride];
  double r[2][16];
};

template <int W>
__device__ inline bool lm_wide256_run8(const LmParams &p, int first, uint64_t pid, LmWide256Shared &sh,
                                       bool vec) {
  constexpr int S = kLmW256Stride;
  constexpr int RL = 15 - W, NOP = 16 - W;  // the long tile row; operands = column blocks 0 .. RL
  constexpr int NT = 17;                    // tiles: (RL, 0 .. RL), then (W, 0 .. W)
  LmProblem *pr = p.prob + pid;
  const int lane = lane_id();
  const int half = lane >> 5, lp = lane & 31, kk = lane >> 4, cc = lane & 15;
  const uint64_t n = p.n, m = p.m;
  const double *A = p.Aw + pid * m * n, *y = p.yw + pid * m, *th = p.theta + pid * n;
  double tv[8];  // theta at the lane's column pair of the four 64-column blocks
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const uint64_t e = 64 * b + 2 * static_cast<uint64_t>(lp);
    tv[2 * b] = e < n ? th[e] : 0.0;
    tv[2 * b + 1] = e + 1 < n ? th[e + 1] : 0.0;
  }
  const uint64_t nstep = (m + 15) / 16;
  const int rr = 2 * W + half;  // the row of a group this lane's half stages
  const bool long_row = 16 * static_cast<uint64_t>(RL) < n;  // tile row 15 - W has columns below n
  double a[8], yv;
  auto fetch = [&](uint64_t s) {
    const uint64_t i = 16 * s + rr;
    const bool in = i < m;
    const double *row = A + (in ? i : 0) * n;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint64_t e = 64 * b + 2 * static_cast<uint64_t>(lp);
      if (vec) {  // n even: rows start 16-byte aligned; the matrix is read once: streamed (nt)
        const v2d_nt u = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt *>(row + (e < n ? e : 0)));
        a[2 * b] = (in && e < n) ? u.x : 0.0;
        a[2 * b + 1] = (in && e < n) ? u.y : 0.0;
      } else {
        a[2 * b] = (in && e < n) ? __builtin_nontemporal_load(row + (e < n ? e : 0)) : 0.0;
        a[2 * b + 1] = (in && e + 1 < n) ? __builtin_nontemporal_load(row + (e + 1 < n ? e + 1 : 0)) : 0.0;
      }
    }
    yv = y[in ? i : 0];
  };
  auto stage = [&](uint64_t s, int buf) {
    double z = __builtin_fma(a[1], tv[1], a[0] * tv[0]);
#pragma unroll
    for (int b = 1; b < 4; b++) z = __builtin_fma(a[2 * b + 1], tv[2 * b + 1], __builtin_fma(a[2 * b], tv[2 * b], z));
    butterfly_levels<16>([&](auto off) { z = z + lane_xor<decltype(off)::value>(z); });
    const bool in = 16 * s + rr < m;
    const double tz = det_tanh(z);
    const double res = in ? yv - tz : 0.0;
    const double wgt = 1 - tz * tz;
    double *row = &sh.J[buf][rr * S];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      double2 jv;
      jv.x = -(wgt * a[2 * b]);
      jv.y = -(wgt * a[2 * b + 1]);
      *reinterpret_cast<double2 *>(row + 64 * b + 2 * lp) = jv;
    }
    if (lp == 0) sh.r[buf][rr] = res;
  };
  v4d acc[NT];
#pragma unroll
  for (int c = 0; c < NT; c++) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
  double gacc[2] = {0.0, 0.0};  // column blocks RL and W
  double facc[4] = {0.0, 0.0, 0.0, 0.0};

  fetch(0);
  stage(0, 0);
  if (nstep > 1) fetch(1);
  __syncthreads();
  for (uint64_t s = 0; s < nstep; s++) {
    const int buf = static_cast<int>(s & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const double *row = &sh.J[buf][(4 * ks + kk) * S];
      double op[NOP];
#pragma unroll
      for (int b = 0; b <= W; b++) op[b] = row[16 * b + cc];
      // g: chain kk of column 16 b + cc takes the rows = kk (mod 4) in order
      const double rv = sh.r[buf][4 * ks + kk];
      gacc[1] = __builtin_fma(op[W], rv, gacc[1]);
      if (long_row) {  // (wave-uniform: a tile row past n is all zeros and stays out of the matrix pipe)
#pragma unroll
        for (int b = W + 1; b < NOP; b++) op[b] = row[16 * b + cc];
        gacc[0] = __builtin_fma(op[RL], rv, gacc[0]);
#pragma unroll
        for (int c = 0; c <= RL; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[RL], op[c], acc[c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c <= W; c++)
        acc[RL + 1 + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[W], op[c], acc[RL + 1 + c], 0, 0, 0);
    }
    if constexpr (W == 4) {  // f: chain (s mod 4, parity) takes its rows of the group in order
      const int sm = static_cast<int>(s & 3);
      double fw = sm == 0 ? facc[0] : sm == 1 ? facc[1] : sm == 2 ? facc[2] : facc[3];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const double rq = sh.r[buf][2 * k + half];
        fw = __builtin_fma(rq, rq, fw);
      }
      facc[0] = sm == 0 ? fw : facc[0];
      facc[1] = sm == 1 ? fw : facc[1];
      facc[2] = sm == 2 ? fw : facc[2];
      facc[3] = sm == 3 ? fw : facc[3];
    }
    if (s + 1 < nstep) {
      stage(s + 1, buf ^ 1);
      if (s + 2 < nstep) fetch(s + 2);
    }
    __syncthreads();
  }
  // ---- publish H = 2 J^T J (both triangles), g = 2 J^T r, f
  double *H = p.Hw + pid * n * n;
  bool offd = false;  // some off-diagonal entry exceeds is_diagonal's threshold (:295-307)
  auto put_tile = [&](int tr, int tc, const v4d &tile) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const uint64_t row = 16 * static_cast<uint64_t>(tr) + kk + 4 * rg, col = 16 * static_cast<uint64_t>(tc) + cc;
      if (row < n && col < n) {
        const double v = 2 * tile[rg];
        H[row * n + col] = v;
        if (tr != tc) H[col * n + row] = v;
        offd |= row != col && v > 2.220446049250313e-16 * 1e12;
      }
    }
  };
#pragma unroll
  for (int c = 0; c <= RL; c++) put_tile(RL, c, acc[c]);
#pragma unroll
  for (int c = 0; c <= W; c++) put_tile(W, c, acc[RL + 1 + c]);
  auto put_g = [&](double gv, int b) {
    const double g0 = __shfl(gv, cc, 64), g1 = __shfl(gv, cc + 16, 64);
    const double g2 = __shfl(gv, cc + 32, 64), g3 = __shfl(gv, cc + 48, 64);
    const uint64_t col = 16 * static_cast<uint64_t>(b) + cc;
    if (kk == 0 && col < n) p.gg[pid * n + col] = 2 * (((g0 + g1) + g2) + g3);
  };
  put_g(gacc[0], RL);
  put_g(gacc[1], W);
  if constexpr (W == 4) {
    double f = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      f = f + __shfl(facc[w], 0, 64);
      f = f + __shfl(facc[w], 32, 64);
    }
    if (lane == 0) lm_publish_state(p, pr, first, f);
  }
  return offd;
}

__global__ __launch_bounds__(512, 2) void lm_wide256x8_tanh_eval_kernel(LmParams p, int first) {
  extern __shared__ __align__(16) unsigned char lm_w256_smem[];
  LmWide256Shared &sh = *reinterpret_cast<LmWide256Shared *>(lm_w256_smem);
  const uint64_t pid = blockIdx.x;
  if (!first && p.prob[pid].done) return;
  const bool vec = (p.n & 1) == 0;
  bool offd;
  switch (__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6)) {
    case 0: offd = lm_wide256_run8<0>(p, first, pid, sh, vec); break;
    case 1: offd = lm_wide256_run8<1>(p, first, pid, sh, vec); break;
    case 2: offd = lm_wide256_run8<2>(p, first, pid, sh, vec); break;
    case 3: offd = lm_wide256_run8<3>(p, first, pid, sh, vec); break;
    case 4: offd = lm_wide256_run8<4>(p, first, pid, sh, vec); break;
    case 5: offd = lm_wide256_run8<5>(p, first, pid, sh, vec); break;
    case 6: offd = lm_wide256_run8<6>(p, first, pid, sh, vec); break;
    default: offd = lm_wide256_run8<7>(p, first, pid, sh, vec); break;
  }
  const int any = __syncthreads_or(offd);  // is_diagonal's verdict, for the step
  if (threadIdx.x == 0) p.prob[pid].upper = 2 | (any ? 1 : 0);
}

// ---- 128 < n <= 1024 on the matrix cores. H no longer fits one set of accumulators, so the rows
// are passed over once per 128 x 128 super-block of its lower triangle (as lm_wide_tanh_eval_kernel
// does on the VALU), after a first phase that needs them all: z = A theta, tanh, residuals r and
// weights (kept in global memory, rw) and f. A pass (bj, bk) stages, sixteen rows at a time, the
// scaled Jacobian columns of block bj (and of bk, if another) in LDS and feeds v_mfma_f64_16x16x4:
// a diagonal super-block exactly as the n <= 128 kernel does (its 36 lower tiles, nine per
// wave), any other one as four 64 x 64 quadrants of sixteen tiles, one per wave; g rides on the
// operands of the passes (bj, 0). Every H_jk is still one k-ordered fma chain over the rows, g_j
// four chains by row mod 4, f eight chains, z the 32-lane chains of the order-1 oracle: same bits.
struct LmWideMfmaShared {
  double Jj[16 * kLmW128Stride];  // phase 1 keeps theta (up to 1024 doubles) here
  double Jk[16 * kLmW128Stride];
  double r[16];
  double part[8];
};

template <int W, bool DIAG>
__device__ __attribute__((noinline)) bool lm_wide_mfma_pass(const LmParams &p, uint64_t pid, LmWideMfmaShared &sh, uint64_t bj,
                                         uint64_t bk, bool vec) {
  constexpr int S = kLmW128Stride;
  constexpr int R = 7 - W;                  // DIAG: the wave's long tile row (its short one is W)
  constexpr int QJ = W >> 1, QK = W & 1;    // !DIAG: the wave's 64 x 64 quadrant
  constexpr int NACC = DIAG ? 9 : 16;
  const int lane = lane_id();
  const int half = lane >> 5, lp = lane & 31, kk = lane >> 4, cc = lane & 15;
  const uint64_t n = p.n, m = p.m;
  const double *A = p.Aw + pid * m * n;
  const double *rg = p.rw + pid * 2 * m, *wg = rg + m;
  const uint64_t nstep = (m + 15) / 16;
  const int rr[2] = {2 * W + half, 2 * (W + 4) + half};
  const bool with_g = bk == 0;
  double aj[2][4], ak[2][4], wv[2], rv2[2];
  auto fetch = [&](uint64_t s) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint64_t i = 16 * s + rr[q];
      const bool in = i < m;
      const double *row = A + (in ? i : 0) * n;
      auto get4 = [&](uint64_t base, double (&out)[4]) {
        const uint64_t e0 = base + 2 * static_cast<uint64_t>(lp), e2 = e0 + 64;
        if (vec) {
          const v2d_nt u = *reinterpret_cast<const v2d_nt *>(row + (e0 < n ? e0 : 0));
          const v2d_nt v = *reinterpret_cast<const v2d_nt *>(row + (e2 < n ? e2 : 0));
          out[0] = (in && e0 < n) ? u.x : 0.0;
          out[1] = (in && e0 < n) ? u.y : 0.0;
          out[2] = (in && e2 < n) ? v.x : 0.0;
          out[3] = (in && e2 < n) ? v.y : 0.0;
        } else {
          out[0] = (in && e0 < n) ? row[e0 < n ? e0 : 0] : 0.0;
          out[1] = (in && e0 + 1 < n) ? row[e0 + 1 < n ? e0 + 1 : 0] : 0.0;
          out[2] = (in && e2 < n) ? row[e2 < n ? e2 : 0] : 0.0;
          out[3] = (in && e2 + 1 < n) ? row[e2 + 1 < n ? e2 + 1 : 0] : 0.0;
        }
      };
      get4(128 * bj, aj[q]);
      if constexpr (!DIAG) get4(128 * bk, ak[q]);
      wv[q] = in ? wg[i] : 0.0;
      rv2[q] = in ? rg[i] : 0.0;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const double wgt = wv[q];
      double2 lo, hi;
      lo.x = -(wgt * aj[q][0]);
      lo.y = -(wgt * aj[q][1]);
      hi.x = -(wgt * aj[q][2]);
      hi.y = -(wgt * aj[q][3]);
      *reinterpret_cast<double2 *>(&sh.Jj[rr[q] * S + 2 * lp]) = lo;
      *reinterpret_cast<double2 *>(&sh.Jj[rr[q] * S + 64 + 2 * lp]) = hi;
      if constexpr (!DIAG) {
        lo.x = -(wgt * ak[q][0]);
        lo.y = -(wgt * ak[q][1]);
        hi.x = -(wgt * ak[q][2]);
        hi.y = -(wgt * ak[q][3]);
        *reinterpret_cast<double2 *>(&sh.Jk[rr[q] * S + 2 * lp]) = lo;
        *reinterpret_cast<double2 *>(&sh.Jk[rr[q] * S + 64 + 2 * lp]) = hi;
      }
      if (lp == 0) sh.r[rr[q]] = rv2[q];
    }
  };
  v4d acc[NACC];
#pragma unroll
  for (int c = 0; c < NACC; c++) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
  double gacc[4] = {0.0, 0.0, 0.0, 0.0};
  fetch(0);
  for (uint64_t s = 0; s < nstep; s++) {
    stage();
    if (s + 1 < nstep) fetch(s + 1);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const double rv = sh.r[4 * ks + kk];
      if constexpr (DIAG) {
        constexpr int GB = 6 - 2 * W;
        const double *row = &sh.Jj[(4 * ks + kk) * S];
        double op[R + 1];
#pragma unroll
        for (int b = 0; b <= R; b++) op[b] = row[16 * b + cc];
        if (with_g) {
          gacc[0] = __builtin_fma(op[GB], rv, gacc[0]);
          gacc[1] = __builtin_fma(op[GB + 1], rv, gacc[1]);
        }
#pragma unroll
        for (int c = 0; c <= R; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[R], op[c], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c <= W; c++)
          acc[R + 1 + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[W], op[c], acc[R + 1 + c], 0, 0, 0);
      } else {
        const double *rowj = &sh.Jj[(4 * ks + kk) * S + 64 * QJ], *rowk = &sh.Jk[(4 * ks + kk) * S + 64 * QK];
        double opj[4], opk[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          opj[b] = rowj[16 * b + cc];
          opk[b] = rowk[16 * b + cc];
        }
        if (with_g && QK == 0) {
#pragma unroll
          for (int b = 0; b < 4; b++) gacc[b] = __builtin_fma(opj[b], rv, gacc[b]);
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++)
            acc[4 * a + b] = __builtin_amdgcn_mfma_f64_16x16x4f64(opj[a], opk[b], acc[4 * a + b], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- publish the super-block (and its mirror image), and g for the columns of block bj
  double *H = p.Hw + pid * n * n;
  bool offd = false;  // some off-diagonal entry published here exceeds is_diagonal's threshold (:295-307)
  auto put_tile = [&](uint64_t rb, uint64_t cb, const v4d &tile, bool mirror) {
#pragma unroll
    for (int rg4 = 0; rg4 < 4; rg4++) {
      const uint64_t row = 128 * bj + 16 * rb + kk + 4 * rg4, col = 128 * bk + 16 * cb + cc;
      if (row < n && col < n) {
        const double v = 2 * tile[rg4];
        H[row * n + col] = v;
        if (mirror) H[col * n + row] = v;
        offd |= row != col && v > 2.220446049250313e-16 * 1e12;
      }
    }
  };
  auto put_g = [&](double gv, uint64_t colblock) {
    const double g0 = __shfl(gv, cc, 64), g1 = __shfl(gv, cc + 16, 64);
    const double g2 = __shfl(gv, cc + 32, 64), g3 = __shfl(gv, cc + 48, 64);
    const uint64_t col = 128 * bj + 16 * colblock + cc;
    if (kk == 0 && col < n) p.gg[pid * n + col] = 2 * (((g0 + g1) + g2) + g3);
  };
  if constexpr (DIAG) {
#pragma unroll
    for (int c = 0; c <= R; c++) put_tile(R, c, acc[c], R != c);
#pragma unroll
    for (int c = 0; c <= W; c++) put_tile(W, c, acc[R + 1 + c], W != c);
    if (with_g) {
      put_g(gacc[0], 6 - 2 * W);
      put_g(gacc[1], 7 - 2 * W);
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) put_tile(4 * QJ + a, 4 * QK + b, acc[4 * a + b], true);
    if (with_g && QK == 0) {
#pragma unroll
      for (int b = 0; b < 4; b++) put_g(gacc[b], 4 * QJ + b);
    }
  }
  return offd;
}

__global__ __launch_bounds__(256, 2) void lm_wide_mfma_tanh_eval_kernel(LmParams p, int first) {
  __shared__ __align__(16) LmWideMfmaShared sh;
  const uint64_t pid = blockIdx.x;
  LmProblem *pr = p.prob + pid;
  if (!first && pr->done) return;
  const int t = threadIdx.x, lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, lp = lane & 31;
  const uint64_t n = p.n, m = p.m;
  const bool vec = (n & 1) == 0;
  const double *A = p.Aw + pid * m * n, *y = p.yw + pid * m, *th = p.theta + pid * n;
  double *rg = p.rw + pid * 2 * m, *wg = rg + m;
  // ---- phase 1: z_i = A_i theta (a half-wave per row: each lane's fma chain runs over its column
  // pair of every 64-column block, then the butterfly 16 .. 1), tanh, r, weights, f. Wave w takes
  // the sixteen-row groups s = w (mod 4): the f chain (s mod 4, row parity) then lives in one
  // half-wave's registers, rows in order.
  double *theta = sh.Jj;
  for (uint64_t j = t; j < kLmWideMaxN; j += 256) theta[j] = j < n ? th[j] : 0.0;
  __syncthreads();
  const uint64_t nstep = (m + 15) / 16, nblk = (n + 63) / 64;
  double fw = 0.0;
  for (uint64_t s = w; s < nstep; s += 4) {
    for (int k0 = 0; k0 < 8; k0 += 2) {  // two rows of the half at a time: their chains interleave
      double z[2], yv[2];
      bool in[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint64_t i = 16 * s + 2 * (k0 + q) + half;
        in[q] = i < m;
        const double *row = A + (in[q] ? i : 0) * n;
        yv[q] = y[in[q] ? i : 0];
        double acc = 0.0;
        for (uint64_t c = 0; c < nblk; c++) {
          const uint64_t e0 = 64 * c + 2 * static_cast<uint64_t>(lp);
          double a0, a1;
          if (vec) {
            const v2d_nt u = *reinterpret_cast<const v2d_nt *>(row + (e0 < n ? e0 : 0));
            a0 = (in[q] && e0 < n) ? u.x : 0.0;
            a1 = (in[q] && e0 < n) ? u.y : 0.0;
          } else {
            a0 = (in[q] && e0 < n) ? row[e0 < n ? e0 : 0] : 0.0;
            a1 = (in[q] && e0 + 1 < n) ? row[e0 + 1 < n ? e0 + 1 : 0] : 0.0;
          }
          const double2 tv = *reinterpret_cast<const double2 *>(&theta[e0]);
          acc = c == 0 ? __builtin_fma(a1, tv.y, a0 * tv.x) : __builtin_fma(a1, tv.y, __builtin_fma(a0, tv.x, acc));
        }
        z[q] = acc;
      }
      butterfly_levels<16>([&](auto off) {
#pragma unroll
        for (int q = 0; q < 2; q++) z[q] = z[q] + lane_xor<decltype(off)::value>(z[q]);
      });
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint64_t i = 16 * s + 2 * (k0 + q) + half;
        const double tz = det_tanh(z[q]);
        const double res = in[q] ? yv[q] - tz : 0.0;
        if (lp == 0 && in[q]) {
          rg[i] = res;
          wg[i] = 1 - tz * tz;
        }
        fw = __builtin_fma(res, res, fw);
      }
    }
  }
  if (lp == 0) sh.part[2 * w + half] = fw;
  __syncthreads();  // r and the weights of every row are in place; theta's space is free
  // ---- phase 2: the super-blocks of the lower triangle
  const uint64_t SB = (n + 127) / 128;
  bool offd = false;
  for (uint64_t bj = 0; bj < SB; bj++)
    for (uint64_t bk = 0; bk <= bj; bk++) {
      if (bj == bk) {
        switch (w) {
          case 0: offd |= lm_wide_mfma_pass<0, true>(p, pid, sh, bj, bk, vec); break;
          case 1: offd |= lm_wide_mfma_pass<1, true>(p, pid, sh, bj, bk, vec); break;
          case 2: offd |= lm_wide_mfma_pass<2, true>(p, pid, sh, bj, bk, vec); break;
          default: offd |= lm_wide_mfma_pass<3, true>(p, pid, sh, bj, bk, vec); break;
        }
      } else {
        switch (w) {
          case 0: offd |= lm_wide_mfma_pass<0, false>(p, pid, sh, bj, bk, vec); break;
          case 1: offd |= lm_wide_mfma_pass<1, false>(p, pid, sh, bj, bk, vec); break;
          case 2: offd |= lm_wide_mfma_pass<2, false>(p, pid, sh, bj, bk, vec); break;
          default: offd |= lm_wide_mfma_pass<3, false>(p, pid, sh, bj, bk, vec); break;
        }
      }
    }
  const int any = __syncthreads_or(offd);  // is_diagonal's verdict on the published matrix, for the step
  if (t == 0) {
    double f = 0.0;
    for (int k = 0; k < 8; k++) f = f + sh.part[k];
    lm_publish_state(p, pr, first, f);
    pr->upper = 2 | (any ? 1 : 0);
  }
}

__global__ void lm_count_unfinished_kernel(LmParams p, unsigned long long *count) {
  const uint64_t pid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (pid < p.batch && !p.prob[pid].done) atomicAdd(count, 1ull);
}

}  // namespace nlsg
