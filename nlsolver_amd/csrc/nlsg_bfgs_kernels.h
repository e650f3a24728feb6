// nlsolver_amd/csrc/nlsg_bfgs_kernels.h — gfx950 kernels of the batched BFGS engine.
//
// Replaces (nlsolver.h): BFGS::solve 3196-3285, update_inverse_hessian 3130-3168,
// cvsrch 1673-1793, cstep 1527-1671, math::dot / norm 58-99.
//
// Data layout: inverse Hessians [batch][n][n] fp64 row-major (8 MiB per problem at
// n = 1024, 32 GiB for batch = 4096); vectors [batch][n]. All problems advance in
// lock step, one launch per phase:
//   bfgs_search_kernel   one wave per problem: stop tests, reset guard, the whole
//                        More-Thuente search (<= 20 f+g evaluations), s, x, g, y, rho
//   bfgs_hy_kernel       t = H y            (reads H once)            — HBM bound
//   bfgs_update_kernel   H -= rho (s t^T + t s^T + denom s s^T) fused with the next
//                        direction d = -H' g (reads H, writes H)      — HBM bound
// A freshly reset H (identity, 3212 / 3253-3260) is never materialised: a per-problem
// flag makes the two streaming kernels treat H as I (no read).
// Every reduction is the lane tree of DESIGN.md (element e -> lane (e%128)/2, in-lane
// sequential, 64-lane xor butterfly), mirrored by oracle/oracle_bfgs.c (tree = 1).
#pragma once

#include "nlsg_common.h"

namespace nlsg {

struct BfgsProblem {  // per-problem scalars
  double prev_norm, cur_norm, rho, fval;
  double denom;  // symmetric restatement: rho y^T t + 1 of the current update (bfgs_sym_reduce_kernel)
  uint64_t iter, fcalls, gcalls;
  int32_t done, identity;
};

struct BfgsParams {
  double *H;                  // [batch][n][ldh]
  uint64_t ldh;               // doubles per row of H: n — or, reference order with a row of a multiple of 4 KiB,
                              // n + 16: the lane-per-row passes touch 64 rows at one column offset, which
                              // at such a stride is ONE memory channel (0.65 of the rate at n = 1008, measured)
  double *x, *g, *dir, *s, *y, *t;  // [batch][n]
  BfgsProblem *prob;          // [batch]
  const double *qd, *qb;      // objective parameters d, b [n]
  const double *zero;
  uint64_t batch, n, max_iter;
  double grad_eps, alpha, qc;
  int32_t model;  // kBfgsQuad, or the nlsg_objective minimised with a finite-difference gradient
  int32_t seq;    // NLSG_BFGS_REFERENCE_ORDER: every sum in index order (nlsg_common.h wave_sum_seq)
  // symmetric restatement (NLSG_BFGS_SYMMETRIC): the upper 128 x 128 blocks of H, and the block
  // partials of the two products (see "symmetric restatement" below)
  double *Hs;    // [batch][nstored][128][128]
  double *part;  // [batch][nb][nb][128]
  uint32_t nb, nstored;
};

// ---- wave-level vector helpers (vectors replicated in every lane layout) --------
template <int CHUNKS>
__device__ inline double wave_dot(const double (&a)[CHUNKS][2], const double (&b)[CHUNKS][2]) {
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    acc = acc + a[c][0] * b[c][0];
    acc = acc + a[c][1] * b[c][1];
  }
  return wave_sum(acc);
}
template <int CHUNKS>
__device__ inline double wave_total(const double (&a)[CHUNKS][2]) {
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    acc = acc + a[c][0];
    acc = acc + a[c][1];
  }
  return wave_sum(acc);
}
// (the serial sums go through the wave's LDS buffer: wave_sum_seq_buf / wave_objective_seq_buf,
// nlsg_common.h; `buf`: 128 CHUNKS doubles)
template <int CHUNKS>
__device__ inline void bfgs_stage_terms(const double (&t)[CHUNKS][2], double *buf) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++)
    *reinterpret_cast<double2 *>(buf + 128 * c + 2 * lane) = make_double2(t[c][0], t[c][1]);
}
// two sums side by side (two independent chains: their additions interleave); buf: 2 x 128 CHUNKS
template <int CHUNKS>
__device__ inline void wave_sum_seq_lds2(const double (&ta)[CHUNKS][2], const double (&tb)[CHUNKS][2], uint64_t n,
                                         double *buf, double &sa, double &sb) {
  double *bufb = buf + 128 * CHUNKS;
  bfgs_stage_terms<CHUNKS>(ta, buf);
  bfgs_stage_terms<CHUNKS>(tb, bufb);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double a = 0.0, b = 0.0;
  const int m = static_cast<int>(n);
#pragma unroll 8
  for (int e = 0; e < m; e++) {
    a = a + buf[e];
    b = b + bufb[e];
  }
  __builtin_amdgcn_wave_barrier();
  sa = a;
  sb = b;
}

// math::dot (nlsolver.h:58-67) in the engine's summation order: the lane tree, or — reference
// order (`seq`, wave-uniform; lds = the wave's buffer) — the products added in index order
template <int CHUNKS>
__device__ inline double bfgs_dot(const double (&a)[CHUNKS][2], const double (&b)[CHUNKS][2],
                                  uint64_t n, int seq, double *lds) {
  if (!seq) return wave_dot<CHUNKS>(a, b);
  double t[CHUNKS][2];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    t[c][0] = a[c][0] * b[c][0];
    t[c][1] = a[c][1] * b[c][1];
  }
  return wave_sum_seq_buf<CHUNKS>(t, n, lds);
}

// The G6 quadratic and its gradient (operation order of oracle_bfgs.c quad_f / quad_g).
template <int CHUNKS>
__device__ inline double quad_f(const double (&x)[CHUNKS][2], const double (&d)[CHUNKS][2],
                                const double (&b)[CHUNKS][2], double c, uint64_t n = 0, int seq = 0,
                                double *lds = nullptr) {
  if (seq) {  // reference order: the three sums of oracle_bfgs.c quad_f_raw, each in index order
    double tq[CHUNKS][2], tl[CHUNKS][2];
#pragma unroll
    for (int k = 0; k < CHUNKS; k++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        tq[k][h] = d[k][h] * x[k][h] * x[k][h];
        tl[k][h] = b[k][h] * x[k][h];
      }
    double qq, lin;
    wave_sum_seq_lds2<CHUNKS>(tq, tl, n, lds, qq, lin);
    const double sx = wave_sum_seq_buf<CHUNKS>(x, n, lds);
    return 0.5 * qq + 0.5 * c * (sx * sx) - lin;
  }
  double aq = 0.0, al = 0.0;
#pragma unroll
  for (int k = 0; k < CHUNKS; k++) {
    aq = aq + d[k][0] * x[k][0] * x[k][0];
    aq = aq + d[k][1] * x[k][1] * x[k][1];
    al = al + b[k][0] * x[k][0];
    al = al + b[k][1] * x[k][1];
  }
  const double qq = wave_sum(aq);
  const double sx = wave_total<CHUNKS>(x);
  const double lin = wave_sum(al);
  return 0.5 * qq + 0.5 * c * (sx * sx) - lin;
}
template <int CHUNKS>
__device__ inline void quad_g(const double (&x)[CHUNKS][2], const double (&d)[CHUNKS][2],
                              const double (&b)[CHUNKS][2], double c, uint64_t n,
                              double (&g)[CHUNKS][2], int seq = 0, double *lds = nullptr) {
  const double sx = seq ? wave_sum_seq_buf<CHUNKS>(x, n, lds) : wave_total<CHUNKS>(x);
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < CHUNKS; k++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint64_t e = static_cast<uint64_t>(k) * 128 + 2 * static_cast<uint64_t>(lane) + h;
      const double v = d[k][h] * x[k][h] + c * sx - b[k][h];
      g[k][h] = (e < n) ? v : 0.0;  // lanes past n hold zeros in every vector
    }
  }
}

// ---- what is minimised ---------------------------------------------------------------
// kBfgsQuad: the G6 quadratic with its analytic gradient functor. Otherwise a built-in
// objective (Objective<OBJ>) with the reference's DEFAULT gradient, fin_diff =
// finite_difference_gradient<Callable, scalar_t, 1> (nlsolver.h:1385-1413, 2849-2855): per
// coordinate four probes x_d + {-2,-1,1,2} eps weighted {1,-8,8,-1}, divided by 12 eps,
// eps = DBL_EPSILON * 10e7. The probes go through the counting wrapper (3218-3224), so each
// counts as a function call. One wave evaluates them one after the other, each a full
// evaluation in the lane-tree order (oracle_bfgs.c model_g).
constexpr int kBfgsQuad = -1;

// LDS of the search / init kernels: only the reference-order finite-difference gradient uses any — per
// wave the point and its objective terms, 2 x 128 CHUNKS doubles (the launch passes 0 bytes otherwise)
__host__ __device__ constexpr size_t bfgs_fd_seq_lds_bytes(int chunks) {
  return 4 * 2 * 128 * static_cast<size_t>(chunks) * sizeof(double);  // (serial_sum_lds reads ahead inside it)
}

template <int MODEL, int CHUNKS>
struct BfgsModel {  // finite differences on Objective<MODEL>
  uint64_t n;
  int seq;
  double *lds;  // this wave's 2 x 128 CHUNKS doubles (reference order only)
  template <bool VEC>
  __device__ inline void load(const BfgsParams &p) {
    extern __shared__ __align__(16) double bfgs_smem[];
    n = p.n;
    seq = p.seq;
    lds = bfgs_smem + (threadIdx.x >> 6) * (2 * 128 * CHUNKS);
  }
  __device__ inline double value(const double (&x)[CHUNKS][2]) const {
    if (!seq) return wave_objective<MODEL, CHUNKS>(x, n);
    return wave_objective_seq_buf<MODEL, CHUNKS>(x, n, lds);  // (whole-vector bodies: rejected at engine creation)
  }
  __device__ inline double f(const double (&x)[CHUNKS][2], uint64_t &fcalls) const {
    fcalls++;
    return value(x);
  }
  // fin_diff in REFERENCE ORDER, a probe per LANE. Probe (d, s) is the objective at x + delta_s e_d
  // with its terms added in index order. Only the terms that contain x_d differ from the base
  // point's (t_d; for a chain objective t_{d-1} too), and the running sum up to the first of them is
  // the base point's own prefix sum — bit for bit, the additions are the same ones. So:
  //   * the base terms t_e and the point go to LDS once;
  //   * the prefix sums S_e = (..(0 + t_0) + ..) + t_e are ONE serial chain per gradient (not one
  //     per probe), advanced pass by pass; lane L captures the one its coordinate starts from;
  //   * a pass handles 64 coordinates, lane L = coordinate d0 + L, its four probes side by side
  //     (four independent chains: the additions pipeline): start value, the one or two modified
  //     terms, then the tail t_{d+1} .. t_{nt-1} read from LDS at a wave-uniform address (lanes
  //     whose tail has not begun yet sit out);
  //   * the pass's 64 gradient entries return to the wave layout through two lane gathers.
  // 4 n probes cost n / 64 passes of ~n additions per lane instead of 4 n serial sums of n terms
  // each by the whole wave: every probe's value has the bits of wave_objective_seq at that point
  // (tests: the reference's own runs, tests/golden/bfgs_fd.json, and oracle tree 0 on batches).
  __device__ inline void grad_seq(const double (&x)[CHUNKS][2], double (&g)[CHUNKS][2]) const {
    using O = Objective<MODEL>;
    constexpr double eps = 2.220446049250313e-16 * 10e7;
    constexpr double coeff[4] = {1, -8, 8, -1}, coeff2[4] = {-2, -1, 1, 2};
    constexpr double dd_val = 12 * eps;
    constexpr int off = O::kChain ? 2 : 1;  // a coordinate's first modified term is t_{d - off + 1}
    const int lane = lane_id();
    const int D = static_cast<int>(n), nt = static_cast<int>(O::n_terms(n));
    double *xs = lds, *ts = lds + 128 * CHUNKS;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      double xn = 0.0;
      if (O::kChain) {
        const double same = lane_down1(x[c][0]);
        double next = 0.0;
        if (c + 1 < CHUNKS) next = lane_first(x[c + 1][0]);
        xn = (lane == 63) ? next : same;
      }
      *reinterpret_cast<double2 *>(xs + 128 * c + 2 * lane) = make_double2(x[c][0], x[c][1]);
      *reinterpret_cast<double2 *>(ts + 128 * c + 2 * lane) =
          make_double2(O::term(x[c][0], x[c][1]), O::term(x[c][1], xn));
      g[c][0] = g[c][1] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double run = 0.0;  // S_{e_run - 1}
    int e_run = 0;
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + lane;
      const int dc = d < D ? d : D - 1;
      const double xd = xs[dc], xm = xs[dc > 0 ? dc - 1 : 0], xp = xs[dc + 1 < D ? dc + 1 : dc];
      // the prefix chain up to the last start value this pass needs
      double a = 0.0;
      const int cap = d - off, e_hi = d0 + 63 - off < nt - 1 ? d0 + 63 - off : nt - 1;
#pragma unroll 4
      for (; e_run <= e_hi; e_run++) {
        run = run + ts[e_run];
        a = cap == e_run ? run : a;
      }
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
      // the tail: t_e for e > d. Inside the pass's own window the lanes join one by one.
      int e = d0 + 1;
      const int e_win = d0 + 64 < nt ? d0 + 64 : nt;
#pragma unroll 4
      for (; e < e_win; e++) {
        const double te = ts[e];
        if (e > d) {
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] = acc[q] + te;
        }
      }
#pragma unroll 8
      for (; e < nt; e++) {
        const double te = ts[e];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = acc[q] + te;
      }
      double ga = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) ga = ga + coeff[q] * O::finish(acc[q], n);
      const double gd = ga / dd_val;
      // coordinate d0 + L sits in lane L; the wave layout wants 128 c + 2 l + k in lane l
      const double r0 = __shfl(gd, (2 * lane) & 63, 64), r1 = __shfl(gd, (2 * lane + 1) & 63, 64);
      const int c0 = d0 >> 7, half = (d0 >> 6) & 1;
#pragma unroll
      for (int c = 0; c < CHUNKS; c++)
        if (c == c0 && (lane >> 5) == half) {
          const int e0 = 128 * c + 2 * lane;
          g[c][0] = e0 < D ? r0 : 0.0;
          g[c][1] = e0 + 1 < D ? r1 : 0.0;
        }
    }
    __builtin_amdgcn_wave_barrier();  // (the next call's stores come after this call's reads)
  }
  __device__ inline void grad(const double (&x)[CHUNKS][2], double (&g)[CHUNKS][2],
                              uint64_t &fcalls, uint64_t &gcalls) const {
    gcalls++;
    if constexpr (!Objective<MODEL>::kWhole) {
      if (seq) {  // wave-uniform
        grad_seq(x, g);
        fcalls += 4 * n;
        return;
      }
    }
    constexpr double eps = 2.220446049250313e-16 * 10e7;
    constexpr double coeff[4] = {1, -8, 8, -1}, coeff2[4] = {-2, -1, 1, 2};
    constexpr double dd_val = 12 * eps;
    const int lane = lane_id();
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) g[c][0] = g[c][1] = 0.0;
    for (uint64_t d = 0; d < n; d++) {
      const int cc = static_cast<int>(d >> 7), kk = static_cast<int>(d & 1);
      const bool mine = lane == static_cast<int>((d & 127) >> 1);
      double acc = 0.0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        double xp[CHUNKS][2];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++)
#pragma unroll
          for (int k = 0; k < 2; k++)
            xp[c][k] = (mine && c == cc && k == kk) ? x[c][k] + coeff2[s] * eps : x[c][k];
        acc = acc + coeff[s] * value(xp);
      }
      fcalls += 4;
      const double gd = acc / dd_val;
#pragma unroll
      for (int c = 0; c < CHUNKS; c++)
#pragma unroll
        for (int k = 0; k < 2; k++)
          if (mine && c == cc && k == kk) g[c][k] = gd;
    }
  }
};

template <int CHUNKS>
struct BfgsModel<kBfgsQuad, CHUNKS> {
  double qd[CHUNKS][2], qb[CHUNKS][2];
  double qc;
  uint64_t n;
  int seq;
  double *lds;  // this wave's 2 x 128 CHUNKS doubles (reference order only)
  template <bool VEC>
  __device__ inline void load(const BfgsParams &p) {
    extern __shared__ __align__(16) double bfgs_smem[];
    n = p.n;
    seq = p.seq;
    lds = bfgs_smem + (threadIdx.x >> 6) * (2 * 128 * CHUNKS);
    qc = p.qc;
    load_row<CHUNKS, VEC>(p.qd, n, p.zero, qd);
    load_row<CHUNKS, VEC>(p.qb, n, p.zero, qb);
  }
  __device__ inline double f(const double (&x)[CHUNKS][2], uint64_t &fcalls) const {
    fcalls++;
    return quad_f<CHUNKS>(x, qd, qb, qc, n, seq, lds);
  }
  __device__ inline void grad(const double (&x)[CHUNKS][2], double (&g)[CHUNKS][2],
                              uint64_t &, uint64_t &gcalls) const {
    gcalls++;
    quad_g<CHUNKS>(x, qd, qb, qc, n, g, seq, lds);
  }
};

// ---- More-Thuente (scalar code, identical in every lane) -------------------------
__device__ inline double mt_max_abs3(double x, double y, double z) {
  return fmax(fabs(x), fmax(fabs(y), fabs(z)));
}
__device__ inline double mt_min(double a, double b) { return b < a ? b : a; }
__device__ inline double mt_max(double a, double b) { return a < b ? b : a; }
__device__ inline double mt_clamp(double v, double lo, double hi) {
  return v < lo ? lo : (hi < v ? hi : v);
}

// cstep, nlsolver.h:1527-1671
__device__ inline int mt_cstep(double &stx, double &fx, double &dx, double &sty, double &fy,
                               double &dy, double &stp, double fp, double dp, int &brackt,
                               double stpmin, double stpmax, int &info) {
  info = 0;
  int bound;
  if ((brackt & ((stp <= mt_min(stx, sty)) || (stp >= mt_max(stx, sty)))) ||
      (dx * (stp - stx) >= 0.0) || (stpmax < stpmin))
    return -1;
  const double sgnd = dp * (dx / fabs(dx));
  double stpf = 0, stpc, stpq;
  if (fp > fx) {
    info = 1;
    bound = 1;
    const double theta = 3. * (fx - fp) / (stp - stx) + dx + dp;
    const double s = mt_max_abs3(theta, dx, dp);
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta;
    const double q = ((gamma - dx) + gamma) + dp;
    const double r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.) * (stp - stx);
    if (fabs(stpc - stx) < fabs(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2;
    brackt = 1;
  } else if (sgnd < 0.0) {
    info = 2;
    bound = 0;
    const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = mt_max_abs3(theta, dx, dp);
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + dx;
    const double r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = 1;
  } else if (fabs(dp) < fabs(dx)) {
    info = 3;
    bound = 1;
    const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = mt_max_abs3(theta, dx, dp);
    double gamma = s * sqrt(mt_max(0., (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = (gamma + (dx - dp)) + gamma;
    const double r = p / q;
    if ((r < 0.0) & (gamma != 0.0))
      stpc = stp + r * (stx - stp);
    else if (stp > stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt)
      stpf = (fabs(stp - stpc) < fabs(stp - stpq)) ? stpc : stpq;
    else
      stpf = (fabs(stp - stpc) > fabs(stp - stpq)) ? stpc : stpq;
  } else {
    info = 4;
    bound = 0;
    if (brackt) {
      const double theta = 3 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = mt_max_abs3(theta, dy, dp);
      double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + dy;
      const double r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stpf = mt_clamp(stpf, stpmin, stpmax);
  stp = stpf;
  if (brackt & bound) {
    if (sty > stx)
      stp = mt_min(stx + 0.66 * (sty - stx), stp);
    else
      stp = mt_max(stx + 0.66 * (sty - stx), stp);
  }
  return 0;
}

template <int CHUNKS, bool VEC>
__device__ inline void load_vec(const double *p, uint64_t n, const double *zero,
                                double (&v)[CHUNKS][2]) {
  load_row<CHUNKS, VEC>(p, n, zero, v);
}

// One wave per problem: everything of an iteration except the two H passes.
// Long vectors (CHUNKS >= 4) want the whole register file of a SIMD for one wave; short ones
// (every finite-difference model) leave room for four, which is what hides the latency of the
// 4 n dependent evaluations per gradient.
template <int CHUNKS, bool VEC, int MODEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CHUNKS >= 4 ? 1 : 4))) void
bfgs_search_kernel(BfgsParams p) {
  const uint64_t pid = static_cast<uint64_t>(blockIdx.x) * 4 +
                       __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (pid >= p.batch) return;
  BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const int lane = lane_id();
  const uint64_t n = p.n;
  double x[CHUNKS][2], g[CHUNKS][2], dir[CHUNKS][2];
  BfgsModel<MODEL, CHUNKS> model;
  model.template load<VEC>(p);
  load_vec<CHUNKS, VEC>(p.x + pid * n, n, p.zero, x);
  load_vec<CHUNKS, VEC>(p.g + pid * n, n, p.zero, g);
  load_vec<CHUNKS, VEC>(p.dir + pid * n, n, p.zero, dir);
  uint64_t iter = pr->iter, fcalls = pr->fcalls, gcalls = pr->gcalls;
  double prev_norm = pr->prev_norm, cur_norm = pr->cur_norm;

  // stop tests, nlsolver.h:3239-3246
  if (iter >= p.max_iter || cur_norm < p.grad_eps || fabs(cur_norm - prev_norm) < p.grad_eps ||
      isinf(cur_norm)) {
    const double fv = model.f(x, fcalls);
    if (lane == 0) {
      pr->fval = fv;
      pr->fcalls = fcalls;
      pr->done = 1;
    }
    return;
  }
  // direction: d = -H g, computed by the previous update pass (or -g while H = I)
  // H in memory is valid after the previous update pass; only the very first
  // iteration (H = I, :3212) and the reset guard below use the identity shortcut
  int identity = (iter == 0) ? 1 : 0;
  if (identity) {
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      dir[c][0] = -g[c][0];
      dir[c][1] = -g[c][1];
    }
  }
  const double phi = bfgs_dot<CHUNKS>(g, dir, n, p.seq, model.lds);
  if ((phi > 0) || isnan(phi) || cur_norm > prev_norm) {  // reset guard, :3253-3260
    identity = 1;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      dir[c][0] = -g[c][0];
      dir[c][1] = -g[c][1];
    }
  }
  double pg[CHUNKS][2];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    pg[c][0] = g[c][0];
    pg[c][1] = g[c][1];
  }
  // more_thuente_search (1880-1891) -> cvsrch (1673-1793)
  const double f0 = model.f(x, fcalls);
  double stp = p.alpha;
  {
    int info = 0, infoc = 1;
    const double xtol = 1e-15, ftol = 1e-4, gtol = 1e-2, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4;
    const int maxfev = 20;
    int nfev = 0;
    const double dginit = bfgs_dot<CHUNKS>(g, dir, n, p.seq, model.lds);
    if (!(dginit >= 0.0)) {
      int brackt = 0, stage1 = 1;
      const double finit = f0, dgtest = ftol * dginit;
      double width = stpmax - stpmin, width1 = 2 * width;
      double stx = 0.0, fx = finit, dgx = dginit, sty = 0.0, fy = finit, dgy = dginit;
      double stmin, stmax;
      for (;;) {
        if (brackt) {
          stmin = mt_min(stx, sty);
          stmax = mt_max(stx, sty);
        } else {
          stmin = stx;
          stmax = stp + xtrapf * (stp - stx);
        }
        stp = mt_clamp(stp, stpmin, stpmax);
        if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (nfev >= maxfev - 1) ||
            (infoc == 0) || (brackt && ((stmax - stmin) <= (xtol * stmax))))
          stp = stx;
        double tmp[CHUNKS][2];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
          tmp[c][0] = x[c][0] + stp * dir[c][0];
          tmp[c][1] = x[c][1] + stp * dir[c][1];
        }
        const double fcur = model.f(tmp, fcalls);
        model.grad(tmp, g, fcalls, gcalls);
        nfev++;
        const double dg = bfgs_dot<CHUNKS>(g, dir, n, p.seq, model.lds);
        const double ftest1 = finit + stp * dgtest;
        if ((brackt & ((stp <= stmin) | (stp >= stmax))) | (infoc == 0)) info = 6;
        if ((stp == stpmax) & (fcur <= ftest1) & (dg <= dgtest)) info = 5;
        if ((stp == stpmin) & ((fcur > ftest1) | (dg >= dgtest))) info = 4;
        if (nfev >= maxfev) info = 3;
        if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
        if ((fcur <= ftest1) & (fabs(dg) <= gtol * (-dginit))) info = 1;
        if (info != 0) break;
        if (stage1 & (fcur <= ftest1) & (dg >= mt_min(ftol, gtol) * dginit)) stage1 = 0;
        if (stage1 & (fcur <= fx) & (fcur > ftest1)) {
          const double fm = fcur - stp * dgtest;
          double fxm = fx - stx * dgtest, fym = fy - sty * dgtest;
          const double dgm = dg - dgtest;
          double dgxm = dgx - dgtest, dgym = dgy - dgtest;
          mt_cstep(stx, fxm, dgxm, sty, fym, dgym, stp, fm, dgm, brackt, stmin, stmax, infoc);
          fx = fxm + stx * dgtest;
          fy = fym + sty * dgtest;
          dgx = dgxm + dgtest;
          dgy = dgym + dgtest;
        } else {
          mt_cstep(stx, fx, dgx, sty, fy, dgy, stp, fcur, dg, brackt, stmin, stmax, infoc);
        }
        if (brackt) {
          if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
          width1 = width;
          width = fabs(sty - stx);
        }
      }
    }
  }
  // s = rate d; x += s; g = grad(x); norms; y; rho  (3266-3278)
  double s[CHUNKS][2], y[CHUNKS][2];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      s[c][h] = dir[c][h] * stp;
      x[c][h] = x[c][h] + s[c][h];
    }
  }
  model.grad(x, g, fcalls, gcalls);
  prev_norm = cur_norm;
  cur_norm = sqrt(bfgs_dot<CHUNKS>(g, g, n, p.seq, model.lds));
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    y[c][0] = g[c][0] - pg[c][0];
    y[c][1] = g[c][1] - pg[c][1];
  }
  double rho = bfgs_dot<CHUNKS>(y, s, n, p.seq, model.lds);
  rho = 1 / rho;
  store_row<CHUNKS, VEC>(p.x + pid * n, n, x);
  store_row<CHUNKS, VEC>(p.g + pid * n, n, g);
  store_row<CHUNKS, VEC>(p.s + pid * n, n, s);
  store_row<CHUNKS, VEC>(p.y + pid * n, n, y);
  if (lane == 0) {
    pr->prev_norm = prev_norm;
    pr->cur_norm = cur_norm;
    pr->rho = rho;
    pr->iter = iter + 1;
    pr->fcalls = fcalls;
    pr->gcalls = gcalls;
    pr->identity = identity;
  }
}

// g = grad(x0), norms as at nlsolver.h:3234-3237
template <int CHUNKS, bool VEC, int MODEL>
__global__ __launch_bounds__(256) void bfgs_init_kernel(BfgsParams p) {
  const uint64_t pid = static_cast<uint64_t>(blockIdx.x) * 4 +
                       __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  if (pid >= p.batch) return;
  const uint64_t n = p.n;
  double x[CHUNKS][2], g[CHUNKS][2];
  BfgsModel<MODEL, CHUNKS> model;
  model.template load<VEC>(p);
  load_vec<CHUNKS, VEC>(p.x + pid * n, n, p.zero, x);
  uint64_t fcalls = 0, gcalls = 0;
  model.grad(x, g, fcalls, gcalls);
  store_row<CHUNKS, VEC>(p.g + pid * n, n, g);
  if (lane_id() == 0) {
    BfgsProblem *pr = p.prob + pid;
    pr->prev_norm = 1e9;
    pr->cur_norm = 1e8;
    pr->rho = 0.0;
    pr->fval = 0.0;
    pr->iter = 0;
    pr->fcalls = fcalls;
    pr->gcalls = gcalls;
    pr->done = 0;
    pr->identity = 1;
  }
}

typedef double bfgs_v2d __attribute__((ext_vector_type(2)));
// the blocks of H are touched once per pass: streamed past the caches' retention (nt)
__device__ inline double2 bfgs_stream_load(const double *p) {
  const bfgs_v2d v = __builtin_nontemporal_load(reinterpret_cast<const bfgs_v2d *>(p));
  return make_double2(v.x, v.y);
}
__device__ inline void bfgs_stream_store(double *p, double2 v) {
  bfgs_v2d w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<bfgs_v2d *>(p));
}
// a row of H in the lane layout of load_row / store_row, streamed (nt)
template <int CHUNKS, bool VEC>
__device__ inline void bfgs_load_h_row(const double *__restrict__ row, uint64_t D,
                                       const double *__restrict__ zero, double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (VEC) {
      const double2 t = bfgs_stream_load((e0 < D) ? row + e0 : zero);
      v[c][0] = t.x;
      v[c][1] = t.y;
    } else {
      v[c][0] = __builtin_nontemporal_load((e0 < D) ? row + e0 : zero);
      v[c][1] = __builtin_nontemporal_load((e0 + 1 < D) ? row + e0 + 1 : zero);
    }
  }
}
template <int CHUNKS, bool VEC>
__device__ inline void bfgs_store_h_row(double *__restrict__ row, uint64_t D, const double (&v)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    if (VEC) {
      if (e0 < D) bfgs_stream_store(row + e0, make_double2(v[c][0], v[c][1]));
    } else {
      if (e0 < D) __builtin_nontemporal_store(v[c][0], row + e0);
      if (e0 + 1 < D) __builtin_nontemporal_store(v[c][1], row + e0 + 1);
    }
  }
}

constexpr int kBfgsRowsPerWave = 8;  // rows a wave streams per launch (vector kept in regs)

// t = H y (first loop of update_inverse_hessian, 3139-3142). Block = 4 waves = 32 rows.
// (Reference order: bfgs_hy_seq_kernel / bfgs_update_seq_kernel below.)
template <int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void bfgs_hy_kernel(BfgsParams p, uint32_t blocks_per_problem) {
  const uint64_t pid = blockIdx.x / blocks_per_problem;
  const BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const uint64_t n = p.n;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const uint64_t row0 =
      (static_cast<uint64_t>(blockIdx.x % blocks_per_problem) * 4 + wid) * kBfgsRowsPerWave;
  if (row0 >= n) return;
  const int lane = lane_id();
  double y[CHUNKS][2];
  load_vec<CHUNKS, VEC>(p.y + pid * n, n, p.zero, y);
  double *t = p.t + pid * n;
  if (pr->identity) {  // H = I: t = y exactly, nothing to stream
    for (int r = 0; r < kBfgsRowsPerWave; r++) {
      const uint64_t j = row0 + r;
      if (j < n && lane == 0) t[j] = p.y[pid * n + j];
    }
    return;
  }
  const double *Hp = p.H + pid * n * n;
#pragma unroll 2
  for (int r = 0; r < kBfgsRowsPerWave; r++) {
    const uint64_t j = row0 + r;
    if (j >= n) break;
    double h[CHUNKS][2];
    bfgs_load_h_row<CHUNKS, VEC>(Hp + j * n, n, p.zero, h);
    const double v = wave_dot<CHUNKS>(y, h);  // dot(grad_diff, H row), :3140
    if (lane == 0) t[j] = v;
  }
}

// rank-2 update (3151-3164) fused with the next direction d = -H' g (3248-3251)
template <int CHUNKS, bool VEC>
__global__ __launch_bounds__(256) void bfgs_update_kernel(BfgsParams p,
                                                        uint32_t blocks_per_problem) {
  const uint64_t pid = blockIdx.x / blocks_per_problem;
  const BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const uint64_t n = p.n;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const uint64_t row0 =
      (static_cast<uint64_t>(blockIdx.x % blocks_per_problem) * 4 + wid) * kBfgsRowsPerWave;
  if (row0 >= n) return;
  const int lane = lane_id();
  double s[CHUNKS][2], t[CHUNKS][2], y[CHUNKS][2], g[CHUNKS][2];
  load_vec<CHUNKS, VEC>(p.s + pid * n, n, p.zero, s);
  load_vec<CHUNKS, VEC>(p.t + pid * n, n, p.zero, t);
  load_vec<CHUNKS, VEC>(p.y + pid * n, n, p.zero, y);
  load_vec<CHUNKS, VEC>(p.g + pid * n, n, p.zero, g);
  const double rho = pr->rho;
  double denom = wave_dot<CHUNKS>(y, t);  // :3143-3145
  denom = (denom * rho) + 1.0;
  const bool identity = pr->identity != 0;
  double *Hp = p.H + pid * n * n;
  double *dir = p.dir + pid * n;
#pragma unroll 2
  for (int r = 0; r < kBfgsRowsPerWave; r++) {
    const uint64_t j = row0 + r;
    if (j >= n) break;
    double h[CHUNKS][2];
    // H = I is never materialised: identity rows are synthesised
    bfgs_load_h_row<CHUNKS, VEC>(Hp + j * n, identity ? 0 : n, p.zero, h);
    const double sj = p.s[pid * n + j], tj = p.t[pid * n + j];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint64_t i = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane) + k;
        const double hij = identity ? (i == j ? 1.0 : 0.0) : h[c][k];
        const double v =
            hij - rho * (s[c][k] * tj + t[c][k] * sj + denom * s[c][k] * sj);  // :3156-3163
        h[c][k] = (i < n) ? v : 0.0;
      }
    }
    bfgs_store_h_row<CHUNKS, VEC>(Hp + j * n, n, h);
    const double dj = -wave_dot<CHUNKS>(h, g);  // :3249-3250 with the updated row
    if (lane == 0) dir[j] = dj;
  }
}

// ---- the H passes in REFERENCE ORDER at streaming speed ---------------------------------------
// Every row's dot in index order — sum_i H[j][i] v[i], products rounded, added left to right — is a
// serial chain per ROW, but the rows are independent: a LANE per row. A wave owns 64 rows and walks
// their columns in tiles of sixteen: the tile is loaded the coalesced way (eight instructions, each
// eight rows x 128 bytes), transposed through the wave's own LDS tile (row stride 17 doubles), and
// lane l then reads its row's sixteen entries and extends its chain by sixteen products; the next
// tile's loads are in flight meanwhile. ~0.1 instructions per matrix element instead of the ~2.5 of
// one-row-at-a-time (a readlane and an addition per element by the whole wave): the passes are back
// to being bound by the stream of H. The vectors of the products (and, for the update, s and t) sit
// in the block's LDS. Same additions in the same order as wave_sum_seq: the same bits.
constexpr int kBfgsSeqRows = 64;
// columns per tile: 16 = 128-byte row segments. (32 — 256-byte segments, 76 KiB of LDS per block —
// measured the same in the read-only pass and 5 % slower in the update: not the segment size.)
constexpr int kBfgsSeqColsHy = 16, kBfgsSeqColsUpdate = 16;
__host__ __device__ constexpr size_t bfgs_seq_h_lds_bytes(uint64_t n, int cols) {
  return (n + 4 * static_cast<size_t>(kBfgsSeqRows) * (cols + 1)) * sizeof(double);
}
__host__ __device__ constexpr uint32_t bfgs_seq_blocks_per_problem(uint64_t n) {
  return static_cast<uint32_t>((n + 4 * kBfgsSeqRows - 1) / (4 * kBfgsSeqRows));
}

// the tile of rows row0 .. row0+63, columns c0 .. c0+TC-1 as the wave loads it: instruction q, lane l
// -> row (128 / TC) q + l / (TC / 2), columns 2 (l % (TC / 2)), +1. VEC: n is even (16-byte aligned pairs).
template <bool VEC, int TC>
__device__ inline void bfgs_seq_load_tile(const double *__restrict__ Hp, uint64_t n, uint64_t ld, uint64_t row0,
                                          uint64_t c0, double2 (&v)[TC / 2]) {
  constexpr int LPR = TC / 2, RPI = 64 / LPR;  // lanes per row, rows per instruction
  const int lane = lane_id();
  const uint64_t col = c0 + 2 * static_cast<uint64_t>(lane % LPR);
#pragma unroll
  for (int q = 0; q < TC / 2; q++) {
    const uint64_t row = row0 + RPI * q + (lane / LPR);
    v[q] = make_double2(0.0, 0.0);
    if (row < n && col < n) {
      const double *src = Hp + row * ld + col;
      if (VEC) {
        v[q] = bfgs_stream_load(src);
      } else {
        v[q].x = __builtin_nontemporal_load(src);
        if (col + 1 < n) v[q].y = __builtin_nontemporal_load(src + 1);
      }
    }
  }
}

// t = H y (update_inverse_hessian's first loop, 3139-3142), every row's dot in index order
template <bool VEC>
__global__ __launch_bounds__(256) void bfgs_hy_seq_kernel(BfgsParams p, uint32_t blocks_per_problem) {
  extern __shared__ __align__(16) double bfgs_seq_smem[];
  const uint64_t pid = blockIdx.x / blocks_per_problem;
  const BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const uint64_t n = p.n;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6), lane = lane_id();
  const uint64_t row0 = (static_cast<uint64_t>(blockIdx.x % blocks_per_problem) * 4 + wid) * kBfgsSeqRows;
  double *t = p.t + pid * n;
  const double *y = p.y + pid * n;
  if (pr->identity) {  // H = I: t = y exactly
    if (row0 + lane < n) t[row0 + lane] = y[row0 + lane];
    return;
  }
  constexpr int TC = kBfgsSeqColsHy, ST = TC + 1, LPR = TC / 2, RPI = 64 / LPR;
  double *ys = bfgs_seq_smem, *tile = bfgs_seq_smem + n + wid * (kBfgsSeqRows * ST);
  for (uint64_t i = threadIdx.x; i < n; i += 256) ys[i] = y[i];
  __syncthreads();
  if (row0 >= n) return;
  const double *Hp = p.H + pid * n * p.ldh;
  const uint64_t ld = p.ldh;
  double acc = 0.0;
  auto consume = [&](const double2 (&v)[TC / 2], uint64_t c0) {  // the tile at columns c0 .. c0+TC-1: TC more products
#pragma unroll
    for (int q = 0; q < TC / 2; q++) {
      double *dst = tile + (RPI * q + lane / LPR) * ST + 2 * (lane % LPR);
      dst[0] = v[q].x;
      dst[1] = v[q].y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int m = n - c0 < TC ? static_cast<int>(n - c0) : TC;  // wave-uniform
    if (m == TC) {
#pragma unroll
      for (int j = 0; j < TC; j++) acc = acc + ys[c0 + j] * tile[lane * ST + j];
    } else {
      for (int j = 0; j < m; j++) acc = acc + ys[c0 + j] * tile[lane * ST + j];
    }
    __builtin_amdgcn_wave_barrier();  // the next tile's stores come after these reads
  };
  // two tiles in flight while a third is consumed
  double2 va[TC / 2], vb[TC / 2];
  bfgs_seq_load_tile<VEC, TC>(Hp, n, ld, row0, 0, va);
  bfgs_seq_load_tile<VEC, TC>(Hp, n, ld, row0, TC, vb);
  for (uint64_t c0 = 0; c0 < n; c0 += 2 * TC) {
    consume(va, c0);
    bfgs_seq_load_tile<VEC, TC>(Hp, n, ld, row0, c0 + 2 * TC, va);
    if (c0 + TC < n) consume(vb, c0 + TC);
    bfgs_seq_load_tile<VEC, TC>(Hp, n, ld, row0, c0 + 3 * TC, vb);
  }
  if (row0 + lane < n) t[row0 + lane] = acc;
}

// denom = rho y^T t + 1 (3143-3145), the sum in index order: once per problem, between the two passes
__global__ __launch_bounds__(64) void bfgs_denom_seq_kernel(BfgsParams p) {
  extern __shared__ __align__(16) double bfgs_seq_smem[];
  const uint64_t pid = blockIdx.x, n = p.n;
  BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const double *y = p.y + pid * n, *t = p.t + pid * n;
  for (uint64_t i = threadIdx.x; i < n; i += 64) bfgs_seq_smem[i] = y[i] * t[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double acc = 0.0;
#pragma unroll 8
  for (uint64_t i = 0; i < n; i++) acc = acc + bfgs_seq_smem[i];
  if (threadIdx.x == 0) pr->denom = (acc * pr->rho) + 1.0;
}

// rank-2 update (3151-3164) fused with the next direction d = -H' g (3248-3251), reference order.
// The wave's rows are fixed, so s_j and t_j of a lane's eight rows stay in registers; s_i and t_i of
// a tile's columns travel with the tile's loads; only g (the dot's other factor, read at a uniform
// address) sits in LDS.
template <bool VEC>
__global__ __launch_bounds__(256) void bfgs_update_seq_kernel(BfgsParams p, uint32_t blocks_per_problem) {
  extern __shared__ __align__(16) double bfgs_seq_smem[];
  const uint64_t pid = blockIdx.x / blocks_per_problem;
  const BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const uint64_t n = p.n;
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6), lane = lane_id();
  const uint64_t row0 = (static_cast<uint64_t>(blockIdx.x % blocks_per_problem) * 4 + wid) * kBfgsSeqRows;
  constexpr int TC = kBfgsSeqColsUpdate, ST = TC + 1, LPR = TC / 2, RPI = 64 / LPR, NQ = TC / 2;
  double *gs = bfgs_seq_smem, *tile = gs + n + wid * (kBfgsSeqRows * ST);
  const double *sp = p.s + pid * n, *tp = p.t + pid * n;
  for (uint64_t i = threadIdx.x; i < n; i += 256) gs[i] = p.g[pid * n + i];
  __syncthreads();
  if (row0 >= n) return;
  const double rho = pr->rho, denom = pr->denom;
  const bool identity = pr->identity != 0;
  double *Hp = p.H + pid * n * p.ldh;
  const uint64_t ld = p.ldh;
  double sj[NQ], tj[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const uint64_t row = row0 + RPI * q + (lane / LPR);
    sj[q] = row < n ? sp[row] : 0.0;
    tj[q] = row < n ? tp[row] : 0.0;
  }
  struct Tile {
    double2 h[NQ];
    double si0, si1, ti0, ti1;
  };
  auto load = [&](Tile &T, uint64_t c0) {
    const uint64_t col = c0 + 2 * static_cast<uint64_t>(lane % LPR);
    T.si0 = col < n ? sp[col] : 0.0;
    T.ti0 = col < n ? tp[col] : 0.0;
    T.si1 = col + 1 < n ? sp[col + 1] : 0.0;
    T.ti1 = col + 1 < n ? tp[col + 1] : 0.0;
    if (!identity) bfgs_seq_load_tile<VEC, TC>(Hp, n, ld, row0, c0, T.h);
  };
  double acc = 0.0;
  auto consume = [&](const Tile &T, uint64_t c0) {
    const uint64_t col = c0 + 2 * static_cast<uint64_t>(lane % LPR);
    const bool c_in = col < n, c1_in = col + 1 < n;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const uint64_t row = row0 + RPI * q + (lane / LPR);
      // H = I is never materialised: identity rows are synthesised
      const double h0 = identity ? (col == row ? 1.0 : 0.0) : T.h[q].x;
      const double h1 = identity ? (col + 1 == row ? 1.0 : 0.0) : T.h[q].y;
      const double w0 = h0 - rho * (T.si0 * tj[q] + T.ti0 * sj[q] + denom * T.si0 * sj[q]);  // :3156-3163
      const double w1 = h1 - rho * (T.si1 * tj[q] + T.ti1 * sj[q] + denom * T.si1 * sj[q]);
      double *dst = tile + (RPI * q + lane / LPR) * ST + 2 * (lane % LPR);
      dst[0] = w0;
      dst[1] = w1;
      if (row < n && c_in) {
        double *out = Hp + row * ld + col;
        if (VEC) {
          bfgs_stream_store(out, make_double2(w0, w1));
        } else {
          __builtin_nontemporal_store(w0, out);
          if (c1_in) __builtin_nontemporal_store(w1, out + 1);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int m = n - c0 < TC ? static_cast<int>(n - c0) : TC;  // wave-uniform
    if (m == TC) {
#pragma unroll
      for (int j = 0; j < TC; j++) acc = acc + tile[lane * ST + j] * gs[c0 + j];
    } else {
      for (int j = 0; j < m; j++) acc = acc + tile[lane * ST + j] * gs[c0 + j];
    }
    __builtin_amdgcn_wave_barrier();
  };
  Tile A, B;
  load(A, 0);
  load(B, TC);
  for (uint64_t c0 = 0; c0 < n; c0 += 2 * TC) {
    consume(A, c0);
    load(A, c0 + 2 * TC);
    if (c0 + TC < n) consume(B, c0 + TC);
    load(B, c0 + 3 * TC);
  }
  if (row0 + lane < n) p.dir[pid * n + row0 + lane] = -acc;  // :3249-3250 with the updated row
}

// ---- symmetric restatement of the rank-2 update (NLSG_BFGS_SYMMETRIC) ----------------------------
// H is symmetric in exact arithmetic; the reference's literal update is not bitwise symmetric only
// because of how its last term associates: (denom * s[i]) * s[j] (nlsolver.h:3156-3163). Restated as
//     H[j][i] -= rho * ((s[i] t[j] + t[i] s[j]) + denom * (s[i] s[j]))
// both (j, i) and (i, j) get the same bits (the two products of the first bracket commute under
// the addition, s[i] s[j] commutes), so H stays bitwise symmetric from the identity on and only its
// upper 128 x 128 blocks need to exist: 36 of 64 at n = 1024 (4.5 MiB instead of 8 per problem),
// each streamed ONCE per pass — 13.5 MiB per iteration and problem instead of 24.
//
// A stored block (I, J), I <= J, T[r][c] = H[128 I + r][128 J + c], serves two row blocks of a
// product v = H u: directly, v_I += T u_J (a lane-tree dot per row, as in the literal kernels),
// and — if I < J — transposed, v_J += T^T u_I (per column, rows summed in order by the wave that
// owns them, then the block's four waves in order). Every (row block K, column block c) pair of
// the full matrix thus yields one 128-vector partial, written to part[K][c]; v_K is their sum for
// c = 0 .. nb-1 in order (bfgs_sym_reduce_kernel). Fixed orders throughout: oracle_bfgs.c, tree = 2,
// restates them and the kernels match it bit for bit; against the literal arithmetic (and the
// reference) results differ at rounding level (f within 1e-12 on the G6 runs, tested).
constexpr int kBfgsSymB = 128;

// block (I, J), I <= J, in the packed upper triangle of nb x nb blocks
__host__ __device__ inline uint32_t bfgs_sym_block(uint32_t I, uint32_t J, uint32_t nb) {
  return I * nb - I * (I - 1) / 2 + (J - I);
}

// u[128 blk + 2 lane + k], zero past n
__device__ inline void bfgs_sym_slice(const double *u, uint64_t n, uint32_t blk, double (&v)[2]) {
  const uint64_t e = static_cast<uint64_t>(blk) * kBfgsSymB + 2 * static_cast<uint64_t>(lane_id());
  v[0] = e < n ? u[e] : 0.0;
  v[1] = e + 1 < n ? u[e + 1] : 0.0;
}

struct BfgsSymShared {
  double direct[kBfgsSymB];      // per row of the block: T[r] . u_J
  double transp[4][kBfgsSymB];   // per wave and column: sum over the wave's rows of T[r][c] u_I[r]
};

// the block's two partial vectors -> part[I][J] (direct) and part[J][I] (transposed)
__device__ inline void bfgs_sym_store_partials(const BfgsParams &p, uint64_t pid, uint32_t I, uint32_t J,
                                               const BfgsSymShared &sh) {
  const uint32_t t = threadIdx.x;
  if (t >= kBfgsSymB) return;
  double *base = p.part + pid * p.nb * p.nb * kBfgsSymB;
  base[(static_cast<uint64_t>(I) * p.nb + J) * kBfgsSymB + t] = sh.direct[t];
  if (I != J)
    base[(static_cast<uint64_t>(J) * p.nb + I) * kBfgsSymB + t] =
        ((sh.transp[0][t] + sh.transp[1][t]) + sh.transp[2][t]) + sh.transp[3][t];
}

// t = H y, one workgroup per stored block (first loop of update_inverse_hessian, 3139-3142)
constexpr int kBfgsSymFlight = 16;  // rows (KiB) a wave keeps in flight

__global__ __launch_bounds__(256) void bfgs_sym_hy_kernel(BfgsParams p) {
  __shared__ BfgsSymShared sh;
  const uint64_t pid = blockIdx.x / p.nstored;
  const BfgsProblem *pr = p.prob + pid;
  if (pr->done || pr->identity) return;  // H = I: t = y, written by the reduce kernel
  uint32_t I = 0, rem = blockIdx.x % p.nstored;
  while (rem >= p.nb - I) {
    rem -= p.nb - I;
    I++;
  }
  const uint32_t J = I + rem;
  const uint64_t n = p.n;
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const double *y = p.y + pid * n;
  double yJ[2], yI[2];
  bfgs_sym_slice(y, n, J, yJ);
  bfgs_sym_slice(y, n, I, yI);  // element r of the row block sits in lane r / 2: read per row
  const double *T = p.Hs + (pid * p.nstored + blockIdx.x % p.nstored) * (kBfgsSymB * kBfgsSymB) +
                    2 * lane;
  double tp0 = 0.0, tp1 = 0.0;
  for (int r0 = 32 * wid; r0 < 32 * wid + 32; r0 += kBfgsSymFlight) {
    double2 h[kBfgsSymFlight];
#pragma unroll
    for (int q = 0; q < kBfgsSymFlight; q++)
      h[q] = bfgs_stream_load(T + (r0 + q) * kBfgsSymB);
#pragma unroll
    for (int q = 0; q < kBfgsSymFlight; q++) {
      const double yr = lane_broadcast(yI[q & 1], (r0 + q) >> 1);
      double acc = 0.0;
      acc = acc + h[q].x * yJ[0];
      acc = acc + h[q].y * yJ[1];
      const double d = wave_sum(acc);
      if (lane == 0) sh.direct[r0 + q] = d;
      tp0 = tp0 + h[q].x * yr;
      tp1 = tp1 + h[q].y * yr;
    }
  }
  sh.transp[wid][2 * lane] = tp0;
  sh.transp[wid][2 * lane + 1] = tp1;
  __syncthreads();
  bfgs_sym_store_partials(p, pid, I, J, sh);
}

// v = sum of the column blocks' partials in order; one workgroup per problem.
// UPDATE = false: t = H y (or y while H = I), then denom = rho y^T t + 1 (3143-3145, lane-tree dot).
// UPDATE = true:  d = -H' g, the next search direction (3248-3251).
template <bool UPDATE>
__global__ __launch_bounds__(256) void bfgs_sym_reduce_kernel(BfgsParams p) {
  __shared__ double tl[1024];
  const uint64_t pid = blockIdx.x;
  BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  const uint64_t n = p.n;
  const double *base = p.part + pid * p.nb * p.nb * kBfgsSymB;
  double *out = (UPDATE ? p.dir : p.t) + pid * n;
  for (uint64_t e = threadIdx.x; e < 1024; e += 256) {
    double acc = 0.0;
    if (e < n) {
      if (!UPDATE && pr->identity) {
        acc = p.y[pid * n + e];
      } else {
        const uint64_t K = e / kBfgsSymB, r = e % kBfgsSymB;
        acc = base[(K * p.nb) * kBfgsSymB + r];
        for (uint32_t c = 1; c < p.nb; c++) acc = acc + base[(K * p.nb + c) * kBfgsSymB + r];
      }
      if (UPDATE) acc = -acc;
      out[e] = acc;
    }
    tl[e] = acc;
  }
  if (UPDATE) return;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = lane_id();
    const double *y = p.y + pid * n;
    double acc = 0.0;  // wave_dot<8>(y, t): chunks past n add exact zeros
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint64_t e = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
      const double y0 = e < n ? y[e] : 0.0, y1 = e + 1 < n ? y[e + 1] : 0.0;
      acc = acc + y0 * tl[e];
      acc = acc + y1 * tl[e + 1];
    }
    const double dot = wave_sum(acc);
    if (lane == 0) pr->denom = (dot * pr->rho) + 1.0;
  }
}

// rank-2 update of one stored block (3151-3164, restated) fused with its share of the next
// direction d = -H' g
__global__ __launch_bounds__(256) void bfgs_sym_update_kernel(BfgsParams p) {
  __shared__ BfgsSymShared sh;
  const uint64_t pid = blockIdx.x / p.nstored;
  const BfgsProblem *pr = p.prob + pid;
  if (pr->done) return;
  uint32_t I = 0, rem = blockIdx.x % p.nstored;
  while (rem >= p.nb - I) {
    rem -= p.nb - I;
    I++;
  }
  const uint32_t J = I + rem;
  const uint64_t n = p.n;
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const double *s = p.s + pid * n, *t = p.t + pid * n, *g = p.g + pid * n;
  double sJ[2], tJ[2], gJ[2], sI[2], tI[2], gI[2];
  bfgs_sym_slice(s, n, J, sJ);
  bfgs_sym_slice(t, n, J, tJ);
  bfgs_sym_slice(g, n, J, gJ);
  bfgs_sym_slice(s, n, I, sI);  // element r of the row block sits in lane r / 2: read per row
  bfgs_sym_slice(t, n, I, tI);
  bfgs_sym_slice(g, n, I, gI);
  const double rho = pr->rho, denom = pr->denom;
  const bool identity = pr->identity != 0;
  double *T = p.Hs + (pid * p.nstored + blockIdx.x % p.nstored) * (kBfgsSymB * kBfgsSymB) + 2 * lane;
  const uint64_t col = static_cast<uint64_t>(J) * kBfgsSymB + 2 * static_cast<uint64_t>(lane);
  double tp0 = 0.0, tp1 = 0.0;
  for (int r0 = 32 * wid; r0 < 32 * wid + 32; r0 += kBfgsSymFlight) {
    double2 h[kBfgsSymFlight];
    if (identity) {  // H = I is never materialised: identity blocks are synthesised (wave-uniform)
#pragma unroll
      for (int q = 0; q < kBfgsSymFlight; q++) {
        const uint64_t row = static_cast<uint64_t>(I) * kBfgsSymB + r0 + q;
        h[q].x = (row < n && row == col) ? 1.0 : 0.0;
        h[q].y = (row < n && row == col + 1) ? 1.0 : 0.0;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kBfgsSymFlight; q++)
        h[q] = bfgs_stream_load(T + (r0 + q) * kBfgsSymB);
    }
#pragma unroll
    for (int q = 0; q < kBfgsSymFlight; q++) {
      const int src = (r0 + q) >> 1;
      const double sr = lane_broadcast(sI[q & 1], src), tr = lane_broadcast(tI[q & 1], src);
      const double gr = lane_broadcast(gI[q & 1], src);
      h[q].x = h[q].x - rho * ((sJ[0] * tr + tJ[0] * sr) + denom * (sJ[0] * sr));
      h[q].y = h[q].y - rho * ((sJ[1] * tr + tJ[1] * sr) + denom * (sJ[1] * sr));
      bfgs_stream_store(T + (r0 + q) * kBfgsSymB, h[q]);
      double acc = 0.0;
      acc = acc + h[q].x * gJ[0];
      acc = acc + h[q].y * gJ[1];
      const double d = wave_sum(acc);
      if (lane == 0) sh.direct[r0 + q] = d;
      tp0 = tp0 + h[q].x * gr;
      tp1 = tp1 + h[q].y * gr;
    }
  }
  sh.transp[wid][2 * lane] = tp0;
  sh.transp[wid][2 * lane + 1] = tp1;
  __syncthreads();
  bfgs_sym_store_partials(p, pid, I, J, sh);
}

__global__ void bfgs_count_unfinished_kernel(BfgsParams p, unsigned long long *count) {
  const uint64_t pid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (pid < p.batch && !p.prob[pid].done) atomicAdd(count, 1ull);
}
__global__ void bfgs_count_identity_kernel(BfgsParams p, unsigned long long *count) {
  const uint64_t pid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (pid < p.batch && !p.prob[pid].done && p.prob[pid].identity) atomicAdd(count, 1ull);
}

}  // namespace nlsg
