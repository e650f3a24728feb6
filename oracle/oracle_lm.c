/* oracle/oracle_lm.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Levenberg-Marquardt as the reference implements it (damped Newton with an
 * always-accepted step, nlsolver.h:3465-3544) driven as an NLLS solver by
 * Gauss-Newton functors (f = sum r^2, Grad = 2 J^T r, Hess = 2 J^T J), its linear
 * algebra (math::cholesky / forwardsolve_inplace / backsolve_inplace_t / is_diagonal
 * / get_update_with_hessian, nlsolver.h:251-330) and tinyqr (givens_rotation,
 * rotate_matrix, qr_impl, qr_decomposition, back_solve, lm; tinyqr.h:86-139,
 * 253-310, 437-470).
 *
 * order = 0: sequential sums and libm exp/tanh = the arithmetic of the reference
 *            run (pinned by tests/golden/lm.json);
 * order = 1: the HIP kernel's summation order, fma chains where the kernel uses
 *            fp64 MFMA, and the deterministic exp/tanh below (bit-exact with it).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---- deterministic exp / tanh (no libm), cf. oracle_math.c ----------------- */
static double bits2d(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}
double orc_exp(double x) {
  static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                      invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                      P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.0) return INFINITY;
  if (x < -708.0) return 0.0; /* results below the normal range are flushed */
  const int k = (int)fma(invln2, x, x < 0 ? -0.5 : 0.5);
  const double hi = fma(-(double)k, ln2HI, x), lo = (double)k * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = fma(-t, fma(t, fma(t, fma(t, fma(t, P5, P4), P3), P2), P1), r);
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  return y * bits2d((uint64_t)(1023 + k) << 52); /* y * 2^k, k in [-1021, 1023] */
}
/* tanh from the rational form of the exponential: with 2|x| = k ln2 + r and fdlibm's
 * c = r - r^2 P(r^2), exp(r) = (2 + 2r - c) / (2 - c), so with s = 2^k and B = 2 - c
 *   tanh|x| = (e - 1) / (e + 1) = ((s - 1) B + 2 s r) / ((s + 1) B + 2 s r)
 * — ONE division (1 - 2 / (exp(2|x|) + 1) takes two), no cancellation for small |x| (k = 0:
 * r / (B + r)), and 1 exactly from |x| = 22 on. Same arithmetic as nlsg_math.h det_tanh. */
double orc_tanh(double x) {
  static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                      invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                      P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double ax = fabs(x);
  ax = ax < 22.0 ? ax : 22.0; /* a NaN takes the cap too: the conversion below stays defined */
  const double X = 2.0 * ax;
  const int k = (int)fma(invln2, X, 0.5);
  const double dk = (double)k;
  const double hi = fma(-dk, ln2HI, X), lo = dk * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = fma(-t, fma(t, fma(t, fma(t, fma(t, P5, P4), P3), P2), P1), r);
  const double B = 2.0 - c;
  const double s = ldexp(1.0, k), sr2 = ldexp(r, k + 1); /* 2^k and 2 s r, both exact */
  const double num = fma(s - 1.0, B, sr2), den = fma(s + 1.0, B, sr2);
  const double tt = num / den;
  return x != x ? x : copysign(tt, x); /* tanh(-0) = -0 */
}

/* ---- linear algebra of the reference LM ------------------------------------- */
/* order = 1 (the kernel's arithmetic): every accumulate-multiply of the factorisation and of the
 * two substitutions is ONE fused multiply-add, sums in the same order as below */
static void cholesky_order(double *A, size_t n, int order) { /* nlsolver.h:251-269 */
  for (size_t i = 0; i < n; ++i) {
    for (size_t j = 0; j < i; ++j) {
      double sum = 0;
      for (size_t k = 0; k < j; ++k)
        sum = order ? fma(A[i * n + k], A[j * n + k], sum) : sum + A[i * n + k] * A[j * n + k];
      A[i * n + j] = (1.0 / A[j * n + j] * (A[i * n + j] - sum));
    }
    double sum = 0;
    for (size_t k = 0; k < i; ++k)
      sum = order ? fma(A[i * n + k], A[i * n + k], sum) : sum + A[i * n + k] * A[i * n + k];
    A[i * n + i] = sqrt(A[i * n + i] - sum);
  }
}
void orc_cholesky(double *A, size_t n) { cholesky_order(A, n, 0); }
static void forwardsolve(double *update, const double *L, const double *b, size_t n, int order) { /* :282-294 */
  memset(update, 0, n * sizeof(double));
  for (size_t i = 0; i < n; ++i) {
    double sum = 0.0;
    for (size_t j = 0; j < i; ++j)
      sum = order ? fma(L[i * n + j], update[j], sum) : sum + L[i * n + j] * update[j];
    update[i] = (b[i] - sum) / L[i + i * n];
  }
}
/* order = 1 adds the terms of each inner sum from j = n-1 down to i+1: the order in which
 * the solution components become available, which lets the kernel sweep columns in parallel */
static void backsolve_t(const double *U, double *b, size_t n, int order) { /* :270-281 */
  for (int i = (int)n - 1; i >= 0; --i) {
    double sum = 0.0;
    if (order == 0)
      for (size_t j = (size_t)i + 1; j < n; ++j) sum += U[j * n + (size_t)i] * b[j];
    else
      for (size_t j = n; j-- > (size_t)i + 1;) sum = fma(U[j * n + (size_t)i], b[j], sum);
    b[i] = (b[i] - sum) / U[(size_t)i * n + (size_t)i];
  }
}
static int is_diagonal(const double *A, size_t n) { /* :295-307 (positive off-diagonals only) */
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j)
      if (i != j && A[i * n + j] > 2.220446049250313e-16 * 1e12) return 0;
  return 1;
}
void orc_update_with_hessian_order(double *update, double *hess, const double *grad, size_t n,
                                   int order) {
  if (is_diagonal(hess, n)) { /* :310-330 */
    for (size_t i = 0; i < n; i++) update[i] = grad[i] / hess[i * n + i];
    return;
  }
  cholesky_order(hess, n, order);
  forwardsolve(update, hess, grad, n, order);
  backsolve_t(hess, update, n, order);
}
void orc_update_with_hessian(double *update, double *hess, const double *grad, size_t n) {
  orc_update_with_hessian_order(update, hess, grad, n, 0);
}

/* ---- tinyqr ------------------------------------------------------------------ */
/* order = 1: r*r instead of pow(r, 2) (the kernel has no libm) */
static void givens(double a, double b, double *c, double *s, int order) { /* tinyqr.h:86-97 */
  if (fabs(b) > fabs(a)) {
    const double r = a / b;
    const double sv = 1.0 / sqrt((order ? r * r : pow(r, 2)) + 1.0);
    *c = sv * r;
    *s = sv;
    return;
  }
  const double r = b / a;
  const double cv = 1.0 / sqrt((order ? r * r : pow(r, 2)) + 1.0);
  *c = cv;
  *s = cv * r;
}
static void rotate(double *lower, double *upper, double c, double s, size_t p) { /* :126-139 */
  for (; p > 0; --p) {
    const double t1 = *lower, t2 = *upper;
    *lower = c * t1 + s * t2;
    *upper = -s * t1 + c * t2;
    ++lower;
    ++upper;
  }
}
/* qr_decomposition (tinyqr.h:291-310): X column-major n x p. Q out: p rows of length n
 * (Q[i*n + j]); R out: p x p with R[j*p + i] = R(i,j). Qfull is n*n scratch, Rw n*p. */
static void qr_decomposition_order(const double *X, size_t n, size_t p, double tol, double *Q,
                                   double *R, int order);
void orc_qr_decomposition(const double *X, size_t n, size_t p, double tol, double *Q, double *R) {
  qr_decomposition_order(X, n, p, tol, Q, R, 0);
}
static void qr_decomposition_order(const double *X, size_t n, size_t p, double tol, double *Q,
                                   double *R, int order) {
  double *Qf = (double *)calloc(n * n, sizeof(double));
  double *Rw = (double *)calloc(n * p, sizeof(double));
  for (size_t i = 0; i < n; i++) Qf[i * n + i] = 1.0;
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < p; j++) Rw[i * p + j] = X[j * n + i];
  for (size_t j = 0; j < p; j++) /* qr_impl, :253-283 */
    for (size_t i = n - 1; i > j; --i) {
      double c, s;
      givens(Rw[(i - 1) * p + j], Rw[i * p + j], &c, &s, order);
      rotate(Rw + (i - 1) * p, Rw + i * p, c, s, p);
      rotate(Qf + (i - 1) * n, Qf + i * n, c, s, n);
    }
  for (size_t e = 0; e < n * p; e++) Rw[e] = fabs(Rw[e]) < tol ? 0.0 : Rw[e]; /* cleanup */
  for (size_t i = 0; i < p; i++) /* transpose_square on the leading p x p */
    for (size_t j = i + 1; j < p; j++) {
      const double t = Rw[j * p + i];
      Rw[j * p + i] = Rw[i * p + j];
      Rw[i * p + j] = t;
    }
  memcpy(Q, Qf, n * p * sizeof(double));
  memcpy(R, Rw, p * p * sizeof(double));
  free(Qf);
  free(Rw);
}
/* lm(X, y) = back_solve(qr(X)) (tinyqr.h:437-470), tol = 1e-12 as lm() passes it */
void orc_tinyqr_lm(const double *X, const double *y, size_t n, size_t p, double *beta) {
  orc_tinyqr_lm_order(X, y, n, p, beta, 0);
}
/* order = 1 — the device's restatement (nlsg_lm_kernels.h lm_solve_qr): Givens without pow();
 * Q is never formed: lm() only uses it through Q^T y, so y is rotated along with R (same vector
 * mathematically, other roundings); an element update is one rounded product and one fma,
 * lower' = fma(c, lower, s * upper), upper' = fma(c, upper, (-s) * lower); the back-substitution
 * sums are taken from j = p-1 down to i+1 (the order in which the kernel's column sweep produces
 * them). Same rotations in the same per-element order as qr_impl. */
static void tinyqr_lm_corotated(const double *X, const double *y, size_t n, size_t p, double *beta,
                                double tol) {
  double *Rw = (double *)calloc(n * p, sizeof(double));
  double *w = (double *)malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++) {
    w[i] = y[i];
    for (size_t j = 0; j < p; j++) Rw[i * p + j] = X[j * n + i];
  }
  for (size_t j = 0; j < p; j++)
    for (size_t i = n - 1; i > j; --i) {
      double c, s;
      givens(Rw[(i - 1) * p + j], Rw[i * p + j], &c, &s, 1);
      double *lower = Rw + (i - 1) * p, *upper = Rw + i * p;
      for (size_t k = 0; k < p; k++) {
        const double t1 = lower[k], t2 = upper[k];
        lower[k] = fma(c, t1, s * t2);
        upper[k] = fma(c, t2, (-s) * t1);
      }
      const double t1 = w[i - 1], t2 = w[i];
      w[i - 1] = fma(c, t1, s * t2);
      w[i] = fma(c, t2, (-s) * t1);
    }
  for (size_t e = 0; e < n * p; e++) Rw[e] = fabs(Rw[e]) < tol ? 0.0 : Rw[e]; /* cleanup */
  for (size_t i = 0; i < p; i++) beta[i] = 0.0;
  for (size_t i = p; i-- > 0;) {
    double temp = 0.0;
    for (size_t j = p; j-- > i + 1;) temp += Rw[i * p + j] * beta[j];
    beta[i] = (w[i] - temp) / Rw[i * p + i];
  }
  free(Rw);
  free(w);
}
void orc_tinyqr_lm_order(const double *X, const double *y, size_t n, size_t p, double *beta,
                         int order) {
  orc_tinyqr_lm_tol(X, y, n, p, beta, order, 1e-12); /* lm()'s default third argument, :464 */
}
/* tol: lm()'s third argument, handed to qr_decomposition's cleanup (tinyqr.h:278-282, 467) */
void orc_tinyqr_lm_tol(const double *X, const double *y, size_t n, size_t p, double *beta, int order,
                       double tol) {
  if (order) {
    tinyqr_lm_corotated(X, y, n, p, beta, tol);
    return;
  }
  double *Q = (double *)malloc(n * p * sizeof(double)), *R = (double *)malloc(p * p * sizeof(double));
  qr_decomposition_order(X, n, p, tol, Q, R, order);
  for (size_t i = 0; i < p; i++) beta[i] = 0.0;
  for (size_t i = p; i-- > 0;) {
    double temp = 0.0;
    for (size_t j = i + 1; j < p; ++j) temp += R[j * p + i] * beta[j];
    double ytmp = 0;
    for (size_t j = 0; j < n; ++j) ytmp += Q[i * n + j] * y[j];
    beta[i] = (ytmp - temp) / R[i * p + i];
  }
  free(Q);
  free(R);
}

/* ---- synthetic tanh-regression problems ------------------- */
void orc_lm_make_tanh_problem(uint64_t seed, uint64_t problem, size_t m, size_t n, double *A,
                              double *y, double *theta0) {
  const uint64_t kp = orc_ctr_key(seed, problem);
  const uint64_t kA = orc_ctr_key(kp, 0), kT = orc_ctr_key(kp, 1), k0 = orc_ctr_key(kp, 2);
  const double scale = 1.0 / sqrt((double)n);
  double *star = (double *)malloc(n * sizeof(double));
  for (size_t e = 0; e < m * n; e++) A[e] = (2 * orc_u01(orc_ctr_key(kA, e)) - 1) * scale;
  for (size_t j = 0; j < n; j++) star[j] = 2 * orc_u01(orc_ctr_key(kT, j)) - 1;
  for (size_t i = 0; i < m; i++) {
    double z = 0.0;
    for (size_t j = 0; j < n; j++) z += A[i * n + j] * star[j];
    y[i] = tanh(z);
  }
  for (size_t j = 0; j < n; j++) theta0[j] = 0.5 * star[j] + 0.1 * (2 * orc_u01(orc_ctr_key(k0, j)) - 1);
  free(star);
}

/* ---- residual models ----------------------------------------------------------- */
static void residual_jacobian(const orc_nlls *q, const double *x, double *r, double *J, int order) {
  const size_t m = q->m, n = q->n;
  for (size_t i = 0; i < m; i++) {
    if (q->kind == 0) { /* exp model: r = y - p0 exp(p1 t) */
      const double e = exp(x[1] * q->t[i]);
      r[i] = q->y[i] - x[0] * e;
      J[i * n + 0] = -e;
      J[i * n + 1] = -(x[0] * q->t[i] * e);
    } else {
      double z = 0.0, th;
      if (order == 0) {
        for (size_t j = 0; j < n; j++) z += q->A[i * n + j] * x[j];
        th = tanh(z);
      } else { /* kernel order: 32 column pairs per block of 64 columns (columns >= n are 0; the
                * pairs of later blocks continue each lane's fma chain), xor butterfly 16..1 */
        double lane[32], tmp[32];
        for (size_t l = 0; l < 32; l++) {
          for (size_t c0 = 0; c0 == 0 || c0 < n; c0 += 64) {
            const size_t e0 = c0 + 2 * l, e1 = e0 + 1;
            const double a0 = e0 < n ? q->A[i * n + e0] : 0.0, b0 = e0 < n ? x[e0] : 0.0;
            const double a1 = e1 < n ? q->A[i * n + e1] : 0.0, b1 = e1 < n ? x[e1] : 0.0;
            lane[l] = c0 == 0 ? fma(a1, b1, a0 * b0) : fma(a1, b1, fma(a0, b0, lane[l]));
          }
        }
        for (int off = 16; off >= 1; off >>= 1) {
          for (int l = 0; l < 32; l++) tmp[l] = lane[l] + lane[l ^ off];
          memcpy(lane, tmp, sizeof lane);
        }
        z = lane[0];
        th = orc_tanh(z);
      }
      r[i] = q->y[i] - th;
      const double w = 1 - th * th;
      for (size_t j = 0; j < n; j++) J[i * n + j] = -(w * q->A[i * n + j]);
    }
  }
}

/* f = sum r^2, g = 2 J^T r, H = 2 J^T J in the requested summation order */
static double gn_all(const orc_nlls *q, const double *x, double *g, double *H, double *r, double *J,
                     int order) {
  const size_t m = q->m, n = q->n;
  residual_jacobian(q, x, r, J, order);
  double f;
  if (order == 0) {
    f = 0.0;
    for (size_t i = 0; i < m; i++) f += r[i] * r[i];
    for (size_t j = 0; j < n; j++) {
      double acc = 0.0;
      for (size_t i = 0; i < m; i++) acc += J[i * n + j] * r[i];
      g[j] = 2 * acc;
    }
    for (size_t j = 0; j < n; j++)
      for (size_t k = 0; k < n; k++) {
        double acc = 0.0;
        for (size_t i = 0; i < m; i++) acc += J[i * n + j] * J[i * n + k];
        H[j * n + k] = 2 * acc;
      }
  } else {
    double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < m; i++) { /* 8 accumulators: idx = 2*((i%64)/16) + i%2 */
      const size_t rb = i % 64;
      part[2 * (rb / 16) + rb % 2] = fma(r[i], r[i], part[2 * (rb / 16) + rb % 2]);
    }
    f = 0.0;
    for (int k = 0; k < 8; k++) f += part[k];
    for (size_t j = 0; j < n; j++) { /* 4 accumulators by i mod 4, then ((a0+a1)+a2)+a3 */
      double a[4] = {0, 0, 0, 0};
      for (size_t i = 0; i < m; i++) a[i % 4] = fma(J[i * n + j], r[i], a[i % 4]);
      g[j] = 2 * (((a[0] + a[1]) + a[2]) + a[3]);
    }
    for (size_t j = 0; j < n; j++) /* fp64 MFMA: one fma chain over the rows, in order */
      for (size_t k = 0; k < n; k++) {
        double acc = 0.0;
        for (size_t i = 0; i < m; i++) acc = fma(J[i * n + j], J[i * n + k], acc);
        H[j * n + k] = 2 * acc;
      }
  }
  return f;
}

/* LevenbergMarquardt::solve<true> (nlsolver.h:3465-3544). solver: 0 = the reference's
 * get_update_with_hessian (Cholesky / diagonal shortcut); 1 = tinyqr::lm on the damped
 * matrix (the composition BASELINE config 4 names). lambda is in/out (the reference
 * keeps it as a mutable member, :3436/3541). */
orc_status orc_lm_solve(const orc_nlls *q, double *x, double *lambda, double up, double down,
                        size_t max_iter, double f_delta, int solver, int order, double *f_log,
                        size_t f_cap) {
  const size_t m = q->m, n = q->n;
  double *g = (double *)malloc(n * sizeof(double)), *H = (double *)malloc(n * n * sizeof(double));
  double *r = (double *)malloc(m * sizeof(double)), *J = (double *)malloc(m * n * sizeof(double));
  double *upd = (double *)malloc(n * sizeof(double)), *Xc = (double *)malloc(n * n * sizeof(double));
  size_t iter = 0, fc = 0, gc = 0, hc = 0, nlog = 0;
  double cur = gn_all(q, x, g, H, r, J, order); /* g_lam, h_lam, f_lam at x0 (:3513-3516) */
  gc++;
  hc++;
  fc++;
  if (f_log && nlog < f_cap) f_log[nlog] = cur;
  nlog++;
  double prev = 0.0;
  for (;;) {
    const double delta = fabs(prev - cur);
    if (iter >= max_iter || delta < f_delta || isnan(prev)) break; /* :3520-3527 */
    for (size_t i = 0; i < n; i++) H[i * n + i] += *lambda;          /* :3529-3531 */
    if (solver == 0) {
      orc_update_with_hessian_order(upd, H, g, n, order);
    } else {
      for (size_t i = 0; i < n; i++) /* column-major copy of the row-major damped matrix */
        for (size_t j = 0; j < n; j++) Xc[j * n + i] = H[i * n + j];
      orc_tinyqr_lm_order(Xc, g, n, n, upd, order);
    }
    for (size_t i = 0; i < n; i++) x[i] -= upd[i]; /* :3534 (always accepted) */
    prev = cur;
    cur = gn_all(q, x, g, H, r, J, order);
    fc++;
    gc++;
    hc++;
    if (f_log && nlog < f_cap) f_log[nlog] = cur;
    nlog++;
    iter++;
    *lambda = cur < prev ? *lambda / down : *lambda * up; /* :3541-3542 */
  }
  orc_status st = {cur, iter, fc, gc, hc};
  free(g);
  free(H);
  free(r);
  free(J);
  free(upd);
  free(Xc);
  return st;
}

/* ---- the reference's DEFAULT LevenbergMarquardt<Callable>(f): Grad = fin_diff, Hess = fin_diff_h,
 * i.e. finite_difference_gradient<.,.,1> (nlsolver.h:1385-1413) and
 * finite_difference_hessian<.,.,1> (1413-1517), every probe counted as a function call
 * (3479-3510), on a built-in objective (oracle_objective.c). order = 0: sequential sums (the
 * reference arithmetic); order = 1: the kernel's lane tree and back-substitution order. */
typedef struct {
  int obj, order;
  size_t n, fcalls;
  double *f_log;
  size_t f_cap, nlog;
} fd_ctx;
static double fd_f(fd_ctx *c, const double *x) {
  const double v = c->order ? orc_objective_tree(c->obj, x, c->n) : orc_objective_seq(c->obj, x, c->n);
  c->fcalls++;
  if (c->f_log && c->nlog < c->f_cap) c->f_log[c->nlog] = v;
  c->nlog++;
  return v;
}
static void fd_gradient(fd_ctx *c, double *x, double *g) { /* accuracy 1, :1385-1413 */
  const double eps = 2.220446049250313e-16 * 10e7;
  static const double coeff[4] = {1, -8, 8, -1}, coeff2[4] = {-2, -1, 1, 2};
  const double dd_val = 12 * eps;
  for (size_t d = 0; d < c->n; d++) {
    double acc = 0.0;
    for (int s = 0; s < 4; s++) {
      const double tmp = x[d];
      x[d] += coeff2[s] * eps;
      acc += coeff[s] * fd_f(c, x);
      x[d] = tmp;
    }
    g[d] = acc / dd_val;
  }
}
static void fd_hessian(fd_ctx *c, double *x, double *hess) { /* accuracy 1, :1446-1515, literal */
  const double eps = pow(2.220446049250313e-16, 1.0 / 4.0);
  const double denom = (600.0 * eps * eps), two_eps = 2 * eps, three_eps = 3 * eps, four_eps = 4 * eps;
  const size_t p = c->n;
  for (size_t i = 0; i < p; i++) {
    const double temp_i = x[i];
    for (size_t j = 0; j < p; j++) {
      double result = 0.0, temp = 0.0;
      const double temp_j = x[j];
      x[i] += eps;      x[j] -= two_eps;   temp += fd_f(c, x);
      x[i] += eps;      x[j] += eps;       temp += fd_f(c, x);
      x[i] -= four_eps; x[j] += two_eps;   temp += fd_f(c, x);
      x[i] += eps;      x[j] += eps;       temp += fd_f(c, x);
      result -= 63 * temp;
      temp = 0.0;
      x[j] -= four_eps;                    temp += fd_f(c, x);
      x[i] -= eps;      x[j] += eps;       temp += fd_f(c, x);
      x[i] += three_eps; x[j] += three_eps; temp += fd_f(c, x);
      x[i] += eps;      x[j] -= eps;       temp += fd_f(c, x);
      result += 63 * temp;
      temp = 0.0;
      x[j] -= three_eps;                   temp += fd_f(c, x);
      x[i] -= four_eps; x[j] += four_eps;  temp += fd_f(c, x);
      x[j] -= four_eps;                    temp -= fd_f(c, x);
      x[i] += four_eps; x[j] += four_eps;  temp -= fd_f(c, x);
      result += 44 * temp;
      temp = 0.0;
      x[i] -= three_eps; x[j] -= three_eps; temp += fd_f(c, x);
      x[i] += two_eps;  x[j] += two_eps;   temp += fd_f(c, x);
      x[j] -= two_eps;                     temp -= fd_f(c, x);
      x[i] -= two_eps;  x[j] += two_eps;   temp -= fd_f(c, x);
      result += 74 * temp;
      x[i] = temp_i;
      x[j] = temp_j;
      hess[i * p + j] = result / denom;
    }
  }
}
orc_status orc_lm_fd(int obj, double *x, size_t n, double *lambda, double up, double down,
                     size_t max_iter, double f_delta, int order, double *f_log, size_t f_cap) {
  double *g = (double *)malloc(n * sizeof(double)), *H = (double *)malloc(n * n * sizeof(double));
  double *upd = (double *)malloc(n * sizeof(double));
  fd_ctx c = {obj, order, n, 0, f_log, f_cap, 0};
  size_t iter = 0, gc = 0, hc = 0;
  fd_gradient(&c, x, g); /* g_lam, h_lam, f_lam at x0 (:3513-3516) */
  gc++;
  fd_hessian(&c, x, H);
  hc++;
  double prev = 0.0, cur = fd_f(&c, x);
  for (;;) {
    const double delta = fabs(prev - cur);
    if (iter >= max_iter || delta < f_delta || isnan(prev)) break; /* :3520-3527 */
    for (size_t i = 0; i < n; i++) H[i * n + i] += *lambda;          /* :3529-3531 */
    orc_update_with_hessian_order(upd, H, g, n, order);
    for (size_t i = 0; i < n; i++) x[i] -= upd[i]; /* :3534 */
    prev = cur;
    cur = fd_f(&c, x);
    iter++;
    fd_gradient(&c, x, g);
    gc++;
    fd_hessian(&c, x, H);
    hc++;
    *lambda = cur < prev ? *lambda / down : *lambda * up; /* :3541-3542 */
  }
  orc_status st = {cur, iter, c.fcalls, gc, hc};
  free(g);
  free(H);
  free(upd);
  return st;
}
