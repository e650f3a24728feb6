/* oracle/oracle_bfgs.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * BFGS with the More-Thuente line search, restated from nlsolver.h:
 *   cstep 1527-1671, cvsrch 1673-1793, more_thuente_search 1880-1891,
 *   update_inverse_hessian 3130-3168, BFGS::solve 3196-3285, math::dot/norm 58-99.
 * One implementation, two summation orders (`tree`):
 *   tree = 0  sequential left-to-right sums = the reference arithmetic (pinned by
 *             tests/golden/bfgs.json);
 *   tree = 1  the fixed lane tree the HIP kernel uses (element e -> lane (e%128)/2,
 *             in-lane sequential, 64-lane xor butterfly). The kernel matches it
 *             bit for bit.
 *   tree = 2  the symmetric restatement (NLSG_BFGS_SYMMETRIC, nlsg_bfgs_kernels.h): vector
 *             reductions as tree = 1; the update's last term as denom * (s[i] s[j]), which
 *             keeps H bitwise symmetric; the products H y and H g summed the way the kernels
 *             stream the upper 128 x 128 blocks (sym_matvec below). Differs from tree = 0 / 1
 *             at rounding level only.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---- reductions ---------------------------------------------------------- */
static double reduce_terms(const double *t, size_t n, int tree) {
  if (!tree) {
    double s = 0;
    for (size_t i = 0; i < n; i++) s += t[i];
    return s;
  }
  double lane[64], tmp[64];
  memset(lane, 0, sizeof lane);
  for (size_t e = 0; e < n; e++) lane[(e % 128) / 2] += t[e];
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
    memcpy(lane, tmp, sizeof lane);
  }
  return lane[0];
}

typedef struct {
  size_t n;
  int tree;
  double *scratch; /* n */
} ctx_t;

static double dot_(const ctx_t *c, const double *a, const double *b) { /* math::dot :58-67 */
  for (size_t i = 0; i < c->n; i++) c->scratch[i] = a[i] * b[i];
  return reduce_terms(c->scratch, c->n, c->tree);
}

/* ---- the G6 quadratic: f = 1/2 sum d x^2 + 1/2 c (sum x)^2 - sum b x ---------- */
static double quad_f_raw(const ctx_t *c, const orc_quad *q, const double *x) {
  for (size_t i = 0; i < c->n; i++) c->scratch[i] = q->d[i] * x[i] * x[i];
  const double qq = reduce_terms(c->scratch, c->n, c->tree);
  const double sx = reduce_terms(x, c->n, c->tree);
  for (size_t i = 0; i < c->n; i++) c->scratch[i] = q->b[i] * x[i];
  const double lin = reduce_terms(c->scratch, c->n, c->tree);
  return 0.5 * qq + 0.5 * q->c * (sx * sx) - lin;
}
static void quad_g_raw(const ctx_t *c, const orc_quad *q, const double *x, double *g) {
  const double sx = reduce_terms(x, c->n, c->tree);
  for (size_t i = 0; i < c->n; i++) g[i] = q->d[i] * x[i] + q->c * sx - q->b[i];
}

/* ---- what BFGS minimises: the quadratic with its analytic gradient functor, or a built-in
 * objective (oracle_objective.c) with the reference's DEFAULT gradient, fin_diff =
 * finite_difference_gradient<Callable, scalar_t, 1> (nlsolver.h:1385-1413, 2849-2855). The
 * probes go through the counting wrapper f_lam (3218-3224), so they count as function calls. */
typedef struct {
  const orc_quad *q; /* NULL: built-in objective `obj` + finite differences */
  int obj;
} model_t;

static void log_f(orc_bfgs_counters *cnt, double v) {
  cnt->f_calls++;
  if (cnt->f_log && cnt->f_count < cnt->f_cap) cnt->f_log[cnt->f_count] = v;
  cnt->f_count++;
}
static double model_f(const ctx_t *c, const model_t *m, const double *x, orc_bfgs_counters *cnt) {
  const double v = m->q ? quad_f_raw(c, m->q, x)
                        : (c->tree ? orc_objective_tree(m->obj, x, c->n)
                                   : orc_objective_seq(m->obj, x, c->n));
  log_f(cnt, v);
  return v;
}
static void model_g(const ctx_t *c, const model_t *m, double *x, double *g, orc_bfgs_counters *cnt) {
  cnt->g_calls++;
  if (m->q) {
    quad_g_raw(c, m->q, x, g);
    return;
  }
  /* accuracy = 1: coeff {1, -8, 8, -1}, coeff2 {-2, -1, 1, 2}, dd = 12 (:1390-1399) */
  const double eps = 2.220446049250313e-16 * 10e7;
  static const double coeff[4] = {1, -8, 8, -1}, coeff2[4] = {-2, -1, 1, 2};
  const double dd_val = 12 * eps;
  for (size_t d = 0; d < c->n; d++) {
    double acc = 0.0; /* std::fill(grad, 0) :1402 */
    for (int s = 0; s < 4; s++) {
      const double tmp = x[d];
      x[d] += coeff2[s] * eps;
      acc += coeff[s] * model_f(c, m, x, cnt);
      x[d] = tmp;
    }
    g[d] = acc / dd_val;
  }
}

/* ---- cstep, nlsolver.h:1527-1671 ------------------------------------------ */
static double max_abs3(double x, double y, double z) { /* :1520-1523 */
  return fmax(fabs(x), fmax(fabs(y), fabs(z)));
}
static double dmin(double a, double b) { return b < a ? b : a; }   /* std::min */
static double dmax(double a, double b) { return a < b ? b : a; }   /* std::max */
static double dclamp(double v, double lo, double hi) {             /* std::clamp */
  return v < lo ? lo : (hi < v ? hi : v);
}

static int cstep(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy,
                 double *stp, double fp, double dp, int *brackt, double stpmin, double stpmax,
                 int *info) {
  *info = 0;
  int bound;
  if ((*brackt & ((*stp <= dmin(*stx, *sty)) || (*stp >= dmax(*stx, *sty)))) ||
      (*dx * (*stp - *stx) >= 0.0) || (stpmax < stpmin))
    return -1;
  const double sgnd = dp * (*dx / fabs(*dx));
  double stpf = 0, stpc, stpq;
  if (fp > *fx) { /* case 1 */
    *info = 1;
    bound = 1;
    const double theta = 3. * (*fx - fp) / (*stp - *stx) + *dx + dp;
    const double s = max_abs3(theta, *dx, dp);
    double gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    if (*stp < *stx) gamma = -gamma;
    const double p = (gamma - *dx) + theta;
    const double q = ((gamma - *dx) + gamma) + dp;
    const double r = p / q;
    stpc = *stx + r * (*stp - *stx);
    stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.) * (*stp - *stx);
    if (fabs(stpc - *stx) < fabs(stpq - *stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2;
    *brackt = 1;
  } else if (sgnd < 0.0) { /* case 2 */
    *info = 2;
    bound = 0;
    const double theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    const double s = max_abs3(theta, *dx, dp);
    double gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    if (*stp > *stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + *dx;
    const double r = p / q;
    stpc = *stp + r * (*stx - *stp);
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (fabs(stpc - *stp) > fabs(stpq - *stp))
      stpf = stpc;
    else
      stpf = stpq;
    *brackt = 1;
  } else if (fabs(dp) < fabs(*dx)) { /* case 3 */
    *info = 3;
    bound = 1;
    const double theta = 3 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    const double s = max_abs3(theta, *dx, dp);
    double gamma = s * sqrt(dmax(0., (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
    if (*stp > *stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = (gamma + (*dx - dp)) + gamma;
    const double r = p / q;
    if ((r < 0.0) & (gamma != 0.0))
      stpc = *stp + r * (*stx - *stp);
    else if (*stp > *stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (*brackt) {
      stpf = (fabs(*stp - stpc) < fabs(*stp - stpq)) ? stpc : stpq;
    } else {
      stpf = (fabs(*stp - stpc) > fabs(*stp - stpq)) ? stpc : stpq;
    }
  } else { /* case 4 */
    *info = 4;
    bound = 0;
    if (*brackt) {
      const double theta = 3 * (fp - *fy) / (*sty - *stp) + *dy + dp;
      const double s = max_abs3(theta, *dy, dp);
      double gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
      if (*stp > *sty) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + *dy;
      const double r = p / q;
      stpc = *stp + r * (*sty - *stp);
      stpf = stpc;
    } else if (*stp > *stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > *fx) { /* :1644-1658 */
    *sty = *stp;
    *fy = fp;
    *dy = dp;
  } else {
    if (sgnd < 0.0) {
      *sty = *stx;
      *fy = *fx;
      *dy = *dx;
    }
    *stx = *stp;
    *fx = fp;
    *dx = dp;
  }
  stpf = dclamp(stpf, stpmin, stpmax);
  *stp = stpf;
  if (*brackt & bound) { /* :1663-1670 */
    if (*sty > *stx)
      *stp = dmin(*stx + 0.66 * (*sty - *stx), *stp);
    else
      *stp = dmax(*stx + 0.66 * (*sty - *stx), *stp);
  }
  return 0;
}

/* ---- cvsrch, nlsolver.h:1673-1793; returns the final step through *stp. The
 * gradient vector is overwritten with the gradient at the last trial point. */
static void cvsrch(const ctx_t *c, const model_t *q, const double *x, double f0, double *gradient,
                   double *stp, const double *dir, double *tmp, orc_bfgs_counters *cnt) {
  int info = 0, infoc = 1;
  const double xtol = 1e-15, ftol = 1e-4, gtol = 1e-2, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4;
  const int maxfev = 20;
  int nfev = 0;
  const double dginit = dot_(c, gradient, dir);
  if (dginit >= 0.0) return; /* :1692-1694 */
  int brackt = 0, stage1 = 1;
  const double finit = f0, dgtest = ftol * dginit;
  double width = stpmax - stpmin, width1 = 2 * width;
  double stx = 0.0, fx = finit, dgx = dginit, sty = 0.0, fy = finit, dgy = dginit;
  double stmin, stmax;
  for (;;) {
    if (brackt) { /* :1716-1722 */
      stmin = dmin(stx, sty);
      stmax = dmax(stx, sty);
    } else {
      stmin = stx;
      stmax = *stp + xtrapf * (*stp - stx);
    }
    *stp = dclamp(*stp, stpmin, stpmax);
    if ((brackt && ((*stp <= stmin) || (*stp >= stmax))) || (nfev >= maxfev - 1) ||
        (infoc == 0) || (brackt && ((stmax - stmin) <= (xtol * stmax))))
      *stp = stx; /* :1728-1734 */
    for (size_t i = 0; i < c->n; i++) tmp[i] = x[i] + *stp * dir[i]; /* :1737 */
    const double fcur = model_f(c, q, tmp, cnt);
    model_g(c, q, tmp, gradient, cnt);
    nfev++;
    const double dg = dot_(c, gradient, dir);
    const double ftest1 = finit + *stp * dgtest;
    if ((brackt & ((*stp <= stmin) | (*stp >= stmax))) | (infoc == 0)) info = 6;
    if ((*stp == stpmax) & (fcur <= ftest1) & (dg <= dgtest)) info = 5;
    if ((*stp == stpmin) & ((fcur > ftest1) | (dg >= dgtest))) info = 4;
    if (nfev >= maxfev) info = 3;
    if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
    if ((fcur <= ftest1) & (fabs(dg) <= gtol * (-dginit))) info = 1;
    if (info != 0) return;
    if (stage1 & (fcur <= ftest1) & (dg >= dmin(ftol, gtol) * dginit)) stage1 = 0;
    if (stage1 & (fcur <= fx) & (fcur > ftest1)) { /* :1762-1778 */
      const double fm = fcur - *stp * dgtest;
      double fxm = fx - stx * dgtest, fym = fy - sty * dgtest;
      const double dgm = dg - dgtest;
      double dgxm = dgx - dgtest, dgym = dgy - dgtest;
      cstep(&stx, &fxm, &dgxm, &sty, &fym, &dgym, stp, fm, dgm, &brackt, stmin, stmax, &infoc);
      fx = fxm + stx * dgtest;
      fy = fym + sty * dgtest;
      dgx = dgxm + dgtest;
      dgy = dgym + dgtest;
    } else {
      cstep(&stx, &fx, &dgx, &sty, &fy, &dgy, stp, fcur, dg, &brackt, stmin, stmax, &infoc);
    }
    if (brackt) { /* :1784-1790 */
      if (fabs(sty - stx) >= 0.66 * width1) *stp = stx + 0.5 * (sty - stx);
      width1 = width;
      width = fabs(sty - stx);
    }
  }
}

/* v = H u as the symmetric kernels sum it (bfgs_sym_hy_kernel / bfgs_sym_update_kernel +
 * bfgs_sym_reduce_kernel): the matrix in 128 x 128 blocks, row block K's value = the partials of
 * its column blocks c = 0 .. nb-1 added in order. c >= K: the stored block (K, c) used directly —
 * a lane-tree dot of the row with u's slice. c < K: the stored block (c, K) used transposed — per
 * column, the products of its 128 rows summed in order inside each quarter (a wave's 32 rows),
 * then the four quarters in order. H must be bitwise symmetric; entries past n are zeros. */
#define SYMB 128
static void sym_matvec(const double *H, const double *u, double *out, size_t n) {
  const size_t nb = (n + SYMB - 1) / SYMB;
  for (size_t K = 0; K < nb; K++)
    for (size_t r = 0; r < SYMB && K * SYMB + r < n; r++) {
      const size_t row = K * SYMB + r;
      double total = 0.0;
      for (size_t c = 0; c < nb; c++) {
        double partial;
        if (c >= K) {
          double lane[64], tmp[64];
          for (int l = 0; l < 64; l++) {
            double acc = 0.0;
            for (int k = 0; k < 2; k++) {
              const size_t col = c * SYMB + 2 * (size_t)l + k;
              const double h = col < n ? H[row * n + col] : 0.0, uv = col < n ? u[col] : 0.0;
              acc = acc + h * uv;
            }
            lane[l] = acc;
          }
          for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
            memcpy(lane, tmp, sizeof lane);
          }
          partial = lane[0];
        } else {
          double q[4];
          for (int w = 0; w < 4; w++) {
            double acc = 0.0;
            for (size_t rr = 32 * (size_t)w; rr < 32 * (size_t)w + 32; rr++) {
              const size_t srow = c * SYMB + rr; /* a row of the stored block (c, K) */
              const double h = srow < n ? H[srow * n + row] : 0.0, uv = srow < n ? u[srow] : 0.0;
              acc = acc + h * uv;
            }
            q[w] = acc;
          }
          partial = ((q[0] + q[1]) + q[2]) + q[3];
        }
        total = c == 0 ? partial : total + partial;
      }
      out[row] = total;
    }
}

/* update_inverse_hessian, nlsolver.h:3130-3168 (literal, incl. the sign of the
 * s s^T term). */
void orc_update_inverse_hessian(double *H, const double *s, const double *y, double *t, double rho,
                                size_t n, int tree) {
  double *scratch = (double *)malloc(n * sizeof(double));
  ctx_t c = {n, tree, scratch};
  for (size_t i = 0; i < n; i++) t[i] = dot_(&c, y, H + i * n);
  double denom = dot_(&c, y, t);
  denom = (denom * rho) + 1.0;
  for (size_t j = 0; j < n; j++)
    for (size_t i = 0; i < n; i++)
      H[j * n + i] = H[j * n + i] - rho * (s[i] * t[j] + t[i] * s[j] + denom * s[i] * s[j]);
  free(scratch);
}

/* BFGS::solve<true>, nlsolver.h:3196-3285. x is in/out. */
static orc_status bfgs_solve(const model_t *q, double *x, size_t n, size_t max_iter, double grad_eps,
                             double alpha, int tree, orc_bfgs_counters *cnt) {
  double *H = (double *)calloc(n * n, sizeof(double));
  double *dir = (double *)calloc(n, sizeof(double)), *g = (double *)calloc(n, sizeof(double));
  double *pg = (double *)calloc(n, sizeof(double)), *y = (double *)calloc(n, sizeof(double));
  double *s = (double *)calloc(n, sizeof(double)), *tmp = (double *)calloc(n, sizeof(double));
  double *t = (double *)calloc(n, sizeof(double)), *scratch = (double *)calloc(n, sizeof(double));
  ctx_t c = {n, tree, scratch};
  orc_bfgs_counters local = {0};
  if (!cnt) cnt = &local;
  for (size_t i = 0; i < n; i++) H[i + i * n] = 1.0; /* :3212 */
  size_t iter = 0;
  model_g(&c, q, x, g, cnt); /* :3234 */
  double prev_norm = 1e9, cur_norm = 1e8; /* :3236-3237 */
  double fval;
  for (;;) {
    if (iter >= max_iter || cur_norm < grad_eps || fabs(cur_norm - prev_norm) < grad_eps ||
        isinf(cur_norm)) { /* :3239-3246 */
      fval = model_f(&c, q, x, cnt);
      break;
    }
    if (tree == 2) {
      sym_matvec(H, g, dir, n);
      for (size_t j = 0; j < n; j++) dir[j] = -dir[j];
    } else {
      for (size_t j = 0; j < n; j++) dir[j] = -dot_(&c, H + j * n, g); /* :3248-3251 */
    }
    const double phi = dot_(&c, g, dir);
    if ((phi > 0) || isnan(phi) || cur_norm > prev_norm) { /* :3253-3260 */
      memset(H, 0, n * n * sizeof(double));
      for (size_t i = 0; i < n; i++) {
        H[i + i * n] = 1.0;
        dir[i] = -g[i];
      }
    }
    memcpy(pg, g, n * sizeof(double)); /* :3261 */
    /* more_thuente_search overload without f value: evaluates f(x) first (:1885) */
    const double f0 = model_f(&c, q, x, cnt);
    double rate = alpha;
    cvsrch(&c, q, x, f0, g, &rate, dir, tmp, cnt);
    for (size_t i = 0; i < n; i++) s[i] = dir[i] * rate; /* :3266 */
    for (size_t i = 0; i < n; i++) x[i] += s[i];         /* :3268 */
    model_g(&c, q, x, g, cnt);                           /* :3271 */
    prev_norm = cur_norm;
    cur_norm = sqrt(dot_(&c, g, g)); /* math::norm :91-99 */
    for (size_t i = 0; i < n; i++) y[i] = g[i] - pg[i]; /* :3275 */
    double rho = dot_(&c, y, s);
    rho = 1 / rho; /* :3277-3278 */
    /* update_inverse_hessian with this context's summation order */
    if (tree == 2)
      sym_matvec(H, y, t, n);
    else
      for (size_t i = 0; i < n; i++) t[i] = dot_(&c, y, H + i * n);
    double denom = dot_(&c, y, t);
    denom = (denom * rho) + 1.0;
    for (size_t j = 0; j < n; j++)
      for (size_t i = 0; i < n; i++)
        H[j * n + i] = tree == 2
                           ? H[j * n + i] - rho * ((s[i] * t[j] + t[i] * s[j]) + denom * (s[i] * s[j]))
                           : H[j * n + i] - rho * (s[i] * t[j] + t[i] * s[j] + denom * s[i] * s[j]);
    if (cnt->H_out) memcpy(cnt->H_out, H, n * n * sizeof(double)); /* tests: the last inverse Hessian */
    iter++;
  }
  orc_status st = {fval, iter, cnt->f_calls, cnt->g_calls, 0};
  free(H);
  free(dir);
  free(g);
  free(pg);
  free(y);
  free(s);
  free(tmp);
  free(t);
  free(scratch);
  return st;
}

/* the G6 quadratic with its analytic gradient functor */
orc_status orc_bfgs_quad(const orc_quad *q, double *x, size_t n, size_t max_iter, double grad_eps,
                         double alpha, int tree, orc_bfgs_counters *cnt) {
  const model_t m = {q, 0};
  return bfgs_solve(&m, x, n, max_iter, grad_eps, alpha, tree, cnt);
}

/* a built-in objective (ORC_OBJ_*) with the default finite-difference gradient */
orc_status orc_bfgs_fd(int obj, double *x, size_t n, size_t max_iter, double grad_eps, double alpha,
                       int tree, orc_bfgs_counters *cnt) {
  const model_t m = {NULL, obj};
  return bfgs_solve(&m, x, n, max_iter, grad_eps, alpha, tree, cnt);
}
