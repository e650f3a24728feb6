/* oracle/oracle_nm.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Nelder-Mead, restated from nlsolver.h: max_abs_vec 1894-1904, simplex ctor 1905-1950
 * (effective simplex of SURVEY B1: the write to element [n][n] is out of bounds in the
 * reference and is dropped here), update_centroid 1965-1984, simplex_transform 1986-2007,
 * shrink 2009-2035, std_err 2037-2052, NelderMead::solve 2166-2299 (incl. B2 eps
 * mutation, B3 second-worst rule, B4 contraction with the reflect transform, the stale
 * all-zero centroid of iteration 1 when vertex 0 is the initial worst).
 *
 * order = 0: sequential sums (reference arithmetic, pinned by tests/golden/nm.json);
 * order = 1: objective and std_err use the HIP kernel's trees (everything else in the
 *            reference is already element-wise or sequential per coordinate).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static double nm_std_err(const double *x, size_t n, int order) {
  if (order == 0) return orc_std_err_serial(x, n);
  /* one wave: lane l adds x[l], x[l+64], ... in order, xor butterfly; two passes */
  double lane[64], tmp[64];
  memset(lane, 0, sizeof lane);
  for (size_t i = 0; i < n; i++) lane[i % 64] += x[i];
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
    memcpy(lane, tmp, sizeof lane);
  }
  const double mean = lane[0] / (double)n;
  memset(lane, 0, sizeof lane);
  for (size_t i = 0; i < n; i++) {
    const double d = x[i] - mean;
    lane[i % 64] += d * d;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
    memcpy(lane, tmp, sizeof lane);
  }
  return sqrt(lane[0] / (double)(n - 1));
}

static double dclampv(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }

/* simplex_transform<reflect, bound>, nlsolver.h:1986-2007 */
static void transform(const double *point, const double *centroid, double *result, double coef,
                      int reflect, int bound, const double *upper, const double *lower, size_t n) {
  for (size_t i = 0; i < n; i++) {
    double t = reflect ? centroid[i] + coef * (centroid[i] - point[i])
                       : centroid[i] + coef * (point[i] - centroid[i]);
    if (bound) t = dclampv(t, lower[i], upper[i]);
    result[i] = t;
  }
}

/* One NelderMead::solve<minimize, bound> (2166-2299). *eps is the solver's member (mutated,
 * B2). x in/out. */
orc_status orc_nm_solve(int obj, int minimize, int bound, double *x, size_t n, const double *upper,
                        const double *lower, double step, double alpha, double gamma, double rho,
                        double sigma, double *eps, size_t max_iter, size_t no_change_best_tol,
                        int order, orc_eval_log *log) {
  const size_t nv = n + 1;
  double *S = (double *)malloc(nv * n * sizeof(double)); /* vertices, row-major */
  double *scores = (double *)malloc(nv * sizeof(double));
  double *centroid = (double *)calloc(n, sizeof(double)); /* starts as zeros (:2195) */
  double *tr = (double *)malloc(n * sizeof(double)), *te = (double *)malloc(n * sizeof(double));
  double *tc = (double *)malloc(n * sizeof(double));
  size_t fcalls = 0;
  const double fm = minimize ? 1.0 : -1.0;
#define NM_F(ptr, out)                                                                    \
  do {                                                                                    \
    const double raw_ = order ? orc_objective_tree(obj, (ptr), n) : orc_objective_seq(obj, (ptr), n); \
    if (log) {                                                                            \
      if (log->count < log->capacity) {                                                   \
        memcpy(log->xs + log->count * log->D, (ptr), n * sizeof(double));                 \
        log->fs[log->count] = raw_;                                                       \
      }                                                                                   \
      log->count++;                                                                       \
    }                                                                                     \
    fcalls++;                                                                             \
    (out) = fm * raw_;                                                                    \
  } while (0)
  /* simplex ctor, 1910-1950 */
  for (size_t v = 0; v < nv; v++) memcpy(S + v * n, x, n * sizeof(double));
  if (step < 0) {
    double inf_norm = fabs(x[0]); /* max_abs_vec, 1894-1904 */
    for (size_t i = 1; i < n; i++) {
      const double t = fabs(x[i]);
      if (inf_norm < t) inf_norm = t;
    }
    const double a = inf_norm < 1.0 ? 1.0 : inf_norm;
    const double scale = a < 10 ? a : 10;
    for (size_t i = 1; i < n; i++) S[i * n + i] += scale; /* i == n: out of bounds, dropped */
    const double nn = (double)n;
    for (size_t i = 0; i < n; i++) S[i] = x[i] + ((1.0 - sqrt(nn + 1.0)) / nn * scale);
  } else {
    for (size_t i = 1; i < n; i++) S[i * n + i] += step;
  }
  for (size_t v = 0; v < nv; v++) NM_F(S + v * n, scores[v]); /* 2184-2186 */
  *eps = *eps * (scores[0] * *eps);                             /* 2189 (B2) */
  size_t best, worst = 0, second_worst = 0, prev_worst = 0, last_best = 99999999, no_change = 0;
  size_t iter = 0;
  int shrunk = 0;
  orc_status st;
  for (;;) {
    best = 0;
    prev_worst = worst;
    worst = 0;
    second_worst = 0;
    const double se = nm_std_err(scores, nv, order);
    for (size_t i = 1; i < nv; i++) { /* 2208-2221 (B3) */
      if (scores[i] < scores[best]) {
        best = i;
      } else if (scores[i] > scores[worst]) {
        second_worst = worst;
        worst = i;
      }
    }
    if (last_best == best) { /* 2223-2230 */
      no_change++;
    } else {
      no_change = 0;
      last_best = best;
    }
    if (iter >= max_iter || se < *eps || no_change >= no_change_best_tol) { /* 2233-2237 */
      memcpy(x, S + best * n, n * sizeof(double));
      st.f_value = scores[best];
      st.iteration = iter;
      st.function_calls_used = fcalls;
      st.gradient_evals_used = 0;
      st.hessian_evals_used = 0;
      break;
    }
    iter++;
    if (prev_worst != worst || shrunk) { /* update_centroid, 1965-1984 */
      memset(centroid, 0, n * sizeof(double));
      for (size_t v = 0; v < nv; v++) {
        if (v == worst) continue;
        for (size_t j = 0; j < n; j++) centroid[j] += S[v * n + j];
      }
      for (size_t j = 0; j < n; j++) centroid[j] /= (double)(nv - 1);
      shrunk = 0;
    }
    double ref_score, exp_score, cont_score;
    transform(S + worst * n, centroid, tr, alpha, 1, bound, upper, lower, n); /* 2245 */
    NM_F(tr, ref_score);
    if (ref_score >= scores[best] && ref_score < scores[second_worst]) { /* 2251 */
      memcpy(S + worst * n, tr, n * sizeof(double));
      scores[worst] = ref_score;
    } else if (ref_score < scores[best]) { /* expand, 2255-2265 */
      transform(tr, centroid, te, gamma, 0, bound, upper, lower, n);
      NM_F(te, exp_score);
      memcpy(S + worst * n, exp_score < ref_score ? te : tr, n * sizeof(double));
      scores[worst] = exp_score < ref_score ? exp_score : ref_score;
    } else { /* contraction, 2266-2297 (B4: reflect transform for both kinds) */
      transform(ref_score < scores[worst] ? tr : S + worst * n, centroid, tc, rho, 1, bound, upper,
                lower, n);
      NM_F(tc, cont_score);
      if (cont_score < (ref_score < scores[worst] ? ref_score : scores[worst])) {
        memcpy(S + worst * n, tc, n * sizeof(double));
        scores[worst] = cont_score;
      } else { /* shrink, 2009-2035, and rescoring 2288-2294 */
        for (size_t v = 0; v < nv; v++) {
          if (v == best) continue;
          for (size_t j = 0; j < n; j++)
            S[v * n + j] = S[best * n + j] + sigma * (S[v * n + j] - S[best * n + j]);
        }
        for (size_t v = 0; v < nv; v++) {
          if (v == best) continue;
          NM_F(S + v * n, scores[v]);
        }
        shrunk = 1;
      }
    }
  }
#undef NM_F
  free(S);
  free(scores);
  free(centroid);
  free(tr);
  free(te);
  free(tc);
  return st;
}

/* minimize()/maximize() incl. restarts (2127-2163): res.add() semantics of 2084-2091 */
orc_status orc_nm_run(int obj, int minimize, int bound, double *x, size_t n, const double *upper,
                      const double *lower, double step, double alpha, double gamma, double rho,
                      double sigma, double *eps, size_t max_iter, size_t no_change_best_tol,
                      size_t restarts, int order, orc_eval_log *log) {
  orc_status res = orc_nm_solve(obj, minimize, bound, x, n, upper, lower, step, alpha, gamma, rho,
                                sigma, eps, max_iter, no_change_best_tol, order, log);
  for (size_t r = 0; r < restarts; r++) {
    const orc_status more = orc_nm_solve(obj, minimize, bound, x, n, upper, lower, step, alpha,
                                         gamma, rho, sigma, eps, max_iter, no_change_best_tol,
                                         order, log);
    res.function_calls_used += more.function_calls_used;
    res.iteration += more.iteration;
    res.f_value = more.f_value;
  }
  return res;
}
