/* oracle/oracle_nmpso.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * The Nelder-Mead / PSO hybrid, restated from NelderMeadPSO (nlsolver.h:3546-3920): solve
 * 3623-3685, init_solver_state 3686-3738, apply_simplex 3739-3822, apply_pso 3823-3866,
 * update_centroid 3867-3884, shrink 3885-3902, simplex_std_err 3903-3918, with
 * simplex_transform 1986-2007 and max_abs_vec 1894-1904.
 *
 * Behaviour of the reference kept literally (each is visible in its runs):
 *   H1  the last simplex particle keeps x: init writes particle_positions[n][n], one element past
 *       the vector (3713-3716; out of bounds, dropped here, like NelderMead's B1 — and like there
 *       the reference only survives it for even n);
 *   H2  `best_val` is read once before the loop (3658) and never updated, so the no-change
 *       counter counts iterations whose best value EQUALS the first particle's initial value;
 *   H3  apply_pso works on a COPY of the particle's velocity (3838-3840: only `particle` is a
 *       reference), so velocities keep their initial values for the whole run;
 *   H4  the "better particle of each pair" is order[ns] for the first pair and the particle of
 *       rank 2m+1 for pair m >= 1 (3830-3836);
 *   H5  the bounded overloads clamp the velocity with lower[i] / upper[i] where i is the
 *       particle's rank (3853-3855; out of bounds for vectors of n elements). Not restated: with
 *       bound != 0 the clamp uses the coordinate index j, the evident intent.
 * std::sort's order among equal values is unspecified; here ties keep their current order.
 *
 *   orc_nmpso_serial  reference arithmetic and xorshift draw order; pinned to reference runs
 *                     (tests/golden/nmpso.json, unbounded overloads, even n).
 *   orc_nmpso_sync    what the GPU executes: counter-keyed draws, the wave's objective tree and
 *                     the wave tree for simplex_std_err.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static size_t hyb_last_shrinks; /* shrink steps of the most recent run (test coverage aid) */
size_t orc_nmpso_last_shrinks(void) { return hyb_last_shrinks; }

typedef struct {
  int obj, order; /* order: 0 reference arithmetic, 1 device trees */
  size_t n, fcalls;
  double fm;
  double *f_log;
  size_t f_cap, logged;
} hyb_ctx;

static double hyb_f(hyb_ctx *c, const double *x) {
  const double raw = c->order ? orc_objective_tree(c->obj, x, c->n) : orc_objective_seq(c->obj, x, c->n);
  if (c->f_log && c->logged < c->f_cap) c->f_log[c->logged] = raw;
  c->logged++;
  c->fcalls++;
  return c->fm * raw;
}

/* stable insertion sort of `order` by value; NaN sorts last */
static int hyb_less(double a, double b) { return (a < b) || (b != b && a == a); }
static void hyb_sort(size_t *order, size_t count, const double *values) {
  for (size_t i = 1; i < count; i++) {
    const size_t id = order[i];
    size_t k = i;
    while (k > 0 && hyb_less(values[id], values[order[k - 1]])) {
      order[k] = order[k - 1];
      k--;
    }
    order[k] = id;
  }
}

/* simplex_std_err (3903-3918) over the first `count` sorted particles */
static double hyb_std_err(const size_t *order, size_t count, const double *values, int tree) {
  if (!tree) {
    double mean = 0, result = 0;
    for (size_t i = 0; i < count; i++) mean += values[order[i]];
    mean /= (double)count;
    for (size_t i = 0; i < count; i++) result += pow(values[order[i]] - mean, 2);
    result /= (double)(count - 1);
    return sqrt(result);
  }
  /* one wave: lane l adds elements l, l+64, ... in order, xor butterfly; two passes */
  double lane[64], tmp[64];
  memset(lane, 0, sizeof lane);
  for (size_t i = 0; i < count; i++) lane[i % 64] += values[order[i]];
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
    memcpy(lane, tmp, sizeof lane);
  }
  const double mean = lane[0] / (double)count;
  memset(lane, 0, sizeof lane);
  for (size_t i = 0; i < count; i++) {
    const double d = values[order[i]] - mean;
    lane[i % 64] += d * d;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
    memcpy(lane, tmp, sizeof lane);
  }
  return sqrt(lane[0] / (double)(count - 1));
}

static double hyb_clamp(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }

static void hyb_transform(const double *point, const double *centroid, double *result, double coef,
                          int reflect, int bound, const double *upper, const double *lower, size_t n) {
  for (size_t i = 0; i < n; i++) {
    double t = reflect ? centroid[i] + coef * (centroid[i] - point[i])
                       : centroid[i] + coef * (point[i] - centroid[i]);
    if (bound) t = hyb_clamp(t, lower[i], upper[i]);
    result[i] = t;
  }
}

/* gen != NULL: serial draws (reference order); gen == NULL: keyed by (seed, instance). */
static orc_status hyb_solve(int obj, int minimize, int bound, double *x, size_t n,
                            const double *upper_in, const double *lower_in, orc_xorshift *gen,
                            uint64_t seed, uint64_t instance, double alpha, double gamma, double rho,
                            double sigma, double inertia, double cog, double soc, double eps,
                            size_t max_iter, size_t no_change_best_iter, double *f_log, size_t f_cap) {
  const int sync = gen == NULL;
  hyb_ctx c = {obj, sync, n, 0, minimize ? 1.0 : -1.0, f_log, f_cap, 0};
  if (n < 2) { /* :3627-3637 */
    orc_status bad = {999999, 0, 0, 0, 0};
    return bad;
  }
  const size_t ns = n + 1, np = 2 * n, total = ns + np;
  double *upper = (double *)malloc(n * sizeof(double)), *lower = (double *)malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++) {
    if (bound) {
      upper[i] = upper_in[i];
      lower[i] = lower_in[i];
    } else { /* :3587-3593 */
      const double temp = fabs(2.5 * x[i]);
      lower[i] = -temp;
      upper[i] = temp;
    }
  }
  double *pos = (double *)malloc(total * n * sizeof(double));
  double *vel = (double *)calloc(total * n, sizeof(double));
  double *val = (double *)malloc(total * sizeof(double));
  size_t *order = (size_t *)malloc(total * sizeof(size_t));
  double *centroid = (double *)malloc(n * sizeof(double)), *tr = (double *)malloc(n * sizeof(double));
  double *te = (double *)malloc(n * sizeof(double)), *tc = (double *)malloc(n * sizeof(double));
  double *pair = (double *)malloc(n * sizeof(double));
  const uint64_t kc = orc_ctr_key(seed, instance);
  /* init_solver_state (3686-3738) */
  {
    double inf_norm = fabs(x[0]);
    for (size_t i = 1; i < n; i++) {
      const double t = fabs(x[i]);
      if (inf_norm < t) inf_norm = t;
    }
    const double a = inf_norm < 1.0 ? 1.0 : inf_norm;
    const double scale = a < 10 ? a : 10;
    for (size_t i = 0; i < ns; i++) memcpy(pos + i * n, x, n * sizeof(double));
    for (size_t i = 1; i < n; i++) pos[i * n + i] = x[i] + scale; /* i == n: H1 */
    const double nn = (double)n;
    for (size_t i = 0; i < n; i++) pos[i] = x[i] + ((1.0 - sqrt(nn + 1.0)) / nn * scale);
    const uint64_t kinit = orc_ctr_key(kc, 0);
    for (size_t i = ns; i < total; i++) {
      const uint64_t kp = orc_ctr_key(kinit, i - ns);
      for (size_t j = 0; j < n; j++) {
        const double temp = fabs(upper[j] - lower[j]);
        const double u1 = sync ? orc_u01(orc_ctr_key(kp, 2 * j)) : orc_xorshift_next(gen);
        pos[i * n + j] = lower[j] + ((upper[j] - lower[j]) * u1);
        const double u2 = sync ? orc_u01(orc_ctr_key(kp, 2 * j + 1)) : orc_xorshift_next(gen);
        vel[i * n + j] = -temp + (u2 * temp);
      }
    }
    for (size_t i = 0; i < total; i++) val[i] = hyb_f(&c, pos + i * n);
  }
  for (size_t i = 0; i < total; i++) order[i] = i;
  hyb_last_shrinks = 0;
  size_t iter = 0, no_change = 0;
  const double best_val = val[0]; /* H2 */
  for (;;) {
    hyb_sort(order, total, val);
    const int same = best_val == val[order[0]];
    no_change += (size_t)same;
    no_change *= (size_t)same;
    if (iter >= max_iter || no_change >= no_change_best_iter ||
        hyb_std_err(order, ns, val, sync) < eps) {
      memcpy(x, pos + order[0] * n, n * sizeof(double));
      break;
    }
    /* apply_simplex (3739-3822) */
    {
      const double best_score = val[order[0]];
      const size_t worst = order[ns - 1], second = order[ns - 2];
      for (size_t j = 0; j < n; j++) centroid[j] = 0.0;
      for (size_t i = 0; i < ns - 1; i++)
        for (size_t j = 0; j < n; j++) centroid[j] += pos[order[i] * n + j];
      for (size_t j = 0; j < n; j++) centroid[j] /= (double)(ns - 1);
      hyb_transform(pos + worst * n, centroid, tr, alpha, 1, bound, upper, lower, n);
      const double ref_score = hyb_f(&c, tr);
      if (ref_score >= best_score && ref_score < val[second]) {
        memcpy(pos + worst * n, tr, n * sizeof(double));
        val[worst] = ref_score;
      } else if (ref_score < best_score) {
        hyb_transform(tr, centroid, te, gamma, 0, bound, upper, lower, n);
        const double exp_score = hyb_f(&c, te);
        memcpy(pos + worst * n, exp_score < ref_score ? te : tr, n * sizeof(double));
        val[worst] = exp_score < ref_score ? exp_score : ref_score;
      } else {
        const double worst_score = val[worst];
        hyb_transform(ref_score < worst_score ? tr : pos + worst * n, centroid, tc, rho, 0, bound,
                      upper, lower, n);
        const double cont_score = hyb_f(&c, tc);
        if (cont_score < (worst_score < ref_score ? worst_score : ref_score)) { /* std::min */
          memcpy(pos + worst * n, tc, n * sizeof(double));
          val[worst] = cont_score;
        } else {
          hyb_last_shrinks++;
          const double *best = pos + order[0] * n; /* shrink (3885-3902) */
          for (size_t i = 1; i < ns; i++) {
            double *cur = pos + order[i] * n;
            for (size_t j = 0; j < n; j++) cur[j] = best[j] + sigma * (cur[j] - best[j]);
          }
          for (size_t i = 1; i < ns; i++) val[order[i]] = hyb_f(&c, pos + order[i] * n);
          hyb_sort(order, total, val);
        }
      }
    }
    /* apply_pso (3823-3866) */
    {
      const uint64_t kit = orc_ctr_key(kc, iter + 1);
      int flip = 0;
      size_t best_in_pair = order[ns];
      const double *best = pos + order[0] * n;
      for (size_t i = ns; i < total; i++) {
        const size_t id = order[i];
        if (flip) best_in_pair = order[i + 1]; /* H4 */
        flip = (int)((i - ns) % 2);
        double *particle = pos + id * n;
        const double *velocity = vel + id * n; /* H3: never written back */
        memcpy(pair, pos + best_in_pair * n, n * sizeof(double));
        const uint64_t kp = orc_ctr_key(kit, i - ns);
        for (size_t j = 0; j < n; j++) {
          const double r_p = sync ? orc_u01(orc_ctr_key(kp, 2 * j)) : orc_xorshift_next(gen);
          const double r_g = sync ? orc_u01(orc_ctr_key(kp, 2 * j + 1)) : orc_xorshift_next(gen);
          double temp = (inertia * velocity[j]) + cog * r_p * (pair[j] - particle[j]) +
                        soc * r_g * (best[j] - particle[j]);
          if (bound) temp = hyb_clamp(temp, lower[j], upper[j]); /* H5 */
          particle[j] += temp;
        }
        val[id] = hyb_f(&c, particle);
      }
    }
    iter++;
  }
  orc_status st = {val[order[0]], iter, c.fcalls, 0, 0};
  free(upper);
  free(lower);
  free(pos);
  free(vel);
  free(val);
  free(order);
  free(centroid);
  free(tr);
  free(te);
  free(tc);
  free(pair);
  return st;
}

orc_status orc_nmpso_serial(int obj, int minimize, int bound, double *x, size_t n,
                            const double *upper, const double *lower, orc_xorshift *gen, double alpha,
                            double gamma, double rho, double sigma, double inertia, double cog,
                            double soc, double eps, size_t max_iter, size_t no_change_best_iter,
                            double *f_log, size_t f_cap) {
  return hyb_solve(obj, minimize, bound, x, n, upper, lower, gen, 0, 0, alpha, gamma, rho, sigma,
                   inertia, cog, soc, eps, max_iter, no_change_best_iter, f_log, f_cap);
}

/* Draws of instance `instance`: kc = key(seed, instance); initialisation uses key(kc, 0),
 * iteration it uses key(kc, it + 1); PSO particle of rank r (0 .. 2n-1) has kp = key(that, r)
 * and coordinate j takes draws 2j and 2j + 1 of kp. */
orc_status orc_nmpso_sync(int obj, int minimize, int bound, double *x, size_t n, const double *upper,
                          const double *lower, uint64_t seed, uint64_t instance, double alpha,
                          double gamma, double rho, double sigma, double inertia, double cog,
                          double soc, double eps, size_t max_iter, size_t no_change_best_iter,
                          double *f_log, size_t f_cap) {
  return hyb_solve(obj, minimize, bound, x, n, upper, lower, NULL, seed, instance, alpha, gamma, rho,
                   sigma, inertia, cog, soc, eps, max_iter, no_change_best_iter, f_log, f_cap);
}
