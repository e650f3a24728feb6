/* oracle/oracle_sann.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Simulated annealing, restated from SANN::solve (nlsolver.h:2777-2814) with rnorm
 * (2479-2485).
 *   orc_sann_serial  the reference's arithmetic and draw order (xorshift, libm log / cos /
 *                    exp): pinned to reference runs (tests/golden/sann.json).
 *   orc_sann_sync    what the GPU executes: the same chain with counter-keyed draws, the
 *                    deterministic log / cos / exp and the wave's objective tree.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static void log_value(double *f_log, size_t f_cap, size_t *count, double f) {
  if (f_log && *count < f_cap) f_log[*count] = f;
  (*count)++;
}

orc_status orc_sann_serial(int obj, int minimize, double *x, size_t D, orc_xorshift *gen,
                           size_t max_iter, size_t temp_iter, double temp_max, double *f_log,
                           size_t f_cap) {
  const double fm = minimize ? 1.0 : -1.0, e_minus_1 = 1.7182818, pi_ = 3.141593; /* :2780, 2481 */
  size_t f_evals = 0, logged = 0, iter = 0;
  double fv = orc_objective_seq(obj, x, D);
  log_value(f_log, f_cap, &logged, fv);
  double best_val = fm * fv; /* :2781 */
  const double scale = 1.0 / temp_max;
  f_evals++;
  double *p = (double *)malloc(D * sizeof(double)), *ptry = (double *)malloc(D * sizeof(double));
  memcpy(p, x, D * sizeof(double));
  memcpy(ptry, x, D * sizeof(double));
  for (;;) {
    if (iter >= max_iter) break; /* :2788-2791 */
    const double t = temp_max / log((double)iter + e_minus_1); /* :2793-2794 */
    for (size_t j = 1; j < temp_iter; j++) {
      const double current_scale = t * scale;
      for (size_t i = 0; i < D; i++) { /* :2798-2801; rnorm draws u1 (log) then u2 (cos) */
        const double u1 = orc_xorshift_next(gen);
        const double u2 = orc_xorshift_next(gen);
        ptry[i] = p[i] + current_scale * (sqrt(-2 * log(u1)) * cos(2 * pi_ * u2));
      }
      fv = orc_objective_seq(obj, ptry, D);
      log_value(f_log, f_cap, &logged, fv);
      const double current_val = fm * fv;
      f_evals++;
      const double difference = current_val - best_val; /* against the best, not the current */
      if ((difference <= 0.0) || (orc_xorshift_next(gen) < exp(-difference / t))) { /* :2805 */
        memcpy(p, ptry, D * sizeof(double));
        if (current_val <= best_val) {
          memcpy(x, p, D * sizeof(double));
          best_val = current_val;
        }
      }
    }
    iter++;
  }
  free(p);
  free(ptry);
  orc_status st = {best_val, iter, f_evals, 0, 0};
  return st;
}

/* Draws of chain `chain`: kc = key(seed, chain); inner step s = iter * (temp_iter - 1) + (j - 1)
 * has ks = key(kc, s); coordinate e uses draw 2e of ks for both uniforms of its normal variate, the acceptance
 * test draw 2 D. */
orc_status orc_sann_sync(int obj, int minimize, double *x, size_t D, uint64_t seed, uint64_t chain,
                         size_t max_iter, size_t temp_iter, double temp_max, double *f_log,
                         size_t f_cap) {
  const double fm = minimize ? 1.0 : -1.0, e_minus_1 = 1.7182818;
  size_t f_evals = 0, logged = 0, iter = 0;
  const uint64_t kc = orc_ctr_key(seed, chain);
  double fv = orc_objective_tree(obj, x, D);
  log_value(f_log, f_cap, &logged, fv);
  double best_val = fm * fv;
  const double scale = 1.0 / temp_max;
  f_evals++;
  double *p = (double *)malloc(D * sizeof(double)), *ptry = (double *)malloc(D * sizeof(double));
  memcpy(p, x, D * sizeof(double));
  memcpy(ptry, x, D * sizeof(double));
  for (;;) {
    if (iter >= max_iter) break;
    const double t = temp_max / orc_log((double)iter + e_minus_1);
    for (size_t j = 1; j < temp_iter; j++) {
      const uint64_t ks = orc_ctr_key(kc, iter * (temp_iter - 1) + (j - 1));
      const double current_scale = t * scale;
      for (size_t i = 0; i < D; i++) {
        const uint64_t z1 = orc_ctr_key(ks, 2 * i); /* one draw per normal variate */
        ptry[i] = p[i] + current_scale * orc_rnorm(z1);
      }
      fv = orc_objective_tree(obj, ptry, D);
      log_value(f_log, f_cap, &logged, fv);
      const double current_val = fm * fv;
      f_evals++;
      const double difference = current_val - best_val;
      if ((difference <= 0.0) ||
          (orc_u01(orc_ctr_key(ks, 2 * D)) < orc_exp(-difference / t))) {
        memcpy(p, ptry, D * sizeof(double));
        if (current_val <= best_val) {
          memcpy(x, p, D * sizeof(double));
          best_val = current_val;
        }
      }
    }
    iter++;
  }
  free(p);
  free(ptry);
  orc_status st = {best_val, iter, f_evals, 0, 0};
  return st;
}
