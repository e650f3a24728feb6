/* oracle/oracle_pso.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Particle Swarm Optimisation, restated from nlsolver.h:2479-2742.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static void log_eval(orc_eval_log *log, const double *x, double f) {
  if (!log) return;
  if (log->count < log->capacity) {
    memcpy(log->xs + log->count * log->D, x, log->D * sizeof(double));
    log->fs[log->count] = f;
  }
  log->count++;
}

/* rnorm, nlsolver.h:2479-2485: u1 feeds log, u2 feeds cos (g++/clang evaluate the
 * two generator() operands left to right); pi_ = 3.141593. */
static double rnorm_serial(orc_xorshift *g) {
  const double pi_ = 3.141593;
  const double u1 = orc_xorshift_next(g);
  const double u2 = orc_xorshift_next(g);
  return sqrt(-2 * log(u1)) * cos(2 * pi_ * u2);
}

/* update_positions, Accelerated (nlsolver.h:2687-2699), and threshold_positions (2701-2715) for one
 * coordinate: the ONE piece of code both the serial restatement (pinned to the reference's runs)
 * and the synchronous one (what the GPU executes) run; they differ in where `normal` comes from. */
static double apso_position(double inertia, double normal, double cog, double pos, double soc,
                            double gbest_j) {
  return inertia * normal + (1 - cog) * pos + soc * gbest_j;
}
static double clamp_position(double p, double lower, double upper) {
  p = p < lower ? lower : p;
  p = p > upper ? upper : p;
  return p;
}
/* the same for a whole particle, as a probe for tests */
void orc_pso_accel_move_from_normals(double *pos, const double *normals, const double *gbest,
                                     const double *lower, const double *upper, size_t D,
                                     double inertia, double cog, double soc, int bounded) {
  for (size_t j = 0; j < D; j++) {
    double p = apso_position(inertia, normals[j], cog, pos[j], soc, gbest[j]);
    if (bounded) p = clamp_position(p, lower[j], upper[j]);
    pos[j] = p;
  }
}

/* PSO::solve (nlsolver.h:2593-2624) with init_solver_state (2626-2657),
 * update_velocities (2658-2677, literal incl. B7), update_positions (2678-2700),
 * threshold_positions (2701-2715), update_best_positions (2716-2741).
 * type: 0 = Vanilla, 1 = Accelerated. bounded = 0: bounds = -+|x_i| and no
 * thresholding (2553-2563). Vanilla requires n <= D (the reference indexes
 * swarm_best_position with the particle index, 2674). */
orc_status orc_pso_serial(int obj, int minimize, int type, int bounded, double *x, size_t D,
                          const double *lower_in, const double *upper_in, orc_xorshift *gen,
                          double inertia, double cog, double soc, size_t n, size_t max_iter,
                          size_t best_val_no_change, double eps, orc_eval_log *log) {
  double *lower = (double *)malloc(D * sizeof(double)), *upper = (double *)malloc(D * sizeof(double));
  for (size_t j = 0; j < D; j++) {
    if (bounded) {
      lower[j] = lower_in[j];
      upper[j] = upper_in[j];
    } else {
      const double t = fabs(x[j]); /* :2556-2559 */
      lower[j] = -t;
      upper[j] = t;
    }
  }
  double *pos = (double *)malloc(n * D * sizeof(double));
  double *vel = (double *)calloc(n * D, sizeof(double));
  double *pbest = (double *)malloc(n * sizeof(double));
  double *gbest = (double *)calloc(D, sizeof(double));
  const double init_inertia = inertia;
  double swarm_best = 100000.0; /* :2631 */
  size_t f_evals = 0, val_no_change = 0, iter = 0;
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < D; j++) { /* :2642-2654 */
      const double temp = fabs(upper[j] - lower[j]);
      pos[i * D + j] = lower[j] + ((upper[j] - lower[j]) * orc_xorshift_next(gen));
      if (type == 0) vel[i * D + j] = -temp + (orc_xorshift_next(gen) * temp);
    }
  for (size_t i = 0; i < n; i++) pbest[i] = 10000; /* :2655-2656 */
  const double fm = minimize ? 1.0 : -1.0;
  int have_gbest = 0;
  for (;;) {
    /* update_best_positions, :2716-2741 */
    size_t best_index = 0;
    int update_happened = 0;
    for (size_t i = 0; i < n; i++) {
      const double fv = orc_objective_seq(obj, pos + i * D, D);
      log_eval(log, pos + i * D, fv);
      const double temp = fm * fv;
      if (temp < swarm_best) {
        swarm_best = temp;
        best_index = i;
        update_happened = 1;
      }
      if (temp < pbest[i]) pbest[i] = temp;
    }
    f_evals += n;
    if (update_happened) {
      memcpy(gbest, pos + best_index * D, D * sizeof(double));
      have_gbest = 1;
    }
    val_no_change = (size_t)(best_index == 0) * (val_no_change + 1); /* :2740 (B9) */
    if (iter >= max_iter || val_no_change >= best_val_no_change ||
        orc_std_err_serial(pbest, n) < eps) { /* :2599-2605 */
      memcpy(x, gbest, D * sizeof(double));
      orc_status st = {swarm_best, iter, f_evals, 0, 0};
      free(lower);
      free(upper);
      free(pos);
      free(vel);
      free(pbest);
      free(gbest);
      (void)have_gbest;
      return st;
    }
    if (type == 0) { /* update_velocities, :2658-2677 (literal, B7) */
      for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < D; j++) {
          const double r_p = orc_xorshift_next(gen), r_g = orc_xorshift_next(gen);
          vel[i * D + j] = (inertia * vel[i * D + j]) +
                           cog * r_p * (pos[i * D + j] - pos[i * D + j]) +
                           soc * r_g * (gbest[i] - pos[i * D + j]);
        }
      for (size_t i = 0; i < n * D; i++) pos[i] += vel[i]; /* :2679-2686 */
    } else {
      inertia = pow(init_inertia, (double)iter); /* :2613 */
      for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < D; j++) /* :2687-2699 */
          pos[i * D + j] = apso_position(inertia, rnorm_serial(gen), cog, pos[i * D + j], soc, gbest[j]);
    }
    if (bounded) { /* :2701-2715 */
      for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < D; j++) pos[i * D + j] = clamp_position(pos[i * D + j], lower[j], upper[j]);
    }
    iter++;
  }
}

/* ------------------------------------------------------------------------- */
/* Synchronous restatement (what the GPU executes)                            */
/* ------------------------------------------------------------------------- */
static double rnorm_ctr(uint64_t kp, size_t j) {
  /* one draw (slot 2j) per normal variate: u1 from all of it, u2 from its low 32 bits */
  return orc_rnorm(orc_ctr_key(kp, 2 * j));
}

void orc_pso_sync_init(orc_pso_sync *s) {
  const uint64_t kg = orc_ctr_key(s->seed, 0);
  const double fm = s->minimize ? 1.0 : -1.0;
  for (size_t i = 0; i < s->n; i++) {
    const uint64_t kp = orc_ctr_key(kg, i);
    double *row = s->pos + i * s->D;
    for (size_t j = 0; j < s->D; j++) { /* :2642-2654 */
      const double lo = s->lower[j], hi = s->upper[j];
      const double temp = fabs(hi - lo);
      row[j] = lo + ((hi - lo) * orc_u01(orc_ctr_key(kp, 2 * j)));
      if (s->type == 0) {
        s->vel[i * s->D + j] = -temp + (orc_u01(orc_ctr_key(kp, 2 * j + 1)) * temp);
        s->pbest_pos[i * s->D + j] = row[j];
      }
    }
    const double f = fm * orc_objective_tree(s->obj, row, s->D);
    s->cur_val[i] = f;
    s->pbest_val[i] = f; /* +inf sentinel: the first value always wins (B8) */
  }
  s->gbest_val = INFINITY;
  s->gbest_idx = 0;
  s->iter = 0;
  s->val_no_change = 0;
  s->fevals = 0;
  s->done = 0;
  s->std_err = NAN;
  s->inertia = s->inertia0;
}

void orc_pso_shard_record(const orc_pso_sync *s, size_t lo, size_t n, double *rec) {
  const double *sc = s->cur_val + lo;
  double bv = INFINITY;
  size_t bi = (size_t)-1;
  for (size_t i = 0; i < n; i++)
    if (sc[i] < bv || (sc[i] == bv && bi == (size_t)-1)) {
      bv = sc[i];
      bi = i;
    }
  const int valid = bi != (size_t)-1;
  const uint64_t gi = valid ? lo + bi : 0;
  double sum = 0.0, m2 = 0.0;
  if (s->eps > 0) {
    sum = orc_tiled_sum(s->pbest_val + lo, n);
    m2 = orc_tiled_sumsq_dev(s->pbest_val + lo, n, sum / (double)n);
  }
  rec[0] = bv;
  memcpy(&rec[1], &gi, 8);
  rec[2] = sum;
  rec[3] = m2;
  rec[4] = valid ? 1.0 : 0.0;
  for (size_t d = 0; d < s->D; d++) rec[ORC_DE_REC_HEADER + d] = valid ? s->pos[gi * s->D + d] : 0.0;
}

int orc_pso_apply_records(orc_pso_sync *s, const double *recs, int world) {
  if (s->done) return 1;
  const size_t stride = ORC_DE_REC_HEADER + s->D;
  /* update_best_positions (:2716-2741) on the scores of the last evaluation: the
   * first occurrence of the minimum wins if it is strictly below the incumbent */
  int win = -1;
  double bv = INFINITY;
  uint64_t bi = 0;
  for (int r = 0; r < world; r++) {
    const double *rec = recs + (size_t)r * stride;
    if (rec[4] != 1.0) continue;
    uint64_t i;
    memcpy(&i, &rec[1], 8);
    if (win < 0 || rec[0] < bv || (rec[0] == bv && i < bi)) {
      bv = rec[0];
      bi = i;
      win = r;
    }
  }
  const int update_happened = win >= 0 && bv < s->gbest_val;
  if (update_happened) {
    s->gbest_val = bv;
    s->gbest_idx = bi;
    memcpy(s->gbest_x, recs + (size_t)win * stride + ORC_DE_REC_HEADER, s->D * sizeof(double));
  }
  s->fevals += s->n;
  s->val_no_change = update_happened ? 0 : s->val_no_change + 1; /* :2740 with B9 repaired */
  double se = NAN;
  if (s->eps > 0) { /* std_err(particle_best_values), :2601 */
    const double n_r = (double)(s->n / (size_t)world);
    double tot = 0.0;
    for (int r = 0; r < world; r++) tot = tot + recs[(size_t)r * stride + 2];
    const double gmean = tot / (double)s->n;
    double m2 = 0.0;
    for (int r = 0; r < world; r++) {
      const double *rec = recs + (size_t)r * stride;
      double term = rec[3];
      if (world > 1) {
        const double dm = rec[2] / n_r - gmean;
        term = term + n_r * (dm * dm);
      }
      m2 = m2 + term;
    }
    se = sqrt(m2 / (double)(s->n - 1));
  }
  s->std_err = se;
  if (s->iter >= s->max_iter || s->val_no_change >= s->best_val_no_change ||
      (s->eps > 0 && se < s->eps)) { /* :2599-2600 */
    s->done = 1;
    return 1;
  }
  return 0;
}

static void pso_particle(orc_pso_sync *s, uint64_t kg, size_t i) {
  const size_t D = s->D;
  const uint64_t kp = orc_ctr_key(kg, i);
  double *row = s->pos + i * D;
  for (size_t j = 0; j < D; j++) {
    double p;
    if (s->type == 1) { /* Accelerated, :2687-2699 */
      p = apso_position(s->inertia, rnorm_ctr(kp, j), s->cog, row[j], s->soc, s->gbest_x[j]);
    } else { /* Vanilla with the intended update (B7 repaired): pbest[j]-pos, gbest[j]-pos */
      const double r_p = orc_u01(orc_ctr_key(kp, 2 * j)), r_g = orc_u01(orc_ctr_key(kp, 2 * j + 1));
      double *v = s->vel + i * D + j;
      *v = (s->inertia * *v) + s->cog * r_p * (s->pbest_pos[i * D + j] - row[j]) +
           s->soc * r_g * (s->gbest_x[j] - row[j]);
      p = row[j] + *v; /* :2683 */
    }
    if (s->bounded) p = clamp_position(p, s->lower[j], s->upper[j]); /* :2701-2715 */
    row[j] = p;
  }
  const double fm = s->minimize ? 1.0 : -1.0;
  const double f = fm * orc_objective_tree(s->obj, row, D);
  s->cur_val[i] = f;
  if (f < s->pbest_val[i]) { /* :2733-2735 */
    s->pbest_val[i] = f;
    if (s->type == 0) memcpy(s->pbest_pos + i * D, row, D * sizeof(double));
  }
}

void orc_pso_shard_move(orc_pso_sync *s, size_t lo, size_t n, int threads) {
  /* Accelerated: inertia = pow(init_inertia, iter) (:2613, libm as in the reference;
   * the device receives the same values as a host-computed table) */
  if (s->type == 1) s->inertia = pow(s->inertia0, (double)s->iter);
  const uint64_t kg = orc_ctr_key(s->seed, s->iter + 1);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (long i = (long)lo; i < (long)(lo + n); i++) pso_particle(s, kg, (size_t)i);
}

void orc_pso_commit(orc_pso_sync *s) { s->iter++; }

void orc_pso_sync_step(orc_pso_sync *s, int threads) {
  if (s->done) return;
  const int world = (int)s->n_shards;
  const size_t shard_n = s->n / s->n_shards;
  const size_t stride = ORC_DE_REC_HEADER + s->D;
  double *recs = (double *)malloc((size_t)world * stride * sizeof(double));
  for (int r = 0; r < world; r++)
    orc_pso_shard_record(s, (size_t)r * shard_n, shard_n, recs + (size_t)r * stride);
  const int done = orc_pso_apply_records(s, recs, world);
  free(recs);
  if (done) return;
  orc_pso_shard_move(s, 0, s->n, threads);
  orc_pso_commit(s);
}
