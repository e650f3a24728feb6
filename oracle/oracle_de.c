/* oracle/oracle_de.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Differential Evolution, restated from nlsolver.h:2302-2477.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static void log_eval(orc_eval_log *log, const double *x, double f) {
  if (!log || log->count >= log->capacity) {
    if (log) log->count++;
    return;
  }
  memcpy(log->xs + log->count * log->D, x, log->D * sizeof(double));
  log->fs[log->count] = f;
  log->count++;
}

/* Where the reference's generator draws come from: the xorshift stream (orc_de_serial, pinned to
 * the goldens) or a replayed list (orc_de_serial_proposal_from_draws: the SAME code below fed
 * with the keyed draws of the synchronous restatement, tests/test_oracle_properties.py). */
typedef struct {
  orc_xorshift *g;
  const double *replay;
  size_t pos, n;
} draw_src;
static double next_draw(draw_src *d) {
  if (d->g) return orc_xorshift_next(d->g);
  return d->pos < d->n ? d->replay[d->pos++] : (d->pos++, 0.0);
}

/* generate_index, nlsolver.h:2325-2329: size_t(u * max). */
static size_t gen_index(size_t max, draw_src *g) {
  return (size_t)(next_draw(g) * (double)max);
}

/* generate_indices, nlsolver.h:2331-2355: three distinct proposals != fixed by
 * rejection; the unordered_set is a membership test over <= 3 values. */
static void gen_indices(size_t fixed, size_t max, draw_src *g, size_t out[4]) {
  out[0] = fixed;
  size_t samples = 1;
  for (;;) {
    const size_t proposal = gen_index(max, g);
    int used = 0;
    for (size_t k = 0; k < samples; k++) used |= (out[k] == proposal);
    if (!used) {
      out[samples++] = proposal;
      if (samples == 4) return;
    }
  }
}

/* propose_new_agent, nlsolver.h:2357-2375: forced dimension, then per coordinate one draw
 * (always taken: the left operand of ||) and the mutant or the base agent's coordinate */
static size_t propose(const double *agents, const size_t ids[4], size_t D, double CR, double F,
                      draw_src *g, double *proposal) {
  const size_t dim = gen_index(D, g);
  for (size_t d = 0; d < D; d++) {
    const double u = next_draw(g);
    if (u < CR || d == dim) {
      proposal[d] = agents[ids[1] * D + d] + F * (agents[ids[2] * D + d] - agents[ids[3] * D + d]);
    } else {
      proposal[d] = agents[ids[0] * D + d];
    }
  }
  return dim;
}

/* The reference's donor pick and proposal for one agent, fed with given draws instead of the
 * xorshift stream: donor_draws (consumed until three distinct donors != fixed are found; returns
 * how many in *donor_used), then cross_draws[0] for the forced dimension and cross_draws[1 .. D]
 * for the coordinates. ids_out[4] = {fixed, r1, r2, r3}; returns the forced dimension. */
size_t orc_de_serial_proposal_from_draws(const double *agents, size_t pop, size_t D, size_t fixed,
                                         double CR, double F, const double *donor_draws,
                                         size_t n_donor, const double *cross_draws, size_t *ids_out,
                                         size_t *donor_used, double *proposal) {
  draw_src dd = {NULL, donor_draws, 0, n_donor};
  gen_indices(fixed, pop, &dd, ids_out);
  *donor_used = dd.pos;
  draw_src dc = {NULL, cross_draws, 0, D + 1};
  return propose(agents, ids_out, D, CR, F, &dc, proposal);
}

orc_status orc_de_serial(int obj, int minimize, int strategy, double *x, size_t D,
                         orc_xorshift *gen, double CR, double F, double eps,
                         size_t pop, size_t max_iter, size_t best_val_no_change,
                         orc_eval_log *log) {
  double *agents = (double *)malloc(pop * D * sizeof(double));
  double *scores = (double *)malloc(pop * sizeof(double));
  double *proposal = (double *)malloc(D * sizeof(double));
  /* init_agents / generate_sequence, nlsolver.h:2302-2323: (u - 0.5) * x0[i],
   * agent-major draw order. */
  draw_src src = {gen, NULL, 0, 0};
  for (size_t a = 0; a < pop; a++)
    for (size_t i = 0; i < D; i++)
      agents[a * D + i] = (next_draw(&src) - 0.5) * x[i];
  const double fm = minimize ? 1.0 : -1.0; /* :2418 */
  for (size_t a = 0; a < pop; a++) {       /* :2423-2425 */
    const double f = orc_objective_seq(obj, agents + a * D, D);
    log_eval(log, agents + a * D, f);
    scores[a] = fm * f;
  }
  size_t fcalls = pop, iter = 0, best_id = 0, val_no_change = 0;
  for (;;) {
    int not_updated = 1;
    for (size_t i = 0; i < pop; i++) { /* :2432-2437 strict '<' vs incumbent */
      if (scores[i] < scores[best_id]) {
        best_id = i;
        not_updated = 0;
      }
    }
    val_no_change = (size_t)not_updated * (val_no_change + 1); /* :2439 */
    if (iter >= max_iter || val_no_change >= best_val_no_change ||
        orc_std_err_serial(scores, pop) < eps) { /* :2441-2447 */
      memcpy(x, agents + best_id * D, D * sizeof(double));
      orc_status st = {scores[best_id], iter, fcalls, 0, 0};
      free(agents);
      free(scores);
      free(proposal);
      return st;
    }
    for (size_t i = 0; i < pop; i++) { /* :2449 */
      size_t ids[4];
      gen_indices(strategy == 1 ? i : best_id, pop, &src, ids); /* :2451-2457 */
      propose(agents, ids, D, CR, F, &src, proposal);           /* :2357-2375 */
      const double f = orc_objective_seq(obj, proposal, D);
      log_eval(log, proposal, f);
      const double score = fm * f; /* :2463 */
      fcalls++;
      if (score < scores[i]) { /* :2466-2471 in-place replacement */
        memcpy(agents + i * D, proposal, D * sizeof(double));
        scores[i] = score;
      }
    }
    iter++;
  }
}

/* ------------------------------------------------------------------------- */
/* Synchronous restatement                                                    */
/* ------------------------------------------------------------------------- */
#define ORC_DE_MAX_TRIES 64

static size_t clamp_index(double u, size_t n) {
  size_t p = (size_t)(u * (double)n); /* nlsolver.h:2328 */
  return p >= n ? n - 1 : p;          /* B10: u can be exactly 1.0 */
}

void orc_de_sync_init(orc_de_sync *s, const double *x0) {
  const uint64_t kg = orc_ctr_key(s->seed, 0);
  const double fm = s->minimize ? 1.0 : -1.0;
  for (size_t a = 0; a < s->pop; a++) {
    const uint64_t ka = orc_ctr_key(kg, a);
    double *row = s->cur + a * s->D;
    for (size_t d = 0; d < s->D; d++) /* nlsolver.h:2309 */
      row[d] = (orc_u01(orc_ctr_key(ka, d)) - 0.5) * x0[d];
    s->scores[a] = fm * orc_objective_tree(s->obj, row, s->D); /* :2423-2425 */
  }
  s->best_id = 0;
  s->iter = 0;
  s->val_no_change = 0;
  s->fcalls = s->pop;
  s->done = 0;
  s->std_err = NAN;
}

static void sync_agent(const orc_de_sync *s, uint64_t kg, size_t a, double *trial) {
  const size_t D = s->D, pop = s->pop;
  const size_t shard_n = pop / s->n_shards;
  const size_t lo = (a / shard_n) * shard_n;
  const uint64_t ka = orc_ctr_key(kg, a);
  const size_t fixed = s->strategy == 1 ? a : (size_t)s->best_id; /* :2451-2457 */
  /* donors: slots D+1+k, rejection as in generate_indices (2331-2355), drawn
   * inside the agent's shard (island model); bounded number of tries. */
  size_t r[3];
  size_t have = 0;
  for (size_t k = 0; k < ORC_DE_MAX_TRIES && have < 3; k++) {
    const size_t p = lo + clamp_index(orc_u01(orc_ctr_key(ka, D + 1 + k)), shard_n);
    int used = (p == fixed);
    for (size_t j = 0; j < have; j++) used |= (r[j] == p);
    if (!used) r[have++] = p;
  }
  for (size_t p = lo; have < 3; p++) { /* fallback: lowest unused indices */
    int used = (p == fixed);
    for (size_t j = 0; j < have; j++) used |= (r[j] == p);
    if (!used) r[have++] = p;
  }
  const size_t dim = clamp_index(orc_u01(orc_ctr_key(ka, D)), D); /* :2364 */
  const double *r1 = s->cur + r[0] * D, *r2 = s->cur + r[1] * D, *r3 = s->cur + r[2] * D;
  const double *base = s->cur + fixed * D;
  for (size_t d = 0; d < D; d++) { /* :2365-2374 */
    const double u = orc_u01(orc_ctr_key(ka, d));
    trial[d] = (u < s->CR || d == dim) ? r1[d] + s->F * (r2[d] - r3[d]) : base[d];
  }
  const double fm = s->minimize ? 1.0 : -1.0;
  const double score = fm * orc_objective_tree(s->obj, trial, D); /* :2463 */
  const int accept = score < s->scores[a];                         /* :2466 */
  memcpy(s->nxt + a * D, accept ? trial : s->cur + a * D, D * sizeof(double));
  if (accept) s->scores[a] = score;
  if (s->trace) {
    uint64_t *t = s->trace + a * 5;
    t[0] = r[0];
    t[1] = r[1];
    t[2] = r[2];
    t[3] = dim;
    t[4] = (uint64_t)accept;
  }
}

/* ---- the turn head, in the record form the multi-GPU path exchanges ---------
 * Record of one shard (ORC_DE_REC_HEADER + D doubles):
 *   [minv, mini (u64 bits), sum, m2, valid, x_best[0..D)]
 * (sum, m2) of a shard come from its tiles of 1024 scores, each reduced in one pass to
 * (sum_t, M2_t about the tile mean) and merged (orc_tiled_m2_merged); several shards merge
 * (n, sum, M2) the same way in rank order. A population of one tile degenerates to the
 * reference's two-pass std_err (nlsolver.h:2037-2052) in tree order. */
static uint64_t dbl_bits(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
static double bits_dbl(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

void orc_de_shard_record(const orc_de_sync *s, size_t lo, size_t n, double *rec) {
  const double *sc = s->scores + lo;
  /* first index of the shard minimum; NaN is never "less" */
  double bv = INFINITY;
  size_t bi = (size_t)-1;
  for (size_t i = 0; i < n; i++)
    if (sc[i] < bv || (sc[i] == bv && bi == (size_t)-1)) {
      bv = sc[i];
      bi = i;
    }
  uint64_t gi = (bi == (size_t)-1) ? s->best_id : lo + bi;
  const uint64_t inc = s->best_id;
  if (inc >= lo && inc < lo + n) { /* the incumbent keeps ties (:2432-2437) */
    const double inc_score = s->scores[inc];
    if (!(bv < inc_score)) {
      gi = inc;
      bv = inc_score;
    }
  }
  const int mine = gi >= lo && gi < lo + n;
  double sum = 0.0, m2 = 0.0;
  if (s->eps > 0) {
    m2 = orc_tiled_m2_merged(sc, n, &sum); /* tiles merged the way shards are */
  }
  rec[0] = bv;
  rec[1] = bits_dbl(gi);
  rec[2] = sum;
  rec[3] = m2;
  rec[4] = mine ? 1.0 : 0.0;
  for (size_t d = 0; d < s->D; d++) rec[ORC_DE_REC_HEADER + d] = mine ? s->cur[gi * s->D + d] : 0.0;
}

int orc_de_apply_records(orc_de_sync *s, const double *recs, int world, double *best_x) {
  if (s->done) return 1;
  const size_t stride = ORC_DE_REC_HEADER + s->D;
  const uint64_t inc = s->best_id;
  int win = -1;
  double bv = INFINITY;
  uint64_t bi = inc;
  for (int r = 0; r < world; r++) {
    const double *rec = recs + (size_t)r * stride;
    if (rec[4] != 1.0) continue;
    const double v = rec[0];
    const uint64_t i = dbl_bits(rec[1]);
    /* lower value wins; on ties the incumbent, then the lower global index */
    const int better = win < 0 || v < bv || (v == bv && bi != inc && (i == inc || i < bi));
    if (better) {
      bv = v;
      bi = i;
      win = r;
    }
  }
  const int not_updated = (bi == inc);
  s->val_no_change = not_updated ? s->val_no_change + 1 : 0; /* :2439 */
  s->best_id = bi;
  if (win >= 0 && best_x) memcpy(best_x, recs + (size_t)win * stride + ORC_DE_REC_HEADER, s->D * sizeof(double));
  /* std_err is only evaluated when it can decide something (eps > 0): for
   * eps <= 0 or NaN the test `std_err < eps` (:2443) is false for every value
   * std_err can take (>= 0 or NaN). */
  double se = NAN;
  if (s->eps > 0) {
    const double n_r = (double)(s->pop / (size_t)world);
    double tot = 0.0;
    for (int r = 0; r < world; r++) tot = tot + recs[(size_t)r * stride + 2];
    const double gmean = tot / (double)s->pop;
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
    se = sqrt(m2 / (double)(s->pop - 1)); /* :2050-2051 */
  }
  s->std_err = se;
  if (s->iter >= s->max_iter || s->val_no_change >= s->best_val_no_change ||
      (s->eps > 0 && se < s->eps)) { /* :2441-2443 */
    s->done = 1;
    return 1;
  }
  return 0;
}

void orc_de_shard_generation(orc_de_sync *s, size_t lo, size_t n, int threads) {
  const uint64_t kg = orc_ctr_key(s->seed, s->iter + 1);
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    double *trial = (double *)malloc(s->D * sizeof(double));
#pragma omp for schedule(static)
    for (long a = (long)lo; a < (long)(lo + n); a++) sync_agent(s, kg, (size_t)a, trial);
    free(trial);
  }
}

void orc_de_commit(orc_de_sync *s) {
  double *t = s->cur;
  s->cur = s->nxt;
  s->nxt = t;
  s->fcalls += s->pop;
  s->iter++;
}

static void sync_step(orc_de_sync *s, int threads) {
  if (s->done) return;
  const int world = (int)s->n_shards;
  const size_t shard_n = s->pop / s->n_shards;
  const size_t stride = ORC_DE_REC_HEADER + s->D;
  double *recs = (double *)malloc((size_t)world * stride * sizeof(double));
  for (int r = 0; r < world; r++) orc_de_shard_record(s, (size_t)r * shard_n, shard_n, recs + (size_t)r * stride);
  const int done = orc_de_apply_records(s, recs, world, NULL);
  free(recs);
  if (done) return;
  orc_de_shard_generation(s, 0, s->pop, threads);
  orc_de_commit(s);
}

void orc_de_sync_step(orc_de_sync *s) { sync_step(s, 1); }
void orc_de_sync_step_omp(orc_de_sync *s, int threads) { sync_step(s, threads); }
