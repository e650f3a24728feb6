/* oracle/oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference algorithms on the hot path
 * (JSzitas/nlsolver, nlsolver.h / tinyqr.h). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (include/, nlsolver_amd/) never links or calls it.
 *
 * Parity status: PINNED. The serial-compat functions (orc_*_serial,
 * orc_xorshift_*, orc_splitmix_*) reproduce bit-for-bit the golden vectors in
 * the JSON files under tests/golden/, which were produced by running the unmodified reference
 * (oracle/_ref/ref_driver, built from /root/reference by oracle/Makefile) in
 * this container; see tests/golden/gen_golden.py and tests/test_oracle_golden.py.
 *
 * Two families of functions live here:
 *   1. "serial" — the reference's own (asynchronous, single RNG stream)
 *      algorithm, restated line by line with file:line citations.
 *   2. "sync"   — the synchronous, counter-RNG, fixed-reduction-tree form of
 *      the same algorithm that a GPU generation kernel can execute. The HIP
 *      kernels must match these bit-for-bit (selection indices, populations,
 *      scores). The relation sync <-> serial is algorithmic (same update rule
 *      per agent, different scheduling/RNG), checked by convergence tests.
 */
#ifndef NLSG_ORACLE_H_
#define NLSG_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG --- */
/* nlsolver.h:1263-1288 rng::splitmix (seed 12374563468 at :1265). */
#define ORC_SPLITMIX_SEED 12374563468ull
uint64_t orc_splitmix_next(uint64_t *state); /* yield_init(), :1273-1278 */
/* nlsolver.h:1343-1381 rng::xorshift<double> (xorshift128+, shifts 23/18/5). */
typedef struct {
  uint64_t x[2];
} orc_xorshift;
void orc_xorshift_init(orc_xorshift *g);   /* ctor :1345-1349 */
double orc_xorshift_next(orc_xorshift *g); /* yield() :1350-1361 */

/* Counter-based generator used by every "sync" restatement and by the HIP
 * kernels: three nested splitmix64 steps (each level is "output #(n+1) of a
 * splitmix64 stream seeded with the parent key", i.e. nlsolver.h:1273-1278
 * applied as a random-access function). */
uint64_t orc_mix64(uint64_t z);
uint64_t orc_ctr_key(uint64_t parent, uint64_t index); /* mix(parent+G*(index+1)) */
double orc_u01(uint64_t bits); /* (double)bits * 2^-64, cf. nlsolver.h:1358 */

/* Deterministic log / cos (no libm): shared definition of the transcendental
 * arithmetic of the synchronous restatements and the HIP kernels. */
double orc_log(double x);
double orc_cos(double y);
double orc_cos_2pi(double x); /* cos(2 pi x), the device's Rastrigin cosine (nlsg_math.h) */
double orc_log_unit(double x); /* log for x in [2^-64, 1]: det_rnorm's table-driven logarithm */
double orc_cos_unit(double y); /* cos for y in [0, 6.3]: det_rnorm's one-polynomial cosine */
double orc_rnorm(uint64_t z1); /* one normal variate from one 64-bit draw (nlsg_math.h det_rnorm) */
void orc_probe_math(int fn, const uint64_t *in, uint64_t *out, size_t n); /* cf. nlsg_probe_math */

/* ------------------------------------------------------------ objectives --- */
enum {
  ORC_OBJ_ROSENBROCK = 0,      /* example.cpp:41-48 generalised to N-D chain  */
  ORC_OBJ_SPHERE = 1,          /* test_functions.h:52-57                      */
  ORC_OBJ_STYBLINSKI_TANG = 2, /* test_functions.h:249-260 (x^4 as products)  */
  ORC_OBJ_RASTRIGIN = 3        /* test_functions.h:69-78, N-D                 */
};
/* Sequential left-to-right evaluation = "the reference arithmetic" of a user
 * functor written the obvious way. */
double orc_objective_seq(int obj, const double *x, size_t D);
/* Same terms, summed in the device's fixed tree: element e belongs to lane
 * (e % 128) / 2 of chunk e / 128; a lane adds its terms in increasing element
 * order; then a 64-lane xor butterfly (32,16,8,4,2,1). */
double orc_objective_tree(int obj, const double *x, size_t D);

/* ---------------------------------------------------------------- std_err --- */
/* nlsolver.h:2037-2052: two serial passes, pow(.,2), divide by n-1. */
double orc_std_err_serial(const double *x, size_t n);
/* Fixed tree used on device: 256-thread block tree over tiles of 1024, then the
 * same block tree over the tile partials. */
double orc_block_tree_sum(const double *v, size_t n);
double orc_tiled_sum(const double *v, size_t n);
double orc_tiled_sumsq_dev(const double *v, size_t n, double mean);
double orc_tiled_m2_merged(const double *v, size_t n, double *sum_out);
double orc_std_err_tree(const double *x, size_t n);

/* --------------------------------------------------------------------- DE --- */
/* Evaluation recorder shared by the serial restatements. */
typedef struct {
  double *xs; /* capacity * D */
  double *fs; /* capacity */
  size_t capacity, count, D;
} orc_eval_log;

typedef struct {
  double f_value;
  uint64_t iteration, function_calls_used, gradient_evals_used, hessian_evals_used;
} orc_status; /* nlsolver.h:2054-2097 */

/* DE::solve<minimize> (nlsolver.h:2414-2476), strategy 0 = best, 1 = random
 * (enum order of nlsolver.h:2377). `gen` is the caller's generator and is
 * advanced exactly as the reference advances it. x is in/out. */
/* The reference's donor pick (generate_indices, nlsolver.h:2331-2355) and proposal
 * (propose_new_agent, :2357-2375) — the code orc_de_serial runs — for ONE agent, fed with given
 * draws; lets a test show that the synchronous restatement's donors / forced dimension / trial
 * are what the reference's logic makes of the same draws. */
size_t orc_de_serial_proposal_from_draws(const double *agents, size_t pop, size_t D, size_t fixed,
                                         double CR, double F, const double *donor_draws,
                                         size_t n_donor, const double *cross_draws, size_t *ids_out,
                                         size_t *donor_used, double *proposal);
orc_status orc_de_serial(int obj, int minimize, int strategy, double *x, size_t D,
                         orc_xorshift *gen, double CR, double F, double eps,
                         size_t pop, size_t max_iter, size_t best_val_no_change,
                         orc_eval_log *log);

/* Synchronous restatement (what the GPU executes). All state is caller-owned. */
typedef struct {
  int obj, minimize, strategy;
  size_t pop, D;      /* global population */
  size_t n_shards;    /* island model: donors are drawn inside the agent's shard */
  double CR, F, eps;
  size_t max_iter, best_val_no_change;
  uint64_t seed;
  /* state */
  double *cur, *nxt;  /* pop*D each, row-major */
  double *scores;     /* pop */
  uint64_t best_id;
  uint64_t iter, val_no_change, fcalls;
  int done;
  double std_err;
  /* optional trace of the last generation: per agent r1,r2,r3,jrand,accept */
  uint64_t *trace; /* pop*5 or NULL */
} orc_de_sync;

void orc_de_sync_init(orc_de_sync *s, const double *x0); /* generation 0 */
/* The turn in the pieces the sharded (multi-GPU) path executes: per-shard record
 * -> [all-gather] -> apply (global best, counters, stop tests) -> per-shard
 * generation -> commit. orc_de_sync_step is exactly this with all shards local. */
#define ORC_DE_REC_HEADER 5
void orc_de_shard_record(const orc_de_sync *s, size_t lo, size_t n, double *rec);
int orc_de_apply_records(orc_de_sync *s, const double *recs, int world, double *best_x);
void orc_de_shard_generation(orc_de_sync *s, size_t lo, size_t n, int threads);
void orc_de_commit(orc_de_sync *s);
/* One reference loop turn: best scan + stop tests (nlsolver.h:2429-2447), then
 * if not done one synchronous generation (2449-2472) and iter++. */
void orc_de_sync_step(orc_de_sync *s);
/* Multi-threaded variant of the generation (OpenMP over agents); identical
 * results. Used only for the CPU baseline timing. */
void orc_de_sync_step_omp(orc_de_sync *s, int threads);

/* -------------------------------------------------------------------- PSO --- */
/* PSO::solve and helpers (nlsolver.h:2479-2742), literal (incl. the sentinels B8,
 * the best_index rule B9 and, for Vanilla, B7). type 0 = Vanilla, 1 = Accelerated. */
/* update_positions (Accelerated, nlsolver.h:2687-2699) + threshold_positions (2701-2715) of one
 * particle from given normal variates: the code both PSO restatements share. */
void orc_pso_accel_move_from_normals(double *pos, const double *normals, const double *gbest,
                                     const double *lower, const double *upper, size_t D,
                                     double inertia, double cog, double soc, int bounded);
orc_status orc_pso_serial(int obj, int minimize, int type, int bounded, double *x, size_t D,
                          const double *lower, const double *upper, orc_xorshift *gen,
                          double inertia, double cog, double soc, size_t n, size_t max_iter,
                          size_t best_val_no_change, double eps, orc_eval_log *log);

/* Synchronous restatement. Deviations from the literal algorithm, each documented in
 * counter RNG; +inf best sentinels (B8); val_no_change driven by
 * update_happened (B9); Vanilla uses the intended pbest/gbest terms (B7); log/cos
 * are orc_log/orc_cos. */
typedef struct {
  int obj, minimize, type, bounded;
  size_t n, D, n_shards;
  double inertia0, cog, soc, eps;
  size_t max_iter, best_val_no_change;
  uint64_t seed;
  const double *lower, *upper; /* D each (init range; thresholds when bounded) */
  double *pos, *vel, *pbest_pos; /* n*D; vel/pbest_pos only for Vanilla */
  double *pbest_val, *cur_val;   /* n */
  double *gbest_x;               /* D */
  double gbest_val;
  uint64_t gbest_idx;
  uint64_t iter, val_no_change, fevals;
  int done;
  double std_err, inertia;
} orc_pso_sync;
void orc_pso_sync_init(orc_pso_sync *s);
void orc_pso_sync_step(orc_pso_sync *s, int threads);
void orc_pso_shard_record(const orc_pso_sync *s, size_t lo, size_t n, double *rec);
int orc_pso_apply_records(orc_pso_sync *s, const double *recs, int world);
void orc_pso_shard_move(orc_pso_sync *s, size_t lo, size_t n, int threads);
void orc_pso_commit(orc_pso_sync *s);

/* ------------------------------------------------------------------- BFGS --- */
/* The convex quadratic of SURVEY.md §8c G6 (diag + rank-1):
 *   f(x) = 1/2 sum d_i x_i^2 + 1/2 c (sum x)^2 - sum b_i x_i. */
typedef struct {
  const double *d, *b; /* n each */
  double c;
} orc_quad;
typedef struct {
  uint64_t f_calls, g_calls;
  double *f_log; /* optional: every objective value in call order */
  size_t f_cap, f_count;
  double *H_out; /* optional: n*n, the inverse Hessian after the last update */
} orc_bfgs_counters;
/* BFGS::solve<true> (nlsolver.h:3196-3285) with more_thuente_search (1527-1891) and
 * update_inverse_hessian (3130-3168). tree = 0: the reference's sequential sums;
 * tree = 1: the HIP kernel's lane tree; tree = 2: the symmetric restatement (oracle_bfgs.c). */
orc_status orc_bfgs_quad(const orc_quad *q, double *x, size_t n, size_t max_iter, double grad_eps,
                         double alpha, int tree, orc_bfgs_counters *cnt);
/* the same on a built-in objective with the reference's default gradient, fin_diff
 * (finite_difference_gradient<.,.,1>, nlsolver.h:1385-1413): 4 n probes per gradient, each
 * counted as a function call (3218-3224). */
orc_status orc_bfgs_fd(int obj, double *x, size_t n, size_t max_iter, double grad_eps, double alpha,
                       int tree, orc_bfgs_counters *cnt);
void orc_update_inverse_hessian(double *H, const double *s, const double *y, double *t, double rho,
                                size_t n, int tree);

/* --------------------------------------------------------------------- LM --- */
double orc_exp(double x);  /* deterministic (no libm), shared with the HIP kernels */
double orc_tanh(double x);
void orc_cholesky(double *A, size_t n);                        /* nlsolver.h:251-269 */
void orc_update_with_hessian(double *update, double *hess, const double *grad, size_t n); /* :310-330 */
void orc_qr_decomposition(const double *X, size_t n, size_t p, double tol, double *Q, double *R);
void orc_tinyqr_lm(const double *X, const double *y, size_t n, size_t p, double *beta);
void orc_tinyqr_lm_tol(const double *X, const double *y, size_t n, size_t p, double *beta, int order,
                       double tol);
void orc_tinyqr_lm_order(const double *X, const double *y, size_t n, size_t p, double *beta,
                         int order);
void orc_lm_make_tanh_problem(uint64_t seed, uint64_t problem, size_t m, size_t n, double *A,
                              double *y, double *theta0);
typedef struct {
  int kind; /* 0: r_i = y_i - p0 exp(p1 t_i) (n = 2); 1: r_i = y_i - tanh(sum_j A_ij theta_j) */
  size_t m, n;
  const double *A, *y, *t;
} orc_nlls;
orc_status orc_lm_solve(const orc_nlls *q, double *x, double *lambda, double up, double down,
                        size_t max_iter, double f_delta, int solver, int order, double *f_log,
                        size_t f_cap);
/* the reference's default LevenbergMarquardt<Callable>(f) (Grad = fin_diff, Hess = fin_diff_h,
 * nlsolver.h:1385-1517, 3428-3545) on a built-in objective; every probe is a counted call */
orc_status orc_lm_fd(int obj, double *x, size_t n, double *lambda, double up, double down,
                     size_t max_iter, double f_delta, int order, double *f_log, size_t f_cap);

/* --------------------------------------------------------------------- NM --- */
/* NelderMead::solve (nlsolver.h:2166-2299) and the minimize/maximize wrappers with
 * restarts (2127-2163). *eps is the solver's mutable member (B2). order: 0 reference
 * arithmetic, 1 kernel trees for the objective and std_err. */
orc_status orc_nm_solve(int obj, int minimize, int bound, double *x, size_t n, const double *upper,
                        const double *lower, double step, double alpha, double gamma, double rho,
                        double sigma, double *eps, size_t max_iter, size_t no_change_best_tol,
                        int order, orc_eval_log *log);
orc_status orc_nm_run(int obj, int minimize, int bound, double *x, size_t n, const double *upper,
                      const double *lower, double step, double alpha, double gamma, double rho,
                      double sigma, double *eps, size_t max_iter, size_t no_change_best_tol,
                      size_t restarts, int order, orc_eval_log *log);

/* ---- simulated annealing (oracle_sann.c), SANN::solve nlsolver.h:2777-2814 ---- */
orc_status orc_sann_serial(int obj, int minimize, double *x, size_t D, orc_xorshift *gen,
                           size_t max_iter, size_t temp_iter, double temp_max, double *f_log,
                           size_t f_cap);
orc_status orc_sann_sync(int obj, int minimize, double *x, size_t D, uint64_t seed, uint64_t chain,
                         size_t max_iter, size_t temp_iter, double temp_max, double *f_log,
                         size_t f_cap);

/* ---- Nelder-Mead / PSO hybrid (oracle_nmpso.c), NelderMeadPSO nlsolver.h:3546-3920 ---- */
orc_status orc_nmpso_serial(int obj, int minimize, int bound, double *x, size_t n,
                            const double *upper, const double *lower, orc_xorshift *gen, double alpha,
                            double gamma, double rho, double sigma, double inertia, double cog,
                            double soc, double eps, size_t max_iter, size_t no_change_best_iter,
                            double *f_log, size_t f_cap);
size_t orc_nmpso_last_shrinks(void); /* shrink steps of the most recent orc_nmpso_* run */
orc_status orc_nmpso_sync(int obj, int minimize, int bound, double *x, size_t n, const double *upper,
                          const double *lower, uint64_t seed, uint64_t instance, double alpha,
                          double gamma, double rho, double sigma, double inertia, double cog,
                          double soc, double eps, size_t max_iter, size_t no_change_best_iter,
                          double *f_log, size_t f_cap);

#ifdef __cplusplus
}
#endif
#endif /* NLSG_ORACLE_H_ */
