/* include/nlsg_c_api.h — C-ABI of the MI355X-native nlsolver iteration engine.
 *
 * Drop-in boundary for the hot path of JSzitas/nlsolver. The
 * reference has no FFI layer: its boundary is the C++ template API
 * `Solver<Callable,[RNG,]T,...>(f,[gen,]params...).minimize(std::vector<T>&)`
 * (nlsolver.h:2390-2410 DE, 2522-2591 PSO, 2110-2165 NelderMead, 3181-3196
 * BFGS, 3443-3465 LevenbergMarquardt). The header-only host API in
 * include/nlsolver_mi/nlsolver.h keeps that surface and forwards the
 * population/batch loops to the entry points declared here; each entry point
 * names the reference loop it replaces.
 *
 * Conventions
 *  - plain C: pointers, sizes, POD structs; no C++/torch types.
 *  - every function returns an nlsg_err (0 = ok) and never throws; the text of
 *    the last error on the calling thread is nlsg_last_error().
 *  - `*_host` pointers are host memory borrowed for the duration of the call;
 *    `*_dev` pointers are device memory on the engine's device. Buffers an
 *    engine allocates are owned by the engine and freed by *_destroy().
 *  - one engine handle per host thread; calls on one handle are serialised by
 *    the caller (the reference's solver objects are not thread-safe either,
 *    nlsolver.h:2104/2189, 3436/3541).
 *  - work is enqueued on the engine's HIP stream (cfg.stream, or a private
 *    stream when NULL) and is asynchronous unless the function says it syncs.
 */
#ifndef NLSG_C_API_H_
#define NLSG_C_API_H_

#ifndef __HIPCC_RTC__ /* the run-time compiler of user objectives brings its own fixed-width types */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define NLSG_ABI_VERSION 1

typedef enum {
  NLSG_OK = 0,
  NLSG_ERR_INVALID_ARG = 1,  /* bad pointer / size / enum                     */
  NLSG_ERR_UNSUPPORTED = 2,  /* valid request the device path does not cover  */
  NLSG_ERR_NO_DEVICE = 3,    /* no usable gfx950 device                       */
  NLSG_ERR_HIP = 4,          /* a HIP runtime call failed                     */
  NLSG_ERR_OOM = 5,          /* device allocation failed                      */
  NLSG_ERR_STATE = 6         /* call order violated (e.g. step before init)   */
} nlsg_err;

/* Built-in device objectives (the reference calls an arbitrary host functor in
 * the inner loop, nlsolver.h:2463; a device kernel cannot). */
typedef enum {
  NLSG_OBJ_ROSENBROCK = 0,      /* example.cpp:41-48, N-D chain                */
  NLSG_OBJ_SPHERE = 1,          /* test_functions.h:52-57                      */
  NLSG_OBJ_STYBLINSKI_TANG = 2, /* test_functions.h:249-260                    */
  NLSG_OBJ_RASTRIGIN = 3,       /* test_functions.h:69-78                      */
  NLSG_OBJ_CUSTOM = 64          /* user source compiled at run time, nlsg_custom_objective */
} nlsg_objective;

/* A user objective of the form  f(x) = finish( sum_i term(x_i, x_{i+1}), D )  — the shape of the
 * built-in ones — given as C++ function BODIES that are compiled for the device at engine
 * creation (hiprtc; SURVEY.md §8f N3: lifts the "built-in objectives only" restriction for the
 * reference's functor contract, README.md:127-136). Available in the bodies: double xi, xn
 * (term; xn = x_{i+1}, only meaningful when chain != 0) / double s, uint64_t D (finish), and
 * the HIP device math library. Sums run over i < D - 1 when chain != 0, else i < D, in the
 * kernels' fixed lane-tree order; arithmetic is compiled without fp contraction.
 *   term_body   e.g. "double t1 = 1 - xi; double t2 = xn - xi * xi; return t1 * t1 + 100 * t2 * t2;"
 *   finish_body e.g. "return s;"  (NULL = that)
 *
 * chain == NLSG_CUSTOM_VECTOR: the WHOLE-VECTOR form for objectives that are not sums of
 * per-coordinate / neighbour terms (most of the reference's test set, test_functions.h:52-330:
 * Ackley, Beale, Himmelblau, Goldstein-Price, Shekel ...). term_body is then the body of
 *     double f(const X &x, uint64_t D)
 * with  x(i)      coordinate i, for an index that is the same in every lane (a literal, a loop
 *                 counter),
 *       x.size()  D,
 *       x.sum(g)  the sum over all coordinates of g(x_i, i) in the kernels' lane-tree order, g any
 *                 callable double(double xi, uint64_t i);
 * every lane runs the body and must return the same value. finish_body is ignored.
 *   e.g. Himmelblau: "double a = x(0) * x(0) + x(1) - 11, b = x(0) + x(1) * x(1) - 7; return a * a + b * b;" */
#define NLSG_CUSTOM_TERMS 0   /* f = finish(sum_i term(x_i))            */
#define NLSG_CUSTOM_CHAIN 1   /* f = finish(sum_{i<D-1} term(x_i, x_{i+1})) */
#define NLSG_CUSTOM_VECTOR 2  /* f = body(x, D)                          */
typedef struct {
  const char *term_body;
  const char *finish_body;
  int32_t chain;         /* NLSG_CUSTOM_TERMS / _CHAIN / _VECTOR */
  int32_t reserved;
} nlsg_custom_objective;

/* enum RecombinationStrategy { best, random } — nlsolver.h:2377 (same order). */
typedef enum { NLSG_DE_BEST = 0, NLSG_DE_RANDOM = 1 } nlsg_de_strategy;

const char *nlsg_last_error(void);
int nlsg_abi_version(void);
/* Number of visible HIP devices whose arch is gfx950 (0 if none / no runtime). */
int nlsg_device_count(void);
/* Host wall-clock (ms) of the calling thread's most recent DE / PSO / BFGS / LM calls:
 * ms_out6 = {create, upload (nlsg_lm_set_data), init, iterate, read-back, destroy}. The last four
 * are the phases of nlsg_*_minimize — the reference's minimize() (nlsolver.h:2404, 2553, 3188,
 * 3457) is one call; this is where its time goes once the loops run on the device. Host-side laps
 * without extra synchronisation (bench.py --workload tts reports them). */
int nlsg_call_timing(double *ms_out6);
/* Engines recycle their device blocks and streams through a per-process cache (what one
 * minimize() releases the next one takes: the reference's one-call-per-solve API would otherwise
 * pay a dozen hipMalloc / hipFree pairs per call). At most $NLSG_POOL_BYTES (default 40 GiB; 0
 * turns the cache off) sit idle per device. nlsg_release_cached frees all of it now;
 * nlsg_cached_bytes reports how much is parked. */
int nlsg_release_cached(void);
uint64_t nlsg_cached_bytes(void);

/* The device's deterministic math primitives (the log / cos of rnorm nlsolver.h:2479-2485, the
 * exp / tanh of the NLLS model, the cosine of Rastrigin test_functions.h:74-76) evaluated on n
 * caller-chosen arguments: in / out are the IEEE-754 bit patterns of the doubles (for
 * NLSG_PROBE_U01 and NLSG_PROBE_RNORM the input is the 64-bit draw itself). Exists so that the
 * bit-for-bit agreement with the CPU restatement can be tested on the primitives directly. */
typedef enum {
  NLSG_PROBE_LOG = 0,
  NLSG_PROBE_COS = 1,
  NLSG_PROBE_EXP = 2,
  NLSG_PROBE_TANH = 3,
  NLSG_PROBE_COS_2PI = 4,
  NLSG_PROBE_U01 = 5,   /* draw -> uniform in [0, 1] (xorshift::yield's conversion, :1358) */
  NLSG_PROBE_RNORM = 6  /* draw -> normal variate (both uniforms from the one draw)       */
} nlsg_probe_fn;
int nlsg_probe_math(int32_t fn, const uint64_t *in_host, uint64_t *out_host, uint64_t n,
                    int32_t device);

/* solver_status<T> (nlsolver.h:2054-2097) plus engine bookkeeping. */
typedef struct {
  double f_value;               /* best objective (sign as the solver minimises) */
  uint64_t iteration;           /* completed generations / iterations            */
  uint64_t function_calls_used;
  uint64_t gradient_evals_used;
  uint64_t hessian_evals_used;
  uint64_t best_index;          /* global row index of the best agent            */
  uint64_t val_no_change;       /* nlsolver.h:2439 counter                        */
  double std_err;               /* last std_err(scores) (NaN if not evaluated)   */
  int32_t done;                 /* a stop test of nlsolver.h:2441-2443 fired     */
  int32_t reserved;
} nlsg_status;

/* ========================================================================== */
/* Differential Evolution — replaces DE::solve (nlsolver.h:2414-2476) and its */
/* helpers init_agents/generate_indices/propose_new_agent (2302-2375).         */
/* ========================================================================== */
typedef struct nlsg_de nlsg_de; /* opaque engine handle */

typedef struct {
  uint32_t struct_size; /* sizeof(nlsg_de_config), for ABI evolution          */
  int32_t device;       /* HIP device ordinal                                  */
  void *stream;         /* hipStream_t to enqueue on; NULL = private stream.   */
                        /* For the null stream pass hipStreamLegacy ((void*)1) */
  int32_t objective;    /* nlsg_objective                                      */
  int32_t minimize;     /* 1 = minimize(), 0 = maximize() (nlsolver.h:2404-10) */
  int32_t strategy;     /* nlsg_de_strategy                                    */
  int32_t trace;        /* 1 = keep (r1,r2,r3,jrand,accept) of the last gen    */
  uint64_t pop;         /* GLOBAL population size (ctor arg pop_size, :2393)   */
  uint64_t dim;         /* x.size()                                            */
  uint64_t shard_lo;    /* first global agent owned by this engine             */
  uint64_t shard_n;     /* agents owned (== pop on one GPU); donors are drawn  */
                        /* inside the shard (island model)      */
  double CR, F, eps;    /* crossover_prob, differential_weight, eps (:2391-92) */
  uint64_t max_iter;    /* :2393                                               */
  uint64_t best_val_no_change; /* :2394                                        */
  uint64_t seed;        /* key of the counter-based generator                  */
} nlsg_de_config;

int nlsg_de_create(const nlsg_de_config *cfg, nlsg_de **out);
/* cfg->objective == NLSG_OBJ_CUSTOM: compiles `obj` for this dim and builds the engine around it.
 * nlsg_rtc_load(path) chooses the hiprtc shared object (NULL / "" / never called: "libhiprtc.so"
 * on the loader path; a process that already holds a HIP runtime should pass that runtime's own). */
int nlsg_rtc_load(const char *hiprtc_path);
int nlsg_de_create_custom(const nlsg_de_config *cfg, const nlsg_custom_objective *obj, nlsg_de **out);
int nlsg_de_destroy(nlsg_de *e);

/* init_agents + initial scoring (nlsolver.h:2315-2323, 2423-2425):
 * agents[a][i] = (u - 0.5) * x0[i]; scores[a] = +-f(agents[a]). Asynchronous. */
int nlsg_de_init(nlsg_de *e, const double *x0_host);

/* Enqueue `turns` turns of the reference's while(true) loop (2429-2475): best
 * scan + stop tests, then one generation (mutation, crossover, evaluation,
 * selection) unless a stop test fired. Once `done` is set further turns are
 * no-ops on device. Asynchronous; no host synchronisation. */
int nlsg_de_step(nlsg_de *e, uint64_t turns);

/* Full solve(): init, turns until done, x_inout <- best agent (2441-2447).
 * Synchronises. `poll_every` turns are enqueued between host checks of the
 * device stop flag (0 = engine default). */
int nlsg_de_minimize(nlsg_de *e, double *x_inout_host, uint64_t poll_every,
                     nlsg_status *out);

/* Synchronise the stream and read the device-resident solver state. */
int nlsg_de_status(nlsg_de *e, nlsg_status *out);
/* Best agent (x has dim entries), its score and global index. Synchronises. */
int nlsg_de_best(nlsg_de *e, double *x_host, double *f, uint64_t *index);
/* Parity-test access: current population shard (shard_n*dim, row-major),
 * scores (shard_n) and the last generation's trace (shard_n*5 u64:
 * r1,r2,r3,jrand,accept; requires cfg.trace). NULL pointers are skipped. */
int nlsg_de_download(nlsg_de *e, double *pop_host, double *scores_host,
                     uint64_t *trace_host);
/* Replace the current shard population and scores (tests, checkpoint/resume). */
int nlsg_de_upload(nlsg_de *e, const double *pop_host, const double *scores_host);

/* Measurement aid for bench.py: launches ONLY the generation kernel `launches`
 * times back to back on the engine's stream (buffers ping-pong, solver state is
 * restored afterwards) bracketed by two hipEvents; returns total milliseconds. */
int nlsg_de_time_generation_kernel(nlsg_de *e, uint32_t launches, float *ms_total);
/* Same bracket around `turns` full turns (scan + generation). */
int nlsg_de_time_turns(nlsg_de *e, uint64_t turns, float *ms_total);

/* ---- multi-GPU exchange (one small record per rank per generation) -------- */
/* A turn on a sharded population is split so the host can run ONE collective
 * between the two halves (RCCL all-gather through torch.distributed):
 *   nlsg_de_turn_begin  : local best scan -> record in `send_dev`
 *   <all-gather of record_doubles() doubles per rank into `gathered_dev`>
 *   nlsg_de_turn_end    : global best (lowest f, incumbent keeps ties, then
 *                         lowest global index), stop tests, generation.
 * Record layout (doubles): [f_best, idx_best(as u64 bits), sum_scores, m2, valid,
 *                           x_best[0..dim)]. */
uint64_t nlsg_de_record_doubles(const nlsg_de *e);
int nlsg_de_turn_begin(nlsg_de *e, double *send_dev);
int nlsg_de_turn_end(nlsg_de *e, const double *gathered_dev, int32_t world);
/* The two halves of nlsg_de_turn_end. For strategy random (nlsg_de_can_speculate) the
 * generation only depends on the exchange through the stop flag and writes the OTHER
 * population / score buffers, so the host may order a turn as
 *   turn_begin -> start all-gather (async) -> turn_generation -> wait -> turn_finalize
 * hiding the collective behind the generation; if the finaliser fires a stop test, the
 * speculative generation's output is never adopted. Strategy best must keep
 *   turn_begin -> all-gather -> turn_finalize -> turn_generation. */
int nlsg_de_turn_finalize(nlsg_de *e, const double *gathered_dev, int32_t world);
int nlsg_de_turn_generation(nlsg_de *e);
int nlsg_de_can_speculate(const nlsg_de *e);

/* The same turns with the exchange issued by the library itself (RCCL all-gather on a second
 * stream, hidden behind the generation for strategy random): no host round trip per turn.
 *   nlsg_comm_load(path)       resolve RCCL from the shared object the process already uses
 *                              (PyTorch's librccl.so; NULL/"" = "librccl.so" on the loader path)
 *   nlsg_comm_unique_id(id)    128-byte ncclUniqueId; one rank calls it, the host broadcasts it
 *   nlsg_de_comm_attach        collective: every rank, same id; the engine's shard must be
 *                              rank * shard_n .. of a population of world * shard_n
 *   nlsg_de_step_sharded       `turns` turns of while(true) (nlsolver.h:2429-2475) on the
 *                              sharded population; results equal turn_begin/.../turn_end */
int nlsg_comm_load(const char *rccl_path);
int nlsg_comm_unique_id(unsigned char *id_out_128);
int nlsg_de_comm_attach(nlsg_de *e, const unsigned char *unique_id_128, int32_t world, int32_t rank);
int nlsg_de_step_sharded(nlsg_de *e, uint64_t turns);
/* Size of the attached communicator and this engine's rank in it, read back from RCCL
 * (ncclCommCount / ncclCommUserRank) — what a run really used, not what it was asked for. */
int nlsg_de_comm_ranks(nlsg_de *e, int32_t *world_out, int32_t *rank_out);

/* ========================================================================== */
/* Particle Swarm Optimisation — replaces PSO::solve (nlsolver.h:2593-2624),   */
/* init_solver_state (2626-2657), update_velocities (2658-2677),               */
/* update_positions (2678-2700), threshold_positions (2701-2715) and            */
/* update_best_positions (2716-2741).                                           */
/* Accelerated is the parity target; Vanilla runs the intended pbest/gbest      */
/* update (the reference's has a zero cognitive term and an out-of-bounds       */
/* read). Best sentinels are +inf (B8), the no-change counter is  */
/* driven by "an update happened" (B9). See DESIGN.md.                          */
/* ========================================================================== */
typedef struct nlsg_pso nlsg_pso;

/* enum PSOType { Vanilla, Accelerated } — nlsolver.h:2496 (same order). */
typedef enum { NLSG_PSO_VANILLA = 0, NLSG_PSO_ACCELERATED = 1 } nlsg_pso_type;

typedef struct {
  uint32_t struct_size;
  int32_t device;
  void *stream;          /* as nlsg_de_config.stream                              */
  int32_t objective;     /* nlsg_objective                                        */
  int32_t minimize;      /* 1 = minimize(), 0 = maximize()                        */
  int32_t type;          /* nlsg_pso_type                                         */
  int32_t bounded;       /* 1 = (x, lower, upper) overloads: threshold_positions  */
                         /* 0 = minimize(x): bounds only seed the initial swarm   */
  uint64_t n_particles;  /* GLOBAL swarm size (ctor arg, nlsolver.h:2525)         */
  uint64_t dim;
  uint64_t shard_lo, shard_n; /* particles owned by this engine                   */
  double inertia, cognitive, social, eps; /* ctor args :2523-2526                 */
  uint64_t max_iter, best_val_no_change;
  uint64_t seed;
} nlsg_pso_config;

int nlsg_pso_create(const nlsg_pso_config *cfg, nlsg_pso **out);
/* cfg->objective == NLSG_OBJ_CUSTOM, as nlsg_de_create_custom */
int nlsg_pso_create_custom(const nlsg_pso_config *cfg, const nlsg_custom_objective *obj, nlsg_pso **out);
int nlsg_pso_destroy(nlsg_pso *e);
/* init_solver_state + the first update_best_positions' evaluations (2626-2657, 2595). */
int nlsg_pso_init(nlsg_pso *e, const double *lower_host, const double *upper_host);
/* `turns` turns of: best update of the last evaluation + stop tests (2599-2605), then
 * velocity/position update, thresholding and evaluation (2606-2621). Asynchronous. */
int nlsg_pso_step(nlsg_pso *e, uint64_t turns);
/* Whole solve(): init with the given bounds, turns until done, x_out <- swarm best. */
int nlsg_pso_minimize(nlsg_pso *e, double *x_out_host, const double *lower_host,
                      const double *upper_host, uint64_t poll_every, nlsg_status *out);
int nlsg_pso_status(nlsg_pso *e, nlsg_status *out);
int nlsg_pso_best(nlsg_pso *e, double *x_host, double *f, uint64_t *index);
/* Parity-test access (shard): positions, velocities (Vanilla), personal-best values,
 * values of the last evaluation. NULL pointers are skipped. */
int nlsg_pso_download(nlsg_pso *e, double *pos_host, double *vel_host, double *pbest_val_host,
                      double *cur_val_host);
int nlsg_pso_time_move_kernel(nlsg_pso *e, uint32_t launches, float *ms_total);
/* sharded turn, as nlsg_de_turn_begin / nlsg_de_turn_end */
uint64_t nlsg_pso_record_doubles(const nlsg_pso *e);
int nlsg_pso_turn_begin(nlsg_pso *e, double *send_dev);
int nlsg_pso_turn_end(nlsg_pso *e, const double *gathered_dev, int32_t world);
/* the same turns ordered by the library (see nlsg_de_comm_attach / nlsg_de_step_sharded) */
int nlsg_pso_comm_attach(nlsg_pso *e, const unsigned char *unique_id_128, int32_t world, int32_t rank);
int nlsg_pso_step_sharded(nlsg_pso *e, uint64_t turns);
int nlsg_pso_comm_ranks(nlsg_pso *e, int32_t *world_out, int32_t *rank_out);

/* ========================================================================== */
/* Batched BFGS — replaces BFGS::solve (nlsolver.h:3196-3285), the More-Thuente */
/* search it calls (cvsrch/cstep 1527-1793, more_thuente_search 1880-1891) and  */
/* update_inverse_hessian (3130-3168), for `batch` independent starts of one    */
/* objective (the reference solves one start per minimize() call).              */
/* ========================================================================== */
typedef struct nlsg_bfgs nlsg_bfgs;

/* Objectives with an analytic gradient functor on device. */
typedef enum {
  /* f(x) = 1/2 sum d_i x_i^2 + 1/2 c (sum x)^2 - sum b_i x_i   */
  NLSG_OBJ_QUAD_DIAG_RANK1 = 16
} nlsg_grad_objective;

/* nlsg_bfgs_config.flags.
 * NLSG_BFGS_SYMMETRIC: the rank-2 update restated so that H stays BITWISE symmetric,
 *   H[j][i] -= rho ((s[i] t[j] + t[i] s[j]) + denom (s[i] s[j]))
 * (the reference's last term associates as (denom s[i]) s[j], nlsolver.h:3156-3163, which is the
 * only thing that makes its H[j][i] and H[i][j] differ in the last bit). Only the upper 128 x 128
 * blocks of H are kept and streamed: 56 % of the memory and of the traffic per iteration at
 * dim = 1024. Same algorithm, same counts on the reference's runs; values agree with the literal
 * arithmetic to rounding (f within 1e-12), not bit for bit. Default (0): the literal update. */
#define NLSG_BFGS_SYMMETRIC 1
/* NLSG_BFGS_REFERENCE_ORDER: every sum of the solve — the objective's terms, math::dot / norm
 * (nlsolver.h:58-99), the rows of H y and H g (3139-3142, 3248-3251) — is taken in INDEX order,
 * as the reference's sequential loops take it, instead of the kernels' lane tree. With it the
 * engine reproduces the reference's own runs bit for bit, the default finite-difference gradient
 * included (fin_diff divides differences of objective values by 12 eps, which turns the last bit
 * of a tree sum into 1e-8 .. 1e-6 of the result). Literal update only, objectives given by their
 * terms (not Rastrigin, not a whole-vector custom body). With the default gradient it is also the
 * FASTER mode — the 4 n probes share the base point's terms and prefix sums, a probe per lane —:
 * 9.4e6 against 2.6e6 iteration-problems/s at Rosenbrock-128D x 4096; with a gradient functor the H
 * passes (a lane per row) cost 1.5 x the tree kernels' time at dim = 1024. */
#define NLSG_BFGS_REFERENCE_ORDER 2

typedef struct {
  uint32_t struct_size;
  int32_t device;
  void *stream;
  int32_t objective;   /* nlsg_grad_objective (analytic gradient functor), or one of */
                       /* NLSG_OBJ_ROSENBROCK / SPHERE / STYBLINSKI_TANG / RASTRIGIN:  */
                       /* reference's default gradient fin_diff (nlsolver.h:         */
                       /* 1385-1413, 2849-2855), evaluated on the device              */
  int32_t flags;       /* 0, NLSG_BFGS_SYMMETRIC or NLSG_BFGS_REFERENCE_ORDER     */
  uint64_t batch;      /* independent problems                                    */
  uint64_t dim;        /* x.size() of each problem (<= 1024)                      */
  uint64_t max_iter;   /* ctor args of nlsolver.h:3181-3185                       */
  double grad_eps, alpha;
  double quad_c;       /* rank-1 weight of NLSG_OBJ_QUAD_DIAG_RANK1               */
} nlsg_bfgs_config;

/* diag_host / lin_host: d[dim], b[dim] of the quadratic (copied); NULL for the others. */
int nlsg_bfgs_create(const nlsg_bfgs_config *cfg, const double *diag_host,
                     const double *lin_host, nlsg_bfgs **out);
/* cfg->objective == NLSG_OBJ_CUSTOM: a user objective (nlsg_custom_objective, as for
 * nlsg_de_create_custom) minimised with the default finite-difference gradient; dim <= 256. */
int nlsg_bfgs_create_custom(const nlsg_bfgs_config *cfg, const nlsg_custom_objective *obj,
                            nlsg_bfgs **out);
int nlsg_bfgs_destroy(nlsg_bfgs *e);
/* x0_host: batch*dim start points (row-major). H = I, g = grad(x0) (3212, 3234). */
int nlsg_bfgs_init(nlsg_bfgs *e, const double *x0_host);
/* `iters` turns of the while(true) loop of 3238-3284 for every unfinished problem:
 * stop tests, direction, reset guard, More-Thuente search, s/y/rho, rank-2 update. */
int nlsg_bfgs_step(nlsg_bfgs *e, uint64_t iters);
/* Number of problems whose stop test has not fired yet. Synchronises. */
int nlsg_bfgs_unfinished(nlsg_bfgs *e, uint64_t *count);
/* Number of unfinished problems whose inverse Hessian is the identity right now (the start,
 * 3212, or the reset guard of 3253-3260 fired in the last turn): the H passes skip their reads
 * for those, so a throughput figure is a dense-H figure only while this is 0. Synchronises. */
int nlsg_bfgs_identity_count(nlsg_bfgs *e, uint64_t *count);
/* x (batch*dim) and per-problem status (batch). Synchronises. NULLs are skipped.
 * status.f_value is f(x) evaluated when the stop test fired (3243). */
int nlsg_bfgs_download(nlsg_bfgs *e, double *x_host, nlsg_status *status_host);
/* Parity-test access: gradient (batch*dim) and inverse Hessian (batch*dim*dim) */
int nlsg_bfgs_download_state(nlsg_bfgs *e, double *g_host, double *h_host);
/* init + steps until every problem is done + download. */
int nlsg_bfgs_minimize(nlsg_bfgs *e, double *x_inout_host, nlsg_status *status_host);
/* hipEvent bracket around `iters` turns: total ms and the share spent in the two
 * kernels that stream the inverse Hessians (t = H y; rank-2 update + next direction). */
int nlsg_bfgs_time_steps(nlsg_bfgs *e, uint64_t iters, float *ms_total, float *ms_hessian);

/* ========================================================================== */
/* Batched Levenberg-Marquardt NLLS — replaces LevenbergMarquardt::solve        */
/* (nlsolver.h:3465-3544) driven with Gauss-Newton functors (f = sum r^2,       */
/* Grad = 2 J^T r, Hess = 2 J^T J), math::get_update_with_hessian (251-330) and */
/* — solver = QR — tinyqr::lm on the damped matrix (tinyqr.h:253-310, 437-470), */
/* for `batch` independent problems of one residual model.                      */
/* ========================================================================== */
typedef struct nlsg_lm nlsg_lm;

typedef enum {
  /* r_i(theta) = y_i - tanh(sum_j A_ij theta_j)   */
  NLSG_OBJ_TANH_REGRESSION = 32
} nlsg_nlls_objective;
/* NLSG_LM_CHOLESKY_REFERENCE_ORDER: the class's own solve with the reference's literal arithmetic
 * — every probe of fin_diff / fin_diff_h sums its objective in INDEX order (the reference's
 * sequential loops) and math::cholesky / forwardsolve_inplace / backsolve_inplace_t round as the
 * reference does (separate multiply and add, nlsolver.h:251-294) — so that the default-functor
 * model reproduces the reference's own runs bit for bit (fin_diff_h divides differences of
 * objective values by 600 eps^2 = 9e-6: the last bit of a tree sum is worth 1e-9 .. 1e-6 of the
 * result). For the finite-difference model on objectives given by their terms (Rosenbrock / Sphere /
 * Styblinski-Tang, custom term bodies; any n the engine takes); also the FASTER evaluation there — a
 * probe per lane on the base point's shared terms: 1.18e7 against 6.7e6 iteration-problems/s at
 * Rosenbrock-16D x 8192, 5 x past 64 parameters. */
typedef enum { NLSG_LM_CHOLESKY = 0, NLSG_LM_QR = 1, NLSG_LM_CHOLESKY_REFERENCE_ORDER = 2 } nlsg_lm_solver;

typedef struct {
  uint32_t struct_size;
  int32_t device;
  void *stream;
  int32_t objective;      /* nlsg_nlls_objective (Gauss-Newton functors on the device), or
                           * NLSG_OBJ_ROSENBROCK / SPHERE / STYBLINSKI_TANG / RASTRIGIN / CUSTOM: the
                           * reference's default functors, Grad = fin_diff and Hess =
                           * fin_diff_h at accuracy 1 (nlsolver.h:3494-3511, 1385-1413,
                           * 1446-1515), every probe evaluated on the device; Cholesky only;
                           * m is ignored and there is no nlsg_lm_set_data            */
  int32_t solver;         /* nlsg_lm_solver                                          */
  uint64_t batch;         /* independent problems                                    */
  uint64_t m;             /* residuals per problem                                   */
  uint64_t n;             /* parameters per problem (<= 1024; NLSG_LM_QR: <= 64)     */
  double lambda, up, down; /* ctor args lambda, upward_mult, downward_mult (:3443-45) */
  uint64_t max_iter;      /* :3446                                                   */
  double f_delta;         /* :3447                                                   */
} nlsg_lm_config;

int nlsg_lm_create(const nlsg_lm_config *cfg, nlsg_lm **out);
/* cfg->objective == NLSG_OBJ_CUSTOM, as nlsg_de_create_custom: LevenbergMarquardt with the
 * default functors on a user objective compiled at run time */
int nlsg_lm_create_custom(const nlsg_lm_config *cfg, const nlsg_custom_objective *obj, nlsg_lm **out);
int nlsg_lm_destroy(nlsg_lm *e);
/* design matrices A [batch][m][n] row-major and targets y [batch][m] (copied to HBM, in chunks
 * that overlap their repacking on the device). From page-locked memory (nlsg_host_alloc) the
 * copy runs at the PCIe link's rate; from pageable memory the runtime stages it (~10-15 GB/s). */
int nlsg_lm_set_data(nlsg_lm *e, const double *a_host, const double *y_host);
/* Page-locked host memory for buffers handed to the library (any engine); free with nlsg_host_free. */
int nlsg_host_alloc(void **out, uint64_t bytes);
int nlsg_host_free(void *ptr);
/* Switch the damped-system solver of an existing engine (the model data stays resident): the
 * next nlsg_lm_minimize uses it. */
int nlsg_lm_set_solver(nlsg_lm *e, int32_t solver);
/* Whole solve() of every problem in one launch: theta [batch][n] in/out, one status per
 * problem (f, iterations, f/grad/hess evaluation counts), final damping per problem
 * (the reference keeps lambda as a mutable member, :3436/3541). NULLs are skipped. */
int nlsg_lm_minimize(nlsg_lm *e, double *theta_inout_host, nlsg_status *status_host,
                     double *lambda_out_host);
/* Same launch bracketed by hipEvents (theta restored from theta0 each time). */
int nlsg_lm_time_solve(nlsg_lm *e, const double *theta0_host, uint32_t repeats, float *ms_total);
/* Measurement aid: `repeats` launches of the evaluation kernel (f, g = 2 J^T r, H = 2 J^T J at
 * theta0 for every problem; the Gauss-Newton functors of nlsolver.h:3513-3516 / 3535-3537). */
int nlsg_lm_time_eval_kernel(nlsg_lm *e, const double *theta0_host, uint32_t repeats, float *ms_total);
/* the same for the QR step kernel alone (after one evaluation at theta0) */
int nlsg_lm_time_qr_kernel(nlsg_lm *e, const double *theta0_host, uint32_t repeats, float *ms_total);

/* ========================================================================== */
/* Batched Nelder-Mead — replaces NelderMead::solve (nlsolver.h:2166-2299), the  */
/* simplex helpers (simplex ctor 1905-1950, update_centroid 1965-1984,           */
/* simplex_transform 1986-2007, shrink 2009-2035, max_abs_vec 1894-1904) and the */
/* minimize/maximize wrappers with restarts (2127-2163), for `batch` independent */
/* starts (one simplex per workgroup).                                           */
/* ========================================================================== */
typedef struct nlsg_nm nlsg_nm;

/* NLSG_NM_REFERENCE_ORDER: the objective's terms and std_err's two sums (nlsolver.h:2037-2052) are
 * added in INDEX order, as the reference's sequential loops add them, instead of the kernels' lane
 * tree (everything else — centroid, transforms, shrink — is per coordinate and already the
 * reference's). With it the engine reproduces the reference's own runs bit for bit at every
 * dimension; without it a run can fork where two vertices are equal under one order and one ulp
 * apart under the other (same algorithm, another tie-break: Rosenbrock-128D from a constant start
 * forks at its 262nd evaluation). Objectives given by their terms (not Rastrigin, not a
 * whole-vector custom body). Costs the serial sums' latency per evaluation and scan. */
#define NLSG_NM_REFERENCE_ORDER 1

typedef struct {
  uint32_t struct_size;
  int32_t device;
  void *stream;
  int32_t objective;   /* nlsg_objective                                          */
  int32_t minimize;    /* 1 = minimize(), 0 = maximize()                           */
  int32_t bounded;     /* 1 = the (x, upper, lower) overloads (2136, 2155)         */
  int32_t flags;       /* 0 or NLSG_NM_REFERENCE_ORDER                             */
  uint64_t batch;
  uint64_t dim;        /* <= 1024 (past 128 the simplex lives in a global workspace) */
  double step, alpha, gamma, rho, sigma, eps; /* ctor args, nlsolver.h:2110-2113   */
  uint64_t max_iter, no_change_best_tol, restarts; /* :2114-2115                   */
} nlsg_nm_config;

int nlsg_nm_create(const nlsg_nm_config *cfg, nlsg_nm **out);
/* cfg->objective == NLSG_OBJ_CUSTOM, as nlsg_de_create_custom */
int nlsg_nm_create_custom(const nlsg_nm_config *cfg, const nlsg_custom_objective *obj, nlsg_nm **out);
int nlsg_nm_destroy(nlsg_nm *e);
/* x [batch][dim] in/out; upper/lower [dim] (shared by the batch; NULL when unbounded; note
 * the reference's argument order: upper first). One status per start; eps_out receives each
 * solver's mutated tolerance (nlsolver.h:2189). NULLs are skipped. Synchronises. */
int nlsg_nm_minimize(nlsg_nm *e, double *x_inout_host, const double *upper_host,
                     const double *lower_host, nlsg_status *status_host, double *eps_out_host);
/* Measurement aid: one solve from x0_host with the kernel's phase counters switched on —
 * cycles_host[batch][8]: shader-clock cycles the start's decision chain spent in 0 the scan
 * (std_err, best / worst / second worst, stop tests), 1 the centroid, 2 the reflection (transform,
 * evaluation, decision), 3 expansion or contraction, 4 shrink + rescoring (with its barriers);
 * [6] iterations, [7] shrinks. x is not returned. */
int nlsg_nm_phase_cycles(nlsg_nm *e, const double *x0_host, uint64_t *cycles_host);
int nlsg_nm_time_solve(nlsg_nm *e, const double *x0_host, uint32_t repeats, float *ms_total);

/* ========================================================================== */
/* Batched simulated annealing — replaces SANN::solve (nlsolver.h:2777-2814)   */
/* with rnorm (2479-2485) and the minimize / maximize wrappers (2766-2773), for */
/* `batch` independent chains (one per wave). Draws are keyed */
/* by (seed, chain, step, slot) instead of coming from one sequential generator */
/* (oracle: orc_sann_sync).                                                     */
/* ========================================================================== */
typedef struct nlsg_sann nlsg_sann;

typedef struct {
  uint32_t struct_size;
  int32_t device;
  void *stream;
  int32_t objective;         /* nlsg_objective                                       */
  int32_t minimize;          /* 1: minimize(), 0: maximize() (f_multiplier, :2779)   */
  uint64_t batch;            /* independent chains                                   */
  uint64_t dim;              /* any (past 1024 the chain is streamed from memory)    */
  uint64_t chain_lo;         /* global id of chain 0 (keys the draws; batch sharding)*/
  uint64_t max_iter;         /* ctor arg max_iter = 5000 (:2760)                     */
  uint64_t temperature_iter; /* ctor arg temperature_iter = 10 (:2761)               */
  double temperature_max;    /* ctor arg temperature_max = 10.0 (:2761)              */
  uint64_t seed;
} nlsg_sann_config;

int nlsg_sann_create(const nlsg_sann_config *cfg, nlsg_sann **out);
/* cfg->objective == NLSG_OBJ_CUSTOM, as nlsg_de_create_custom */
int nlsg_sann_create_custom(const nlsg_sann_config *cfg, const nlsg_custom_objective *obj,
                            nlsg_sann **out);
int nlsg_sann_destroy(nlsg_sann *e);
/* x [batch][dim] in: starts, out: best points (:2808). One status per chain: f_value =
 * f_multiplier * f(x) as the reference returns it (:2790), iteration = max_iter,
 * function_calls_used = 1 + max_iter * (temperature_iter - 1). Synchronises. */
int nlsg_sann_minimize(nlsg_sann *e, double *x_inout_host, nlsg_status *status_host);
/* The same solve bracketed by hipEvents (starts restored from x0 each time). */
int nlsg_sann_time_solve(nlsg_sann *e, const double *x0_host, uint32_t repeats, float *ms_total);

/* ========================================================================== */
/* Batched Nelder-Mead / PSO hybrid — replaces NelderMeadPSO::solve             */
/* (nlsolver.h:3623-3685) with init_solver_state 3686-3738, apply_simplex       */
/* 3739-3822, apply_pso 3823-3866, update_centroid, shrink, simplex_std_err     */
/* 3867-3918 and the four minimize / maximize overloads 3583-3620, for `batch`  */
/* independent instances (one workgroup each). Draws are keyed */
/* by (seed, instance, iteration, particle rank, slot) (oracle: orc_nmpso_sync). */
/* ========================================================================== */
typedef struct nlsg_nmpso nlsg_nmpso;

typedef struct {
  uint32_t struct_size;
  int32_t device;
  void *stream;
  int32_t objective;   /* nlsg_objective                                              */
  int32_t minimize;    /* 1: minimize(), 0: maximize()                                */
  int32_t bounded;     /* 0: minimize(x): bounds -+|2.5 x_i| (3587-3593) seed the PSO
                        * particles only; 1: minimize(x, lower, upper): they also clamp
                        * the simplex points and the velocity (by coordinate)          */
  int32_t reserved;
  uint64_t batch;      /* independent instances                                       */
  uint64_t dim;        /* 2 <= dim <= 1024; 3 dim + 1 particles per instance          */
  uint64_t inst_lo;    /* global id of instance 0 (keys the draws; batch sharding)    */
  double alpha, gamma, rho, sigma;      /* ctor args, defaults 1, 2, 0.5, 0.5 (3564-3566) */
  double inertia, cognitive, social;    /* 0.8, 1.8, 1.8 (3566-3567)                   */
  double eps;                           /* 1e-6 (3568)                                 */
  uint64_t max_iter, no_change_best_iter; /* 1000, 20 (3568-3569)                      */
  uint64_t seed;
} nlsg_nmpso_config;

int nlsg_nmpso_create(const nlsg_nmpso_config *cfg, nlsg_nmpso **out);
/* cfg->objective == NLSG_OBJ_CUSTOM, as nlsg_de_create_custom */
int nlsg_nmpso_create_custom(const nlsg_nmpso_config *cfg, const nlsg_custom_objective *obj,
                             nlsg_nmpso **out);
int nlsg_nmpso_destroy(nlsg_nmpso *e);
/* x [batch][dim] in: starts, out: best particles; lower/upper [dim] shared by the batch (note
 * the reference's order: lower first, 3609-3612; NULL when unbounded). One status per
 * instance (f_value = f_multiplier * f). Synchronises. */
int nlsg_nmpso_minimize(nlsg_nmpso *e, double *x_inout_host, const double *lower_host,
                        const double *upper_host, nlsg_status *status_host);
int nlsg_nmpso_time_solve(nlsg_nmpso *e, const double *x0_host, uint32_t repeats, float *ms_total);

/* ========================================================================== */
/* Batched linear least squares by Givens QR — replaces tinyqr::lm             */
/* (tinyqr.h:461-470: qr_decomposition :291-310 -> qr_impl :253-283 with        */
/* givens_rotation :86-97 and rotate_matrix :126-139, then back_solve :437-459) */
/* for `batch` independent n x p systems, n >= p, 1 <= p <= 64 (the reference   */
/* solves one system per call). Same rotations in the same per-element order;   */
/* Q is never formed (y is rotated along with R); values agree with the          */
/* reference's to rounding, bit for bit with oracle_lm.c order 1.               */
/* ========================================================================== */
/* X_host: [batch][p][n] — every system column-major n x p exactly as tinyqr::lm takes it
 * (X[j * n + i] = element (i, j)); y_host: [batch][n]; beta_host out: [batch][p]; tol: lm()'s
 * third argument (default 1e-12 there), entries of R below it are read as 0. ms_kernel (may be
 * NULL): the kernel's duration by HIP events, transfers excluded. */
int nlsg_tinyqr_lm(const double *X_host, const double *y_host, uint64_t batch, uint64_t n, uint64_t p,
                   double tol, int32_t device, double *beta_host, float *ms_kernel);
/* The same on systems that already live in HBM (device pointers, same layouts); enqueued on
 * `stream` (NULL: the null stream), asynchronous. */
int nlsg_tinyqr_lm_device(const double *X_dev, const double *y_dev, uint64_t batch, uint64_t n,
                          uint64_t p, double tol, int32_t device, void *stream, double *beta_dev);
/* tinyqr::qr_decomposition (tinyqr.h:291-310) and lm (:461-470) in REFERENCE ORDER on the device — a
 * parity mode, not the fast path: the reference's serial rotation order, Q formed, element updates
 * as two products and an add, back_solve's sums in index order, so that Q, R and beta carry the
 * reference's own bits (tests/golden/tinyqr.json). X as for nlsg_tinyqr_lm. Outputs, each optional
 * (NULL): Q [batch][p][n] = the thin Q as the reference returns it (Q[i * n + j]), R [batch][p][p]
 * with R[j * p + i] = R(i, j) after the cleanup |v| < tol -> 0, beta [batch][p] (needs y_host;
 * p <= 64). n >= p, n + p <= 1280, any p. */
int nlsg_tinyqr_qr(const double *X_host, const double *y_host, uint64_t batch, uint64_t n, uint64_t p,
                   double tol, int32_t device, double *Q_host, double *R_host, double *beta_host);

#ifdef __cplusplus
}
#endif
#endif /* NLSG_C_API_H_ */
