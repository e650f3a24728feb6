// include/nlsolver_mi/nlsolver.h — header-only C++17 host API of the MI355X-native
// nlsolver iteration engine.
//
// Same API surface as JSzitas/nlsolver (functor objective + std::vector<T> +
// minimize()/maximize(), solver_status) so user code switches by changing the
// include. Written from scratch; each block cites the reference interface it
// mirrors (file:line into the reference tree).
//
// Two execution paths, selected at COMPILE time by the objective's type:
//   * device objectives (nlsolver::device::Rosenbrock<double>, ... — types that
//     carry `nlsg_objective`): the population loops run as HIP kernels on a
//     gfx950 GPU through the extern "C" boundary include/nlsg_c_api.h
//     (libnlsolver_hip.so, loaded with dlopen). There is NO CPU fallback on this
//     path: a missing library or device throws nlsolver::device_error.
//   * any other callable (lambdas, stateful functors; README.md:127-144): it can
//     only run on the host, exactly as in the reference (config C1 of
//     BASELINE.json, "plumbing, no GPU"): a single-threaded serial loop that
//     consumes the caller's generator draw for draw like nlsolver.h:2414-2476.
#ifndef NLSOLVER_MI_NLSOLVER_H_
#define NLSOLVER_MI_NLSOLVER_H_

#include <dlfcn.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../nlsg_c_api.h"

namespace nlsolver {

// ---------------------------------------------------------------------------
// rng — nlsolver.h:1176-1382
// ---------------------------------------------------------------------------
namespace rng {
// splitmix64 (nlsolver.h:1263-1288): fixed seed, yield() in [0,1], yield_init() raw.
template <typename scalar_t = float>
struct splitmix {
  splitmix() : s_(12374563468ull) {}
  uint64_t yield_init() {
    uint64_t z = (s_ += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  scalar_t yield() {
    return static_cast<scalar_t>(yield_init()) / static_cast<scalar_t>(UINT64_MAX);
  }
  scalar_t operator()() { return yield(); }
  void set_state(uint64_t seed) { s_ = seed; }
  std::vector<scalar_t> get_state() const { return {static_cast<scalar_t>(s_)}; }

 private:
  uint64_t s_;
};

// xorshift128+ (nlsolver.h:1343-1381): state seeded from splitmix, x1 = x0 >> 32,
// output (t+s)/2^64 in [0,1] inclusive.
template <typename scalar_t = float>
struct xorshift {
  xorshift() { reset(); }
  scalar_t yield() {
    uint64_t t = x_[0];
    const uint64_t s = x_[1];
    x_[0] = s;
    t ^= t << 23;
    t ^= t >> 18;
    t ^= s ^ (s >> 5);
    x_[1] = t;
    return static_cast<scalar_t>((t + s) / static_cast<scalar_t>(UINT64_MAX));
  }
  scalar_t operator()() { return yield(); }
  void reset() {
    splitmix<scalar_t> seeder;
    x_[0] = seeder.yield_init();
    x_[1] = x_[0] >> 32;
  }
  void set_state(uint64_t y, uint64_t z) {
    x_[0] = y;
    x_[1] = z;
  }
  std::vector<scalar_t> get_state() const {
    return {static_cast<scalar_t>(x_[0]), static_cast<scalar_t>(x_[1])};
  }

 private:
  uint64_t x_[2]{};
};
}  // namespace rng

// ---------------------------------------------------------------------------
// solver_status — nlsolver.h:2054-2097 (same ctor order, print text, summary order)
// ---------------------------------------------------------------------------
template <typename scalar_t = double>
struct solver_status {
  solver_status(const scalar_t f_val, const size_t iter_used, const size_t f_calls_used,
                const size_t grad_evals_used = 0ul, const size_t hess_evals_used = 0ul)
      : f_value(f_val),
        iteration(iter_used),
        function_calls_used(f_calls_used),
        gradient_evals_used(grad_evals_used),
        hessian_evals_used(hess_evals_used) {}
  void print() const {
    std::cout << "Function calls used: " << function_calls_used << std::endl;
    std::cout << "Algorithm iterations used: " << iteration << std::endl;
    if (gradient_evals_used > 0)
      std::cout << "Gradient evaluations used: " << gradient_evals_used << std::endl;
    if (hessian_evals_used > 0)
      std::cout << "Hessian evaluations used: " << hessian_evals_used << std::endl;
    std::cout << "With final function value of " << f_value << std::endl;
  }
  std::tuple<size_t, size_t, scalar_t, size_t, size_t> get_summary() const {
    return std::make_tuple(function_calls_used, iteration, f_value, gradient_evals_used,
                           hessian_evals_used);
  }
  void add(const solver_status<scalar_t> &other) {
    function_calls_used += other.function_calls_used;
    iteration += other.iteration;
    f_value = other.f_value;
    gradient_evals_used += other.gradient_evals_used;
    hessian_evals_used += other.hessian_evals_used;
  }

 private:
  scalar_t f_value;
  size_t iteration, function_calls_used, gradient_evals_used, hessian_evals_used;
};

// std_err — nlsolver.h:2037-2052 (two passes, pow(.,2), n-1).
template <typename scalar_t = double>
static inline scalar_t std_err(const std::vector<scalar_t> &x) {
  const size_t n = x.size();
  scalar_t mean = 0, acc = 0;
  for (size_t i = 0; i < n; i++) mean += x[i];
  mean /= static_cast<scalar_t>(n);
  for (size_t i = 0; i < n; i++) acc += std::pow(x[i] - mean, 2);
  acc /= static_cast<scalar_t>(n - 1);
  return std::sqrt(acc);
}

// ---------------------------------------------------------------------------
// device objectives + the dlopen'ed C-ABI
// ---------------------------------------------------------------------------
struct device_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

namespace device {
// Tagged objective types: the solver recognises `nlsg_objective` and runs the
// population loops on the GPU. operator() evaluates the same formula on the
// host (for the caller's own use; the solvers never call it on this path).
template <typename T = double>
struct Rosenbrock {  // example.cpp:41-48 generalised to an N-D chain
  static constexpr int nlsg_objective = NLSG_OBJ_ROSENBROCK;
  T operator()(const std::vector<T> &x) const {
    T acc = 0;
    for (size_t i = 0; i + 1 < x.size(); i++) {
      const T t1 = 1 - x[i];
      const T t2 = x[i + 1] - x[i] * x[i];
      acc += t1 * t1 + 100 * t2 * t2;
    }
    return acc;
  }
};
template <typename T = double>
struct Sphere {  // test_functions.h:52-57
  static constexpr int nlsg_objective = NLSG_OBJ_SPHERE;
  T operator()(const std::vector<T> &x) const {
    T acc = 0;
    for (const T v : x) acc += v * v;
    return acc;
  }
};
template <typename T = double>
struct StyblinskiTang {  // test_functions.h:249-260
  static constexpr int nlsg_objective = NLSG_OBJ_STYBLINSKI_TANG;
  T operator()(const std::vector<T> &x) const {
    T acc = 0;
    for (const T v : x) {
      const T v2 = v * v;
      acc += v2 * v2 - 16 * v2 + 5 * v;
    }
    return acc / 2.0;
  }
};
template <typename T = double>
struct Rastrigin {  // test_functions.h:69-78
  static constexpr int nlsg_objective = NLSG_OBJ_RASTRIGIN;
  T operator()(const std::vector<T> &x) const {
    T acc = 0;
    for (const T v : x) acc += v * v - 10 * std::cos(2 * M_PI * v);
    return 10.0 * static_cast<T>(x.size()) + acc;
  }
};

template <typename C, typename = void>
struct is_device_objective : std::false_type {};
template <typename C>
struct is_device_objective<C, std::void_t<decltype(C::nlsg_objective)>> : std::true_type {};

// Lazily bound entry points of libnlsolver_hip.so. Search order: $NLSG_LIBRARY,
// then the default loader path.
class api {
 public:
  static const api &get() {
    static const api instance;
    return instance;
  }
  decltype(&nlsg_last_error) last_error;
  decltype(&nlsg_abi_version) abi_version;
  decltype(&nlsg_de_create) de_create;
  decltype(&nlsg_de_destroy) de_destroy;
  decltype(&nlsg_de_minimize) de_minimize;
  decltype(&nlsg_pso_create) pso_create;
  decltype(&nlsg_pso_destroy) pso_destroy;
  decltype(&nlsg_pso_minimize) pso_minimize;

  void check(int rc) const {
    if (rc != NLSG_OK)
      throw device_error(std::string("nlsg error ") + std::to_string(rc) + ": " + last_error());
  }

 private:
  api() {
    const char *env = std::getenv("NLSG_LIBRARY");
    const char *name = (env && *env) ? env : "libnlsolver_hip.so";
    void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h)
      throw device_error(std::string("cannot load ") + name + " (" + dlerror() +
                         "); device objectives have no CPU fallback");
    bind(h, "nlsg_last_error", last_error);
    bind(h, "nlsg_abi_version", abi_version);
    bind(h, "nlsg_de_create", de_create);
    bind(h, "nlsg_de_destroy", de_destroy);
    bind(h, "nlsg_de_minimize", de_minimize);
    bind(h, "nlsg_pso_create", pso_create);
    bind(h, "nlsg_pso_destroy", pso_destroy);
    bind(h, "nlsg_pso_minimize", pso_minimize);
    if (abi_version() != NLSG_ABI_VERSION)
      throw device_error("libnlsolver_hip.so ABI version mismatch");
  }
  template <typename F>
  static void bind(void *h, const char *sym, F &fn) {
    fn = reinterpret_cast<F>(dlsym(h, sym));
    if (!fn) throw device_error(std::string("missing symbol ") + sym);
  }
};

// 64-bit key for the device's counter-based generator from a reference-style
// generator (T operator()() in [0,1], held by reference: it advances by exactly
// two draws, so distinct host states give distinct device streams).
template <typename RNG>
inline uint64_t seed_from(RNG &generator) {
  auto half = [&]() {
    const double u = static_cast<double>(generator()) * 4294967296.0;
    return u >= 4294967295.0 ? 0xFFFFFFFFull : static_cast<uint64_t>(u);
  };
  const uint64_t hi = half();
  const uint64_t lo = half();
  return (hi << 32) | lo;
}
}  // namespace device

// ---------------------------------------------------------------------------
// DE — nlsolver.h:2377-2477
// ---------------------------------------------------------------------------
enum RecombinationStrategy { best, random };  // nlsolver.h:2377

template <typename Callable, typename RNG, typename scalar_t = double,
          RecombinationStrategy RecombinationType = random>
class DE {
  Callable &f;
  RNG &generator;
  const scalar_t crossover_prob, differential_weight, eps;
  const size_t pop_size, max_iter, best_value_no_change;

 public:
  // same positional arguments and defaults as nlsolver.h:2390-2394
  DE(Callable &f, RNG &generator, const scalar_t crossover_prob = 0.9,
     const scalar_t differential_weight = 0.8, const scalar_t eps = 10e-4,
     const size_t pop_size = 50, const size_t max_iter = 1000,
     const size_t best_val_no_change = 50)
      : f(f),
        generator(generator),
        crossover_prob(crossover_prob),
        differential_weight(differential_weight),
        eps(eps),
        pop_size(pop_size),
        max_iter(max_iter),
        best_value_no_change(best_val_no_change) {}
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) { return solve<true>(x); }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x) { return solve<false>(x); }

 private:
  template <bool minimize>
  solver_status<scalar_t> solve(std::vector<scalar_t> &x) {
    if constexpr (device::is_device_objective<Callable>::value) {
      static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
      return solve_device<minimize>(x);
    } else {
      return solve_host<minimize>(x);
    }
  }

  // GPU path: the whole while(true) loop of nlsolver.h:2429-2475 runs device-resident.
  template <bool minimize>
  solver_status<scalar_t> solve_device(std::vector<scalar_t> &x) {
    const device::api &api = device::api::get();
    nlsg_de_config cfg{};
    cfg.struct_size = sizeof(cfg);
    if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
    cfg.objective = Callable::nlsg_objective;
    cfg.minimize = minimize ? 1 : 0;
    cfg.strategy = RecombinationType == best ? NLSG_DE_BEST : NLSG_DE_RANDOM;
    cfg.pop = cfg.shard_n = pop_size;
    cfg.dim = x.size();
    cfg.CR = crossover_prob;
    cfg.F = differential_weight;
    cfg.eps = eps;
    cfg.max_iter = max_iter;
    cfg.best_val_no_change = best_value_no_change;
    cfg.seed = device::seed_from(generator);
    nlsg_de *eng = nullptr;
    api.check(api.de_create(&cfg, &eng));
    nlsg_status st{};
    const int rc = api.de_minimize(eng, x.data(), 0, &st);
    const std::string msg = rc ? api.last_error() : "";
    api.de_destroy(eng);
    if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
    return solver_status<scalar_t>(st.f_value, st.iteration, st.function_calls_used);
  }

  // Host path for arbitrary callables (config C1): the reference's serial,
  // in-place algorithm; population kept as one row-major matrix.
  template <bool minimize>
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x) {
    const size_t D = x.size(), NP = pop_size;
    constexpr scalar_t sign = minimize ? 1.0 : -1.0;
    std::vector<scalar_t> agents(NP * D), scores(NP);
    std::vector<scalar_t> point(D);  // objective argument (functors take vector&)
    auto row = [&](size_t a) { return agents.data() + a * D; };
    auto evaluate = [&](const scalar_t *src) {
      point.assign(src, src + D);
      return sign * f(point);
    };
    // init_agents (2315-2323): (u - 0.5) * x0[i], agent-major draw order
    for (size_t a = 0; a < NP; a++)
      for (size_t i = 0; i < D; i++) row(a)[i] = (generator() - 0.5) * x[i];
    for (size_t a = 0; a < NP; a++) scores[a] = evaluate(row(a));
    size_t calls = NP, iter = 0, best_id = 0, stale = 0;
    std::vector<scalar_t> trial(D);
    auto draw_index = [&](size_t n) { return static_cast<size_t>(generator() * n); };  // 2325-29
    for (;;) {
      bool moved = false;
      for (size_t i = 0; i < NP; i++)  // strict '<': the incumbent keeps ties (2432-2437)
        if (scores[i] < scores[best_id]) {
          best_id = i;
          moved = true;
        }
      stale = moved ? 0 : stale + 1;  // 2439
      if (iter >= max_iter || stale >= best_value_no_change || std_err(scores) < eps) {
        x.assign(row(best_id), row(best_id) + D);  // 2441-2447
        return solver_status<scalar_t>(scores[best_id], iter, calls);
      }
      for (size_t i = 0; i < NP; i++) {
        // generate_indices (2331-2355): three distinct donors != fixed, by rejection
        const size_t fixed = RecombinationType == random ? i : best_id;
        size_t donor[3];
        for (size_t have = 0; have < 3;) {
          const size_t cand = draw_index(NP);
          bool clash = cand == fixed;
          for (size_t k = 0; k < have; k++) clash = clash || donor[k] == cand;
          if (!clash) donor[have++] = cand;
        }
        // propose_new_agent (2357-2375): forced dimension, then one draw per coordinate
        const size_t forced = draw_index(D);
        const scalar_t *a = row(donor[0]), *b = row(donor[1]), *c = row(donor[2]);
        const scalar_t *keep = row(fixed);
        for (size_t d = 0; d < D; d++) {
          const scalar_t u = generator();
          trial[d] = (u < crossover_prob || d == forced)
                         ? a[d] + differential_weight * (b[d] - c[d])
                         : keep[d];
        }
        const scalar_t score = evaluate(trial.data());
        calls++;
        if (score < scores[i]) {  // greedy in-place replacement (2466-2471)
          std::copy(trial.begin(), trial.end(), row(i));
          scores[i] = score;
        }
      }
      iter++;
    }
  }
};

// README.md:80 uses the (stale) name DESolver for the same class.
template <typename Callable, typename RNG, typename scalar_t = double,
          RecombinationStrategy RecombinationType = random>
using DESolver = DE<Callable, RNG, scalar_t, RecombinationType>;

// ---------------------------------------------------------------------------
// PSO — nlsolver.h:2479-2742
// ---------------------------------------------------------------------------
// rnorm (nlsolver.h:2479-2485): u1 feeds log, u2 feeds cos (two sequenced draws).
template <typename scalar_t, typename RNG>
static inline scalar_t rnorm(RNG &generator) {
  constexpr scalar_t pi_ = 3.141593;
  const scalar_t u1 = generator();
  const scalar_t u2 = generator();
  return std::sqrt(-2 * std::log(u1)) * std::cos(2 * pi_ * u2);
}

enum PSOType { Vanilla, Accelerated };  // nlsolver.h:2496

template <typename Callable, typename RNG, typename scalar_t = double, PSOType Type = Vanilla>
class PSO {
  RNG &generator;
  Callable &f;
  const scalar_t inertia0, cognitive_coef, social_coef;
  const size_t n_particles, max_iter, best_val_no_change;
  const scalar_t eps;

 public:
  // same positional arguments and defaults as nlsolver.h:2522-2526
  PSO(Callable &f, RNG &generator, const scalar_t inertia = 0.8,
      const scalar_t cognitive_coef = 1.8, const scalar_t social_coef = 1.8,
      const size_t n_particles = 10, const size_t max_iter = 5000,
      const size_t best_val_no_change = 50, const scalar_t eps = 10e-4)
      : generator(generator),
        f(f),
        inertia0(inertia),
        cognitive_coef(cognitive_coef),
        social_coef(social_coef),
        n_particles(n_particles),
        max_iter(max_iter),
        best_val_no_change(best_val_no_change),
        eps(eps) {}
  // bounds = -+|x_i|, no thresholding (nlsolver.h:2553-2575)
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) { return free_run<true>(x); }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x) { return free_run<false>(x); }
  // (x, lower, upper) — note the order, opposite to NelderMead (nlsolver.h:2577-2591)
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x, const std::vector<scalar_t> &lower,
                                   const std::vector<scalar_t> &upper) {
    return solve<true, true>(x, lower, upper);
  }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x, const std::vector<scalar_t> &lower,
                                   const std::vector<scalar_t> &upper) {
    return solve<false, true>(x, lower, upper);
  }

 private:
  template <bool minimize>
  solver_status<scalar_t> free_run(std::vector<scalar_t> &x) {
    std::vector<scalar_t> lower(x.size()), upper(x.size());
    for (size_t i = 0; i < x.size(); i++) {
      const scalar_t t = std::abs(x[i]);
      lower[i] = -t;
      upper[i] = t;
    }
    return solve<minimize, false>(x, lower, upper);
  }

  template <bool minimize, bool constrained>
  solver_status<scalar_t> solve(std::vector<scalar_t> &x, const std::vector<scalar_t> &lower,
                                const std::vector<scalar_t> &upper) {
    if constexpr (device::is_device_objective<Callable>::value) {
      static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
      const device::api &api = device::api::get();
      nlsg_pso_config cfg{};
      cfg.struct_size = sizeof(cfg);
      if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
      cfg.objective = Callable::nlsg_objective;
      cfg.minimize = minimize ? 1 : 0;
      cfg.type = Type == Accelerated ? NLSG_PSO_ACCELERATED : NLSG_PSO_VANILLA;
      cfg.bounded = constrained ? 1 : 0;
      cfg.n_particles = cfg.shard_n = n_particles;
      cfg.dim = x.size();
      cfg.inertia = inertia0;
      cfg.cognitive = cognitive_coef;
      cfg.social = social_coef;
      cfg.eps = eps;
      cfg.max_iter = max_iter;
      cfg.best_val_no_change = best_val_no_change;
      cfg.seed = device::seed_from(generator);
      nlsg_pso *eng = nullptr;
      api.check(api.pso_create(&cfg, &eng));
      nlsg_status st{};
      const int rc = api.pso_minimize(eng, x.data(), lower.data(), upper.data(), 0, &st);
      const std::string msg = rc ? api.last_error() : "";
      api.pso_destroy(eng);
      if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
      return solver_status<scalar_t>(st.f_value, st.iteration, st.function_calls_used);
    } else {
      return solve_host<minimize, constrained>(x, lower, upper);
    }
  }

  // Host path for arbitrary callables: the reference's serial algorithm
  // (init_solver_state 2626-2657, solve 2593-2624, update_best_positions
  // 2716-2741 incl. its sentinels and best_index rule). Accelerated is literal;
  // Vanilla uses the intended pbest/gbest terms (the reference's line 2669-2674
  // has a zero cognitive term and reads swarm_best_position out of bounds).
  template <bool minimize, bool constrained>
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x, const std::vector<scalar_t> &lower,
                                     const std::vector<scalar_t> &upper) {
    const size_t D = lower.size(), NP = n_particles;
    constexpr scalar_t sign = minimize ? 1.0 : -1.0;
    std::vector<scalar_t> pos(NP * D), vel(Type == Vanilla ? NP * D : 0),
        pbest_pos(Type == Vanilla ? NP * D : 0);
    std::vector<scalar_t> pbest_val(NP, 10000), gbest, point(D);
    scalar_t swarm_best = 100000.0, inertia = inertia0;
    size_t f_evals = 0, stale = 0, iter = 0;
    for (size_t i = 0; i < NP; i++)
      for (size_t j = 0; j < D; j++) {
        const scalar_t width = std::abs(upper[j] - lower[j]);
        pos[i * D + j] = lower[j] + ((upper[j] - lower[j]) * generator());
        if constexpr (Type == Vanilla) {
          vel[i * D + j] = -width + (generator() * width);
          pbest_pos[i * D + j] = pos[i * D + j];
        }
      }
    for (;;) {
      size_t best_index = 0;
      bool improved = false;
      for (size_t i = 0; i < NP; i++) {
        point.assign(pos.begin() + i * D, pos.begin() + (i + 1) * D);
        const scalar_t val = sign * f(point);
        if (val < swarm_best) {
          swarm_best = val;
          best_index = i;
          improved = true;
        }
        if (val < pbest_val[i]) {
          pbest_val[i] = val;
          if constexpr (Type == Vanilla)
            std::copy(point.begin(), point.end(), pbest_pos.begin() + i * D);
        }
      }
      f_evals += NP;
      if (improved) gbest.assign(pos.begin() + best_index * D, pos.begin() + (best_index + 1) * D);
      stale = (best_index == 0) * (stale + 1);  // nlsolver.h:2740
      if (iter >= max_iter || stale >= best_val_no_change || std_err(pbest_val) < eps) {
        x = gbest;
        return solver_status<scalar_t>(swarm_best, iter, f_evals);
      }
      if constexpr (Type == Accelerated) inertia = std::pow(inertia0, iter);  // :2613
      for (size_t i = 0; i < NP; i++)
        for (size_t j = 0; j < D; j++) {
          scalar_t &p = pos[i * D + j];
          if constexpr (Type == Accelerated) {
            p = inertia * rnorm<scalar_t>(generator) + (1 - cognitive_coef) * p +
                social_coef * gbest[j];
          } else {
            const scalar_t r_p = generator(), r_g = generator();
            scalar_t &v = vel[i * D + j];
            v = (inertia * v) + cognitive_coef * r_p * (pbest_pos[i * D + j] - p) +
                social_coef * r_g * (gbest[j] - p);
            p += v;
          }
          if constexpr (constrained) {
            p = p < lower[j] ? lower[j] : p;
            p = p > upper[j] ? upper[j] : p;
          }
        }
      iter++;
    }
  }
};

// README.md:99 uses the (stale) name PSOSolver for the same class.
template <typename Callable, typename RNG, typename scalar_t = double, PSOType Type = Vanilla>
using PSOSolver = PSO<Callable, RNG, scalar_t, Type>;

}  // namespace nlsolver

#endif  // NLSOLVER_MI_NLSOLVER_H_
