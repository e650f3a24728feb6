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

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../nlsg_c_api.h"
#include "./tinyqr.h"  // namespace tinyqr: qr_decomposition / back_solve / lm (reference: nlsolver.h:46)

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

// A user objective of the shape f(x) = finish(sum_i term(x_i, x_{i+1}), D), written as C++
// function bodies and compiled for the device when the solver runs (nlsg_custom_objective,
// hiprtc) -- the device-side answer to the reference's "any functor"
// contract (README.md:127-136). In the bodies: xi, xn (= x_{i+1}, chain objectives only) /
// s, D. There is no host evaluation of it: the type only works on the device path.
//   Custom<double> f("double t1 = 1 - xi; double t2 = xn - xi * xi; return t1 * t1 + 100 * t2 * t2;",
//                    /*chain=*/true);
//   auto st = DE<Custom<double>, rng::xorshift<double>, double>(f, gen).minimize(x);
template <typename T = double>
struct Custom {
  static constexpr int nlsg_objective = NLSG_OBJ_CUSTOM;
  std::string term_body, finish_body;
  int chain;  // NLSG_CUSTOM_TERMS / NLSG_CUSTOM_CHAIN / NLSG_CUSTOM_VECTOR
  explicit Custom(std::string term_body, bool chain = false, std::string finish_body = "return s;")
      : term_body(std::move(term_body)), finish_body(std::move(finish_body)), chain(chain ? 1 : 0) {}
  // the whole-vector form: `body` is the body of  double f(const X &x, uint64_t D)  with x(i),
  // x.size(), x.sum(g) (include/nlsg_c_api.h), for objectives that are not sums of terms:
  //   auto f = Custom<double>::vector("double a = x(0) * x(0) + x(1) - 11, b = x(0) + x(1) * x(1) - 7;"
  //                                   " return a * a + b * b;");
  static Custom vector(std::string body) {
    Custom c(std::move(body));
    c.chain = NLSG_CUSTOM_VECTOR;
    return c;
  }
};

// Objectives with an analytic gradient on the device (batched BFGS):
// f(x) = 1/2 sum d_i x_i^2 + 1/2 c (sum x)^2 - sum b_i x_i  (SURVEY.md §8c G6).
template <typename T = double>
struct QuadDiagRank1 {
  static constexpr int nlsg_grad_objective = NLSG_OBJ_QUAD_DIAG_RANK1;
  std::vector<T> d, b;
  T c;
  QuadDiagRank1(std::vector<T> d, std::vector<T> b, T c) : d(std::move(d)), b(std::move(b)), c(c) {}
  T operator()(const std::vector<T> &x) const {
    T q = 0, sx = 0, lin = 0;
    for (size_t i = 0; i < x.size(); i++) {
      q += d[i] * x[i] * x[i];
      sx += x[i];
      lin += b[i] * x[i];
    }
    return 0.5 * q + 0.5 * c * (sx * sx) - lin;
  }
};
// Grad tag for device objectives: their gradient is evaluated by the kernels.
struct analytic_grad {};

// NLLS model for the batched Levenberg-Marquardt path:
// r_i(theta) = y_i - tanh(sum_j A_ij theta_j), f = sum r^2; the Gauss-Newton gradient and
// Hessian (2 J^T r, 2 J^T J) are evaluated by the kernels. A: [problems][m][n] row-major.
template <typename T = double>
struct TanhRegression {
  static constexpr int nlsg_nlls_objective = NLSG_OBJ_TANH_REGRESSION;
  size_t m, n;
  std::vector<T> A, y;  // problems*m*n, problems*m
  TanhRegression(size_t m, size_t n, std::vector<T> A, std::vector<T> y)
      : m(m), n(n), A(std::move(A)), y(std::move(y)) {}
  size_t problems() const { return y.size() / m; }
  T operator()(const std::vector<T> &theta) const {  // problem 0
    T acc = 0;
    for (size_t i = 0; i < m; i++) {
      T z = 0;
      for (size_t j = 0; j < n; j++) z += A[i * n + j] * theta[j];
      const T r = y[i] - std::tanh(z);
      acc += r * r;
    }
    return acc;
  }
};
struct gauss_newton {};  // Grad / Hess tag of device NLLS models
template <typename C, typename = void>
struct has_nlls_objective : std::false_type {};
template <typename C>
struct has_nlls_objective<C, std::void_t<decltype(C::nlsg_nlls_objective)>> : std::true_type {};

template <typename C, typename = void>
struct has_grad_objective : std::false_type {};
template <typename C>
struct has_grad_objective<C, std::void_t<decltype(C::nlsg_grad_objective)>> : std::true_type {};

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
  decltype(&nlsg_de_create_custom) de_create_custom;
  decltype(&nlsg_rtc_load) rtc_load;
  decltype(&nlsg_de_destroy) de_destroy;
  decltype(&nlsg_de_minimize) de_minimize;
  decltype(&nlsg_pso_create) pso_create;
  decltype(&nlsg_pso_create_custom) pso_create_custom;
  decltype(&nlsg_pso_destroy) pso_destroy;
  decltype(&nlsg_pso_minimize) pso_minimize;
  decltype(&nlsg_bfgs_create) bfgs_create;
  decltype(&nlsg_bfgs_create_custom) bfgs_create_custom;
  decltype(&nlsg_bfgs_destroy) bfgs_destroy;
  decltype(&nlsg_bfgs_minimize) bfgs_minimize;
  decltype(&nlsg_lm_create) lm_create;
  decltype(&nlsg_lm_create_custom) lm_create_custom;
  decltype(&nlsg_lm_destroy) lm_destroy;
  decltype(&nlsg_lm_set_data) lm_set_data;
  decltype(&nlsg_lm_minimize) lm_minimize;
  decltype(&nlsg_nm_create) nm_create;
  decltype(&nlsg_nm_create_custom) nm_create_custom;
  decltype(&nlsg_nm_destroy) nm_destroy;
  decltype(&nlsg_nm_minimize) nm_minimize;
  decltype(&nlsg_nmpso_create) nmpso_create;
  decltype(&nlsg_nmpso_create_custom) nmpso_create_custom;
  decltype(&nlsg_nmpso_destroy) nmpso_destroy;
  decltype(&nlsg_nmpso_minimize) nmpso_minimize;
  decltype(&nlsg_sann_create) sann_create;
  decltype(&nlsg_sann_create_custom) sann_create_custom;
  decltype(&nlsg_sann_destroy) sann_destroy;
  decltype(&nlsg_sann_minimize) sann_minimize;

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
    bind(h, "nlsg_de_create_custom", de_create_custom);
    bind(h, "nlsg_rtc_load", rtc_load);
    bind(h, "nlsg_de_destroy", de_destroy);
    bind(h, "nlsg_de_minimize", de_minimize);
    bind(h, "nlsg_pso_create", pso_create);
    bind(h, "nlsg_pso_create_custom", pso_create_custom);
    bind(h, "nlsg_pso_destroy", pso_destroy);
    bind(h, "nlsg_pso_minimize", pso_minimize);
    bind(h, "nlsg_bfgs_create", bfgs_create);
    bind(h, "nlsg_bfgs_create_custom", bfgs_create_custom);
    bind(h, "nlsg_bfgs_destroy", bfgs_destroy);
    bind(h, "nlsg_bfgs_minimize", bfgs_minimize);
    bind(h, "nlsg_lm_create", lm_create);
    bind(h, "nlsg_lm_create_custom", lm_create_custom);
    bind(h, "nlsg_lm_destroy", lm_destroy);
    bind(h, "nlsg_lm_set_data", lm_set_data);
    bind(h, "nlsg_lm_minimize", lm_minimize);
    bind(h, "nlsg_nm_create", nm_create);
    bind(h, "nlsg_nm_create_custom", nm_create_custom);
    bind(h, "nlsg_nm_destroy", nm_destroy);
    bind(h, "nlsg_nm_minimize", nm_minimize);
    bind(h, "nlsg_nmpso_create", nmpso_create);
    bind(h, "nlsg_nmpso_create_custom", nmpso_create_custom);
    bind(h, "nlsg_nmpso_destroy", nmpso_destroy);
    bind(h, "nlsg_nmpso_minimize", nmpso_minimize);
    bind(h, "nlsg_sann_create", sann_create);
    bind(h, "nlsg_sann_create_custom", sann_create_custom);
    bind(h, "nlsg_sann_destroy", sann_destroy);
    bind(h, "nlsg_sann_minimize", sann_minimize);
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

// Summation order of the device solves whose results turn on the last bit of a sum: BFGS (dots,
// norms, H y, and with the default gradient fin_diff's differences over 12 eps), the
// default-functor LevenbergMarquardt (fin_diff_h's differences over 600 eps^2) and NelderMead (ties
// between vertices whose values differ only by the order their objective's terms were added in).
//   reference  (default) every sum in index order, separate multiply and add — the reference's
//              sequential loops (NLSG_BFGS_REFERENCE_ORDER / NLSG_LM_CHOLESKY_REFERENCE_ORDER): x, f
//              and every counter are the reference's own, bit for bit, wherever the reference's
//              arithmetic exists on the device (not Rastrigin, whose device cosine is not libm's;
//              not whole-vector Custom bodies; not TanhRegression). With the default functors these
//              are also the FASTER kernels — a probe per lane on the base point's shared terms and
//              prefix sums: BFGS on Rosenbrock-128D x 4096 starts 8.3 ms against 32.5 ms for 20
//              iterations, LM on Rosenbrock-16D x 4096 5.3 against 6.9 ms for 10 —; BFGS with a
//              gradient functor pays 1.16 x on large batches (n 1024 x 4096: 19.1 against 16.4 ms
//              per iteration; H passes with a lane per row), 9 x on ONE start of that size (12 ms
//              against 1.3).
//   tree       the wave's butterfly sums and fused multiply-adds: same algorithm, same branch
//              decisions on the reference's runs, values within 1e-8 .. 1e-6 (fin_diff) or
//              rounding (analytic gradient).
// Set before the solves it should govern: `nlsolver::device::summation() = sum_order::tree`, or the
// environment variable NLSG_SUMMATION = reference | tree (read at first use).
enum class sum_order { reference, tree };
inline sum_order &summation() {
  static sum_order order = [] {
    const char *e = std::getenv("NLSG_SUMMATION");
    const std::string v = e ? e : "";
    if (v == "tree") return sum_order::tree;
    if (!v.empty() && v != "reference")
      throw device_error("NLSG_SUMMATION must be reference or tree, not '" + v + "'");
    return sum_order::reference;
  }();
  return order;
}
inline bool reference_order() { return summation() == sum_order::reference; }
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
    if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
      // $NLSG_HIPRTC names the hiprtc of the HIP runtime in use (default: libhiprtc.so)
      api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
      nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
      api.check(api.de_create_custom(&cfg, &obj, &eng));
    } else {
      api.check(api.de_create(&cfg, &eng));
    }
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
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
        api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
        nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
        api.check(api.pso_create_custom(&cfg, &obj, &eng));
      } else {
        api.check(api.pso_create(&cfg, &eng));
      }
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

// ---------------------------------------------------------------------------
// finite differences + line search + BFGS — nlsolver.h:1383-1412, 1519-1891, 3130-3286
// ---------------------------------------------------------------------------
namespace finite_difference {
// central differences of accuracy 1 (4 evaluations per dimension), the only order the
// default Grad functor uses (nlsolver.h:1385-1412 with accuracy = 1, 2852-2853)
template <typename Callable, typename scalar_t>
void finite_difference_gradient(Callable &f, std::vector<scalar_t> &x, std::vector<scalar_t> &grad) {
  constexpr scalar_t eps = std::numeric_limits<scalar_t>::epsilon() * 10e7;
  constexpr scalar_t coeff[4] = {1, -8, 8, -1}, shift[4] = {-2, -1, 1, 2};
  constexpr scalar_t dd = 12 * eps;
  std::fill(grad.begin(), grad.end(), 0.0);
  for (size_t d = 0; d < x.size(); d++) {
    for (int k = 0; k < 4; k++) {
      const scalar_t keep = x[d];
      x[d] += shift[k] * eps;
      grad[d] += coeff[k] * f(x);
      x[d] = keep;
    }
    grad[d] /= dd;
  }
}
}  // namespace finite_difference

template <typename Callable, typename scalar_t>
struct fin_diff {  // nlsolver.h:2848-2855
  void operator()(Callable &f, std::vector<scalar_t> &x, std::vector<scalar_t> &gradient) {
    finite_difference::finite_difference_gradient<Callable, scalar_t>(f, x, gradient);
  }
};

namespace math {
template <typename T>
inline T dot(const T *x, const T *y, int n) {  // nlsolver.h:58-67
  T s = 0;
  for (int i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}
template <typename T>
inline T norm(const T *x, int n) {  // nlsolver.h:91-99
  T s = 0;
  for (int i = 0; i < n; i++) s += x[i] * x[i];
  return std::sqrt(s);
}
}  // namespace math

namespace linesearch {
// More-Thuente safeguarded step (nlsolver.h:1527-1671). Interval end points and the
// trial are updated in place; `info` reports the case (0 = rejected input).
template <typename T>
struct mt_interval {
  T stx, fx, dx, sty, fy, dy;
  bool brackt;
};
template <typename T>
static int cstep(mt_interval<T> &iv, T &stp, const T fp, const T dp, const T stpmin, const T stpmax,
                 int &info) {
  info = 0;
  T &stx = iv.stx, &fx = iv.fx, &dx = iv.dx, &sty = iv.sty, &fy = iv.fy, &dy = iv.dy;
  bool &brackt = iv.brackt;
  bool bound;
  if ((brackt & ((stp <= std::min<T>(stx, sty)) || (stp >= std::max<T>(stx, sty)))) ||
      (dx * (stp - stx) >= 0.0) || (stpmax < stpmin))
    return -1;
  const T sgnd = dp * (dx / std::fabs(dx));
  auto cubic = [](T theta, T a, T b) {
    const T s = std::max(std::fabs(theta), std::max(std::fabs(a), std::fabs(b)));
    return std::make_pair(s, (theta / s) * (theta / s) - (a / s) * (b / s));
  };
  T stpf = 0, stpc, stpq;
  if (fp > fx) {
    info = 1;
    bound = true;
    const T theta = 3. * (fx - fp) / (stp - stx) + dx + dp;
    auto [s, rad] = cubic(theta, dx, dp);
    T gamma = s * std::sqrt(rad);
    if (stp < stx) gamma = -gamma;
    const T r = ((gamma - dx) + theta) / (((gamma - dx) + gamma) + dp);
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.) * (stp - stx);
    stpf = std::fabs(stpc - stx) < std::fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2;
    brackt = true;
  } else if (sgnd < 0.0) {
    info = 2;
    bound = false;
    const T theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    auto [s, rad] = cubic(theta, dx, dp);
    T gamma = s * std::sqrt(rad);
    if (stp > stx) gamma = -gamma;
    const T r = ((gamma - dp) + theta) / (((gamma - dp) + gamma) + dx);
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
    brackt = true;
  } else if (std::fabs(dp) < std::fabs(dx)) {
    info = 3;
    bound = true;
    const T theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    auto [s, rad] = cubic(theta, dx, dp);
    T gamma = s * std::sqrt(std::max<T>(static_cast<T>(0.), rad));
    if (stp > stx) gamma = -gamma;
    const T r = ((gamma - dp) + theta) / ((gamma + (dx - dp)) + gamma);
    if ((r < 0.0) & (gamma != 0.0))
      stpc = stp + r * (stx - stp);
    else
      stpc = stp > stx ? stpmax : stpmin;
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    const bool closer_c = std::fabs(stp - stpc) < std::fabs(stp - stpq);
    const bool farther_c = std::fabs(stp - stpc) > std::fabs(stp - stpq);
    stpf = brackt ? (closer_c ? stpc : stpq) : (farther_c ? stpc : stpq);
  } else {
    info = 4;
    bound = false;
    if (brackt) {
      const T theta = 3 * (fp - fy) / (sty - stp) + dy + dp;
      auto [s, rad] = cubic(theta, dy, dp);
      T gamma = s * std::sqrt(rad);
      if (stp > sty) gamma = -gamma;
      const T r = ((gamma - dp) + theta) / (((gamma - dp) + gamma) + dy);
      stpf = stp + r * (sty - stp);
    } else {
      stpf = stp > stx ? stpmax : stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = std::clamp(stpf, stpmin, stpmax);
  if (brackt & bound) {
    const T lim = stx + static_cast<T>(0.66) * (sty - stx);
    stp = sty > stx ? std::min<T>(lim, stp) : std::max<T>(lim, stp);
  }
  return 0;
}

// cvsrch + more_thuente_search (nlsolver.h:1673-1793, 1880-1891): evaluates f(x) first,
// then up to 20 trial points; `gradient` ends as the gradient at the last trial.
template <typename Callable, typename Grad, typename T = double>
T more_thuente_search(Callable &f, std::vector<T> &x, std::vector<T> &gradient,
                      const std::vector<T> &dir, std::vector<T> &trial, T alpha, Grad g) {
  const T finit = f(x);
  T stp = alpha;
  constexpr T xtol = 1e-15, ftol = 1e-4, gtol = 1e-2, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4;
  constexpr int maxfev = 20;
  const int n = static_cast<int>(x.size());
  const T dginit = math::dot(gradient.data(), dir.data(), n);
  if (dginit >= 0.0) return stp;
  int info = 0, infoc = 1, nfev = 0;
  bool stage1 = true;
  const T dgtest = ftol * dginit;
  T width = stpmax - stpmin, width1 = 2 * width;
  mt_interval<T> iv{0.0, finit, dginit, 0.0, finit, dginit, false};
  for (;;) {
    T stmin, stmax;
    if (iv.brackt) {
      stmin = std::min<T>(iv.stx, iv.sty);
      stmax = std::max<T>(iv.stx, iv.sty);
    } else {
      stmin = iv.stx;
      stmax = stp + xtrapf * (stp - iv.stx);
    }
    stp = std::clamp(stp, stpmin, stpmax);
    if ((iv.brackt && ((stp <= stmin) || (stp >= stmax))) || (nfev >= maxfev - 1) ||
        (infoc == 0) || (iv.brackt && ((stmax - stmin) <= (xtol * stmax))))
      stp = iv.stx;
    for (size_t i = 0; i < x.size(); i++) trial[i] = x[i] + stp * dir[i];
    const T fcur = f(trial);
    g(trial, gradient);
    nfev++;
    const T dg = math::dot(gradient.data(), dir.data(), n);
    const T ftest1 = finit + stp * dgtest;
    if ((iv.brackt & ((stp <= stmin) | (stp >= stmax))) | (infoc == 0)) info = 6;
    if ((stp == stpmax) & (fcur <= ftest1) & (dg <= dgtest)) info = 5;
    if ((stp == stpmin) & ((fcur > ftest1) | (dg >= dgtest))) info = 4;
    if (nfev >= maxfev) info = 3;
    if (iv.brackt & (stmax - stmin <= xtol * stmax)) info = 2;
    if ((fcur <= ftest1) & (std::fabs(dg) <= gtol * (-dginit))) info = 1;
    if (info != 0) return stp;
    if (stage1 & (fcur <= ftest1) & (dg >= std::min<T>(ftol, gtol) * dginit)) stage1 = false;
    if (stage1 & (fcur <= iv.fx) & (fcur > ftest1)) {
      mt_interval<T> m{iv.stx, iv.fx - iv.stx * dgtest, iv.dx - dgtest,
                       iv.sty, iv.fy - iv.sty * dgtest, iv.dy - dgtest, iv.brackt};
      cstep(m, stp, fcur - stp * dgtest, dg - dgtest, stmin, stmax, infoc);
      iv = {m.stx, m.fx + m.stx * dgtest, m.dx + dgtest, m.sty, m.fy + m.sty * dgtest,
            m.dy + dgtest, m.brackt};
    } else {
      cstep(iv, stp, fcur, dg, stmin, stmax, infoc);
    }
    if (iv.brackt) {
      if (std::fabs(iv.sty - iv.stx) >= 0.66 * width1) stp = iv.stx + 0.5 * (iv.sty - iv.stx);
      width1 = width;
      width = std::fabs(iv.sty - iv.stx);
    }
  }
}
}  // namespace linesearch

template <typename Callable, typename scalar_t = double,
          typename Grad = std::conditional_t<device::has_grad_objective<Callable>::value,
                                             device::analytic_grad, fin_diff<Callable, scalar_t>>>
class BFGS {
  Callable &f;
  Grad g;
  const size_t max_iter;
  const scalar_t grad_eps, alpha;

 public:
  // same positional arguments and defaults as nlsolver.h:3181-3185
  explicit BFGS(Callable &f, Grad g = Grad(), const size_t max_iter = 100,
                const scalar_t grad_eps = 5e-3, const scalar_t alpha = 1)
      : f(f), g(g), max_iter(max_iter), grad_eps(grad_eps), alpha(alpha) {}
  // Device coverage of the default-gradient path (fin_diff on a built-in objective): the
  // objectives whose arithmetic is deterministic on the device (up to the engine's 1024 dimensions).
  static constexpr bool device_fd() {
    if constexpr (device::is_device_objective<Callable>::value &&
                  std::is_same_v<Grad, fin_diff<Callable, scalar_t>>)
      return Callable::nlsg_objective == NLSG_OBJ_ROSENBROCK ||
             Callable::nlsg_objective == NLSG_OBJ_SPHERE ||
             Callable::nlsg_objective == NLSG_OBJ_STYBLINSKI_TANG ||
             Callable::nlsg_objective == NLSG_OBJ_RASTRIGIN ||
             Callable::nlsg_objective == NLSG_OBJ_CUSTOM;
    else
      return false;
  }
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) {
    if constexpr (device::has_grad_objective<Callable>::value) {
      std::vector<std::vector<scalar_t>> one{x};
      auto st = solve_device(one);
      x = one[0];
      return st[0];
    } else if constexpr (device_fd()) {
      if constexpr (Callable::nlsg_objective != NLSG_OBJ_CUSTOM)  // (Custom has no host evaluation)
        if (x.size() > 1024) return solve_host(x);  // beyond the device coverage: host functor path
      std::vector<std::vector<scalar_t>> one{x};
      auto st = solve_device(one);
      x = one[0];
      return st[0];
    } else {
      return solve_host(x);
    }
  }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &) {
    static_assert(sizeof(Callable) == 0, "BFGS currently only supports minimization");  // :3199
    return solver_status<scalar_t>(0, 0, 0);
  }
  // Extension (BASELINE config 3): `xs.size()` independent starts solved in lock step on
  // the GPU; the reference solves one start per minimize() call.
  // Summation order: device::summation().
  std::vector<solver_status<scalar_t>> minimize_batch(std::vector<std::vector<scalar_t>> &xs) {
    return solve_device(xs);
  }
  // The reference's arithmetic exists on the device for the objectives given by their terms
  // (Rastrigin's device cosine is not libm's; a Custom body is the user's own arithmetic).
  static constexpr bool has_reference_order() {
    if constexpr (device::has_grad_objective<Callable>::value)
      return true;
    else if constexpr (device_fd())
      return Callable::nlsg_objective != NLSG_OBJ_RASTRIGIN && Callable::nlsg_objective != NLSG_OBJ_CUSTOM;
    else
      return false;
  }

 private:
  // device::summation(): reference order wherever the reference's arithmetic exists on the device —
  // also for a Custom objective given by its terms, where index order is what the body's own loop
  // on a CPU would do.
  bool use_reference_order() const {
    if constexpr (device::has_grad_objective<Callable>::value) {
      return device::reference_order();
    } else if constexpr (device_fd()) {
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM)
        return f.chain != NLSG_CUSTOM_VECTOR && device::reference_order();
      else
        return has_reference_order() && device::reference_order();
    } else {
      return false;
    }
  }
  std::vector<solver_status<scalar_t>> solve_device(std::vector<std::vector<scalar_t>> &xs) {
    static_assert(device::has_grad_objective<Callable>::value || device_fd(),
                  "minimize_batch needs a device objective: one with an analytic gradient, or "
                  "Rosenbrock / Sphere / StyblinskiTang with the default finite-difference one");
    static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
    const device::api &api = device::api::get();
    const size_t B = xs.size(), n = B ? xs[0].size() : 0;
    nlsg_bfgs_config cfg{};
    cfg.struct_size = sizeof(cfg);
    if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
    if (use_reference_order()) cfg.flags |= NLSG_BFGS_REFERENCE_ORDER;
    cfg.batch = B;
    cfg.dim = n;
    cfg.max_iter = max_iter;
    cfg.grad_eps = grad_eps;
    cfg.alpha = alpha;
    nlsg_bfgs *eng = nullptr;
    if constexpr (device::has_grad_objective<Callable>::value) {
      cfg.objective = Callable::nlsg_grad_objective;
      cfg.quad_c = f.c;
      api.check(api.bfgs_create(&cfg, f.d.data(), f.b.data(), &eng));
    } else {  // fin_diff (nlsolver.h:2849-2855) evaluated on the device
      cfg.objective = Callable::nlsg_objective;
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
        api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
        nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
        api.check(api.bfgs_create_custom(&cfg, &obj, &eng));
      } else {
        api.check(api.bfgs_create(&cfg, nullptr, nullptr, &eng));
      }
    }
    std::vector<scalar_t> flat(B * n);
    for (size_t p = 0; p < B; p++) std::copy(xs[p].begin(), xs[p].end(), flat.begin() + p * n);
    std::vector<nlsg_status> st(B);
    const int rc = api.bfgs_minimize(eng, flat.data(), st.data());
    const std::string msg = rc ? api.last_error() : "";
    api.bfgs_destroy(eng);
    if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
    std::vector<solver_status<scalar_t>> out;
    for (size_t p = 0; p < B; p++) {
      std::copy(flat.begin() + p * n, flat.begin() + (p + 1) * n, xs[p].begin());
      out.emplace_back(st[p].f_value, st[p].iteration, st[p].function_calls_used,
                       st[p].gradient_evals_used);
    }
    return out;
  }

  // Host path for arbitrary callables: BFGS::solve (nlsolver.h:3196-3285) incl. the
  // literal rank-2 update of update_inverse_hessian (3130-3168).
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x) {
    const size_t n = x.size();
    const int ni = static_cast<int>(n);
    std::vector<scalar_t> H(n * n, 0.0), dir(n, 0.0), grad(n, 0.0), prev_grad(n, 0.0), y(n, 0.0),
        s(n, 0.0), trial(n, 0.0), t(n);
    for (size_t i = 0; i < n; i++) H[i + i * n] = 1.0;
    size_t iter = 0, f_calls = 0, g_calls = 0;
    auto f_counted = [&](std::vector<scalar_t> &at) {
      f_calls++;
      return f(at);
    };
    auto g_counted = [&](std::vector<scalar_t> &at, std::vector<scalar_t> &out) {
      g_calls++;
      if constexpr (std::is_same_v<Grad, fin_diff<Callable, scalar_t>>) {
        fin_diff<decltype(f_counted), scalar_t>()(f_counted, at, out);  // counts f calls too
      } else {
        g(f, at, out);
      }
    };
    g_counted(x, grad);
    scalar_t prev_norm = 1e9, cur_norm = 1e8;
    for (;;) {
      if (iter >= max_iter || cur_norm < grad_eps || std::abs(cur_norm - prev_norm) < grad_eps ||
          std::isinf(cur_norm)) {
        const scalar_t f_final = f_counted(x);  // sequenced before f_calls is read
        return solver_status<scalar_t>(f_final, iter, f_calls, g_calls);
      }
      for (size_t j = 0; j < n; j++) dir[j] = -math::dot(H.data() + j * n, grad.data(), ni);
      const scalar_t phi = math::dot(grad.data(), dir.data(), ni);
      if ((phi > 0) || std::isnan(phi) || cur_norm > prev_norm) {
        std::fill(H.begin(), H.end(), 0.0);
        for (size_t i = 0; i < n; i++) {
          H[i + i * n] = 1.0;
          dir[i] = -grad[i];
        }
      }
      prev_grad = grad;
      const scalar_t rate =
          linesearch::more_thuente_search(f_counted, x, grad, dir, trial, alpha, g_counted);
      for (size_t i = 0; i < n; i++) s[i] = dir[i] * rate;
      for (size_t i = 0; i < n; i++) x[i] += s[i];
      g_counted(x, grad);
      prev_norm = cur_norm;
      cur_norm = math::norm(grad.data(), ni);
      for (size_t i = 0; i < n; i++) y[i] = grad[i] - prev_grad[i];
      scalar_t rho = math::dot(y.data(), s.data(), ni);
      rho = 1 / rho;
      for (size_t i = 0; i < n; i++) t[i] = math::dot(y.data(), H.data() + i * n, ni);
      scalar_t denom = math::dot(y.data(), t.data(), ni);
      denom = (denom * rho) + 1.0;
      for (size_t j = 0; j < n; j++)
        for (size_t i = 0; i < n; i++)
          H[j * n + i] = H[j * n + i] - rho * (s[i] * t[j] + t[i] * s[j] + denom * s[i] * s[j]);
      iter++;
    }
  }
};

// ---------------------------------------------------------------------------
// SANN — nlsolver.h:2744-2815
// ---------------------------------------------------------------------------
template <typename Callable, typename RNG, typename scalar_t = double>
class SANN {
  RNG &generator;
  Callable &f;
  size_t f_evals;  // accumulates across calls, like the reference's member (nlsolver.h:2751)
  const size_t max_iter, temperature_iter;
  const scalar_t temperature_max;

 public:
  // same positional arguments and defaults as nlsolver.h:2759-2761
  SANN(Callable &f, RNG &generator, const size_t max_iter = 5000,
       const size_t temperature_iter = 10, const scalar_t temperature_max = 10.0)
      : generator(generator), f(f), f_evals(0), max_iter(max_iter),
        temperature_iter(temperature_iter), temperature_max(temperature_max) {}
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) { return run<true>(x); }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x) { return run<false>(x); }
  // Extension: `xs.size()` independent chains annealed side by side on the GPU (chain b is
  // keyed by (seed, b)); the reference runs one chain per call.
  std::vector<solver_status<scalar_t>> minimize_batch(std::vector<std::vector<scalar_t>> &xs) {
    return run_device<true>(xs);
  }
  std::vector<solver_status<scalar_t>> maximize_batch(std::vector<std::vector<scalar_t>> &xs) {
    return run_device<false>(xs);
  }

 private:
  template <bool minimize>
  solver_status<scalar_t> run(std::vector<scalar_t> &x) {
    if constexpr (device::is_device_objective<Callable>::value) {
      std::vector<std::vector<scalar_t>> one{x};
      auto st = run_device<minimize>(one);
      x = one[0];
      return st[0];
    } else {
      return solve_host<minimize>(x);
    }
  }
  template <bool minimize>
  std::vector<solver_status<scalar_t>> run_device(std::vector<std::vector<scalar_t>> &xs) {
    static_assert(device::is_device_objective<Callable>::value,
                  "the batched chains need a device objective");
    static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
    const device::api &api = device::api::get();
    const size_t B = xs.size(), n = B ? xs[0].size() : 0;
    nlsg_sann_config cfg{};
    cfg.struct_size = sizeof(cfg);
    if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
    cfg.objective = Callable::nlsg_objective;
    cfg.minimize = minimize ? 1 : 0;
    cfg.batch = B;
    cfg.dim = n;
    cfg.max_iter = max_iter;
    cfg.temperature_iter = temperature_iter;
    cfg.temperature_max = temperature_max;
    cfg.seed = device::seed_from(generator);
    nlsg_sann *eng = nullptr;
    if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
      api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
      nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
      api.check(api.sann_create_custom(&cfg, &obj, &eng));
    } else {
      api.check(api.sann_create(&cfg, &eng));
    }
    std::vector<scalar_t> flat(B * n);
    for (size_t b = 0; b < B; b++) std::copy(xs[b].begin(), xs[b].end(), flat.begin() + b * n);
    std::vector<nlsg_status> st(B);
    const int rc = api.sann_minimize(eng, flat.data(), st.data());
    const std::string msg = rc ? api.last_error() : "";
    api.sann_destroy(eng);
    if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
    std::vector<solver_status<scalar_t>> out;
    for (size_t b = 0; b < B; b++) {
      std::copy(flat.begin() + b * n, flat.begin() + (b + 1) * n, xs[b].begin());
      f_evals += st[b].function_calls_used;
      out.emplace_back(st[b].f_value, st[b].iteration, st[b].function_calls_used);
    }
    return out;
  }

  // Host path for arbitrary callables: SANN::solve (nlsolver.h:2777-2814). Trial points are
  // normal steps around the current point scaled by the temperature; a trial is compared with
  // the BEST value so far (not the current point's), accepted when not worse or with
  // probability exp(-difference / t); the uniform draw is only taken when it is worse.
  template <bool minimize>
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x) {
    constexpr scalar_t sign = minimize ? 1.0 : -1.0, e_minus_1 = 1.7182818;
    const size_t n = x.size();
    scalar_t best = sign * f(x);
    f_evals++;
    const scalar_t inv_max = 1.0 / temperature_max;
    std::vector<scalar_t> current = x, trial = x;
    for (size_t iter = 0; iter < max_iter; iter++) {
      const scalar_t t = temperature_max / std::log(static_cast<scalar_t>(iter) + e_minus_1);
      for (size_t j = 1; j < temperature_iter; j++) {
        const scalar_t spread = t * inv_max;
        for (size_t i = 0; i < n; i++) trial[i] = current[i] + spread * rnorm<scalar_t>(generator);
        const scalar_t value = sign * f(trial);
        f_evals++;
        const scalar_t difference = value - best;
        if (difference <= 0.0 || generator() < std::exp(-difference / t)) {
          current = trial;
          if (value <= best) {
            x = current;
            best = value;
          }
        }
      }
    }
    return solver_status<scalar_t>(best, max_iter, f_evals);
  }
};

// ---------------------------------------------------------------------------
// NelderMeadPSO — nlsolver.h:3546-3920
// ---------------------------------------------------------------------------
template <typename Callable, typename RNG, typename scalar_t = double>
class NelderMeadPSO {
  RNG &generator;
  Callable &f;
  const scalar_t alpha, gamma, rho, sigma, inertia, cognitive_coef, social_coef;
  scalar_t eps;
  const size_t max_iter, no_change_best_iter;

 public:
  // same positional arguments and defaults as nlsolver.h:3563-3569
  NelderMeadPSO(Callable &f, RNG &generator, const scalar_t alpha = 1, const scalar_t gamma = 2,
                const scalar_t rho = 0.5, const scalar_t sigma = 0.5, const scalar_t inertia = 0.8,
                const scalar_t cognitive_coef = 1.8, const scalar_t social_coef = 1.8,
                const scalar_t eps = 1e-6, const size_t max_iter = 1000,
                const size_t no_change_best_iter = 20)
      : generator(generator), f(f), alpha(alpha), gamma(gamma), rho(rho), sigma(sigma),
        inertia(inertia), cognitive_coef(cognitive_coef), social_coef(social_coef), eps(eps),
        max_iter(max_iter), no_change_best_iter(no_change_best_iter) {}
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) { return unbounded<true>(x); }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x) { return unbounded<false>(x); }
  // note the argument order (lower, upper), nlsolver.h:3609-3612
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x, const std::vector<scalar_t> &lower,
                                   const std::vector<scalar_t> &upper) {
    return run<true, true>(x, upper, lower);
  }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x, const std::vector<scalar_t> &lower,
                                   const std::vector<scalar_t> &upper) {
    return run<false, true>(x, upper, lower);
  }
  // Extension: `xs.size()` independent instances side by side on the GPU (instance b is keyed
  // by (seed, b)); the reference solves one per call.
  std::vector<solver_status<scalar_t>> minimize_batch(std::vector<std::vector<scalar_t>> &xs) {
    std::vector<scalar_t> none;
    return run_device<true, false>(xs, none, none);
  }

 private:
  template <bool minimize>
  solver_status<scalar_t> unbounded(std::vector<scalar_t> &x) {
    std::vector<scalar_t> lower(x.size()), upper(x.size());
    for (size_t i = 0; i < x.size(); i++) {  // implied bounds, nlsolver.h:3587-3593
      const scalar_t reach = std::abs(2.5 * x[i]);
      lower[i] = -reach;
      upper[i] = reach;
    }
    return run<minimize, false>(x, upper, lower);
  }
  template <bool minimize, bool bound>
  solver_status<scalar_t> run(std::vector<scalar_t> &x, const std::vector<scalar_t> &upper,
                              const std::vector<scalar_t> &lower) {
    if (x.size() < 2) {  // nlsolver.h:3627-3637
      std::cout << "You are trying to optimize a one dimensional function; use NelderMead or PSO: "
                   "the NelderMead-PSO hybrid does not support this."
                << std::endl;
      return solver_status<scalar_t>(999999, 0, 0);
    }
    if constexpr (device::is_device_objective<Callable>::value) {
      std::vector<std::vector<scalar_t>> one{x};
      auto st = run_device<minimize, bound>(one, upper, lower);
      x = one[0];
      return st[0];
    } else {
      return solve_host<minimize, bound>(x, upper, lower);
    }
  }
  template <bool minimize, bool bound>
  std::vector<solver_status<scalar_t>> run_device(std::vector<std::vector<scalar_t>> &xs,
                                                  const std::vector<scalar_t> &upper,
                                                  const std::vector<scalar_t> &lower) {
    static_assert(device::is_device_objective<Callable>::value,
                  "the batched hybrid needs a device objective");
    static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
    const device::api &api = device::api::get();
    const size_t B = xs.size(), n = B ? xs[0].size() : 0;
    nlsg_nmpso_config cfg{};
    cfg.struct_size = sizeof(cfg);
    if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
    cfg.objective = Callable::nlsg_objective;
    cfg.minimize = minimize ? 1 : 0;
    cfg.bounded = bound ? 1 : 0;
    cfg.batch = B;
    cfg.dim = n;
    cfg.alpha = alpha;
    cfg.gamma = gamma;
    cfg.rho = rho;
    cfg.sigma = sigma;
    cfg.inertia = inertia;
    cfg.cognitive = cognitive_coef;
    cfg.social = social_coef;
    cfg.eps = eps;
    cfg.max_iter = max_iter;
    cfg.no_change_best_iter = no_change_best_iter;
    cfg.seed = device::seed_from(generator);
    nlsg_nmpso *eng = nullptr;
    if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
      api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
      nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
      api.check(api.nmpso_create_custom(&cfg, &obj, &eng));
    } else {
      api.check(api.nmpso_create(&cfg, &eng));
    }
    std::vector<scalar_t> flat(B * n);
    for (size_t b = 0; b < B; b++) std::copy(xs[b].begin(), xs[b].end(), flat.begin() + b * n);
    std::vector<nlsg_status> st(B);
    const int rc = api.nmpso_minimize(eng, flat.data(), bound ? lower.data() : nullptr,
                                      bound ? upper.data() : nullptr, st.data());
    const std::string msg = rc ? api.last_error() : "";
    api.nmpso_destroy(eng);
    if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
    std::vector<solver_status<scalar_t>> out;
    for (size_t b = 0; b < B; b++) {
      std::copy(flat.begin() + b * n, flat.begin() + (b + 1) * n, xs[b].begin());
      out.emplace_back(st[b].f_value, st[b].iteration, st[b].function_calls_used);
    }
    return out;
  }

  // Host path for arbitrary callables: NelderMeadPSO::solve (nlsolver.h:3623-3685) with
  // init_solver_state 3686-3738, apply_simplex 3739-3822, apply_pso 3823-3866. 3n + 1
  // particles: every iteration sorts them, runs one Nelder-Mead step on the best n + 1 and a PSO
  // move on the other 2n. Kept as the reference behaves: the last simplex particle stays at x
  // (its write of element [n][n] is out of bounds and dropped); the no-change counter compares
  // with the first particle's initial value, which is never refreshed; the PSO move works on a
  // copy of the velocity, so velocities keep their initial values; the "better particle of a
  // pair" is the pair's second one (the first pair: its first). The bounded overloads clamp the
  // velocity by coordinate (the reference indexes the bounds with the particle's rank, out of
  // range).
  template <bool minimize, bool bound>
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x, const std::vector<scalar_t> &upper,
                                     const std::vector<scalar_t> &lower) {
    constexpr scalar_t sign = minimize ? 1.0 : -1.0;
    const size_t n = x.size(), n_simplex = n + 1, n_all = 3 * n + 1;
    size_t calls = 0;
    auto value = [&](std::vector<scalar_t> &at) {
      calls++;
      return sign * f(at);
    };
    std::vector<std::vector<scalar_t>> where(n_all, x), speed(n_all, std::vector<scalar_t>(n, 0.0));
    std::vector<scalar_t> score(n_all);
    {
      scalar_t inf_norm = std::abs(x[0]);
      for (size_t i = 1; i < n; i++) inf_norm = inf_norm < std::abs(x[i]) ? std::abs(x[i]) : inf_norm;
      const scalar_t a = inf_norm < 1.0 ? 1.0 : inf_norm;
      const scalar_t spread = a < 10 ? a : 10;
      for (size_t i = 1; i < n; i++) where[i][i] = x[i] + spread;
      const auto nn = static_cast<scalar_t>(n);
      for (size_t i = 0; i < n; i++) where[0][i] = x[i] + ((1.0 - std::sqrt(nn + 1.0)) / nn * spread);
    }
    for (size_t i = n_simplex; i < n_all; i++)
      for (size_t j = 0; j < n; j++) {
        const scalar_t width = std::abs(upper[j] - lower[j]);
        where[i][j] = lower[j] + ((upper[j] - lower[j]) * generator());
        speed[i][j] = -width + (generator() * width);
      }
    for (size_t i = 0; i < n_all; i++) score[i] = value(where[i]);
    std::vector<size_t> rank(n_all);
    for (size_t i = 0; i < n_all; i++) rank[i] = i;
    auto by_score = [&](size_t l, size_t r) { return score[l] < score[r]; };
    auto transform = [&](const std::vector<scalar_t> &pt, const std::vector<scalar_t> &c,
                         std::vector<scalar_t> &out, scalar_t coef, bool reflect) {
      for (size_t i = 0; i < n; i++) {
        scalar_t t = reflect ? c[i] + coef * (c[i] - pt[i]) : c[i] + coef * (pt[i] - c[i]);
        if constexpr (bound) t = std::clamp(t, lower[i], upper[i]);
        out[i] = t;
      }
    };
    const scalar_t first_value = score[0];
    size_t iter = 0, unchanged = 0;
    std::vector<scalar_t> centroid(n), refl(n), expd(n), cont(n);
    for (;;) {
      std::sort(rank.begin(), rank.end(), by_score);
      const bool same = first_value == score[rank[0]];
      unchanged = same ? unchanged + 1 : 0;
      scalar_t mean = 0, dev = 0;
      for (size_t i = 0; i < n_simplex; i++) mean += score[rank[i]];
      mean /= static_cast<scalar_t>(n_simplex);
      for (size_t i = 0; i < n_simplex; i++) dev += std::pow(score[rank[i]] - mean, 2);
      dev = std::sqrt(dev / static_cast<scalar_t>(n_simplex - 1));
      if (iter >= max_iter || unchanged >= no_change_best_iter || dev < eps) {
        x = where[rank[0]];
        return solver_status<scalar_t>(score[rank[0]], iter, calls);
      }
      {  // the simplex step on the best n + 1 particles
        const scalar_t best_score = score[rank[0]];
        const size_t worst = rank[n_simplex - 1], second = rank[n_simplex - 2];
        std::fill(centroid.begin(), centroid.end(), 0.0);
        for (size_t i = 0; i + 1 < n_simplex; i++)
          for (size_t j = 0; j < n; j++) centroid[j] += where[rank[i]][j];
        for (auto &c : centroid) c /= static_cast<scalar_t>(n_simplex - 1);
        transform(where[worst], centroid, refl, alpha, true);
        const scalar_t refl_score = value(refl);
        if (refl_score >= best_score && refl_score < score[second]) {
          where[worst] = refl;
          score[worst] = refl_score;
        } else if (refl_score < best_score) {
          transform(refl, centroid, expd, gamma, false);
          const scalar_t expd_score = value(expd);
          where[worst] = expd_score < refl_score ? expd : refl;
          score[worst] = expd_score < refl_score ? expd_score : refl_score;
        } else {
          const scalar_t worst_score = score[worst];
          transform(refl_score < worst_score ? refl : where[worst], centroid, cont, rho, false);
          const scalar_t cont_score = value(cont);
          if (cont_score < std::min(refl_score, worst_score)) {
            where[worst] = cont;
            score[worst] = cont_score;
          } else {
            const std::vector<scalar_t> &best = where[rank[0]];
            for (size_t i = 1; i < n_simplex; i++)
              for (size_t j = 0; j < n; j++)
                where[rank[i]][j] = best[j] + sigma * (where[rank[i]][j] - best[j]);
            for (size_t i = 1; i < n_simplex; i++) score[rank[i]] = value(where[rank[i]]);
            std::sort(rank.begin(), rank.end(), by_score);
          }
        }
      }
      {  // the PSO move of the other 2n particles
        const std::vector<scalar_t> &best = where[rank[0]];
        size_t partner = rank[n_simplex];
        bool take_next = false;
        for (size_t i = n_simplex; i < n_all; i++) {
          const size_t id = rank[i];
          if (take_next) partner = rank[i + 1];
          take_next = ((i - n_simplex) % 2) != 0;
          const std::vector<scalar_t> pair = where[partner];  // a copy, taken before the move
          for (size_t j = 0; j < n; j++) {
            const scalar_t r_p = generator(), r_g = generator();
            scalar_t step = (inertia * speed[id][j]) + cognitive_coef * r_p * (pair[j] - where[id][j]) +
                            social_coef * r_g * (best[j] - where[id][j]);
            if constexpr (bound) step = std::clamp(step, lower[j], upper[j]);
            where[id][j] += step;
          }
          score[id] = value(where[id]);
        }
      }
      iter++;
    }
  }
};

// ---------------------------------------------------------------------------
// NelderMead — nlsolver.h:1894-2300
// ---------------------------------------------------------------------------
template <typename Callable, typename scalar_t = double>
class NelderMead {
  Callable &f;
  const scalar_t step, alpha, gamma, rho, sigma;
  scalar_t eps;  // rescaled by every solve (nlsolver.h:2189)
  const size_t max_iter, no_change_best_tol, restarts;

 public:
  // same positional arguments and defaults as nlsolver.h:2110-2115
  explicit NelderMead(Callable &f, const scalar_t step = -1, const scalar_t alpha = 1,
                      const scalar_t gamma = 2, const scalar_t rho = 0.5,
                      const scalar_t sigma = 0.5, const scalar_t eps = 1e-6,
                      const size_t max_iter = 500, const size_t no_change_best_tol = 20,
                      const size_t restarts = 0)
      : f(f), step(step), alpha(alpha), gamma(gamma), rho(rho), sigma(sigma), eps(eps),
        max_iter(max_iter), no_change_best_tol(no_change_best_tol), restarts(restarts) {}
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) {
    std::vector<scalar_t> none;
    return run<true, false>(x, none, none);
  }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x) {
    std::vector<scalar_t> none;
    return run<false, false>(x, none, none);
  }
  // note the argument order (upper, lower), nlsolver.h:2136 / 2155
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x, const std::vector<scalar_t> &upper,
                                   const std::vector<scalar_t> &lower) {
    return run<true, true>(x, upper, lower);
  }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &x, const std::vector<scalar_t> &upper,
                                   const std::vector<scalar_t> &lower) {
    return run<false, true>(x, upper, lower);
  }

 private:
  template <bool minimize, bool bound>
  solver_status<scalar_t> run(std::vector<scalar_t> &x, const std::vector<scalar_t> &upper,
                              const std::vector<scalar_t> &lower) {
    if constexpr (device::is_device_objective<Callable>::value) {
      static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
      const device::api &api = device::api::get();
      nlsg_nm_config cfg{};
      cfg.struct_size = sizeof(cfg);
      if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
      cfg.objective = Callable::nlsg_objective;
      cfg.minimize = minimize ? 1 : 0;
      cfg.bounded = bound ? 1 : 0;
      // device::summation(): the objective's terms and std_err's sums in the reference's index order
      // wherever its arithmetic exists on the device — the reference's runs bit for bit
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
        if (f.chain != NLSG_CUSTOM_VECTOR && device::reference_order()) cfg.flags |= NLSG_NM_REFERENCE_ORDER;
      } else if constexpr (Callable::nlsg_objective != NLSG_OBJ_RASTRIGIN) {
        if (device::reference_order()) cfg.flags |= NLSG_NM_REFERENCE_ORDER;
      }
      cfg.batch = 1;
      cfg.dim = x.size();
      cfg.step = step;
      cfg.alpha = alpha;
      cfg.gamma = gamma;
      cfg.rho = rho;
      cfg.sigma = sigma;
      cfg.eps = eps;
      cfg.max_iter = max_iter;
      cfg.no_change_best_tol = no_change_best_tol;
      cfg.restarts = restarts;
      nlsg_nm *eng = nullptr;
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
        api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
        nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
        api.check(api.nm_create_custom(&cfg, &obj, &eng));
      } else {
        api.check(api.nm_create(&cfg, &eng));
      }
      nlsg_status st{};
      double eps_after = eps;
      const int rc = api.nm_minimize(eng, x.data(), bound ? upper.data() : nullptr,
                                     bound ? lower.data() : nullptr, &st, &eps_after);
      const std::string msg = rc ? api.last_error() : "";
      api.nm_destroy(eng);
      if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
      eps = eps_after;
      return solver_status<scalar_t>(st.f_value, st.iteration, st.function_calls_used);
    } else {
      auto res = solve_host<minimize, bound>(x, upper, lower);
      for (size_t i = 0; i < restarts; i++) res.add(solve_host<minimize, bound>(x, upper, lower));
      return res;
    }
  }

  // Host path: NelderMead::solve (nlsolver.h:2166-2299) with the effective initial simplex
  // of the reference (its write to element [n][n] is out of bounds and is dropped), the eps
  // rescaling, the running second-worst rule, contraction with the reflect transform and
  // the centroid that is only recomputed when the worst vertex changed.
  template <bool minimize, bool bound>
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x, const std::vector<scalar_t> &upper,
                                     const std::vector<scalar_t> &lower) {
    const size_t n = x.size(), nv = n + 1;
    std::vector<std::vector<scalar_t>> S(nv, x);
    scalar_t spread = step;
    if (step < 0) {
      scalar_t inf_norm = std::abs(x[0]);
      for (size_t i = 1; i < n; i++) inf_norm = inf_norm < std::abs(x[i]) ? std::abs(x[i]) : inf_norm;
      const scalar_t a = inf_norm < 1.0 ? 1.0 : inf_norm;
      spread = a < 10 ? a : 10;
    }
    for (size_t i = 1; i < n; i++) S[i][i] += spread;
    if (step < 0) {
      const auto nn = static_cast<scalar_t>(n);
      for (size_t i = 0; i < n; i++) S[0][i] = x[i] + ((1.0 - std::sqrt(nn + 1.0)) / nn * spread);
    }
    size_t calls = 0;
    auto value = [&](std::vector<scalar_t> &at) {
      constexpr scalar_t sign = minimize ? 1.0 : -1.0;
      calls++;
      return sign * f(at);
    };
    auto transform = [&](const std::vector<scalar_t> &pt, const std::vector<scalar_t> &c,
                         std::vector<scalar_t> &out, scalar_t coef, bool reflect) {
      for (size_t i = 0; i < n; i++) {
        scalar_t t = reflect ? c[i] + coef * (c[i] - pt[i]) : c[i] + coef * (pt[i] - c[i]);
        if constexpr (bound) t = std::clamp(t, lower[i], upper[i]);
        out[i] = t;
      }
    };
    std::vector<scalar_t> scores(nv);
    for (size_t v = 0; v < nv; v++) scores[v] = value(S[v]);
    eps = eps * (scores[0] * eps);
    size_t best, worst = 0, second = 0, prev_worst = 0, last_best = 99999999, stale = 0, iter = 0;
    std::vector<scalar_t> centroid(n), refl(n), expd(n), cont(n);
    bool shrunk = false;
    for (;;) {
      best = 0;
      prev_worst = worst;
      worst = 0;
      second = 0;
      const scalar_t spread_f = std_err(scores);
      for (size_t i = 1; i < nv; i++) {
        if (scores[i] < scores[best]) {
          best = i;
        } else if (scores[i] > scores[worst]) {
          second = worst;
          worst = i;
        }
      }
      if (last_best == best) {
        stale++;
      } else {
        stale = 0;
        last_best = best;
      }
      if (iter >= max_iter || spread_f < eps || stale >= no_change_best_tol) {
        x = S[best];
        return solver_status<scalar_t>(scores[best], iter, calls);
      }
      iter++;
      if (prev_worst != worst || shrunk) {
        std::fill(centroid.begin(), centroid.end(), 0.0);
        for (size_t v = 0; v < nv; v++)
          if (v != worst)
            for (size_t j = 0; j < n; j++) centroid[j] += S[v][j];
        for (auto &c : centroid) c /= static_cast<scalar_t>(nv - 1);
        shrunk = false;
      }
      transform(S[worst], centroid, refl, alpha, true);
      const scalar_t r_score = value(refl);
      if (r_score >= scores[best] && r_score < scores[second]) {
        S[worst] = refl;
        scores[worst] = r_score;
      } else if (r_score < scores[best]) {
        transform(refl, centroid, expd, gamma, false);
        const scalar_t e_score = value(expd);
        S[worst] = e_score < r_score ? expd : refl;
        scores[worst] = e_score < r_score ? e_score : r_score;
      } else {
        const bool outside = r_score < scores[worst];
        transform(outside ? refl : S[worst], centroid, cont, rho, true);
        const scalar_t c_score = value(cont);
        if (c_score < (outside ? r_score : scores[worst])) {
          S[worst] = cont;
          scores[worst] = c_score;
        } else {
          for (size_t v = 0; v < nv; v++)
            if (v != best)
              for (size_t j = 0; j < n; j++) S[v][j] = S[best][j] + sigma * (S[v][j] - S[best][j]);
          for (size_t v = 0; v < nv; v++)
            if (v != best) scores[v] = value(S[v]);
          shrunk = true;
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// LevenbergMarquardt — nlsolver.h:251-330, 1414-1517, 3428-3545
// ---------------------------------------------------------------------------
namespace finite_difference {
// 16-point cross stencil of the default Hess functor (nlsolver.h:1447-1515 with accuracy = 1):
// weights -63, +63, +44, +74 over the offsets below, divided by 600 eps^2, eps = epsilon^(1/4).
// The offsets are applied to the saved x_i, x_j (the reference walks x incrementally).
template <typename Callable, typename scalar_t>
void finite_difference_hessian(Callable &f, std::vector<scalar_t> &x, std::vector<scalar_t> &hess) {
  const scalar_t eps = std::pow(std::numeric_limits<scalar_t>::epsilon(), static_cast<scalar_t>(0.25));
  static constexpr int st[16][3] = {
      {1, -2, -63}, {2, -1, -63}, {-2, 1, -63}, {-1, 2, -63}, {-1, -2, 63}, {-2, -1, 63},
      {1, 2, 63},   {2, 1, 63},   {2, -2, 44},  {-2, 2, 44},  {-2, -2, -44}, {2, 2, -44},
      {-1, -1, 74}, {1, 1, 74},   {1, -1, -74}, {-1, 1, -74}};
  const size_t p = x.size();
  for (size_t i = 0; i < p; i++)
    for (size_t j = 0; j < p; j++) {
      const scalar_t xi = x[i], xj = x[j];
      scalar_t acc = 0;
      for (const auto &k : st) {
        x[i] = xi;
        x[j] = xj;
        x[i] += k[0] * eps;
        x[j] += k[1] * eps;
        acc += k[2] * f(x);
      }
      x[i] = xi;
      x[j] = xj;
      hess[i * p + j] = acc / (600.0 * eps * eps);
    }
}
}  // namespace finite_difference

template <typename Callable, typename scalar_t>
struct fin_diff_h {  // nlsolver.h:2856-2863
  void operator()(Callable &f, std::vector<scalar_t> &x, std::vector<scalar_t> &hessian) {
    finite_difference::finite_difference_hessian<Callable, scalar_t>(f, x, hessian);
  }
};

namespace math {
// get_update_with_hessian (nlsolver.h:310-330): diagonal shortcut, else in-place Cholesky +
// forward / transposed back substitution (251-294). `hess` is overwritten.
template <typename T>
void get_update_with_hessian(std::vector<T> &update, std::vector<T> &hess, std::vector<T> &grad) {
  const size_t n = grad.size();
  bool diagonal = true;
  for (size_t i = 0; i < n && diagonal; i++)
    for (size_t j = 0; j < n; j++)
      if (i != j && hess[i * n + j] > std::numeric_limits<T>::epsilon() * 1e12) {
        diagonal = false;
        break;
      }
  if (diagonal) {
    for (size_t i = 0; i < n; i++) update[i] = grad[i] / hess[i * n + i];
    return;
  }
  for (size_t i = 0; i < n; ++i) {  // cholesky
    for (size_t j = 0; j < i; ++j) {
      T sum = 0;
      for (size_t k = 0; k < j; ++k) sum += hess[i * n + k] * hess[j * n + k];
      hess[i * n + j] = (1.0 / hess[j * n + j] * (hess[i * n + j] - sum));
    }
    T sum = 0;
    for (size_t k = 0; k < i; ++k) sum += hess[i * n + k] * hess[i * n + k];
    hess[i * n + i] = std::sqrt(hess[i * n + i] - sum);
  }
  std::fill(update.begin(), update.end(), 0.0);
  for (size_t i = 0; i < n; ++i) {  // L z = g
    T sum = 0.0;
    for (size_t j = 0; j < i; ++j) sum += hess[i * n + j] * update[j];
    update[i] = (grad[i] - sum) / hess[i + i * n];
  }
  for (size_t i = n; i-- > 0;) {  // L^T u = z
    T sum = 0.0;
    for (size_t j = i + 1; j < n; ++j) sum += hess[j * n + i] * update[j];
    update[i] = (update[i] - sum) / hess[i * n + i];
  }
}
}  // namespace math

template <typename Callable, typename scalar_t,
          typename Grad = std::conditional_t<device::has_nlls_objective<Callable>::value,
                                             device::gauss_newton, fin_diff<Callable, scalar_t>>,
          typename Hess = std::conditional_t<device::has_nlls_objective<Callable>::value,
                                             device::gauss_newton, fin_diff_h<Callable, scalar_t>>>
class LevenbergMarquardt {
  Callable &f;
  Grad g;
  Hess h;
  scalar_t lambda;  // mutable across calls, like the reference's member (nlsolver.h:3436)
  const scalar_t upward_mult, downward_mult;
  const size_t max_iter;
  const scalar_t f_delta;

 public:
  // same positional arguments and defaults as nlsolver.h:3443-3447
  explicit LevenbergMarquardt(Callable &f, const scalar_t lambda = 10,
                              const scalar_t upward_mult = 10, const scalar_t downward_mult = 10,
                              const size_t max_iter = 100, const scalar_t f_delta = 1e-12,
                              Grad g = Grad(), Hess h = Hess())
      : f(f), g(g), h(h), lambda(lambda), upward_mult(upward_mult), downward_mult(downward_mult),
        max_iter(max_iter), f_delta(f_delta) {}
  // Device coverage of the default functors (fin_diff + fin_diff_h on a built-in objective,
  // nlsolver.h:3494-3511): the objectives whose arithmetic is deterministic on the device, up
  // to the engine's 1024 parameters (past 64: a workgroup per problem instead of a wave).
  static constexpr bool device_fd() {
    if constexpr (device::is_device_objective<Callable>::value &&
                  std::is_same_v<Grad, fin_diff<Callable, scalar_t>> &&
                  std::is_same_v<Hess, fin_diff_h<Callable, scalar_t>>)
      return Callable::nlsg_objective == NLSG_OBJ_ROSENBROCK ||
             Callable::nlsg_objective == NLSG_OBJ_SPHERE ||
             Callable::nlsg_objective == NLSG_OBJ_STYBLINSKI_TANG ||
             Callable::nlsg_objective == NLSG_OBJ_RASTRIGIN ||
             Callable::nlsg_objective == NLSG_OBJ_CUSTOM;
    else
      return false;
  }
  solver_status<scalar_t> minimize(std::vector<scalar_t> &x) {
    if constexpr (device::has_nlls_objective<Callable>::value) {
      std::vector<std::vector<scalar_t>> one{x};
      auto st = solve_device(one);
      x = one[0];
      return st[0];
    } else if constexpr (device_fd()) {
      if constexpr (Callable::nlsg_objective != NLSG_OBJ_CUSTOM)  // (Custom has no host evaluation)
        if (x.size() > 1024) return solve_host(x);  // beyond the device coverage: host functor path
      std::vector<std::vector<scalar_t>> one{x};
      auto st = solve_device(one);
      x = one[0];
      return st[0];
    } else {
      return solve_host(x);
    }
  }
  solver_status<scalar_t> maximize(std::vector<scalar_t> &) {
    static_assert(sizeof(Callable) == 0,
                  "LevenbergMarquardt currently only supports minimization");  // :3468
    return solver_status<scalar_t>(0, 0, 0);
  }
  // Extension (BASELINE config 4): one start per problem of the model, all solved by one launch.
  // Summation order of the default-functor model: device::summation().
  std::vector<solver_status<scalar_t>> minimize_batch(std::vector<std::vector<scalar_t>> &thetas) {
    return solve_device(thetas);
  }
  // The reference's arithmetic exists on the device for the default functors on the objectives
  // given by their terms (not Rastrigin: its device cosine is not libm's; not Custom; not the
  // TanhRegression model, whose tanh is the device's).
  static constexpr bool has_reference_order() {
    if constexpr (device_fd())
      return Callable::nlsg_objective != NLSG_OBJ_RASTRIGIN && Callable::nlsg_objective != NLSG_OBJ_CUSTOM;
    else
      return false;
  }

 private:
  std::vector<solver_status<scalar_t>> solve_device(std::vector<std::vector<scalar_t>> &thetas) {
    static_assert(device::has_nlls_objective<Callable>::value || device_fd(),
                  "minimize_batch needs a device NLLS model, or Rosenbrock / Sphere / "
                  "StyblinskiTang with the default finite-difference functors");
    static_assert(std::is_same_v<scalar_t, double>, "the device path computes in fp64");
    const device::api &api = device::api::get();
    const size_t B = thetas.size();
    nlsg_lm_config cfg{};
    cfg.struct_size = sizeof(cfg);
    if (const char *d = std::getenv("NLSG_DEVICE")) cfg.device = std::atoi(d);
    cfg.batch = B;
    size_t n = 0;
    if constexpr (device::has_nlls_objective<Callable>::value) {
      if (B != f.problems()) throw device_error("one start per problem of the model is required");
      cfg.objective = Callable::nlsg_nlls_objective;
      cfg.m = f.m;
      n = f.n;
    } else {  // the objective itself, differentiated by fin_diff / fin_diff_h on the device
      cfg.objective = Callable::nlsg_objective;
      n = B ? thetas[0].size() : 0;
    }
    cfg.n = n;
    // (reference order: also for a Custom objective given by its terms, where index order is what
    // the body's own loop would do)
    bool ref = false;
    if constexpr (device_fd()) {
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM)
        ref = f.chain != NLSG_CUSTOM_VECTOR && device::reference_order();
      else
        ref = has_reference_order() && device::reference_order();
    }
    cfg.solver = ref ? NLSG_LM_CHOLESKY_REFERENCE_ORDER : NLSG_LM_CHOLESKY;
    cfg.lambda = lambda;
    cfg.up = upward_mult;
    cfg.down = downward_mult;
    cfg.max_iter = max_iter;
    cfg.f_delta = f_delta;
    nlsg_lm *eng = nullptr;
    bool made = false;
    if constexpr (device::is_device_objective<Callable>::value) {
      if constexpr (Callable::nlsg_objective == NLSG_OBJ_CUSTOM) {
        api.check(api.rtc_load(std::getenv("NLSG_HIPRTC")));
        nlsg_custom_objective obj{f.term_body.c_str(), f.finish_body.c_str(), f.chain, 0};
        api.check(api.lm_create_custom(&cfg, &obj, &eng));
        made = true;
      }
    }
    if (!made) api.check(api.lm_create(&cfg, &eng));
    std::vector<scalar_t> flat(B * n), lam(B);
    for (size_t p = 0; p < B; p++) std::copy(thetas[p].begin(), thetas[p].end(), flat.begin() + p * n);
    std::vector<nlsg_status> st(B);
    int rc = 0;
    if constexpr (device::has_nlls_objective<Callable>::value)
      rc = api.lm_set_data(eng, f.A.data(), f.y.data());
    if (!rc) rc = api.lm_minimize(eng, flat.data(), st.data(), lam.data());
    const std::string msg = rc ? api.last_error() : "";
    api.lm_destroy(eng);
    if (rc) throw device_error("nlsg error " + std::to_string(rc) + ": " + msg);
    std::vector<solver_status<scalar_t>> out;
    for (size_t p = 0; p < B; p++) {
      std::copy(flat.begin() + p * n, flat.begin() + (p + 1) * n, thetas[p].begin());
      out.emplace_back(st[p].f_value, st[p].iteration, st[p].function_calls_used,
                       st[p].gradient_evals_used, st[p].hessian_evals_used);
    }
    if (B == 1) lambda = lam[0];
    return out;
  }

  // Host path: LevenbergMarquardt::solve (nlsolver.h:3465-3544): damped Newton, the step is
  // always accepted, lambda / down on decrease else * up.
  solver_status<scalar_t> solve_host(std::vector<scalar_t> &x) {
    const size_t n = x.size();
    size_t iter = 0, f_calls = 0, g_calls = 0, h_calls = 0;
    std::vector<scalar_t> gradient(n, 0.0), hessian(n * n, 0.0), update(n);
    auto f_counted = [&](std::vector<scalar_t> &at) {
      f_calls++;
      return f(at);
    };
    auto g_counted = [&](std::vector<scalar_t> &at, std::vector<scalar_t> &out) {
      g_calls++;
      if constexpr (std::is_same_v<Grad, fin_diff<Callable, scalar_t>>)
        fin_diff<decltype(f_counted), scalar_t>()(f_counted, at, out);
      else
        g(f, at, out);
    };
    auto h_counted = [&](std::vector<scalar_t> &at, std::vector<scalar_t> &out) {
      h_calls++;
      if constexpr (std::is_same_v<Hess, fin_diff_h<Callable, scalar_t>>)
        fin_diff_h<decltype(f_counted), scalar_t>()(f_counted, at, out);
      else
        h(f, at, out);
    };
    g_counted(x, gradient);
    h_counted(x, hessian);
    scalar_t previous = 0.0, current = f_counted(x);
    for (;;) {
      const scalar_t delta = std::abs(previous - current);
      if (iter >= max_iter || delta < f_delta || std::isnan(previous))
        return solver_status<scalar_t>(current, iter, f_calls, g_calls, h_calls);
      for (size_t i = 0; i < n; i++) hessian[i * n + i] += lambda;
      math::get_update_with_hessian(update, hessian, gradient);
      for (size_t i = 0; i < n; i++) x[i] -= update[i];
      previous = current;
      current = f_counted(x);
      iter++;
      g_counted(x, gradient);
      h_counted(x, hessian);
      lambda = current < previous ? lambda / downward_mult : lambda * upward_mult;
    }
  }
};

}  // namespace nlsolver

#endif  // NLSOLVER_MI_NLSOLVER_H_
