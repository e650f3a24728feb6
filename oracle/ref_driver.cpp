// oracle/ref_driver.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Driver around the *unmodified* reference headers. It is compiled only by
// oracle/Makefile (target `ref`) with `-I$(REFERENCE_DIR)` (default
// /root/reference), i.e. `#include "nlsolver.h"` below resolves to the
// reference's own file where it lies; no reference source is copied into this
// repository. The resulting binary goes to oracle/_ref/ (git-ignored).
//
// Uses:
//   * tests/golden/gen_golden.py runs it to produce the committed golden
//     vectors (tests/golden/*.json) that pin oracle/ (the C restatement).
//   * bench.py may time its `bench-de` mode as `cpu_baseline.kind =
//     "reference"` on the GPU box's host cores.
//
// Every objective functor in this file is the driver's own code; the reference
// library treats it as an opaque callable (nlsolver.h:2383 `Callable &f`).
// Doubles are printed as C99 hexfloat strings ("%a") so goldens are bit-exact.
//
// Build flags: generic math path only (-DNO_MANUAL_VECTORIZATION, no
// -march=native), see SURVEY.md §8(c).

#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "nlsolver.h"  // resolved through -I$(REFERENCE_DIR)

using nlsolver::DE;
using nlsolver::rng::xorshift;
using DEStrat = nlsolver::RecombinationStrategy;

// ----------------------------------------------------------------------------
// helpers
// ----------------------------------------------------------------------------
static void put_hex(double v) { std::printf("\"%a\"", v); }
static void put_vec(const std::vector<double> &v) {
  std::printf("[");
  for (size_t i = 0; i < v.size(); i++) {
    if (i) std::printf(",");
    put_hex(v[i]);
  }
  std::printf("]");
}
// FNV-1a over the raw bits of a sequence of doubles: a compact bit-exact pin
// for the large dumps.
struct Fnv {
  uint64_t h = 1469598103934665603ull;
  void add(double v) {
    uint64_t b;
    std::memcpy(&b, &v, 8);
    for (int k = 0; k < 8; k++) {
      h ^= (b >> (8 * k)) & 0xffu;
      h *= 1099511628211ull;
    }
  }
};

// N-dimensional Rosenbrock, term order identical to example.cpp's 2-D functor
// (t1*t1 + 100*t2*t2), summed left to right over i = 0..D-2.
struct RosenbrockND {
  double operator()(std::vector<double> &x) {
    double acc = 0.0;
    for (size_t i = 0; i + 1 < x.size(); i++) {
      const double t1 = 1 - x[i];
      const double t2 = (x[i + 1] - x[i] * x[i]);
      acc += t1 * t1 + 100 * t2 * t2;
    }
    return acc;
  }
};
// Same objective, but records every evaluation (point + value). The solver
// holds the functor by reference, so the record is complete and ordered.
struct RecordingRosenbrock {
  std::vector<std::vector<double>> xs;
  std::vector<double> fs;
  RosenbrockND inner;
  double operator()(std::vector<double> &x) {
    const double v = inner(x);
    xs.push_back(x);
    fs.push_back(v);
    return v;
  }
};
// README.md objective (README.md:83-90): note t1 = x[0], not 1 - x[0].
struct ReadmeRosenbrock {
  double operator()(std::vector<double> &x) {
    const double t1 = x[0];
    const double t2 = (x[1] - x[0] * x[0]);
    return t1 * t1 + 100 * t2 * t2;
  }
};

// ----------------------------------------------------------------------------
// rng: first n outputs of splitmix::yield_init and xorshift<double>
// ----------------------------------------------------------------------------
static int cmd_rng(int n) {
  nlsolver::rng::splitmix<double> sm;
  std::printf("{\"splitmix_yield_init\":[");
  for (int i = 0; i < n; i++)
    std::printf("%s\"%" PRIu64 "\"", i ? "," : "", sm.yield_init());
  std::printf("],\"xorshift_double\":[");
  xorshift<double> g;
  for (int i = 0; i < n; i++) {
    if (i) std::printf(",");
    put_hex(g());
  }
  std::printf("]}\n");
  return 0;
}

// ----------------------------------------------------------------------------
// de: run DE<…> on Rosenbrock-ND and print status + final x.
//   de <strategy:random|best> <D> <pop> <max_iter> <eps> <no_change> <x0> [CR F]
//   x0 is a comma separated list, or a single value replicated D times.
// With trace=1 every objective evaluation is dumped (small cases) or hashed.
// ----------------------------------------------------------------------------
static std::vector<double> parse_x0(const char *s, size_t D) {
  std::vector<double> out;
  std::stringstream ss(s);
  std::string tok;
  while (std::getline(ss, tok, ',')) out.push_back(std::strtod(tok.c_str(), nullptr));
  if (out.size() == 1 && D > 1) out.assign(D, out[0]);
  if (out.size() != D) {
    std::fprintf(stderr, "x0 has %zu entries, expected %zu\n", out.size(), D);
    std::exit(2);
  }
  return out;
}

template <DEStrat S>
static int run_de(size_t D, size_t pop, size_t max_iter, double eps,
                  size_t no_change, std::vector<double> x, double CR, double F,
                  int trace) {
  RecordingRosenbrock f;
  xorshift<double> gen;
  auto solver = DE<RecordingRosenbrock, xorshift<double>, double, S>(
      f, gen, CR, F, eps, pop, max_iter, no_change);
  auto res = solver.minimize(x);
  auto [fcalls, iters, fval, g, h] = res.get_summary();
  (void)g;
  (void)h;
  std::printf("{\"strategy\":\"%s\",\"D\":%zu,\"pop\":%zu,\"max_iter\":%zu,",
              S == DEStrat::best ? "best" : "random", D, pop, max_iter);
  std::printf("\"eps\":");
  put_hex(eps);
  std::printf(",\"no_change\":%zu,\"CR\":", no_change);
  put_hex(CR);
  std::printf(",\"F\":");
  put_hex(F);
  std::printf(",\"fcalls\":%zu,\"iters\":%zu,\"f\":", fcalls, iters);
  put_hex(fval);
  std::printf(",\"x\":");
  put_vec(x);
  // next two draws of the caller's generator: pins how far the stream advanced
  std::printf(",\"rng_after\":[");
  put_hex(gen());
  std::printf(",");
  put_hex(gen());
  std::printf("]");
  if (trace >= 1) {
    // scores of every evaluation in call order (init agents then trials)
    std::printf(",\"eval_f\":");
    put_vec(f.fs);
    Fnv hh;
    for (auto &v : f.xs)
      for (double d : v) hh.add(d);
    std::printf(",\"eval_x_fnv\":\"%" PRIu64 "\"", hh.h);
  }
  if (trace >= 2) {
    std::printf(",\"eval_x\":[");
    for (size_t i = 0; i < f.xs.size(); i++) {
      if (i) std::printf(",");
      put_vec(f.xs[i]);
    }
    std::printf("]");
  }
  std::printf("}\n");
  return 0;
}

// README objective variant of config C1 (no trace, D=2).
static int run_de_readme(size_t pop) {
  ReadmeRosenbrock f;
  xorshift<double> gen;
  auto solver = DE<ReadmeRosenbrock, xorshift<double>, double, DEStrat::random>(
      f, gen, 0.9, 0.8, 10e-4, pop);
  std::vector<double> x = {5, 7};
  auto res = solver.minimize(x);
  auto [fcalls, iters, fval, g, h] = res.get_summary();
  (void)g;
  (void)h;
  std::printf("{\"fcalls\":%zu,\"iters\":%zu,\"f\":", fcalls, iters);
  put_hex(fval);
  std::printf(",\"x\":");
  put_vec(x);
  std::printf("}\n");
  return 0;
}

// ----------------------------------------------------------------------------
// bench-de: time the reference DE (strategy random) on Rosenbrock-ND.
// candidate-evals/s = pop * generations / wall time of the generation loop
// (initial evaluation is included in wall time and in the eval count).
// ----------------------------------------------------------------------------
static int bench_de(size_t D, size_t pop, size_t gens) {
  RosenbrockND f;
  xorshift<double> gen;
  auto solver = DE<RosenbrockND, xorshift<double>, double, DEStrat::random>(
      f, gen, 0.9, 0.8, 0.0, pop, gens, gens + 1);
  std::vector<double> x(D, 4.096);
  auto t0 = std::chrono::steady_clock::now();
  auto res = solver.minimize(x);
  auto t1 = std::chrono::steady_clock::now();
  const double s = std::chrono::duration<double>(t1 - t0).count();
  auto [fcalls, iters, fval, g, h] = res.get_summary();
  (void)g;
  (void)h;
  std::printf(
      "{\"D\":%zu,\"pop\":%zu,\"generations\":%zu,\"fcalls\":%zu,"
      "\"seconds\":%.6f,\"candidate_evals_per_s\":%.6e,\"f\":%.17g}\n",
      D, pop, iters, fcalls, s, static_cast<double>(fcalls) / s, fval);
  return 0;
}

#include "ref_driver_more.inc"
#include "ref_driver_stat.inc"

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: ref_driver <rng|de|de-readme|bench-de|...> ...\n");
    return 2;
  }
  const std::string cmd = argv[1];
  if (cmd == "rng") return cmd_rng(argc > 2 ? std::atoi(argv[2]) : 16);
  if (cmd == "de") {
    if (argc < 10) {
      std::fprintf(stderr,
                   "de <random|best> D pop max_iter eps no_change x0 trace [CR F]\n");
      return 2;
    }
    const std::string strat = argv[2];
    const size_t D = std::strtoull(argv[3], nullptr, 10);
    const size_t pop = std::strtoull(argv[4], nullptr, 10);
    const size_t max_iter = std::strtoull(argv[5], nullptr, 10);
    const double eps = std::strtod(argv[6], nullptr);
    const size_t no_change = std::strtoull(argv[7], nullptr, 10);
    auto x0 = parse_x0(argv[8], D);
    const int trace = std::atoi(argv[9]);
    const double CR = argc > 10 ? std::strtod(argv[10], nullptr) : 0.9;
    const double F = argc > 11 ? std::strtod(argv[11], nullptr) : 0.8;
    if (strat == "best")
      return run_de<DEStrat::best>(D, pop, max_iter, eps, no_change, x0, CR, F, trace);
    return run_de<DEStrat::random>(D, pop, max_iter, eps, no_change, x0, CR, F, trace);
  }
  if (cmd == "de-readme")
    return run_de_readme(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 40);
  if (cmd == "bench-de") {
    const size_t D = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 128;
    const size_t pop = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 65536;
    const size_t gens = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 10;
    return bench_de(D, pop, gens);
  }
  {
    bool handled = false;
    const int rc = stat_main(cmd, argc, argv, handled);
    if (handled) return rc;
  }
  return more_main(cmd, argc, argv);
}
