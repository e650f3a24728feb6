// tests/cpp/header_device.cpp — the drop-in header with a DEVICE objective:
// same user code as against the reference, the objective type is
// nlsolver::device::Rosenbrock<double>, the population loops run on the GPU
// through libnlsolver_hip.so (dlopen; $NLSG_LIBRARY). No CPU fallback.
//   header_device <best|random> <D> <pop> <max_iter> <eps> <no_change> <x0> [custom]
// With "custom" the objective is nlsolver::device::Custom<double> spelling the same Rosenbrock
// chain as source text, compiled for the device at solve time.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nlsolver_mi/nlsolver.h"

using nlsolver::DE;
using nlsolver::rng::xorshift;
using DEStrat = nlsolver::RecombinationStrategy;
using Objective = nlsolver::device::Rosenbrock<double>;

static Objective make(Objective *) { return Objective(); }
static nlsolver::device::Custom<double> make(nlsolver::device::Custom<double> *) {
  return nlsolver::device::Custom<double>(
      "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;", true);
}
static double host_value(Objective &f, std::vector<double> &x) { return f(x); }
static double host_value(nlsolver::device::Custom<double> &, std::vector<double> &x) {
  Objective f;  // the custom type has no host evaluation; report the built-in's
  return f(x);
}

template <DEStrat S, typename Objective>
static int run(size_t D, size_t pop, size_t max_iter, double eps, size_t no_change, double x0) {
  Objective f = make(static_cast<Objective *>(nullptr));
  xorshift<double> gen;
  std::vector<double> x(D, x0);
  try {
    auto solver = DE<Objective, xorshift<double>, double, S>(f, gen, 0.9, 0.8, eps, pop, max_iter,
                                                             no_change);
    auto res = solver.minimize(x);
    auto [fcalls, iters, fv, g, h] = res.get_summary();
    (void)g;
    (void)h;
    std::printf("{\"fcalls\":%zu,\"iters\":%zu,\"f\":\"%a\",\"f_host\":\"%a\",\"x\":[", fcalls,
                iters, fv, host_value(f, x));
    for (size_t i = 0; i < D; i++) std::printf("%s\"%a\"", i ? "," : "", x[i]);
    const double a = gen(), b = gen();
    std::printf("],\"rng_after\":[\"%a\",\"%a\"]}\n", a, b);
  } catch (const nlsolver::device_error &e) {
    std::printf("{\"device_error\":\"%s\"}\n", e.what());
    return 3;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: header_device <best|random> D pop max_iter eps no_change x0\n");
    return 2;
  }
  const size_t D = std::strtoull(argv[2], nullptr, 10), pop = std::strtoull(argv[3], nullptr, 10);
  const size_t max_iter = std::strtoull(argv[4], nullptr, 10);
  const double eps = std::strtod(argv[5], nullptr);
  const size_t no_change = std::strtoull(argv[6], nullptr, 10);
  const double x0 = std::strtod(argv[7], nullptr);
  const bool best = !std::strcmp(argv[1], "best");
  if (argc > 8 && !std::strcmp(argv[8], "custom")) {
    using C = nlsolver::device::Custom<double>;
    return best ? run<DEStrat::best, C>(D, pop, max_iter, eps, no_change, x0)
                : run<DEStrat::random, C>(D, pop, max_iter, eps, no_change, x0);
  }
  return best ? run<DEStrat::best, Objective>(D, pop, max_iter, eps, no_change, x0)
              : run<DEStrat::random, Objective>(D, pop, max_iter, eps, no_change, x0);
}
