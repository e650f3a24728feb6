// tests/cpp/header_pso.cpp — PSO through the drop-in header.
//   header_pso host                       host-functor path: the reference's goldens
//   header_pso device <accel|vanilla> D particles max_iter eps no_change bound [bounded] [custom]
// "custom": the objective is device::Custom<double> (the Rosenbrock chain as source text)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nlsolver_mi/nlsolver.h"

using nlsolver::PSO;
using nlsolver::PSOType;
using nlsolver::rng::xorshift;

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

template <typename S, typename G>
static void report(const char *name, nlsolver::solver_status<double> res, G &gen,
                   const std::vector<double> &x, bool last = false) {
  (void)sizeof(S);
  auto [fcalls, iters, f, g, h] = res.get_summary();
  (void)g;
  (void)h;
  std::printf("\"%s\":{\"fcalls\":%zu,\"iters\":%zu,\"f\":\"%a\",\"x\":[", name, fcalls, iters, f);
  for (size_t i = 0; i < x.size(); i++) std::printf("%s\"%a\"", i ? "," : "", x[i]);
  const double a = gen(), b = gen();
  std::printf("],\"rng_after\":[\"%a\",\"%a\"]}%s\n", a, b, last ? "" : ",");
}

static int host() {
  std::printf("{\n");
  {
    RosenbrockND f;
    xorshift<double> gen;
    auto s = PSO<RosenbrockND, xorshift<double>, double, PSOType::Accelerated>(f, gen, 0.8, 1.8, 1.8,
                                                                               10, 50, 1000, 0);
    std::vector<double> x = {3, 3};
    auto r = s.minimize(x);
    report<int>("accel_2d_x0_3_3", r, gen, x);
  }
  {
    RosenbrockND f;
    xorshift<double> gen;
    auto s = PSO<RosenbrockND, xorshift<double>, double, PSOType::Accelerated>(f, gen, 0.8, 1.8, 1.8,
                                                                               64, 5, 1000, 0);
    std::vector<double> x(256, 0.3);
    auto r = s.minimize(x);
    report<int>("accel_256d_64p", r, gen, x);
  }
  {
    RosenbrockND f;
    xorshift<double> gen;
    auto s = PSO<RosenbrockND, xorshift<double>, double, PSOType::Accelerated>(f, gen, 0.8, 1.8, 1.8,
                                                                               16, 20, 1000, 0);
    std::vector<double> x(8, 2.0), lo(8, -1.5), hi(8, 1.5);
    auto r = s.minimize(x, lo, hi);
    report<int>("accel_8d_bounded", r, gen, x);
  }
  {
    RosenbrockND f;
    xorshift<double> gen;
    auto s = PSO<RosenbrockND, xorshift<double>, double, PSOType::Accelerated>(f, gen);  // defaults
    std::vector<double> x = {3, 3};
    auto r = s.minimize(x);
    report<int>("accel_2d_default_stops", r, gen, x);
  }
  {
    RosenbrockND f;
    xorshift<double> gen;
    auto s = nlsolver::PSOSolver<RosenbrockND, xorshift<double>, double>(f, gen);  // Vanilla
    std::vector<double> x = {3, 3};
    auto r = s.minimize(x);
    report<int>("vanilla_2d_intended_update", r, gen, x, true);
  }
  std::printf("}\n");
  return 0;
}

static nlsolver::device::Rosenbrock<double> make(nlsolver::device::Rosenbrock<double> *) { return {}; }
static nlsolver::device::Custom<double> make(nlsolver::device::Custom<double> *) {
  return nlsolver::device::Custom<double>(
      "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;", true);
}
template <PSOType T, typename Objective>
static int dev(size_t D, size_t np, size_t max_iter, double eps, size_t no_change, double bound,
               bool bounded) {
  Objective f = make(static_cast<Objective *>(nullptr));
  xorshift<double> gen;
  std::vector<double> x(D, bound), lo(D, -bound), hi(D, bound);
  try {
    auto s = PSO<Objective, xorshift<double>, double, T>(f, gen, 0.8, 1.8, 1.8, np, max_iter,
                                                         no_change, eps);
    auto r = bounded ? s.minimize(x, lo, hi) : s.minimize(x);
    std::printf("{");
    report<int>("run", r, gen, x, true);
    std::printf("}\n");
  } catch (const nlsolver::device_error &e) {
    std::printf("{\"device_error\":\"%s\"}\n", e.what());
    return 3;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 2 && !std::strcmp(argv[1], "host")) return host();
  if (argc >= 9 && !std::strcmp(argv[1], "device")) {
    const size_t D = std::strtoull(argv[3], nullptr, 10), np = std::strtoull(argv[4], nullptr, 10);
    const size_t max_iter = std::strtoull(argv[5], nullptr, 10);
    const double eps = std::strtod(argv[6], nullptr);
    const size_t no_change = std::strtoull(argv[7], nullptr, 10);
    const double bound = std::strtod(argv[8], nullptr);
    const bool bounded = argc > 9 && std::atoi(argv[9]) != 0;
    const bool accel = !std::strcmp(argv[2], "accel");
    if (argc > 10 && !std::strcmp(argv[10], "custom")) {
      using C = nlsolver::device::Custom<double>;
      return accel ? dev<PSOType::Accelerated, C>(D, np, max_iter, eps, no_change, bound, bounded)
                   : dev<PSOType::Vanilla, C>(D, np, max_iter, eps, no_change, bound, bounded);
    }
    using R = nlsolver::device::Rosenbrock<double>;
    return accel ? dev<PSOType::Accelerated, R>(D, np, max_iter, eps, no_change, bound, bounded)
                 : dev<PSOType::Vanilla, R>(D, np, max_iter, eps, no_change, bound, bounded);
  }
  std::fprintf(stderr, "usage: header_pso host | device <accel|vanilla> D np max_iter eps no_change bound [bounded]\n");
  return 2;
}
