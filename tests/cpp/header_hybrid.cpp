// tests/cpp/header_hybrid.cpp — NelderMeadPSO through the drop-in header.
//   header_hybrid host objective n max_iter eps no_change x0 x0_step minimize
//       host-functor path with the reference's xorshift generator (tests/golden/nmpso.json)
//   header_hybrid device|device-custom instances n max_iter eps no_change x0 x0_step
//       device::Rosenbrock / device::Custom instances, batched (start b = x0 + x0_step * (i + b))
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nlsolver_mi/nlsolver.h"

using nlsolver::rng::xorshift;

struct Objective {
  int kind;  // 0 Rosenbrock chain, 1 sphere, 2 Styblinski-Tang
  double operator()(std::vector<double> &x) {
    double acc = 0.0;
    const size_t n = x.size();
    if (kind == 0) {
      for (size_t i = 0; i + 1 < n; i++) {
        const double t1 = 1 - x[i];
        const double t2 = (x[i + 1] - x[i] * x[i]);
        acc += t1 * t1 + 100 * t2 * t2;
      }
    } else if (kind == 1) {
      for (size_t i = 0; i < n; i++) acc += x[i] * x[i];
    } else {
      for (size_t i = 0; i < n; i++) {
        const double x2 = x[i] * x[i];
        acc += x2 * x2 - 16 * x2 + 5 * x[i];
      }
      acc = acc / 2.0;
    }
    return acc;
  }
};

static void print_status(nlsolver::solver_status<double> res, const std::vector<double> &x) {
  auto [fcalls, iters, f, g, h] = res.get_summary();
  (void)g;
  (void)h;
  std::printf("{\"fcalls\":%zu,\"iters\":%zu,\"f\":\"%a\",\"x\":[", fcalls, iters, f);
  for (size_t i = 0; i < x.size(); i++) std::printf("%s\"%a\"", i ? "," : "", x[i]);
  std::printf("]");
}

template <typename F>
static int run_device(F &f, char **argv) {
  const size_t B = std::strtoull(argv[2], nullptr, 10), n = std::strtoull(argv[3], nullptr, 10);
  xorshift<double> gen;
  auto solver = nlsolver::NelderMeadPSO<F, xorshift<double>, double>(
      f, gen, 1, 2, 0.5, 0.5, 0.8, 1.8, 1.8, std::strtod(argv[5], nullptr),
      std::strtoull(argv[4], nullptr, 10), std::strtoull(argv[6], nullptr, 10));
  const double x0 = std::strtod(argv[7], nullptr), step = std::strtod(argv[8], nullptr);
  std::vector<std::vector<double>> xs(B, std::vector<double>(n));
  for (size_t b = 0; b < B; b++)
    for (size_t i = 0; i < n; i++) xs[b][i] = x0 + step * static_cast<double>(i + b);
  try {
    auto sts = solver.minimize_batch(xs);
    std::printf("[");
    for (size_t b = 0; b < B; b++) {
      if (b) std::printf(",");
      print_status(sts[b], xs[b]);
      std::printf("}");
    }
    std::printf("]\n");
  } catch (const nlsolver::device_error &e) {
    std::printf("{\"device_error\":\"%s\"}\n", e.what());
    return 3;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 10 && !std::strcmp(argv[1], "host")) {
    Objective f{std::atoi(argv[2])};
    const size_t n = std::strtoull(argv[3], nullptr, 10);
    xorshift<double> gen;
    auto solver = nlsolver::NelderMeadPSO<Objective, xorshift<double>, double>(
        f, gen, 1, 2, 0.5, 0.5, 0.8, 1.8, 1.8, std::strtod(argv[5], nullptr),
        std::strtoull(argv[4], nullptr, 10), std::strtoull(argv[6], nullptr, 10));
    std::vector<double> x(n);
    for (size_t i = 0; i < n; i++)
      x[i] = std::strtod(argv[7], nullptr) + std::strtod(argv[8], nullptr) * static_cast<double>(i);
    auto res = std::atoi(argv[9]) ? solver.minimize(x) : solver.maximize(x);
    print_status(res, x);
    std::printf(",\"next_draw\":\"%a\"}\n", gen());
    return 0;
  }
  if (argc >= 9 && !std::strcmp(argv[1], "device")) {
    nlsolver::device::Rosenbrock<double> f;
    return run_device(f, argv);
  }
  if (argc >= 9 && !std::strcmp(argv[1], "device-custom")) {
    nlsolver::device::Custom<double> f(
        "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;", true);
    return run_device(f, argv);
  }
  std::fprintf(stderr, "usage: header_hybrid host|device|device-custom ...\n");
  return 2;
}
