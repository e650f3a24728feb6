// tests/cpp/header_nm_lm.cpp — NelderMead and LevenbergMarquardt through the drop-in header.
//   nm-host D step max_iter eps no_change restarts x0 x0_step bounded upper lower minimize
//   nm-device (same arguments; objective = nlsolver::device::Rosenbrock<double>)
//   nm-device-custom (same arguments; objective = nlsolver::device::Custom<double>, compiled at run time)
//   lm-host-exp [lambda max_iter f_delta]      reference-style GN functors on the exp model
//   lm-device m n problems max_iter            device TanhRegression model, batched
//   lm-device-fd n lambda max_iter f_delta x0 x0_step   default functors on device::Rosenbrock
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "nlsolver_mi/nlsolver.h"

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

static void print_status(const nlsolver::solver_status<double> &st, const std::vector<double> &x) {
  auto [fcalls, iters, f, g, h] = st.get_summary();
  std::printf("{\"fcalls\":%zu,\"iters\":%zu,\"gcalls\":%zu,\"hcalls\":%zu,\"f\":\"%a\",\"x\":[",
              fcalls, iters, g, h, f);
  for (size_t i = 0; i < x.size(); i++) std::printf("%s\"%a\"", i ? "," : "", x[i]);
  std::printf("]}");
}

template <typename F>
static int run_nm(F &f, char **a) {
  const size_t D = std::strtoull(a[0], nullptr, 10);
  auto solver = nlsolver::NelderMead<F, double>(f, std::strtod(a[1], nullptr), 1, 2, 0.5, 0.5,
                                                std::strtod(a[3], nullptr),
                                                std::strtoull(a[2], nullptr, 10),
                                                std::strtoull(a[4], nullptr, 10),
                                                std::strtoull(a[5], nullptr, 10));
  std::vector<double> x(D), up(D, std::strtod(a[9], nullptr)), lo(D, std::strtod(a[10], nullptr));
  for (size_t i = 0; i < D; i++)
    x[i] = std::strtod(a[6], nullptr) + std::strtod(a[7], nullptr) * static_cast<double>(i);
  const bool bounded = std::atoi(a[8]) != 0, minimize = std::atoi(a[11]) != 0;
  try {
    auto st = bounded ? (minimize ? solver.minimize(x, up, lo) : solver.maximize(x, up, lo))
                      : (minimize ? solver.minimize(x) : solver.maximize(x));
    print_status(st, x);
    std::printf("\n");
  } catch (const nlsolver::device_error &e) {
    std::printf("{\"device_error\":\"%s\"}\n", e.what());
    return 3;
  }
  return 0;
}

// exp model of the LM golden (oracle/ref_driver_more.inc make_exp_model), GN functors
struct ExpModel {
  std::vector<double> t, y;
  ExpModel() {
    for (size_t i = 0; i < 8; i++) {
      const double ti = 0.25 * static_cast<double>(i);
      t.push_back(ti);
      y.push_back(2.0 * std::exp(-0.7 * ti) + 0.02 * std::sin(3.0 * static_cast<double>(i) + 1.0));
    }
  }
  void rj(const std::vector<double> &x, std::vector<double> &r, std::vector<double> &J) const {
    r.assign(8, 0.0);
    J.assign(16, 0.0);
    for (size_t i = 0; i < 8; i++) {
      const double e = std::exp(x[1] * t[i]);
      r[i] = y[i] - x[0] * e;
      J[i * 2 + 0] = -e;
      J[i * 2 + 1] = -(x[0] * t[i] * e);
    }
  }
  double operator()(std::vector<double> &x) {
    std::vector<double> r, J;
    rj(x, r, J);
    double acc = 0.0;
    for (double v : r) acc += v * v;
    return acc;
  }
};
struct ExpGrad {
  void operator()(ExpModel &f, std::vector<double> &x, std::vector<double> &g) {
    std::vector<double> r, J;
    f.rj(x, r, J);
    for (size_t j = 0; j < 2; j++) {
      double acc = 0.0;
      for (size_t i = 0; i < 8; i++) acc += J[i * 2 + j] * r[i];
      g[j] = 2 * acc;
    }
  }
};
struct ExpHess {
  void operator()(ExpModel &f, std::vector<double> &x, std::vector<double> &h) {
    std::vector<double> r, J;
    f.rj(x, r, J);
    for (size_t j = 0; j < 2; j++)
      for (size_t k = 0; k < 2; k++) {
        double acc = 0.0;
        for (size_t i = 0; i < 8; i++) acc += J[i * 2 + j] * J[i * 2 + k];
        h[j * 2 + k] = 2 * acc;
      }
  }
};

template <typename F>
static int run_lm_fd_with(F &f, int argc, char **argv) {
  // lm-device-fd n lambda max_iter f_delta x0 x0_step [objective [batch]]: default functors on a
  // device objective
  const size_t n = std::strtoull(argv[2], nullptr, 10);
  auto solver = nlsolver::LevenbergMarquardt<F, double>(
      f, std::strtod(argv[3], nullptr), 10, 10, std::strtoull(argv[4], nullptr, 10),
      std::strtod(argv[5], nullptr));
  std::vector<double> x(n);
  for (size_t i = 0; i < n; i++)
    x[i] = std::strtod(argv[6], nullptr) + (argc > 7 ? std::strtod(argv[7], nullptr) : 0.0) * static_cast<double>(i);
  try {
    if (argc > 9 && !std::strcmp(argv[9], "batch")) {  // minimize_batch() with that one start
      std::vector<std::vector<double>> xs{x};
      auto st = solver.minimize_batch(xs);
      print_status(st[0], xs[0]);
    } else {
      auto st = solver.minimize(x);
      print_status(st, x);
    }
    std::printf("\n");
  } catch (const nlsolver::device_error &e) {
    std::printf("{\"device_error\":\"%s\"}\n", e.what());
    return 3;
  }
  return 0;
}
static int run_lm_fd(int argc, char **argv) {
  if (!std::strcmp(argv[1], "lm-device-fd-custom")) {
    nlsolver::device::Custom<double> f(
        "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;", true);
    return run_lm_fd_with(f, argc, argv);
  }
  const std::string which = argc > 8 ? argv[8] : "rosenbrock";
  if (which == "sphere") {
    nlsolver::device::Sphere<double> f;
    return run_lm_fd_with(f, argc, argv);
  }
  if (which == "styblinski_tang") {
    nlsolver::device::StyblinskiTang<double> f;
    return run_lm_fd_with(f, argc, argv);
  }
  nlsolver::device::Rosenbrock<double> f;
  return run_lm_fd_with(f, argc, argv);
}

int main(int argc, char **argv) {
  if (argc >= 14 && !std::strcmp(argv[1], "nm-host")) {
    RosenbrockND f;
    return run_nm(f, argv + 2);
  }
  if (argc >= 14 && !std::strcmp(argv[1], "nm-device")) {
    nlsolver::device::Rosenbrock<double> f;
    return run_nm(f, argv + 2);
  }
  if (argc >= 14 && !std::strcmp(argv[1], "nm-device-custom")) {  // the same chain as source text
    nlsolver::device::Custom<double> f(
        "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;", true);
    return run_nm(f, argv + 2);
  }
  if (argc >= 2 && !std::strcmp(argv[1], "lm-host-exp")) {
    ExpModel f;
    auto solver = nlsolver::LevenbergMarquardt<ExpModel, double, ExpGrad, ExpHess>(
        f, argc > 2 ? std::strtod(argv[2], nullptr) : 10, 10, 10,
        argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 100,
        argc > 4 ? std::strtod(argv[4], nullptr) : 1e-12);
    std::vector<double> x = {1.0, -0.1};
    auto st = solver.minimize(x);
    print_status(st, x);
    std::printf("\n");
    return 0;
  }
  if (argc >= 2 && !std::strcmp(argv[1], "lm-host-findiff")) {
    RosenbrockND f;  // default fin_diff / fin_diff_h functors (example.cpp style)
    auto solver = nlsolver::LevenbergMarquardt<RosenbrockND, double>(f);
    std::vector<double> x = {2, 7};
    auto st = solver.minimize(x);
    print_status(st, x);
    std::printf("\n");
    return 0;
  }
  if (argc >= 6 && !std::strcmp(argv[1], "lm-device")) {
    const size_t m = std::strtoull(argv[2], nullptr, 10), n = std::strtoull(argv[3], nullptr, 10);
    const size_t B = std::strtoull(argv[4], nullptr, 10);
    // deterministic synthetic data: A_ij = cos-pattern / sqrt(n), theta* = sin-pattern
    std::vector<double> A(B * m * n), y(B * m);
    std::vector<std::vector<double>> th(B, std::vector<double>(n));
    for (size_t p = 0; p < B; p++) {
      std::vector<double> star(n);
      for (size_t j = 0; j < n; j++) star[j] = std::sin(0.37 * static_cast<double>(j + 3 * p) + 0.1);
      for (size_t i = 0; i < m; i++) {
        double z = 0.0;
        for (size_t j = 0; j < n; j++) {
          const double a = std::cos(0.11 * static_cast<double>(i * n + j) + 1.3 * p) / std::sqrt(static_cast<double>(n));
          A[(p * m + i) * n + j] = a;
          z += a * star[j];
        }
        y[p * m + i] = std::tanh(z);
      }
      for (size_t j = 0; j < n; j++) th[p][j] = 0.5 * star[j] + 0.05 * std::cos(static_cast<double>(j));
    }
    nlsolver::device::TanhRegression<double> f(m, n, A, y);
    auto solver = nlsolver::LevenbergMarquardt<decltype(f), double>(
        f, 10, 10, 10, std::strtoull(argv[5], nullptr, 10), 0.0);
    try {
      auto sts = solver.minimize_batch(th);
      std::printf("[");
      for (size_t p = 0; p < B; p++) {
        if (p) std::printf(",");
        print_status(sts[p], th[p]);
      }
      std::printf("]\n");
    } catch (const nlsolver::device_error &e) {
      std::printf("{\"device_error\":\"%s\"}\n", e.what());
      return 3;
    }
    return 0;
  }
  if (argc >= 7 && (!std::strcmp(argv[1], "lm-device-fd") || !std::strcmp(argv[1], "lm-device-fd-custom")))
    return run_lm_fd(argc, argv);
  std::fprintf(stderr, "usage: header_nm_lm nm-host|nm-device|lm-host-exp|lm-host-findiff|lm-device|lm-device-fd ...\n");
  return 2;
}
