// tests/cpp/header_bfgs.cpp — BFGS through the drop-in header.
//   header_bfgs host n max_iter grad_eps alpha x0 x0_step    host functor + analytic Grad functor
//   header_bfgs findiff                                      default fin_diff gradient (example.cpp style)
//   header_bfgs device n batch max_iter grad_eps alpha       device objective, batched starts
//   header_bfgs device-fd n max_iter grad_eps alpha x0 x0_step [objective [batch]]  device objective, default Grad
//   header_bfgs device-one n max_iter grad_eps alpha x0 x0_step   one start of the device quadratic
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "nlsolver_mi/nlsolver.h"

struct QuadDR1 {  // the G6 quadratic as a plain host functor
  std::vector<double> d, b;
  double c = 0.01;
  explicit QuadDR1(size_t n) : d(n), b(n) {
    for (size_t i = 0; i < n; i++) {
      d[i] = n > 1 ? 1.0 + 9.0 * static_cast<double>(i) / static_cast<double>(n - 1) : 1.0;
      b[i] = std::sin(0.1 * static_cast<double>(i));
    }
  }
  double operator()(std::vector<double> &x) {
    double q = 0.0, sx = 0.0, lin = 0.0;
    for (size_t i = 0; i < x.size(); i++) {
      q += d[i] * x[i] * x[i];
      sx += x[i];
      lin += b[i] * x[i];
    }
    return 0.5 * q + 0.5 * c * (sx * sx) - lin;
  }
};
struct QuadGrad {
  void operator()(QuadDR1 &f, std::vector<double> &x, std::vector<double> &g) {
    double sx = 0.0;
    for (size_t i = 0; i < x.size(); i++) sx += x[i];
    for (size_t i = 0; i < x.size(); i++) g[i] = f.d[i] * x[i] + f.c * sx - f.b[i];
  }
};
class Rosenbrock {  // example.cpp:41-48
 public:
  double operator()(std::vector<double> &x) {
    const double t1 = 1 - x[0];
    const double t2 = (x[1] - x[0] * x[0]);
    return t1 * t1 + 100 * t2 * t2;
  }
};

static void print_status(const nlsolver::solver_status<double> &st, const std::vector<double> &x) {
  auto [fcalls, iters, f, gcalls, h] = st.get_summary();
  (void)h;
  std::printf("{\"fcalls\":%zu,\"iters\":%zu,\"gcalls\":%zu,\"f\":\"%a\",\"x\":[", fcalls, iters,
              gcalls, f);
  for (size_t i = 0; i < x.size(); i++) std::printf("%s\"%a\"", i ? "," : "", x[i]);
  std::printf("]}");
}

template <typename F>
static int run_device_fd(F &prob, char **argv, bool batch) {
  const size_t n = std::strtoull(argv[2], nullptr, 10);
  auto solver = nlsolver::BFGS<F, double>(prob, {}, std::strtoull(argv[3], nullptr, 10),
                                          std::strtod(argv[4], nullptr), std::strtod(argv[5], nullptr));
  std::vector<double> x(n);
  for (size_t i = 0; i < n; i++)
    x[i] = std::strtod(argv[6], nullptr) + std::strtod(argv[7], nullptr) * static_cast<double>(i);
  try {
    if (batch) {
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

int main(int argc, char **argv) {
  if (argc >= 8 && !std::strcmp(argv[1], "host")) {
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    QuadDR1 f(n);
    QuadGrad g;
    auto solver = nlsolver::BFGS<QuadDR1, double, QuadGrad>(
        f, g, std::strtoull(argv[3], nullptr, 10), std::strtod(argv[4], nullptr),
        std::strtod(argv[5], nullptr));
    std::vector<double> x(n);
    for (size_t i = 0; i < n; i++)
      x[i] = std::strtod(argv[6], nullptr) + std::strtod(argv[7], nullptr) * static_cast<double>(i);
    auto st = solver.minimize(x);
    print_status(st, x);
    std::printf("\n");
    return 0;
  }
  if (argc >= 2 && !std::strcmp(argv[1], "findiff")) {
    Rosenbrock prob;
    auto solver = nlsolver::BFGS<Rosenbrock, double>(prob);  // example.cpp style, default Grad
    std::vector<double> x = {2, 7};
    auto st = solver.minimize(x);
    print_status(st, x);
    std::printf("\n");
    return 0;
  }
  if (argc >= 7 && !std::strcmp(argv[1], "device")) {
    const size_t n = std::strtoull(argv[2], nullptr, 10), B = std::strtoull(argv[3], nullptr, 10);
    QuadDR1 host(n);
    nlsolver::device::QuadDiagRank1<double> f(host.d, host.b, host.c);
    auto solver = nlsolver::BFGS<decltype(f), double>(f, {}, std::strtoull(argv[4], nullptr, 10),
                                                      std::strtod(argv[5], nullptr),
                                                      std::strtod(argv[6], nullptr));
    std::vector<std::vector<double>> xs(B, std::vector<double>(n));
    for (size_t p = 0; p < B; p++)
      for (size_t i = 0; i < n; i++) xs[p][i] = 1.0 + 0.01 * static_cast<double>(p) * std::cos(0.3 * i);
    try {
      auto sts = solver.minimize_batch(xs);
      std::printf("[");
      for (size_t p = 0; p < B; p++) {
        if (p) std::printf(",");
        print_status(sts[p], xs[p]);
      }
      std::printf("]\n");
    } catch (const nlsolver::device_error &e) {
      std::printf("{\"device_error\":\"%s\"}\n", e.what());
      return 3;
    }
    return 0;
  }
  if (argc >= 8 && !std::strcmp(argv[1], "device-fd-custom")) {
    // the same with the objective given as source text
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    nlsolver::device::Custom<double> prob(
        "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;", true);
    auto solver = nlsolver::BFGS<decltype(prob), double>(prob, {}, std::strtoull(argv[3], nullptr, 10),
                                                         std::strtod(argv[4], nullptr),
                                                         std::strtod(argv[5], nullptr));
    std::vector<double> x(n);
    for (size_t i = 0; i < n; i++)
      x[i] = std::strtod(argv[6], nullptr) + std::strtod(argv[7], nullptr) * static_cast<double>(i);
    try {
      auto st = solver.minimize(x);
      print_status(st, x);
      std::printf("\n");
    } catch (const nlsolver::device_error &e) {
      std::printf("{\"device_error\":\"%s\"}\n", e.what());
      return 3;
    }
    return 0;
  }
  if (argc >= 8 && !std::strcmp(argv[1], "device-fd")) {
    // the reference's default-gradient call, objective type swapped for the device one:
    // BFGS<Rosenbrock, double>(prob).minimize(x) (example.cpp:171-173); optional: the objective's
    // name, then "batch" to go through minimize_batch() with that one start
    const std::string which = argc > 8 ? argv[8] : "rosenbrock";
    const bool batch = argc > 9 && !std::strcmp(argv[9], "batch");
    if (which == "sphere") {
      nlsolver::device::Sphere<double> prob;
      return run_device_fd(prob, argv, batch);
    }
    if (which == "styblinski_tang") {
      nlsolver::device::StyblinskiTang<double> prob;
      return run_device_fd(prob, argv, batch);
    }
    nlsolver::device::Rosenbrock<double> prob;
    return run_device_fd(prob, argv, batch);
  }
  if (argc >= 8 && !std::strcmp(argv[1], "device-one")) {
    // BFGS<QuadDiagRank1>(f).minimize(x): ONE start of the G6 quadratic through the reference's call
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    QuadDR1 host(n);
    nlsolver::device::QuadDiagRank1<double> f(host.d, host.b, host.c);
    return run_device_fd(f, argv, false);
  }
  std::fprintf(stderr, "usage: header_bfgs host|findiff|device|device-fd ...\n");
  return 2;
}
