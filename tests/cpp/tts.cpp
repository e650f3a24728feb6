// tests/cpp/tts.cpp — time-to-solution through the drop-in header (bench.py --workload tts).
//
// What a user of the drop-in sees: construct the solver exactly as against the reference, call
// minimize() and wait. Each case is timed twice in one process — COLD (the first device call of the
// process: dlopen of libnlsolver_hip.so, HIP runtime and code-object load, then engine creation,
// allocation, the solve, read-back, destruction) and WARM (the same call again) — with the
// library's own phase laps (nlsg_call_timing: create / upload / init / iterate / read-back /
// destroy) beside the wall time of the whole minimize(). The reference solving the same problem
// on one core is timed by oracle/_ref/ref_driver tts-* (bench.py puts the two side by side).
//
//   tts de   D pop [max_iter eps no_change]      DE<device::Rosenbrock, xorshift, double, random>
//   tts pso  D particles [max_iter eps no_change] PSO<..., Accelerated>
//   tts bfgs n batch [max_iter grad_eps]         BFGS on the G6 quadratic, batch starts in lock step
//   tts lm   m n batch [max_iter f_delta]        LevenbergMarquardt on the tanh regression (§8d C4)
// Defaults are the reference's constructor defaults (nlsolver.h:2390-2394, 2522-2526, 3181-3185,
// 3443-3447): "to their default stops".
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nlsolver_mi/nlsolver.h"

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) {
  return std::chrono::duration<double, std::milli>(clk::now() - t0).count();
}

typedef int (*timing_fn)(double *);
static timing_fn bind_timing() {
  const char *env = std::getenv("NLSG_LIBRARY");
  void *h = dlopen((env && *env) ? env : "libnlsolver_hip.so", RTLD_NOW | RTLD_LOCAL);
  return h ? reinterpret_cast<timing_fn>(dlsym(h, "nlsg_call_timing")) : nullptr;
}

struct Lap {
  double wall_ms = 0, phase[6] = {0, 0, 0, 0, 0, 0};
  size_t iters = 0, fcalls = 0;
  double f = 0;
};
static void put_lap(const char *name, const Lap &l) {
  std::printf("\"%s\":{\"wall_ms\":%.4f,\"create_ms\":%.4f,\"upload_ms\":%.4f,\"init_ms\":%.4f,"
              "\"iterate_ms\":%.4f,\"readback_ms\":%.4f,\"destroy_ms\":%.4f,\"iters\":%zu,"
              "\"fcalls\":%zu,\"f\":%.17g}",
              name, l.wall_ms, l.phase[0], l.phase[1], l.phase[2], l.phase[3], l.phase[4], l.phase[5],
              l.iters, l.fcalls, l.f);
}
template <typename Call>
static Lap timed(Call &&call) {
  Lap l;
  const auto t0 = clk::now();
  call(l);
  if (l.wall_ms == 0) l.wall_ms = ms_since(t0);  // unless the call timed its own inner region
  static timing_fn fn = bind_timing();  // after the first call: the library is loaded by then
  if (fn) fn(l.phase);
  return l;
}

static uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint64_t key(uint64_t parent, uint64_t index) {
  return mix64(parent + 0x9E3779B97F4A7C15ull * (index + 1));
}
static double u01(uint64_t k, uint64_t slot) { return static_cast<double>(key(k, slot)) * 0x1p-64; }

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: tts de|pso|bfgs|lm ...\n");
    return 2;
  }
  const std::string what = argv[1];
  auto arg_u = [&](int i, size_t dflt) { return argc > i ? std::strtoull(argv[i], nullptr, 10) : dflt; };
  auto arg_d = [&](int i, double dflt) { return argc > i ? std::strtod(argv[i], nullptr) : dflt; };
  try {
    Lap cold, warm;
    if (what == "de" || what == "pso") {
      const size_t D = arg_u(2, 128), n = arg_u(3, 65536);
      const size_t max_iter = arg_u(4, what == "de" ? 1000 : 5000), no_change = arg_u(6, 50);
      const double eps = arg_d(5, 10e-4);
      auto once = [&](Lap &l) {
        nlsolver::device::Rosenbrock<double> f;
        nlsolver::rng::xorshift<double> gen;
        std::vector<double> x;
        nlsolver::solver_status<double> st(0, 0, 0);
        if (what == "de") {
          x = D == 2 ? std::vector<double>{5, 7} : std::vector<double>(D, 4.096);
          auto solver = nlsolver::DE<decltype(f), decltype(gen), double>(f, gen, 0.9, 0.8, eps, n,
                                                                         max_iter, no_change);
          st = solver.minimize(x);
        } else {
          x.assign(D, 2.048);
          auto solver = nlsolver::PSO<decltype(f), decltype(gen), double, nlsolver::Accelerated>(
              f, gen, 0.8, 1.8, 1.8, n, max_iter, no_change, eps);
          st = solver.minimize(x);
        }
        auto [fc, it, fv, g, h] = st.get_summary();
        (void)g;
        (void)h;
        l.iters = it;
        l.fcalls = fc;
        l.f = fv;
      };
      cold = timed(once);
      warm = timed(once);
      std::printf("{\"solver\":\"%s\",\"D\":%zu,\"n\":%zu,", what.c_str(), D, n);
    } else if (what == "bfgs") {
      const size_t n = arg_u(2, 1024), B = arg_u(3, 4096), max_iter = arg_u(4, 100);
      const double grad_eps = arg_d(5, 5e-3);
      std::vector<double> d(n), b(n);
      for (size_t i = 0; i < n; i++) {
        d[i] = n > 1 ? 1.0 + 9.0 * static_cast<double>(i) / static_cast<double>(n - 1) : 1.0;
        b[i] = std::sin(0.1 * static_cast<double>(i));
      }
      auto once = [&](Lap &l) {
        nlsolver::device::QuadDiagRank1<double> f(d, b, 0.01);
        auto solver = nlsolver::BFGS<decltype(f), double>(f, {}, max_iter, grad_eps, 1.0);
        std::vector<std::vector<double>> xs(B, std::vector<double>(n));
        for (size_t p = 0; p < B; p++)
          for (size_t i = 0; i < n; i++) xs[p][i] = 1.0 + 0.25 * std::sin(static_cast<double>(i + 31 * p));
        const auto t0 = clk::now();  // the starts are the caller's data: not part of the call
        auto sts = solver.minimize_batch(xs);
        l.wall_ms = ms_since(t0);
        for (auto &st : sts) {
          auto [fc, it, fv, g, h] = st.get_summary();
          (void)g;
          (void)h;
          l.iters += it;
          l.fcalls += fc;
          l.f += fv;
        }
      };
      cold = timed(once);
      warm = timed(once);
      std::printf("{\"solver\":\"bfgs\",\"n\":%zu,\"batch\":%zu,", n, B);
    } else if (what == "lm") {
      const size_t m = arg_u(2, 512), n = arg_u(3, 64), B = arg_u(4, 8192), max_iter = arg_u(5, 100);
      const double f_delta = arg_d(6, 1e-12);
      // A_ij = (2u-1)/sqrt(n), theta* = 2u-1, y = tanh(A theta*),
      // theta0 = 0.5 theta* + 0.1 (2u-1); keyed by (seed, problem, slot) like the reference driver's
      std::vector<double> A(B * m * n), y(B * m);
      std::vector<std::vector<double>> th0(B, std::vector<double>(n));
      const double scale = 1.0 / std::sqrt(static_cast<double>(n));
      for (size_t p = 0; p < B; p++) {
        const uint64_t kp = key(12374563468ull, p), kA = key(kp, 0), kT = key(kp, 1), k0 = key(kp, 2);
        double *Ap = A.data() + p * m * n;
        for (size_t e = 0; e < m * n; e++) Ap[e] = (2 * u01(kA, e) - 1) * scale;
        std::vector<double> star(n);
        for (size_t j = 0; j < n; j++) star[j] = 2 * u01(kT, j) - 1;
        for (size_t i = 0; i < m; i++) {
          double z = 0.0;
          for (size_t j = 0; j < n; j++) z += Ap[i * n + j] * star[j];
          y[p * m + i] = std::tanh(z);
        }
        for (size_t j = 0; j < n; j++) th0[p][j] = 0.5 * star[j] + 0.1 * (2 * u01(k0, j) - 1);
      }
      nlsolver::device::TanhRegression<double> f(m, n, std::move(A), std::move(y));
      auto once = [&](Lap &l) {
        auto solver = nlsolver::LevenbergMarquardt<decltype(f), double>(f, 10, 10, 10, max_iter, f_delta);
        auto th = th0;
        const auto t0 = clk::now();
        auto sts = solver.minimize_batch(th);
        l.wall_ms = ms_since(t0);
        for (auto &st : sts) {
          auto [fc, it, fv, g, h] = st.get_summary();
          (void)g;
          (void)h;
          l.iters += it;
          l.fcalls += fc;
          l.f += fv;
        }
      };
      cold = timed(once);
      warm = timed(once);
      std::printf("{\"solver\":\"lm\",\"m\":%zu,\"n\":%zu,\"batch\":%zu,", m, n, B);
    } else {
      std::fprintf(stderr, "unknown solver %s\n", what.c_str());
      return 2;
    }
    put_lap("cold", cold);
    std::printf(",");
    put_lap("warm", warm);
    std::printf("}\n");
  } catch (const nlsolver::device_error &e) {
    std::printf("{\"device_error\":\"%s\"}\n", e.what());
    return 3;
  }
  return 0;
}
