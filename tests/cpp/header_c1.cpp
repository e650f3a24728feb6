// tests/cpp/header_c1.cpp — config C1 through the drop-in header (host functor
// path, no GPU): the user code below is written exactly as against the
// reference (example.cpp:41-48, 186-188; README.md:83-104), only the include
// differs. Prints hexfloat JSON that tests/test_header_cpp.py compares with the
// golden vectors produced by the reference itself.
#include <cstdio>

#include "nlsolver_mi/nlsolver.h"

using nlsolver::DE;
using nlsolver::DESolver;
using nlsolver::rng::xorshift;
using DEStrat = nlsolver::RecombinationStrategy;

class Rosenbrock {  // example.cpp:41-48
 public:
  double operator()(std::vector<double> &x) {
    const double t1 = 1 - x[0];
    const double t2 = (x[1] - x[0] * x[0]);
    return t1 * t1 + 100 * t2 * t2;
  }
};
class ReadmeRosenbrock {  // README.md:83-90
 public:
  double operator()(std::vector<double> &x) {
    const double t1 = x[0];
    const double t2 = (x[1] - x[0] * x[0]);
    return t1 * t1 + 100 * t2 * t2;
  }
};
struct ConstRefSphere {  // test_functions.h:55 style: const std::vector<T>&
  double operator()(const std::vector<double> &x) { return x[0] * x[0] + x[1] * x[1]; }
};

template <typename S, typename G>
static void report(const char *name, S &solver, G &gen, std::vector<double> x, bool last = false) {
  auto res = solver.minimize(x);
  auto [fcalls, iters, f, g, h] = res.get_summary();
  std::printf("\"%s\":{\"fcalls\":%zu,\"iters\":%zu,\"f\":\"%a\",\"x\":[\"%a\",\"%a\"],", name,
              fcalls, iters, f, x[0], x[1]);
  const double a = gen(), b = gen();
  std::printf("\"rng_after\":[\"%a\",\"%a\"],\"grad\":%zu,\"hess\":%zu}%s\n", a, b, g, h,
              last ? "" : ",");
}

int main() {
  std::printf("{\n");
  {
    Rosenbrock prob;
    xorshift<double> gen;
    auto s = DE<Rosenbrock, xorshift<double>, double, DEStrat::random>(prob, gen, 0.9, 0.8, 10e-4, 40);
    report("c1_random_pop40_x0_5_7", s, gen, {5, 7});
  }
  {
    Rosenbrock prob;
    xorshift<double> gen;
    auto s = DE<Rosenbrock, xorshift<double>, double>(prob, gen);  // all defaults
    report("random_pop50_x0_5_7", s, gen, {5, 7});
  }
  {
    Rosenbrock prob;
    xorshift<double> gen;
    auto s = DE<Rosenbrock, xorshift<double>, double, DEStrat::best>(prob, gen);  // example.cpp:186
    report("example_best_pop50_x0_2_7", s, gen, {2, 7});
  }
  {
    ReadmeRosenbrock prob;
    xorshift<double> gen;
    auto s = DESolver<ReadmeRosenbrock, xorshift<double>, double>(prob, gen, 0.9, 0.8, 10e-4, 40);
    report("readme_objective_pop40", s, gen, {5, 7});
  }
  {
    // lambdas and const-ref functors are accepted too (README.md:136, test_functions.h:55)
    auto lam = [](std::vector<double> &x) { return (x[0] - 1) * (x[0] - 1) + x[1] * x[1]; };
    xorshift<double> gen;
    auto s = DE<decltype(lam), xorshift<double>, double>(lam, gen);
    report("lambda", s, gen, {3, 3});
    ConstRefSphere sp;
    auto s2 = DE<ConstRefSphere, xorshift<double>, double, DEStrat::best>(sp, gen);
    std::vector<double> x = {2, 2};
    auto r = s2.maximize(x);  // maximize() compiles and runs
    (void)r;
    auto s3 = DE<ConstRefSphere, xorshift<double>, double>(sp, gen);
    report("const_ref_sphere", s3, gen, {2, 2}, true);
  }
  std::printf("}\n");
  return 0;
}
