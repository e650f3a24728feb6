// tests/cpp/header_tinyqr.cpp — drives include/nlsolver_mi/tinyqr.h the way a user of the
// reference's tinyqr.h would: qr_decomposition / back_solve / lm on systems read from stdin.
//   header_tinyqr host            one system:  n p, then X (column-major, n*p), then y (n)
//   header_tinyqr device          batch n p, then the systems back to back, then the y's
// Values travel as C99 hexfloats (bit-exact); the Python tests compare with the reference's
// golden outputs (host) and with the oracle (device).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nlsolver_mi/nlsolver.h"  // pulls in tinyqr.h, as the reference's nlsolver.h does

static double read_double() {
  char tok[64];
  if (std::scanf("%63s", tok) != 1) std::exit(3);
  return std::strtod(tok, nullptr);
}
static void put(const char *name, const std::vector<double> &v) {
  std::printf("%s", name);
  for (double d : v) std::printf(" %a", d);
  std::printf("\n");
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (!std::strcmp(argv[1], "host")) {
    size_t n, p;
    if (std::scanf("%zu %zu", &n, &p) != 2) return 3;
    std::vector<double> X(n * p), y(n);
    for (double &v : X) v = read_double();
    for (double &v : y) v = read_double();
    const tinyqr::QR<double> qr = tinyqr::qr_decomposition(X, n, p);  // default tol 1e-8
    put("Q", qr.Q);
    put("R", qr.R);
    put("beta_tol1e-8", tinyqr::back_solve(qr.Q, qr.R, y, n, p));
    put("beta", tinyqr::lm(X, y));  // default tol 1e-12
    return 0;
  }
  if (!std::strcmp(argv[1], "device")) {
    size_t batch, n, p;
    if (std::scanf("%zu %zu %zu", &batch, &n, &p) != 3) return 3;
    std::vector<double> X(batch * n * p), y(batch * n);
    for (double &v : X) v = read_double();
    for (double &v : y) v = read_double();
    try {
      put("beta", tinyqr::device::lm(X, y, batch));
    } catch (const tinyqr::device::device_error &e) {
      std::fprintf(stderr, "device_error: %s\n", e.what());
      return 4;
    }
    return 0;
  }
  return 2;
}
