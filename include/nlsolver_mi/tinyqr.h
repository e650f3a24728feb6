// include/nlsolver_mi/tinyqr.h — header-only C++17 counterpart of the reference's tinyqr.h for
// the least-squares path: Givens QR of an n x p matrix, the
// triangular solve on top of it, and lm() = both.
//
// Same names, argument order, layouts and defaults as the reference, so user code switches by
// changing the include:
//   tinyqr::QR<T>                                           tinyqr.h:286-290
//   tinyqr::qr_decomposition(X, n, p, tol = 1e-8)           tinyqr.h:291-310
//   tinyqr::back_solve(Q, R, y, nrow, ncol)                 tinyqr.h:437-459
//   tinyqr::lm(X, y, tol = 1e-12)                           tinyqr.h:461-470
// X is column-major n x p (X[j * n + i] = element (i, j)); Q comes back as p rows of length n
// (Q[i * n + k] = Q(k, i)), R as a p x p block with R[j * p + i] = R(i, j).
//
// Two execution paths:
//   * the functions above run on the host, one system per call, and reproduce the reference's
//     results bit for bit (tests/golden/lm.json, tinyqr.json: outputs of the unmodified
//     reference) — same rotations, same expression order;
//   * tinyqr::device::lm(X, y, batch, tol) solves `batch` independent systems side by side on a
//     gfx950 GPU through the extern "C" boundary (nlsg_tinyqr_lm in include/nlsg_c_api.h,
//     libnlsolver_hip.so loaded with dlopen). No CPU fallback: a missing library or device throws.
// The eigenvalue helpers of the reference's file (qr_algorithm, QRSolver — used by CMAES only) are
// outside the hot path and not provided.
#ifndef NLSOLVER_MI_TINYQR_H_
#define NLSOLVER_MI_TINYQR_H_

#include <dlfcn.h>

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../nlsg_c_api.h"

namespace tinyqr {

template <typename scalar_t>
struct QR {
  std::vector<scalar_t> Q;
  std::vector<scalar_t> R;
};

namespace internal {
// A plane rotation that maps (a, b) to (hypot, 0): the pair givens_rotation returns
// (tinyqr.h:86-97), scaled by the larger of the two so that the quotient stays in [-1, 1].
template <typename scalar_t>
struct plane_rotation {
  scalar_t c, s;
  plane_rotation(const scalar_t a, const scalar_t b) {
    const bool b_larger = std::abs(b) > std::abs(a);
    const scalar_t q = b_larger ? a / b : b / a;
    const scalar_t w = static_cast<scalar_t>(1.0 / std::sqrt(std::pow(q, 2) + 1.0));
    c = b_larger ? w * q : w;
    s = b_larger ? w : w * q;
  }
  // rows `upper_row` (index i-1) and `lower_row` (index i) of a row-major matrix, `len` entries
  // each (rotate_matrix, tinyqr.h:126-139: first row takes c x + s y, second -s x + c y)
  void mix(scalar_t *upper_row, scalar_t *lower_row, const size_t len) const {
    for (size_t k = 0; k < len; k++) {
      const scalar_t x = upper_row[k], y = lower_row[k];
      upper_row[k] = c * x + s * y;
      lower_row[k] = -s * x + c * y;
    }
  }
};

// Eliminates the sub-diagonal of the n x p working matrix Rt (row-major) column by column, rows
// bottom-up (qr_impl, tinyqr.h:253-283), carrying every rotation over to `Qt` (n x n, starts as the
// identity: ends as Q transposed) when it is given.
template <typename scalar_t>
void eliminate(std::vector<scalar_t> &Rt, std::vector<scalar_t> *Qt, const size_t n, const size_t p) {
  for (size_t col = 0; col < p; col++) {
    for (size_t row = n - 1; row > col; row--) {
      scalar_t *above = Rt.data() + (row - 1) * p, *below = Rt.data() + row * p;
      const plane_rotation<scalar_t> g(above[col], below[col]);
      g.mix(above, below, p);
      if (Qt) g.mix(Qt->data() + (row - 1) * n, Qt->data() + row * n, n);
    }
  }
}
}  // namespace internal

template <typename scalar_t>
[[maybe_unused]] QR<scalar_t> qr_decomposition(const std::vector<scalar_t> &X, const size_t n,
                                               const size_t p, const scalar_t tol = 1e-8) {
  QR<scalar_t> out;
  out.Q.assign(n * n, static_cast<scalar_t>(0.0));
  for (size_t i = 0; i < n; i++) out.Q[i * n + i] = static_cast<scalar_t>(1.0);
  out.R.assign(n * p, static_cast<scalar_t>(0.0));
  for (size_t j = 0; j < p; j++)  // the working matrix is X transposed into row-major
    for (size_t i = 0; i < n; i++) out.R[i * p + j] = X[j * n + i];
  internal::eliminate(out.R, &out.Q, n, p);
  for (scalar_t &v : out.R)  // what is left of the annihilated entries
    if (std::abs(v) < tol) v = static_cast<scalar_t>(0.0);
  for (size_t i = 0; i < p; i++)  // the leading p x p block, transposed in place
    for (size_t j = i + 1; j < p; j++) std::swap(out.R[i * p + j], out.R[j * p + i]);
  out.Q.resize(n * p);
  out.R.resize(p * p);
  return out;
}

// R x = Q^T y by back substitution; Q^T y is formed coefficient by coefficient
template <typename scalar_t>
std::vector<scalar_t> back_solve(const std::vector<scalar_t> &Q, const std::vector<scalar_t> &R,
                                 const std::vector<scalar_t> &y, const size_t nrow,
                                 const size_t ncol) {
  std::vector<scalar_t> x(ncol, static_cast<scalar_t>(0.0));
  for (size_t i = ncol; i-- > 0;) {
    scalar_t known = 0.0;
    for (size_t j = i + 1; j < ncol; j++) known += R[j * ncol + i] * x[j];
    scalar_t qty = 0;
    for (size_t k = 0; k < nrow; k++) qty += Q[i * nrow + k] * y[k];
    x[i] = (qty - known) / R[i * ncol + i];
  }
  return x;
}

template <typename scalar_t>
[[maybe_unused]] std::vector<scalar_t> lm(const std::vector<scalar_t> &X,
                                          const std::vector<scalar_t> &y,
                                          const scalar_t tol = 1e-12) {
  const size_t nrow = y.size();
  const size_t ncol = X.size() / nrow;
  const QR<scalar_t> qr = qr_decomposition(X, nrow, ncol, tol);
  return back_solve(qr.Q, qr.R, y, nrow, ncol);
}

// ---------------------------------------------------------------------------
// batched systems on the GPU
// ---------------------------------------------------------------------------
namespace device {
struct device_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

namespace detail {
struct entry_points {
  decltype(&nlsg_tinyqr_lm) lm = nullptr;
  decltype(&nlsg_last_error) last_error = nullptr;
  entry_points() {
    const char *env = std::getenv("NLSG_LIBRARY");
    const char *name = (env && *env) ? env : "libnlsolver_hip.so";
    void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h)
      throw device_error(std::string("cannot load ") + name + " (" + dlerror() +
                         "); tinyqr::device has no CPU fallback");
    lm = reinterpret_cast<decltype(lm)>(dlsym(h, "nlsg_tinyqr_lm"));
    last_error = reinterpret_cast<decltype(last_error)>(dlsym(h, "nlsg_last_error"));
    if (!lm || !last_error) throw device_error("libnlsolver_hip.so lacks nlsg_tinyqr_lm");
  }
};
}  // namespace detail

// `batch` systems at once: X holds them back to back (system b: X[b * n * p + j * n + i]), y alike
// (y[b * n + i]); n = y.size() / batch, p = X.size() / y.size(). Returns batch * p coefficients.
// Requires n >= p and p <= 64.
[[maybe_unused]] inline std::vector<double> lm(const std::vector<double> &X,
                                               const std::vector<double> &y, const size_t batch,
                                               const double tol = 1e-12, const int gpu = 0) {
  static const detail::entry_points api;
  if (batch == 0 || y.empty() || y.size() % batch || X.size() % y.size())
    throw device_error("tinyqr::device::lm: X and y do not describe `batch` systems of one shape");
  const size_t n = y.size() / batch, p = X.size() / y.size();
  std::vector<double> beta(batch * p);
  const int rc = api.lm(X.data(), y.data(), batch, n, p, tol, gpu, beta.data(), nullptr);
  if (rc != NLSG_OK)
    throw device_error(std::string("nlsg error ") + std::to_string(rc) + ": " + api.last_error());
  return beta;
}
}  // namespace device

}  // namespace tinyqr
#endif  // NLSOLVER_MI_TINYQR_H_
