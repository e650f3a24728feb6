// nlsolver_amd/csrc/nlsg_tinyqr.hip — host side of the batched tinyqr::lm + C-ABI.
#include "nlsg_tinyqr_kernels.h"

#include <algorithm>

using namespace nlsg;

namespace {

int check_shape(uint64_t batch, uint64_t n, uint64_t p) {
  if (batch < 1 || n < 1 || p < 1) return fail(NLSG_ERR_INVALID_ARG, "batch, n and p must be >= 1");
  if (p > kTqrMaxP)
    return fail(NLSG_ERR_UNSUPPORTED, "p = %llu columns > %d is not covered by the device path",
                (unsigned long long)p, kTqrMaxP);
  if (n < p)
    return fail(NLSG_ERR_INVALID_ARG, "n = %llu rows < p = %llu columns: tinyqr::lm needs n >= p",
                (unsigned long long)n, (unsigned long long)p);
  if (batch > 0x7fffffffull) return fail(NLSG_ERR_UNSUPPORTED, "batch too large for one launch grid");
  if (n >= (1ull << 30))  // the kernel counts steps (n + p - 2 of them) in 32 bits
    return fail(NLSG_ERR_UNSUPPORTED, "n = %llu rows is past the device path's 2^30", (unsigned long long)n);
  return NLSG_OK;
}

// The opt-in for more than 64 KiB of dynamic LDS, to the largest p the kernel serves. The
// attribute applies to the device that is current when it is set, so it is set after every
// hipSetDevice (cheap, constant value) — set once per process it left a second device without it.
int allow_lds() {
  NLSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tinyqr_lm_kernel<kTqrThreads>),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(tqr_lds_bytes(kTqrMaxP))));
  return NLSG_OK;
}

void launch(const double *X, const double *y, double *beta, uint64_t batch, uint64_t n, uint64_t p,
            double tol, hipStream_t stream) {
  TqrParams q;
  q.X = X;
  q.y = y;
  q.beta = beta;
  q.batch = batch;
  q.n = n;
  q.p = static_cast<uint32_t>(p);
  q.ring = tqr_ring_rows(q.p);
  q.stride = tqr_stride(q.p);
  q.tol = tol;
  const dim3 grid(static_cast<unsigned>(batch));
  if (p <= 8)
    hipLaunchKernelGGL((tinyqr_lm_kernel<128, 8>), grid, dim3(128), tqr_lds_bytes(q.p), stream, q);
  else if (p <= 32)
    hipLaunchKernelGGL((tinyqr_lm_kernel<256, 32>), grid, dim3(256), tqr_lds_bytes(q.p), stream, q);
  else
    hipLaunchKernelGGL(tinyqr_lm_kernel<kTqrThreads>, grid, dim3(kTqrThreads), tqr_lds_bytes(q.p), stream, q);
}

}  // namespace

extern "C" {

int nlsg_tinyqr_lm_device(const double *X_dev, const double *y_dev, uint64_t batch, uint64_t n,
                          uint64_t p, double tol, int32_t device, void *stream, double *beta_dev) {
  if (!X_dev || !y_dev || !beta_dev) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  int rc = check_shape(batch, n, p);
  if (rc) return rc;
  rc = check_device(device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(device));
  rc = allow_lds();
  if (rc) return rc;
  launch(X_dev, y_dev, beta_dev, batch, n, p, tol, stream ? borrowed_stream(stream) : nullptr);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_tinyqr_lm(const double *X_host, const double *y_host, uint64_t batch, uint64_t n, uint64_t p,
                   double tol, int32_t device, double *beta_host, float *ms_kernel) {
  if (!X_host || !y_host || !beta_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  int rc = check_shape(batch, n, p);
  if (rc) return rc;
  rc = check_device(device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(device));
  rc = allow_lds();
  if (rc) return rc;
  double *X = nullptr, *y = nullptr, *beta = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t he = pool_malloc(reinterpret_cast<void **>(&X), batch * n * p * sizeof(double));
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&y), batch * n * sizeof(double));
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&beta), batch * p * sizeof(double));
  if (he == hipSuccess) he = hipEventCreate(&e0);
  if (he == hipSuccess) he = hipEventCreate(&e1);
  if (he == hipSuccess) he = hipMemcpy(X, X_host, batch * n * p * sizeof(double), hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(y, y_host, batch * n * sizeof(double), hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipEventRecord(e0, nullptr);
  if (he == hipSuccess) {
    launch(X, y, beta, batch, n, p, tol, nullptr);
    he = launches_status();
  }
  if (he == hipSuccess) he = hipEventRecord(e1, nullptr);
  if (he == hipSuccess) he = hipEventSynchronize(e1);
  if (he == hipSuccess && ms_kernel) he = hipEventElapsedTime(ms_kernel, e0, e1);
  if (he == hipSuccess) he = hipMemcpy(beta_host, beta, batch * p * sizeof(double), hipMemcpyDeviceToHost);
  pool_free(X);
  pool_free(y);
  pool_free(beta);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  if (he != hipSuccess)
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP, "nlsg_tinyqr_lm failed: %s",
                hipGetErrorString(he));
  return NLSG_OK;
}

int nlsg_tinyqr_qr(const double *X_host, const double *y_host, uint64_t batch, uint64_t n, uint64_t p,
                   double tol, int32_t device, double *Q_host, double *R_host, double *beta_host) {
  if (!X_host || (!Q_host && !R_host && !beta_host)) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (beta_host && !y_host) return fail(NLSG_ERR_INVALID_ARG, "beta needs y");
  if (batch < 1 || n < 1 || p < 1) return fail(NLSG_ERR_INVALID_ARG, "batch, n and p must be >= 1");
  if (n < p)
    return fail(NLSG_ERR_INVALID_ARG, "n = %llu rows < p = %llu columns: tinyqr needs n >= p",
                (unsigned long long)n, (unsigned long long)p);
  constexpr uint64_t kMaxCols = 64ull * 20;
  if (n + p > kMaxCols)
    return fail(NLSG_ERR_UNSUPPORTED, "n + p = %llu > %llu: the reference-order mode keeps a row of [R | Q] "
                "in one wave's registers", (unsigned long long)(n + p), (unsigned long long)kMaxCols);
  if (beta_host && p > 64)
    return fail(NLSG_ERR_UNSUPPORTED, "reference-order beta is built for p <= 64 (Q and R: any p)");
  int rc = check_device(device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(device));
  const uint64_t W = n + p, per = n * W * sizeof(double);
  const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(batch, (2ull << 30) / per));
  double *X = nullptr, *y = nullptr, *work = nullptr, *Q = nullptr, *R = nullptr, *beta = nullptr;
  hipError_t he = pool_malloc(reinterpret_cast<void **>(&X), batch * n * p * sizeof(double));
  if (he == hipSuccess && y_host) he = pool_malloc(reinterpret_cast<void **>(&y), batch * n * sizeof(double));
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&work), chunk * per);
  if (he == hipSuccess && Q_host) he = pool_malloc(reinterpret_cast<void **>(&Q), batch * n * p * sizeof(double));
  if (he == hipSuccess && R_host) he = pool_malloc(reinterpret_cast<void **>(&R), batch * p * p * sizeof(double));
  if (he == hipSuccess && beta_host) he = pool_malloc(reinterpret_cast<void **>(&beta), batch * p * sizeof(double));
  if (he == hipSuccess) he = hipMemcpy(X, X_host, batch * n * p * sizeof(double), hipMemcpyHostToDevice);
  if (he == hipSuccess && y_host) he = hipMemcpy(y, y_host, batch * n * sizeof(double), hipMemcpyHostToDevice);
  for (uint64_t s0 = 0; he == hipSuccess && s0 < batch; s0 += chunk) {
    TqrRefParams q;
    q.X = X;
    q.y = y;
    q.work = work;
    q.Q = Q;
    q.R = R;
    q.beta = beta;
    q.n = n;
    q.p = p;
    q.sys0 = s0;
    q.tol = tol;
    const dim3 grid(static_cast<unsigned>(std::min<uint64_t>(chunk, batch - s0)));
    const size_t lds = beta_host ? (p * p + 2 * p) * sizeof(double) : 0;
    if (W <= 128)
      hipLaunchKernelGGL(tinyqr_reference_kernel<2>, grid, dim3(64), lds, nullptr, q);
    else if (W <= 512)
      hipLaunchKernelGGL(tinyqr_reference_kernel<8>, grid, dim3(64), lds, nullptr, q);
    else
      hipLaunchKernelGGL(tinyqr_reference_kernel<20>, grid, dim3(64), lds, nullptr, q);
    he = launches_status();
    if (he == hipSuccess) he = hipDeviceSynchronize();  // the workspace is reused by the next chunk
  }
  if (he == hipSuccess && Q_host) he = hipMemcpy(Q_host, Q, batch * n * p * sizeof(double), hipMemcpyDeviceToHost);
  if (he == hipSuccess && R_host) he = hipMemcpy(R_host, R, batch * p * p * sizeof(double), hipMemcpyDeviceToHost);
  if (he == hipSuccess && beta_host) he = hipMemcpy(beta_host, beta, batch * p * sizeof(double), hipMemcpyDeviceToHost);
  pool_free(X);
  pool_free(y);
  pool_free(work);
  pool_free(Q);
  pool_free(R);
  pool_free(beta);
  if (he != hipSuccess)
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP, "nlsg_tinyqr_qr failed: %s",
                hipGetErrorString(he));
  return NLSG_OK;
}

}  // extern "C"
