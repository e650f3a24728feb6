// nlsolver_amd/csrc/nlsg_lm.hip — host side of the batched LM engine + C-ABI.
#include <new>
#include <vector>

#include "nlsg_lm_kernels.h"

using namespace nlsg;

struct nlsg_lm {
  nlsg_lm_config cfg;
  LmParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *A_dev = nullptr, *y_dev = nullptr, *zero_dev = nullptr;
  bool has_data = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {
int lm_check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(NLSG_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n)
    return fail(NLSG_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
  hipDeviceProp_t prop;
  NLSG_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NLSG_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                device, prop.gcnArchName);
  return NLSG_OK;
}

int upload_theta(nlsg_lm *e, const double *theta_host) {
  const uint64_t B = e->p.batch, n = e->p.n;
  std::vector<double> padded(B * kLmN, 0.0);
  for (uint64_t b = 0; b < B; b++)
    for (uint64_t j = 0; j < n; j++) padded[b * kLmN + j] = theta_host[b * n + j];
  NLSG_HIP(hipMemcpy(e->p.theta, padded.data(), padded.size() * sizeof(double),
                     hipMemcpyHostToDevice));
  return NLSG_OK;
}

void launch_solve(nlsg_lm *e) {
  const dim3 grid(static_cast<unsigned>(e->p.batch)), block(256);
  if (e->cfg.solver == NLSG_LM_QR)
    hipLaunchKernelGGL(lm_solve_kernel<true>, grid, block, sizeof(LmShared) + sizeof(LmQrShared),
                       e->stream, e->p);
  else
    hipLaunchKernelGGL(lm_solve_kernel<false>, grid, block, sizeof(LmShared), e->stream, e->p);
}
}  // namespace

extern "C" {

int nlsg_lm_create(const nlsg_lm_config *cfg, nlsg_lm **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_lm_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_lm_config size mismatch (%u vs %zu)", cfg->struct_size,
                sizeof(nlsg_lm_config));
  if (cfg->objective != NLSG_OBJ_TANH_REGRESSION)
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (cfg->solver != NLSG_LM_CHOLESKY && cfg->solver != NLSG_LM_QR)
    return fail(NLSG_ERR_INVALID_ARG, "unknown solver %d", cfg->solver);
  if (cfg->n < 1 || cfg->n > kLmN || cfg->m < 1 || cfg->batch < 1)
    return fail(NLSG_ERR_INVALID_ARG, "need 1 <= n <= 64, m >= 1, batch >= 1");
  if (cfg->batch > 0x7fffffffull) return fail(NLSG_ERR_UNSUPPORTED, "batch too large");
  int rc = lm_check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));
  nlsg_lm *e = new (std::nothrow) nlsg_lm();
  if (!e) return fail(NLSG_ERR_OOM, "host allocation failed");
  e->cfg = *cfg;
  if (cfg->stream) {
    e->stream = static_cast<hipStream_t>(cfg->stream);
  } else {
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
      delete e;
      return fail(NLSG_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(he));
    }
    e->own_stream = true;
  }
  LmParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  const uint64_t B = cfg->batch, m = cfg->m;
  hipError_t he = hipSuccess;
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&e->A_dev), B * m * kLmN * 8);
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&e->y_dev), B * m * 8);
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&p.theta), B * kLmN * 8);
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&p.prob), B * sizeof(LmProblem));
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he == hipSuccess)
    he = hipFuncSetAttribute(reinterpret_cast<const void *>(lm_solve_kernel<false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(LmShared));
  if (he == hipSuccess)
    he = hipFuncSetAttribute(reinterpret_cast<const void *>(lm_solve_kernel<true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             sizeof(LmShared) + sizeof(LmQrShared));
  if (he != hipSuccess) {
    nlsg_lm_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device setup failed: %s", hipGetErrorString(he));
  }
  p.A = e->A_dev;
  p.y = e->y_dev;
  p.zero = e->zero_dev;
  p.batch = B;
  p.m = m;
  p.n = cfg->n;
  p.max_iter = cfg->max_iter;
  p.lambda0 = cfg->lambda;
  p.up = cfg->up;
  p.down = cfg->down;
  p.f_delta = cfg->f_delta;
  *out = e;
  return NLSG_OK;
}

int nlsg_lm_destroy(nlsg_lm *e) {
  if (!e) return NLSG_OK;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  hipFree(e->A_dev);
  hipFree(e->y_dev);
  hipFree(e->p.theta);
  hipFree(e->p.prob);
  hipFree(e->zero_dev);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
  delete e;
  return NLSG_OK;
}

int nlsg_lm_set_data(nlsg_lm *e, const double *a_host, const double *y_host) {
  if (!e || !a_host || !y_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  const uint64_t B = e->p.batch, m = e->p.m, n = e->p.n;
  if (n == kLmN) {
    NLSG_HIP(hipMemcpy(e->A_dev, a_host, B * m * kLmN * 8, hipMemcpyHostToDevice));
  } else {  // repack rows to the 64-column device layout, one problem at a time
    std::vector<double> row(m * kLmN);
    for (uint64_t b = 0; b < B; b++) {
      std::fill(row.begin(), row.end(), 0.0);
      for (uint64_t i = 0; i < m; i++)
        for (uint64_t j = 0; j < n; j++) row[i * kLmN + j] = a_host[(b * m + i) * n + j];
      NLSG_HIP(hipMemcpy(e->A_dev + b * m * kLmN, row.data(), m * kLmN * 8, hipMemcpyHostToDevice));
    }
  }
  NLSG_HIP(hipMemcpy(e->y_dev, y_host, B * m * 8, hipMemcpyHostToDevice));
  e->has_data = true;
  return NLSG_OK;
}

int nlsg_lm_minimize(nlsg_lm *e, double *theta_inout_host, nlsg_status *status_host,
                     double *lambda_out_host) {
  if (!e || !theta_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->has_data) return fail(NLSG_ERR_STATE, "nlsg_lm_set_data has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  int rc = upload_theta(e, theta_inout_host);
  if (rc) return rc;
  launch_solve(e);
  NLSG_HIP(hipGetLastError());
  NLSG_HIP(hipStreamSynchronize(e->stream));
  const uint64_t B = e->p.batch, n = e->p.n;
  std::vector<double> padded(B * kLmN);
  NLSG_HIP(hipMemcpy(padded.data(), e->p.theta, padded.size() * 8, hipMemcpyDeviceToHost));
  for (uint64_t b = 0; b < B; b++)
    for (uint64_t j = 0; j < n; j++) theta_inout_host[b * n + j] = padded[b * kLmN + j];
  std::vector<LmProblem> pr(B);
  NLSG_HIP(hipMemcpy(pr.data(), e->p.prob, B * sizeof(LmProblem), hipMemcpyDeviceToHost));
  for (uint64_t b = 0; b < B; b++) {
    if (status_host) {
      nlsg_status &st = status_host[b];
      st.f_value = pr[b].f;
      st.iteration = pr[b].iter;
      st.function_calls_used = pr[b].fcalls;  // f, grad and hess are evaluated together
      st.gradient_evals_used = pr[b].fcalls;
      st.hessian_evals_used = pr[b].fcalls;
      st.best_index = b;
      st.val_no_change = 0;
      st.std_err = pr[b].lambda;
      st.done = pr[b].done;
      st.reserved = 0;
    }
    if (lambda_out_host) lambda_out_host[b] = pr[b].lambda;
  }
  return NLSG_OK;
}

int nlsg_lm_time_solve(nlsg_lm *e, const double *theta0_host, uint32_t repeats, float *ms_total) {
  if (!e || !theta0_host || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->has_data) return fail(NLSG_ERR_STATE, "nlsg_lm_set_data has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  float total = 0.f;
  for (uint32_t r = 0; r < repeats; r++) {
    int rc = upload_theta(e, theta0_host);
    if (rc) return rc;
    NLSG_HIP(hipEventRecord(e->ev0, e->stream));
    launch_solve(e);
    NLSG_HIP(hipEventRecord(e->ev1, e->stream));
    NLSG_HIP(hipEventSynchronize(e->ev1));
    NLSG_HIP(hipGetLastError());
    float ms = 0.f;
    NLSG_HIP(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    total += ms;
  }
  *ms_total = total;
  return NLSG_OK;
}

}  // extern "C"
