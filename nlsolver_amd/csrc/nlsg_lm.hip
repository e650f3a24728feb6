// nlsolver_amd/csrc/nlsg_lm.hip — host side of the batched LM engine + C-ABI.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "nlsg_lm_kernels.h"
#include "nlsg_rtc.h"

using namespace nlsg;

struct nlsg_lm {
  nlsg_lm_config cfg;
  LmParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *A_dev = nullptr, *y_dev = nullptr, *zero_dev = nullptr;
  unsigned long long *count_dev = nullptr;
  bool has_data = false;
  bool wide = false;   // n > 64: the workgroup-per-problem kernels (lm_wide_*)
  bool wide_valu = false;  // NLSG_LM_WIDE_MFMA=0: the VALU contraction at every n > 64 (A/B switch)
  bool wide256 = true;     // NLSG_LM_WIDE256=0: the super-block evaluation at 128 < n <= 256 too (A/B switch)
  bool wide_chol = true;   // NLSG_LM_WIDE_CHOL=0: the steps before the blocked one — LDS-resident at
                           // n <= 128, column by column beyond (A/B switch)
  uint64_t ldt = kLmN; // row stride of theta / gg on the device: 64, or n when wide
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  LmRtcKernels rtc;  // objective == NLSG_OBJ_CUSTOM: the kernel hiprtc built for it
};

namespace {
int upload_theta(nlsg_lm *e, const double *theta_host) {
  const uint64_t B = e->p.batch, n = e->p.n;
  if (e->wide) {
    NLSG_HIP(hipMemcpy(e->p.theta, theta_host, B * n * sizeof(double), hipMemcpyHostToDevice));
    return NLSG_OK;
  }
  std::vector<double> padded(B * kLmN, 0.0);
  for (uint64_t b = 0; b < B; b++)
    for (uint64_t j = 0; j < n; j++) padded[b * kLmN + j] = theta_host[b * n + j];
  NLSG_HIP(hipMemcpy(e->p.theta, padded.data(), padded.size() * sizeof(double),
                     hipMemcpyHostToDevice));
  return NLSG_OK;
}

// All problems in lock step; the host polls the number of unfinished problems every few
// iterations. Cholesky solver: one launch per iteration (step k, then evaluation k + 1, one wave
// per problem). QR solver: the step is a kernel of its own (one workgroup per problem) between
// two evaluation launches.
bool lm_fd_objective(int objective) {
  return objective == NLSG_OBJ_ROSENBROCK || objective == NLSG_OBJ_SPHERE ||
         objective == NLSG_OBJ_STYBLINSKI_TANG || objective == NLSG_OBJ_RASTRIGIN ||
         objective == NLSG_OBJ_CUSTOM;
}

// finite-difference model: step k + evaluation k + 1 (first: the evaluation at x0 only)
void launch_fd_iter(nlsg_lm *e, int first) {
  const dim3 grid(static_cast<unsigned>(e->p.batch));
  // (+ 66: the point and a zero behind it, for the reference-order evaluation's lanes)
  const unsigned lds = (64 * lm_fd_chunks(e->p.n) + 128 + 66) * sizeof(double);
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &first};
    launch_module_kernel(e->rtc.iter, grid.x, 64, lds, e->stream, args);
    return;
  }
  if (e->cfg.solver == NLSG_LM_CHOLESKY_REFERENCE_ORDER) {  // (Rastrigin is rejected at creation)
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK:
        hipLaunchKernelGGL((lm_fd_iter_kernel<NLSG_OBJ_ROSENBROCK, true>), grid, dim3(64), lds, e->stream, e->p, first);
        break;
      case NLSG_OBJ_SPHERE:
        hipLaunchKernelGGL((lm_fd_iter_kernel<NLSG_OBJ_SPHERE, true>), grid, dim3(64), lds, e->stream, e->p, first);
        break;
      default:
        hipLaunchKernelGGL((lm_fd_iter_kernel<NLSG_OBJ_STYBLINSKI_TANG, true>), grid, dim3(64), lds, e->stream, e->p, first);
        break;
    }
    return;
  }
  switch (e->cfg.objective) {
    case NLSG_OBJ_ROSENBROCK:
      hipLaunchKernelGGL(lm_fd_iter_kernel<NLSG_OBJ_ROSENBROCK>, grid, dim3(64), lds, e->stream, e->p, first);
      break;
    case NLSG_OBJ_SPHERE:
      hipLaunchKernelGGL(lm_fd_iter_kernel<NLSG_OBJ_SPHERE>, grid, dim3(64), lds, e->stream, e->p, first);
      break;
    case NLSG_OBJ_RASTRIGIN:
      hipLaunchKernelGGL(lm_fd_iter_kernel<NLSG_OBJ_RASTRIGIN>, grid, dim3(64), lds, e->stream, e->p, first);
      break;
    default:
      hipLaunchKernelGGL(lm_fd_iter_kernel<NLSG_OBJ_STYBLINSKI_TANG>, grid, dim3(64), lds, e->stream, e->p, first);
      break;
  }
}

// n > 64: evaluation (f, g, H at the current point) as one launch, a workgroup per problem
template <int OBJ>
void launch_wide_fd_ref(nlsg_lm *e, dim3 grid, int first) {  // NLSG_LM_CHOLESKY_REFERENCE_ORDER past 64 parameters
  hipLaunchKernelGGL((lm_wide_fd_lanes_kernel<OBJ>), grid, dim3(256),
                     static_cast<unsigned>(lm_wide_fd_lanes_lds_bytes(e->p.n)), e->stream, e->p, first);
}
template <int OBJ>
void launch_wide_fd(nlsg_lm *e, dim3 grid, int first) {
  if (e->cfg.solver == NLSG_LM_CHOLESKY_REFERENCE_ORDER) {
    if constexpr (OBJ != NLSG_OBJ_RASTRIGIN) launch_wide_fd_ref<OBJ>(e, grid, first);  // (rejected at creation)
    return;
  }
  switch (lm_wide_chunks(e->p.n)) {
    case 1: hipLaunchKernelGGL((lm_wide_fd_eval_kernel<OBJ, 1>), grid, dim3(lm_wide_fd_threads(1)), 0, e->stream, e->p, first); break;
    case 2: hipLaunchKernelGGL((lm_wide_fd_eval_kernel<OBJ, 2>), grid, dim3(lm_wide_fd_threads(2)), 0, e->stream, e->p, first); break;
    case 4: hipLaunchKernelGGL((lm_wide_fd_eval_kernel<OBJ, 4>), grid, dim3(lm_wide_fd_threads(4)), 0, e->stream, e->p, first); break;
    default: hipLaunchKernelGGL((lm_wide_fd_eval_kernel<OBJ, 8>), grid, dim3(lm_wide_fd_threads(8)), 0, e->stream, e->p, first); break;
  }
}
template <bool EVEN>
void launch_wide_chol_step_t(nlsg_lm *e, dim3 grid) {
  switch (lm_wchol_threads(e->p.n)) {
    case 256: hipLaunchKernelGGL((lm_wide_chol_step_kernel<256, EVEN>), grid, dim3(256), lm_wchol_lds_bytes(256), e->stream, e->p); break;
    case 512: hipLaunchKernelGGL((lm_wide_chol_step_kernel<512, EVEN>), grid, dim3(512), lm_wchol_lds_bytes(512), e->stream, e->p); break;
    default: hipLaunchKernelGGL((lm_wide_chol_step_kernel<1024, EVEN>), grid, dim3(1024), lm_wchol_lds_bytes(1024), e->stream, e->p); break;
  }
}
void launch_wide_chol_step(nlsg_lm *e, dim3 grid) {
  if (e->p.n & 1) launch_wide_chol_step_t<false>(e, grid);
  else launch_wide_chol_step_t<true>(e, grid);
}
void launch_wide_eval(nlsg_lm *e, int first) {
  // finite-difference model: enough workgroups per problem to fill the device when the batch is small
  const unsigned split = e->p.fd ? static_cast<unsigned>(std::min<uint64_t>(
                                       64, std::max<uint64_t>(1, 2048 / e->p.batch))) : 1u;
  const dim3 grid(static_cast<unsigned>(e->p.batch), split);
  if (!e->p.fd) {
    // up to 128 parameters: one pass over A, J^T J on the matrix cores
    if (e->wide_valu)
      hipLaunchKernelGGL(lm_wide_tanh_eval_kernel, grid, dim3(kLmWideThreads), 0, e->stream, e->p, first);
    else if (e->p.n <= 128)
      hipLaunchKernelGGL(lm_wide128x8_tanh_eval_kernel, grid, dim3(512), 0, e->stream, e->p, first);
    else if (e->p.n <= 256 && e->wide256)  // one pass over A still: 136 tiles on eight waves
      hipLaunchKernelGGL(lm_wide256x8_tanh_eval_kernel, grid, dim3(512), sizeof(LmWide256Shared), e->stream, e->p, first);
    else  // super-blocks of 128 x 128, each on the matrix cores
      hipLaunchKernelGGL(lm_wide_mfma_tanh_eval_kernel, grid, dim3(256), 0, e->stream, e->p, first);
    return;
  }
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &first};
    if (e->cfg.solver == NLSG_LM_CHOLESKY_REFERENCE_ORDER)
      launch_module_kernel(e->rtc.iter, grid.x, 256, static_cast<unsigned>(lm_wide_fd_lanes_lds_bytes(e->p.n)),
                           e->stream, args, grid.y);
    else
      launch_module_kernel(e->rtc.iter, grid.x, lm_wide_fd_threads(lm_wide_chunks(e->p.n)), 0, e->stream,
                           args, grid.y);
    return;
  }
  switch (e->cfg.objective) {
    case NLSG_OBJ_ROSENBROCK: launch_wide_fd<NLSG_OBJ_ROSENBROCK>(e, grid, first); break;
    case NLSG_OBJ_SPHERE: launch_wide_fd<NLSG_OBJ_SPHERE>(e, grid, first); break;
    case NLSG_OBJ_RASTRIGIN: launch_wide_fd<NLSG_OBJ_RASTRIGIN>(e, grid, first); break;
    default: launch_wide_fd<NLSG_OBJ_STYBLINSKI_TANG>(e, grid, first); break;
  }
}

int launch_solve(nlsg_lm *e) {
  const dim3 grid(static_cast<unsigned>(e->p.batch));
  const bool qr = e->cfg.solver == NLSG_LM_QR, fd = e->p.fd != 0;
  if (e->wide)
    launch_wide_eval(e, 1);
  else if (fd)
    launch_fd_iter(e, 1);
  else
    hipLaunchKernelGGL(lm_iter_kernel, grid, dim3(64), 0, e->stream, e->p, 1, 0);
  uint64_t launched = 0;
  for (;;) {
    // max_iter iterations plus the turn whose stop test fires
    const uint64_t left = e->p.max_iter + 1 - launched;
    const uint64_t chunk = left < 8 ? left : 8;
    for (uint64_t i = 0; i < chunk; i++) {
      if (e->wide) {
        if (e->cfg.solver == NLSG_LM_CHOLESKY_REFERENCE_ORDER)  // the reference's own order of operations
          hipLaunchKernelGGL(lm_wide_step_kernel<true>, grid, dim3(kLmWideThreads), 0, e->stream, e->p);
        else if (e->wide_chol && !e->wide_valu)  // blocked, the panel sums on the matrix cores
          launch_wide_chol_step(e, grid);
        else if (e->p.n <= 128 && !e->wide_valu)  // the damped matrix in LDS, one thread per row
          hipLaunchKernelGGL(lm_wide128_step_kernel, grid, dim3(128), sizeof(LmWide128StepShared),
                             e->stream, e->p);
        else
          hipLaunchKernelGGL(lm_wide_step_kernel<false>, grid, dim3(kLmWideThreads), 0, e->stream, e->p);
        launch_wide_eval(e, 0);
      } else if (fd) {
        launch_fd_iter(e, 0);
      } else if (qr) {
        hipLaunchKernelGGL(lm_qr_step_kernel<kLmQrThreads>, grid, dim3(kLmQrThreads), sizeof(LmQrShared), e->stream, e->p);
        hipLaunchKernelGGL(lm_iter_kernel, grid, dim3(64), 0, e->stream, e->p, 0, 0);
      } else {
        hipLaunchKernelGGL(lm_iter_kernel, grid, dim3(64), 0, e->stream, e->p, 0, 1);
      }
    }
    launched += chunk;
    if (launched >= e->p.max_iter + 1) break;
    NLSG_HIP(hipMemsetAsync(e->count_dev, 0, 8, e->stream));
    hipLaunchKernelGGL(lm_count_unfinished_kernel,
                       dim3(static_cast<unsigned>((e->p.batch + 255) / 256)), dim3(256), 0,
                       e->stream, e->p, e->count_dev);
    unsigned long long open = 0;
    NLSG_HIP(hipMemcpyAsync(&open, e->count_dev, 8, hipMemcpyDeviceToHost, e->stream));
    NLSG_HIP(hipStreamSynchronize(e->stream));
    if (open == 0) break;
  }
  return NLSG_OK;
}
}  // namespace

static int lm_create(const nlsg_lm_config *cfg, const nlsg_custom_objective *custom, nlsg_lm **out);

extern "C" {

int nlsg_lm_create(const nlsg_lm_config *cfg, nlsg_lm **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_lm_create_custom");
  PhaseClock clk;
  const int rc = lm_create(cfg, nullptr, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

int nlsg_lm_create_custom(const nlsg_lm_config *cfg, const nlsg_custom_objective *obj, nlsg_lm **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  PhaseClock clk;
  const int rc = lm_create(cfg, obj, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

}  // extern "C"

static int lm_create(const nlsg_lm_config *cfg, const nlsg_custom_objective *custom, nlsg_lm **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_lm_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_lm_config size mismatch (%u vs %zu)", cfg->struct_size,
                sizeof(nlsg_lm_config));
  const bool fd = lm_fd_objective(cfg->objective);
  if (cfg->objective != NLSG_OBJ_TANH_REGRESSION && !fd)
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  const bool ref_order = cfg->solver == NLSG_LM_CHOLESKY_REFERENCE_ORDER;
  if (cfg->solver != NLSG_LM_CHOLESKY && cfg->solver != NLSG_LM_QR && !ref_order)
    return fail(NLSG_ERR_INVALID_ARG, "unknown solver %d", cfg->solver);
  if (fd && cfg->solver == NLSG_LM_QR)
    return fail(NLSG_ERR_UNSUPPORTED,
                "the finite-difference model runs the reference's own solve (Cholesky) only");
  if (ref_order && (!fd || cfg->objective == NLSG_OBJ_RASTRIGIN ||
                    (custom && custom->chain == NLSG_CUSTOM_VECTOR)))
    return fail(NLSG_ERR_UNSUPPORTED,
                "NLSG_LM_CHOLESKY_REFERENCE_ORDER covers the default functors on objectives given by "
                "their terms (Rosenbrock / Sphere / Styblinski-Tang, custom term bodies): Rastrigin's "
                "cosine is the device's own, a whole-vector body's x.sum() adds in the lane-tree order");
  if (cfg->n < 1 || (!fd && cfg->m < 1) || cfg->batch < 1)
    return fail(NLSG_ERR_INVALID_ARG, "need n >= 1, m >= 1, batch >= 1");
  const bool wide = cfg->n > kLmN;
  if (cfg->n > kLmWideMaxN)
    return fail(NLSG_ERR_UNSUPPORTED, "n = %llu > %d (a thread of the step follows at most four rows)",
                (unsigned long long)cfg->n, kLmWideMaxN);
  if (wide && cfg->solver == NLSG_LM_QR)
    return fail(NLSG_ERR_UNSUPPORTED, "the tinyqr solve is built for n <= 64; n = %llu runs the "
                "class's own Cholesky solve", (unsigned long long)cfg->n);
  if (cfg->batch > 0x7fffffffull) return fail(NLSG_ERR_UNSUPPORTED, "batch too large");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));
  nlsg_lm *e = new (std::nothrow) nlsg_lm();
  if (!e) return fail(NLSG_ERR_OOM, "host allocation failed");
  e->cfg = *cfg;
  e->wide = wide;
  {
    const char *sw = std::getenv("NLSG_LM_WIDE_MFMA");
    e->wide_valu = sw && sw[0] == '0';
    const char *w2 = std::getenv("NLSG_LM_WIDE256");
    e->wide256 = !(w2 && w2[0] == '0');
    const char *wc = std::getenv("NLSG_LM_WIDE_CHOL");
    e->wide_chol = !(wc && wc[0] == '0');
  }
  e->ldt = wide ? cfg->n : kLmN;
  if (cfg->stream) {
    e->stream = borrowed_stream(cfg->stream);
  } else {
    hipError_t he = pool_stream_get(&e->stream);
    if (he != hipSuccess) {
      delete e;
      return fail(NLSG_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(he));
    }
    e->own_stream = true;
  }
  LmParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  const uint64_t B = cfg->batch, m = fd ? 0 : cfg->m;  // no design matrix behind an objective
  hipError_t he = hipSuccess;
  const uint64_t nstep = wide ? 0 : (m + 15) / 16;
  p.nstep = nstep;
  if (he == hipSuccess && nstep)
    he = pool_malloc(reinterpret_cast<void **>(&e->A_dev), nstep * B * 16 * kLmN * 8);
  if (he == hipSuccess && nstep)
    he = pool_malloc(reinterpret_cast<void **>(&e->y_dev), nstep * B * 16 * 8);
  if (he == hipSuccess && wide && m) {  // the caller's layout, [batch][m][n]
    he = pool_malloc(reinterpret_cast<void **>(&e->A_dev), B * m * cfg->n * 8);
    if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->y_dev), B * m * 8);
    if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.rw), B * 2 * m * 8);
  }
  if (he == hipSuccess && wide)
    he = pool_malloc(reinterpret_cast<void **>(&p.Hw), B * cfg->n * cfg->n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.theta), B * e->ldt * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.prob), B * sizeof(LmProblem));
  if (he == hipSuccess && !wide) he = pool_malloc(reinterpret_cast<void **>(&p.Hg), B * kLmTri * 8);
  if (he == hipSuccess && !wide) he = hipMemset(p.Hg, 0, B * kLmTri * 8);  // rows past n are never written
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.gg), B * e->ldt * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->count_dev), 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he == hipSuccess)
    he = hipFuncSetAttribute(reinterpret_cast<const void *>(lm_qr_step_kernel<kLmQrThreads>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(LmQrShared));
  if (he == hipSuccess)
    he = hipFuncSetAttribute(reinterpret_cast<const void *>(lm_wide128_step_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(LmWide128StepShared));
  if (he == hipSuccess)
    he = hipFuncSetAttribute(reinterpret_cast<const void *>(lm_wide256x8_tanh_eval_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(LmWide256Shared));
  for (int ev = 0; ev < 2 && he == hipSuccess; ev++) {  // 139 KB at 1024 threads, 70 KB at 512
    he = hipFuncSetAttribute(ev ? reinterpret_cast<const void *>(lm_wide_chol_step_kernel<1024, true>)
                                : reinterpret_cast<const void *>(lm_wide_chol_step_kernel<1024, false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lm_wchol_lds_bytes(1024));
    if (he == hipSuccess)
      he = hipFuncSetAttribute(ev ? reinterpret_cast<const void *>(lm_wide_chol_step_kernel<512, true>)
                                  : reinterpret_cast<const void *>(lm_wide_chol_step_kernel<512, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lm_wchol_lds_bytes(512));
  }
  if (he != hipSuccess) {
    nlsg_lm_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device setup failed: %s", hipGetErrorString(he));
  }
  if (custom) {
    const uint64_t n = cfg->n;
    const int rc2 = rtc_build_lm(custom, !wide ? 0 : lm_wide_chunks(n), ref_order, &e->rtc);
    if (rc2) {
      nlsg_lm_destroy(e);
      return rc2;
    }
  }
  p.A = e->A_dev;
  p.y = e->y_dev;
  p.Aw = e->A_dev;
  p.yw = e->y_dev;
  p.zero = e->zero_dev;
  p.batch = B;
  p.m = m;
  p.n = cfg->n;
  p.max_iter = cfg->max_iter;
  p.lambda0 = cfg->lambda;
  p.up = cfg->up;
  p.down = cfg->down;
  p.f_delta = cfg->f_delta;
  p.fd = fd ? 1 : 0;
  // the matrix-core evaluations (but the four-wave A/B form) hand is_diagonal's verdict to the step
  p.verdict = (wide && !fd && !e->wide_valu) ? 1 : 0;
  p.eps_h = std::pow(DBL_EPSILON, 1.0 / 4.0);  // fin_diff_h's step (:1454)
  e->has_data = fd;  // the model is the objective itself
  *out = e;
  return NLSG_OK;
}

extern "C" {

int nlsg_lm_destroy(nlsg_lm *e) {
  if (!e) return NLSG_OK;
  PhaseClock clk;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  pool_free(e->A_dev);
  pool_free(e->y_dev);
  pool_free(e->p.theta);
  pool_free(e->p.prob);
  pool_free(e->p.Hg);
  pool_free(e->p.Hw);
  pool_free(e->p.rw);
  pool_free(e->p.gg);
  pool_free(e->count_dev);
  pool_free(e->zero_dev);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  rtc_release(&e->rtc);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  call_timing().destroy_ms = clk.lap();
  return NLSG_OK;
}

int nlsg_lm_set_data(nlsg_lm *e, const double *a_host, const double *y_host) {
  if (!e || !a_host || !y_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (e->p.fd) return fail(NLSG_ERR_STATE, "this engine minimises a built-in objective: no data");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  PhaseClock clk;
  const uint64_t B = e->p.batch, m = e->p.m, n = e->p.n;
  if (e->wide) {  // kept as handed over
    NLSG_HIP(hipMemcpy(e->A_dev, a_host, B * m * n * 8, hipMemcpyHostToDevice));
    NLSG_HIP(hipMemcpy(e->y_dev, y_host, B * m * 8, hipMemcpyHostToDevice));
    e->has_data = true;
    call_timing().upload_ms = clk.lap();
    return NLSG_OK;
  }
  // host layout [problem][m][n] -> device layout [row group][problem][16][64] (zero padded):
  // problems that run side by side read one contiguous stretch of HBM per row group instead
  // of addresses a whole problem (m * 512 bytes) apart, which camp on a few channels.
  // The matrices cross PCIe in chunks of problems through two staging buffers on a copy stream
  // of their own while the previous chunk is repacked: from pinned host memory (nlsg_host_alloc)
  // the copies are plain DMA at the link's rate; from pageable memory the runtime stages them.
  const uint64_t per = m * n * 8;
  uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / per);  // ~64 MiB per staging buffer
  if (chunk > B) chunk = B;
  double *raw[2] = {nullptr, nullptr}, *y_raw = nullptr;
  hipStream_t copy = nullptr;
  hipEvent_t landed[2] = {nullptr, nullptr}, packed[2] = {nullptr, nullptr};
  hipError_t he = pool_stream_get(&copy);
  for (int k = 0; k < 2 && he == hipSuccess; k++) {
    he = pool_malloc(reinterpret_cast<void **>(&raw[k]), chunk * per);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&landed[k], hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&packed[k], hipEventDisableTiming);
  }
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&y_raw), B * m * 8);
  if (he == hipSuccess) he = hipMemcpyAsync(y_raw, y_host, B * m * 8, hipMemcpyHostToDevice, copy);
  uint64_t k = 0;
  for (uint64_t b0 = 0; b0 < B && he == hipSuccess; b0 += chunk, k++) {
    const uint64_t nbp = std::min(chunk, B - b0);
    const int slot = static_cast<int>(k & 1);
    if (k >= 2) he = hipStreamWaitEvent(copy, packed[slot], 0);  // the buffer's previous chunk is repacked
    if (he == hipSuccess)
      he = hipMemcpyAsync(raw[slot], a_host + b0 * m * n, nbp * per, hipMemcpyHostToDevice, copy);
    if (he == hipSuccess) he = hipEventRecord(landed[slot], copy);
    if (he == hipSuccess) he = hipStreamWaitEvent(e->stream, landed[slot], 0);
    if (he == hipSuccess) {
      const uint64_t total = e->p.nstep * nbp * 16 * kLmN;
      hipLaunchKernelGGL(lm_repack_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256),
                         0, e->stream, e->p, raw[slot], y_raw, e->A_dev, e->y_dev, b0, nbp);
      he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipEventRecord(packed[slot], e->stream);
  }
  if (he == hipSuccess) he = hipStreamSynchronize(e->stream);  // the host buffers are borrowed for this call only
  if (copy) hipStreamSynchronize(copy);
  for (int q = 0; q < 2; q++) {
    pool_free(raw[q]);
    if (landed[q]) hipEventDestroy(landed[q]);
    if (packed[q]) hipEventDestroy(packed[q]);
  }
  pool_free(y_raw);
  if (copy) pool_stream_put(e->cfg.device, copy);
  NLSG_HIP(he);
  e->has_data = true;
  call_timing().upload_ms = clk.lap();
  return NLSG_OK;
}

// Page-locked host memory for buffers that cross PCIe at the boundary (design matrices, start
// points): what lives there is copied by DMA at the link's rate instead of being staged.
int nlsg_host_alloc(void **out, uint64_t bytes) {
  if (!out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  NLSG_HIP(hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault));
  return NLSG_OK;
}
int nlsg_host_free(void *ptr) {
  if (ptr) NLSG_HIP(hipHostFree(ptr));
  return NLSG_OK;
}

int nlsg_lm_set_solver(nlsg_lm *e, int32_t solver) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (solver != NLSG_LM_CHOLESKY && solver != NLSG_LM_QR)
    return fail(NLSG_ERR_INVALID_ARG, "unknown solver %d (the reference-order mode is chosen at creation)",
                solver);
  if (e->cfg.solver == NLSG_LM_CHOLESKY_REFERENCE_ORDER)  // the bit-for-bit parity mode is a property of
    return fail(NLSG_ERR_STATE,                           // the engine: it is not left silently
                "this engine was created in NLSG_LM_CHOLESKY_REFERENCE_ORDER; create another engine "
                "for a different solver");
  if (e->wide && solver != NLSG_LM_CHOLESKY)
    return fail(NLSG_ERR_UNSUPPORTED, "the tinyqr solve is built for n <= 64");
  if (e->p.fd && solver != NLSG_LM_CHOLESKY)
    return fail(NLSG_ERR_UNSUPPORTED,
                "the finite-difference model runs the reference's own solve (Cholesky) only");
  e->cfg.solver = solver;
  return NLSG_OK;
}

int nlsg_lm_minimize(nlsg_lm *e, double *theta_inout_host, nlsg_status *status_host,
                     double *lambda_out_host) {
  if (!e || !theta_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->has_data) return fail(NLSG_ERR_STATE, "nlsg_lm_set_data has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  PhaseClock clk;
  int rc = upload_theta(e, theta_inout_host);
  if (rc) return rc;
  call_timing().init_ms = clk.lap();
  rc = launch_solve(e);
  if (rc) return rc;
  NLSG_HIP(launches_status());
  NLSG_HIP(hipStreamSynchronize(e->stream));
  call_timing().iterate_ms = clk.lap();
  const uint64_t B = e->p.batch, n = e->p.n;
  if (e->wide) {
    NLSG_HIP(hipMemcpy(theta_inout_host, e->p.theta, B * n * 8, hipMemcpyDeviceToHost));
  } else {
    std::vector<double> padded(B * kLmN);
    NLSG_HIP(hipMemcpy(padded.data(), e->p.theta, padded.size() * 8, hipMemcpyDeviceToHost));
    for (uint64_t b = 0; b < B; b++)
      for (uint64_t j = 0; j < n; j++) theta_inout_host[b * n + j] = padded[b * kLmN + j];
  }
  std::vector<LmProblem> pr(B);
  NLSG_HIP(hipMemcpy(pr.data(), e->p.prob, B * sizeof(LmProblem), hipMemcpyDeviceToHost));
  for (uint64_t b = 0; b < B; b++) {
    if (status_host) {
      nlsg_status &st = status_host[b];
      st.f_value = pr[b].f;
      st.iteration = pr[b].iter;
      // f, grad and hess are evaluated together; behind the default functors every probe of
      // fin_diff (4 n) and fin_diff_h (16 n^2) is a call of the objective
      st.function_calls_used = pr[b].fcalls * (e->p.fd ? 1 + 4 * n + 16 * n * n : 1);
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
  call_timing().readback_ms = clk.lap();
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
    rc = launch_solve(e);
    if (rc) return rc;
    NLSG_HIP(hipEventRecord(e->ev1, e->stream));
    NLSG_HIP(hipEventSynchronize(e->ev1));
    NLSG_HIP(launches_status());
    float ms = 0.f;
    NLSG_HIP(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    total += ms;
  }
  *ms_total = total;
  return NLSG_OK;
}

// Times `repeats` launches of the dominant kernel of the Cholesky pipeline (f, g, H at theta0
// for every problem) on the engine's stream, HIP events around each launch.
int nlsg_lm_time_eval_kernel(nlsg_lm *e, const double *theta0_host, uint32_t repeats, float *ms_total) {
  if (!e || !theta0_host || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (e->p.fd) return fail(NLSG_ERR_UNSUPPORTED, "Gauss-Newton model only");
  if (!e->has_data) return fail(NLSG_ERR_STATE, "nlsg_lm_set_data has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  int rc = upload_theta(e, theta0_host);
  if (rc) return rc;
  float total = 0.f;
  for (uint32_t r = 0; r < repeats; r++) {
    NLSG_HIP(hipEventRecord(e->ev0, e->stream));
    if (e->wide)
      launch_wide_eval(e, 1);
    else
      hipLaunchKernelGGL(lm_iter_kernel, dim3(static_cast<unsigned>(e->p.batch)), dim3(64), 0,
                         e->stream, e->p, 1, 0);
    NLSG_HIP(hipEventRecord(e->ev1, e->stream));
    NLSG_HIP(hipEventSynchronize(e->ev1));
    NLSG_HIP(launches_status());
    float ms = 0.f;
    NLSG_HIP(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    total += ms;
  }
  *ms_total = total;
  return NLSG_OK;
}

// Times `repeats` launches of the QR step kernel alone (stop tests, damped matrix, Givens QR,
// back-substitution, update) after one evaluation at theta0 has produced H and g.
int nlsg_lm_time_qr_kernel(nlsg_lm *e, const double *theta0_host, uint32_t repeats, float *ms_total) {
  if (!e || !theta0_host || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (e->p.fd) return fail(NLSG_ERR_UNSUPPORTED, "Gauss-Newton model only");
  if (e->wide) return fail(NLSG_ERR_UNSUPPORTED, "the tinyqr step kernel: n <= 64 only");
  if (!e->has_data) return fail(NLSG_ERR_STATE, "nlsg_lm_set_data has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  int rc = upload_theta(e, theta0_host);
  if (rc) return rc;
  const dim3 grid(static_cast<unsigned>(e->p.batch));
  hipLaunchKernelGGL(lm_iter_kernel, grid, dim3(64), 0, e->stream, e->p, 1, 0);
  float total = 0.f;
  for (uint32_t r = 0; r < repeats; r++) {
    NLSG_HIP(hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(lm_qr_step_kernel<kLmQrThreads>, grid, dim3(kLmQrThreads), sizeof(LmQrShared),
                       e->stream, e->p);
    NLSG_HIP(hipEventRecord(e->ev1, e->stream));
    NLSG_HIP(hipEventSynchronize(e->ev1));
    NLSG_HIP(launches_status());
    float ms = 0.f;
    NLSG_HIP(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    total += ms;
  }
  *ms_total = total;
  return NLSG_OK;
}

}  // extern "C"
