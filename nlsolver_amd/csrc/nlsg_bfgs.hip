// nlsolver_amd/csrc/nlsg_bfgs.hip — host side of the batched BFGS engine + C-ABI.
#include <algorithm>
#include <new>
#include <vector>

#include "nlsg_bfgs_kernels.h"
#include "nlsg_rtc.h"

using namespace nlsg;

struct nlsg_bfgs {
  BfgsRtcKernels rtc;  // objective == NLSG_OBJ_CUSTOM: finite-difference kernels hiprtc built
  nlsg_bfgs_config cfg;
  BfgsParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *qd_dev = nullptr, *qb_dev = nullptr, *zero_dev = nullptr;
  unsigned long long *count_dev = nullptr;
  int chunks = 0;
  bool vec = false, initialised = false;
  uint32_t bpp = 0;  // blocks per problem in the H-streaming kernels
  bool symmetric = false;  // NLSG_BFGS_SYMMETRIC: upper blocks of H only
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
};

namespace {

#define BFGS_DISPATCH(KERNEL, grid, ...)                                                         \
  do {                                                                                           \
    const dim3 g_(grid), b_(256);                                                                \
    switch (e->chunks * 2 + (e->vec ? 1 : 0)) {                                                  \
      case 2: hipLaunchKernelGGL((KERNEL<1, false>), g_, b_, 0, e->stream, __VA_ARGS__); break;  \
      case 3: hipLaunchKernelGGL((KERNEL<1, true>), g_, b_, 0, e->stream, __VA_ARGS__); break;   \
      case 4: hipLaunchKernelGGL((KERNEL<2, false>), g_, b_, 0, e->stream, __VA_ARGS__); break;  \
      case 5: hipLaunchKernelGGL((KERNEL<2, true>), g_, b_, 0, e->stream, __VA_ARGS__); break;   \
      case 8: hipLaunchKernelGGL((KERNEL<4, false>), g_, b_, 0, e->stream, __VA_ARGS__); break;  \
      case 9: hipLaunchKernelGGL((KERNEL<4, true>), g_, b_, 0, e->stream, __VA_ARGS__); break;   \
      case 16: hipLaunchKernelGGL((KERNEL<8, false>), g_, b_, 0, e->stream, __VA_ARGS__); break; \
      case 17: hipLaunchKernelGGL((KERNEL<8, true>), g_, b_, 0, e->stream, __VA_ARGS__); break;  \
      default: break;                                                                            \
    }                                                                                            \
  } while (0)

// search / init kernels also depend on what is minimised
#define BFGS_MODEL_CASE(KERNEL, C, V, grid, ...)                                                    \
  const unsigned fd_lds_ = e->p.seq ? static_cast<unsigned>(bfgs_fd_seq_lds_bytes(C)) : 0u;          \
  switch (e->p.model) {                                                                             \
    case kBfgsQuad:                                                                                 \
      hipLaunchKernelGGL((KERNEL<C, V, kBfgsQuad>), dim3(grid), dim3(256), fd_lds_, e->stream, __VA_ARGS__); \
      break;                                                                                        \
    case NLSG_OBJ_ROSENBROCK:                                                                       \
        hipLaunchKernelGGL((KERNEL<C, V, NLSG_OBJ_ROSENBROCK>), dim3(grid), dim3(256), fd_lds_,  e->stream, \
                           __VA_ARGS__);                                                            \
      break;                                                                                        \
    case NLSG_OBJ_SPHERE:                                                                           \
        hipLaunchKernelGGL((KERNEL<C, V, NLSG_OBJ_SPHERE>), dim3(grid), dim3(256), fd_lds_,  e->stream,    \
                           __VA_ARGS__);                                                            \
      break;                                                                                        \
    case NLSG_OBJ_STYBLINSKI_TANG:                                                                  \
        hipLaunchKernelGGL((KERNEL<C, V, NLSG_OBJ_STYBLINSKI_TANG>), dim3(grid), dim3(256), fd_lds_,       \
                           e->stream, __VA_ARGS__);                                                 \
      break;                                                                                        \
    case NLSG_OBJ_RASTRIGIN:                                                                        \
        hipLaunchKernelGGL((KERNEL<C, V, NLSG_OBJ_RASTRIGIN>), dim3(grid), dim3(256), fd_lds_,  e->stream, \
                           __VA_ARGS__);                                                            \
      break;                                                                                        \
    default: break;                                                                                 \
  }
#define BFGS_DISPATCH_MODEL(KERNEL, grid, ...)                           \
  do {                                                                   \
    switch (e->chunks * 2 + (e->vec ? 1 : 0)) {                          \
      case 2: { BFGS_MODEL_CASE(KERNEL, 1, false, grid, __VA_ARGS__) } break; \
      case 3: { BFGS_MODEL_CASE(KERNEL, 1, true, grid, __VA_ARGS__) } break;  \
      case 4: { BFGS_MODEL_CASE(KERNEL, 2, false, grid, __VA_ARGS__) } break; \
      case 5: { BFGS_MODEL_CASE(KERNEL, 2, true, grid, __VA_ARGS__) } break;  \
      case 8: { BFGS_MODEL_CASE(KERNEL, 4, false, grid, __VA_ARGS__) } break; \
      case 9: { BFGS_MODEL_CASE(KERNEL, 4, true, grid, __VA_ARGS__) } break;  \
      case 16: { BFGS_MODEL_CASE(KERNEL, 8, false, grid, __VA_ARGS__) } break; \
      case 17: { BFGS_MODEL_CASE(KERNEL, 8, true, grid, __VA_ARGS__) } break;  \
      default: break;                                                    \
    }                                                                    \
  } while (0)

void launch_iteration(nlsg_bfgs *e, bool timed) {
  const unsigned wave_grid = static_cast<unsigned>((e->p.batch + 3) / 4);
  const unsigned row_grid = static_cast<unsigned>(e->p.batch * e->bpp);
  if (e->p.model == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p};
    launch_module_kernel(e->rtc.search, wave_grid, 256,
                         e->p.seq ? static_cast<unsigned>(bfgs_fd_seq_lds_bytes(e->chunks)) : 0u, e->stream, args);
  } else {
    BFGS_DISPATCH_MODEL(bfgs_search_kernel, wave_grid, e->p);
  }
  if (timed) hipEventRecord(e->ev2, e->stream);
  if (e->symmetric) {
    const dim3 blocks(static_cast<unsigned>(e->p.batch * e->p.nstored)), probs(static_cast<unsigned>(e->p.batch));
    hipLaunchKernelGGL(bfgs_sym_hy_kernel, blocks, dim3(256), 0, e->stream, e->p);
    hipLaunchKernelGGL(bfgs_sym_reduce_kernel<false>, probs, dim3(256), 0, e->stream, e->p);
    hipLaunchKernelGGL(bfgs_sym_update_kernel, blocks, dim3(256), 0, e->stream, e->p);
    hipLaunchKernelGGL(bfgs_sym_reduce_kernel<true>, probs, dim3(256), 0, e->stream, e->p);
  } else if (e->p.seq) {  // reference order: a lane per row (nlsg_bfgs_kernels.h "at streaming speed")
    const uint64_t n = e->p.n;
    const uint32_t bpp = bfgs_seq_blocks_per_problem(n);
    const dim3 grid(static_cast<unsigned>(e->p.batch * bpp)), probs(static_cast<unsigned>(e->p.batch));
    const unsigned lds1 = static_cast<unsigned>(bfgs_seq_h_lds_bytes(n, kBfgsSeqColsHy));
    const unsigned lds3 = static_cast<unsigned>(bfgs_seq_h_lds_bytes(n, kBfgsSeqColsUpdate));
    if (e->vec) {
      hipLaunchKernelGGL(bfgs_hy_seq_kernel<true>, grid, dim3(256), lds1, e->stream, e->p, bpp);
      hipLaunchKernelGGL(bfgs_denom_seq_kernel, probs, dim3(64), static_cast<unsigned>(n * sizeof(double)), e->stream, e->p);
      hipLaunchKernelGGL(bfgs_update_seq_kernel<true>, grid, dim3(256), lds3, e->stream, e->p, bpp);
    } else {
      hipLaunchKernelGGL(bfgs_hy_seq_kernel<false>, grid, dim3(256), lds1, e->stream, e->p, bpp);
      hipLaunchKernelGGL(bfgs_denom_seq_kernel, probs, dim3(64), static_cast<unsigned>(n * sizeof(double)), e->stream, e->p);
      hipLaunchKernelGGL(bfgs_update_seq_kernel<false>, grid, dim3(256), lds3, e->stream, e->p, bpp);
    }
  } else {
    BFGS_DISPATCH(bfgs_hy_kernel, row_grid, e->p, e->bpp);
    BFGS_DISPATCH(bfgs_update_kernel, row_grid, e->p, e->bpp);
  }
  if (timed) hipEventRecord(e->ev3, e->stream);
}

}  // namespace

extern "C" {

static int bfgs_create(const nlsg_bfgs_config *cfg, const double *diag_host, const double *lin_host,
                       const nlsg_custom_objective *custom, nlsg_bfgs **out);

int nlsg_bfgs_create(const nlsg_bfgs_config *cfg, const double *diag_host, const double *lin_host,
                     nlsg_bfgs **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_bfgs_create_custom");
  PhaseClock clk;
  const int rc = bfgs_create(cfg, diag_host, lin_host, nullptr, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

int nlsg_bfgs_create_custom(const nlsg_bfgs_config *cfg, const nlsg_custom_objective *obj,
                            nlsg_bfgs **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  PhaseClock clk;
  const int rc = bfgs_create(cfg, nullptr, nullptr, obj, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

static int bfgs_create(const nlsg_bfgs_config *cfg, const double *diag_host, const double *lin_host,
                       const nlsg_custom_objective *custom, nlsg_bfgs **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_bfgs_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_bfgs_config size mismatch (%u vs %zu)",
                cfg->struct_size, sizeof(nlsg_bfgs_config));
  const bool quad = cfg->objective == NLSG_OBJ_QUAD_DIAG_RANK1;
  const bool fd = cfg->objective == NLSG_OBJ_ROSENBROCK || cfg->objective == NLSG_OBJ_SPHERE ||
                  cfg->objective == NLSG_OBJ_STYBLINSKI_TANG || cfg->objective == NLSG_OBJ_RASTRIGIN ||
                  custom != nullptr;
  if (!quad && !fd) return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (quad && (!diag_host || !lin_host))
    return fail(NLSG_ERR_INVALID_ARG, "the quadratic needs its d and b vectors");
  if (cfg->dim < 1 || cfg->batch < 1) return fail(NLSG_ERR_INVALID_ARG, "dim and batch must be >= 1");
  if (cfg->dim > 1024)
    return fail(NLSG_ERR_UNSUPPORTED, "dim %llu > 1024 is not covered by the device path",
                (unsigned long long)cfg->dim);
  const bool seq = (cfg->flags & NLSG_BFGS_REFERENCE_ORDER) != 0;
  if (cfg->flags & ~(NLSG_BFGS_SYMMETRIC | NLSG_BFGS_REFERENCE_ORDER))
    return fail(NLSG_ERR_INVALID_ARG, "unknown flags 0x%x", cfg->flags);
  if (seq && (cfg->flags & NLSG_BFGS_SYMMETRIC))
    return fail(NLSG_ERR_INVALID_ARG,
                "NLSG_BFGS_REFERENCE_ORDER reproduces the reference's literal arithmetic; the symmetric "
                "restatement is a different one");
  if (seq && custom && custom->chain == NLSG_CUSTOM_VECTOR)
    return fail(NLSG_ERR_UNSUPPORTED,
                "NLSG_BFGS_REFERENCE_ORDER needs an objective given by its terms (x.sum() of a "
                "whole-vector body adds in the lane-tree order)");
  if (seq && cfg->objective == NLSG_OBJ_RASTRIGIN)
    return fail(NLSG_ERR_UNSUPPORTED,
                "NLSG_BFGS_REFERENCE_ORDER: Rastrigin's cosine is the device's deterministic one, not "
                "libm's — there is no reference arithmetic to reproduce");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));
  nlsg_bfgs *e = new (std::nothrow) nlsg_bfgs();
  if (!e) return fail(NLSG_ERR_OOM, "host allocation failed");
  e->cfg = *cfg;
  const uint64_t n = cfg->dim, B = cfg->batch;
  e->chunks = n <= 128 ? 1 : n <= 256 ? 2 : n <= 512 ? 4 : 8;
  e->vec = n % 2 == 0;
  e->bpp = static_cast<uint32_t>((n + 4 * kBfgsRowsPerWave - 1) / (4 * kBfgsRowsPerWave));
  e->symmetric = (cfg->flags & NLSG_BFGS_SYMMETRIC) != 0;
  const uint32_t nb = static_cast<uint32_t>((n + kBfgsSymB - 1) / kBfgsSymB), nstored = nb * (nb + 1) / 2;
  if (B * e->bpp > 0x7fffffffull || B * nstored > 0x7fffffffull) {
    delete e;
    return fail(NLSG_ERR_UNSUPPORTED, "batch * dim too large for one launch grid");
  }
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
  BfgsParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  p.ldh = seq && (n * sizeof(double)) % 4096 == 0 ? n + 16 : n;  // (BfgsParams::ldh)
  auto alloc = [&](void **ptr, size_t bytes) { return pool_malloc(ptr, bytes ? bytes : 8); };
  hipError_t he = hipSuccess;
  const size_t vec_bytes = B * n * sizeof(double);
  if (he == hipSuccess && !e->symmetric)
    he = alloc(reinterpret_cast<void **>(&p.H), B * n * p.ldh * sizeof(double));
  if (he == hipSuccess && e->symmetric)
    he = alloc(reinterpret_cast<void **>(&p.Hs), B * nstored * kBfgsSymB * kBfgsSymB * sizeof(double));
  if (he == hipSuccess && e->symmetric)
    he = alloc(reinterpret_cast<void **>(&p.part), B * nb * nb * kBfgsSymB * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.x), vec_bytes);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.g), vec_bytes);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.dir), vec_bytes);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.s), vec_bytes);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.y), vec_bytes);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.t), vec_bytes);
  if (he == hipSuccess) he = hipMemset(p.dir, 0, vec_bytes);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.prob), B * sizeof(BfgsProblem));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->qd_dev), n * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->qb_dev), n * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->count_dev), 8);
  if (he == hipSuccess)
    he = quad ? hipMemcpy(e->qd_dev, diag_host, n * sizeof(double), hipMemcpyHostToDevice)
              : hipMemset(e->qd_dev, 0, n * sizeof(double));
  if (he == hipSuccess)
    he = quad ? hipMemcpy(e->qb_dev, lin_host, n * sizeof(double), hipMemcpyHostToDevice)
              : hipMemset(e->qb_dev, 0, n * sizeof(double));
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he == hipSuccess) he = hipEventCreate(&e->ev2);
  if (he == hipSuccess) he = hipEventCreate(&e->ev3);
  if (he != hipSuccess) {
    nlsg_bfgs_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device allocation failed: %s", hipGetErrorString(he));
  }
  p.qd = e->qd_dev;
  p.qb = e->qb_dev;
  p.zero = e->zero_dev;
  p.batch = B;
  p.n = n;
  p.max_iter = cfg->max_iter;
  p.grad_eps = cfg->grad_eps;
  p.alpha = cfg->alpha;
  p.qc = cfg->quad_c;
  p.model = quad ? kBfgsQuad : cfg->objective;
  p.seq = seq ? 1 : 0;
  p.nb = nb;
  p.nstored = nstored;
  if (custom) {
    const int rc2 = rtc_build_bfgs(custom, e->chunks, e->vec, &e->rtc);
    if (rc2) {
      nlsg_bfgs_destroy(e);
      return rc2;
    }
  }
  *out = e;
  return NLSG_OK;
}

int nlsg_bfgs_destroy(nlsg_bfgs *e) {
  if (!e) return NLSG_OK;
  PhaseClock clk;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  pool_free(e->p.H);
  pool_free(e->p.Hs);
  pool_free(e->p.part);
  pool_free(e->p.x);
  pool_free(e->p.g);
  pool_free(e->p.dir);
  pool_free(e->p.s);
  pool_free(e->p.y);
  pool_free(e->p.t);
  pool_free(e->p.prob);
  rtc_release(&e->rtc);
  pool_free(e->qd_dev);
  pool_free(e->qb_dev);
  pool_free(e->zero_dev);
  pool_free(e->count_dev);
  for (hipEvent_t ev : {e->ev0, e->ev1, e->ev2, e->ev3})
    if (ev) hipEventDestroy(ev);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  call_timing().destroy_ms = clk.lap();
  return NLSG_OK;
}

int nlsg_bfgs_init(nlsg_bfgs *e, const double *x0_host) {
  if (!e || !x0_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipMemcpyAsync(e->p.x, x0_host, e->p.batch * e->p.n * sizeof(double),
                          hipMemcpyHostToDevice, e->stream));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  if (e->p.model == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p};
    launch_module_kernel(e->rtc.init, static_cast<unsigned>((e->p.batch + 3) / 4), 256,
                         e->p.seq ? static_cast<unsigned>(bfgs_fd_seq_lds_bytes(e->chunks)) : 0u, e->stream, args);
  } else {
    BFGS_DISPATCH_MODEL(bfgs_init_kernel, static_cast<unsigned>((e->p.batch + 3) / 4), e->p);
  }
  NLSG_HIP(launches_status());
  e->initialised = true;
  return NLSG_OK;
}

int nlsg_bfgs_step(nlsg_bfgs *e, uint64_t iters) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_bfgs_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  for (uint64_t k = 0; k < iters; k++) launch_iteration(e, false);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_bfgs_unfinished(nlsg_bfgs *e, uint64_t *count) {
  if (!e || !count) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_bfgs_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipMemsetAsync(e->count_dev, 0, 8, e->stream));
  hipLaunchKernelGGL(bfgs_count_unfinished_kernel,
                     dim3(static_cast<unsigned>((e->p.batch + 255) / 256)), dim3(256), 0, e->stream,
                     e->p, e->count_dev);
  unsigned long long c = 0;
  NLSG_HIP(hipMemcpyAsync(&c, e->count_dev, 8, hipMemcpyDeviceToHost, e->stream));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  *count = c;
  return NLSG_OK;
}

int nlsg_bfgs_identity_count(nlsg_bfgs *e, uint64_t *count) {
  if (!e || !count) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_bfgs_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipMemsetAsync(e->count_dev, 0, 8, e->stream));
  hipLaunchKernelGGL(bfgs_count_identity_kernel,
                     dim3(static_cast<unsigned>((e->p.batch + 255) / 256)), dim3(256), 0, e->stream,
                     e->p, e->count_dev);
  unsigned long long c = 0;
  NLSG_HIP(hipMemcpyAsync(&c, e->count_dev, 8, hipMemcpyDeviceToHost, e->stream));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  *count = c;
  return NLSG_OK;
}

int nlsg_bfgs_download(nlsg_bfgs *e, double *x_host, nlsg_status *status_host) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_bfgs_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  const uint64_t B = e->p.batch, n = e->p.n;
  if (x_host) NLSG_HIP(hipMemcpy(x_host, e->p.x, B * n * sizeof(double), hipMemcpyDeviceToHost));
  if (status_host) {
    std::vector<BfgsProblem> pr(B);
    NLSG_HIP(hipMemcpy(pr.data(), e->p.prob, B * sizeof(BfgsProblem), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < B; i++) {
      nlsg_status &st = status_host[i];
      st.f_value = pr[i].fval;
      st.iteration = pr[i].iter;
      st.function_calls_used = pr[i].fcalls;
      st.gradient_evals_used = pr[i].gcalls;
      st.hessian_evals_used = 0;
      st.best_index = i;
      st.val_no_change = 0;
      st.std_err = pr[i].cur_norm;  // gradient norm at the last iterate
      st.done = pr[i].done;
      st.reserved = 0;
    }
  }
  return NLSG_OK;
}

int nlsg_bfgs_download_state(nlsg_bfgs *e, double *g_host, double *h_host) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_bfgs_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  const uint64_t B = e->p.batch, n = e->p.n;
  if (g_host) NLSG_HIP(hipMemcpy(g_host, e->p.g, B * n * sizeof(double), hipMemcpyDeviceToHost));
  if (h_host && !e->symmetric)
    NLSG_HIP(hipMemcpy2D(h_host, n * sizeof(double), e->p.H, e->p.ldh * sizeof(double), n * sizeof(double),
                         B * n, hipMemcpyDeviceToHost));
  if (h_host && e->symmetric) {  // the full matrix from its upper blocks
    const uint64_t tile = kBfgsSymB * kBfgsSymB;
    std::vector<double> blk(e->p.nstored * tile);
    for (uint64_t b = 0; b < B; b++) {
      NLSG_HIP(hipMemcpy(blk.data(), e->p.Hs + b * e->p.nstored * tile, blk.size() * sizeof(double),
                         hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < n; i++)
        for (uint64_t j = 0; j < n; j++) {
          const uint64_t r = i <= j ? i : j, c = i <= j ? j : i;  // (r, c) in the upper triangle
          const uint32_t I = static_cast<uint32_t>(r / kBfgsSymB), J = static_cast<uint32_t>(c / kBfgsSymB);
          h_host[(b * n + i) * n + j] = blk[bfgs_sym_block(I, J, e->p.nb) * tile +
                                            (r % kBfgsSymB) * kBfgsSymB + c % kBfgsSymB];
        }
    }
  }
  return NLSG_OK;
}

int nlsg_bfgs_minimize(nlsg_bfgs *e, double *x_inout_host, nlsg_status *status_host) {
  if (!e || !x_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  PhaseClock clk;
  int rc = nlsg_bfgs_init(e, x_inout_host);
  if (rc) return rc;
  call_timing().init_ms = clk.lap();
  // every problem stops after at most max_iter + 1 turns (the last one only fires the stop test)
  uint64_t left = e->cfg.max_iter + 1;
  for (;;) {
    const uint64_t chunk = std::min<uint64_t>(left ? left : 1, 8);
    rc = nlsg_bfgs_step(e, chunk);
    if (rc) return rc;
    left = left > chunk ? left - chunk : 0;
    uint64_t open = 0;
    rc = nlsg_bfgs_unfinished(e, &open);
    if (rc) return rc;
    if (open == 0) break;
  }
  call_timing().iterate_ms = clk.lap();
  rc = nlsg_bfgs_download(e, x_inout_host, status_host);
  call_timing().readback_ms = clk.lap();
  return rc;
}

int nlsg_bfgs_time_steps(nlsg_bfgs *e, uint64_t iters, float *ms_total, float *ms_hessian) {
  if (!e || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_bfgs_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  float hess = 0.f;
  NLSG_HIP(hipEventRecord(e->ev0, e->stream));
  for (uint64_t k = 0; k < iters; k++) {
    // one timed iteration per sync: the two inner events are reused
    launch_iteration(e, true);
    NLSG_HIP(hipEventSynchronize(e->ev3));
    float ms = 0.f;
    NLSG_HIP(hipEventElapsedTime(&ms, e->ev2, e->ev3));
    hess += ms;
  }
  NLSG_HIP(hipEventRecord(e->ev1, e->stream));
  NLSG_HIP(hipEventSynchronize(e->ev1));
  NLSG_HIP(launches_status());
  NLSG_HIP(hipEventElapsedTime(ms_total, e->ev0, e->ev1));
  if (ms_hessian) *ms_hessian = hess;
  return NLSG_OK;
}

}  // extern "C"
