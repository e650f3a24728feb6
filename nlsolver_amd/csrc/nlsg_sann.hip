// nlsolver_amd/csrc/nlsg_sann.hip — host side of the batched simulated-annealing engine + C-ABI.
#include <cstdlib>
#include <new>
#include <vector>

#include "nlsg_rtc.h"
#include "nlsg_sann_kernels.h"

using namespace nlsg;

struct nlsg_sann {
  nlsg_sann_config cfg;
  SannParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *zero_dev = nullptr;
  int chunks = 0;
  int group = 0;  // lanes per chain when several chains share a wave (dim <= 64), else 0
  SannRtcKernels rtc;  // objective == NLSG_OBJ_CUSTOM: the kernel hiprtc built for it
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

#define SANN_FOR_CHUNKS(OBJ, chunks, CALL) \
  switch (chunks) {                        \
    case 1: CALL(OBJ, 1); break;           \
    case 2: CALL(OBJ, 2); break;           \
    case 4: CALL(OBJ, 4); break;           \
    case 8: CALL(OBJ, 8); break;           \
    default: break;                        \
  }
#define SANN_FOR_OBJ(obj, chunks, CALL)                                                  \
  switch (obj) {                                                                         \
    case NLSG_OBJ_ROSENBROCK: SANN_FOR_CHUNKS(NLSG_OBJ_ROSENBROCK, chunks, CALL); break; \
    case NLSG_OBJ_SPHERE: SANN_FOR_CHUNKS(NLSG_OBJ_SPHERE, chunks, CALL); break;         \
    case NLSG_OBJ_STYBLINSKI_TANG:                                                       \
      SANN_FOR_CHUNKS(NLSG_OBJ_STYBLINSKI_TANG, chunks, CALL);                           \
      break;                                                                             \
    case NLSG_OBJ_RASTRIGIN: SANN_FOR_CHUNKS(NLSG_OBJ_RASTRIGIN, chunks, CALL); break;   \
    default: break;                                                                      \
  }

template <int OBJ>
void launch_groups(nlsg_sann *e, dim3 grid, uint64_t iter_begin, uint64_t iter_end) {
  const dim3 block(256);
  switch (e->group) {
    case 4:
      hipLaunchKernelGGL((sann_anneal_groups_kernel<OBJ, 4>), grid, block, 0, e->stream, e->p,
                         iter_begin, iter_end);
      break;
    case 8:
      hipLaunchKernelGGL((sann_anneal_groups_kernel<OBJ, 8>), grid, block, 0, e->stream, e->p,
                         iter_begin, iter_end);
      break;
    case 16:
      hipLaunchKernelGGL((sann_anneal_groups_kernel<OBJ, 16>), grid, block, 0, e->stream, e->p,
                         iter_begin, iter_end);
      break;
    default:
      hipLaunchKernelGGL((sann_anneal_groups_kernel<OBJ, 32>), grid, block, 0, e->stream, e->p,
                         iter_begin, iter_end);
      break;
  }
}

void launch_anneal(nlsg_sann *e, uint64_t iter_begin, uint64_t iter_end) {
  // waves: one per chain, or one per 64 / group chains
  const uint64_t per_wave = e->group ? 64 / e->group : 1;
  const uint64_t waves = (e->p.batch + per_wave - 1) / per_wave;
  const dim3 grid(static_cast<unsigned>((waves + 3) / 4)), block(256);
  const bool vec = e->p.D % 2 == 0;
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &iter_begin, &iter_end};
    launch_module_kernel(e->rtc.anneal, grid.x, 256, 0, e->stream, args);
    return;
  }
  if (e->p.D > 1024) {  // chains streamed in segments
#define CALL(OBJ)                                                                                  \
  if (vec)                                                                                         \
    hipLaunchKernelGGL((sann_anneal_long_kernel<OBJ, true>), grid, block, 0, e->stream, e->p,      \
                       iter_begin, iter_end);                                                      \
  else                                                                                             \
    hipLaunchKernelGGL((sann_anneal_long_kernel<OBJ, false>), grid, block, 0, e->stream, e->p,     \
                       iter_begin, iter_end)
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK: CALL(NLSG_OBJ_ROSENBROCK); break;
      case NLSG_OBJ_SPHERE: CALL(NLSG_OBJ_SPHERE); break;
      case NLSG_OBJ_STYBLINSKI_TANG: CALL(NLSG_OBJ_STYBLINSKI_TANG); break;
      default: CALL(NLSG_OBJ_RASTRIGIN); break;
    }
#undef CALL
    return;
  }
  if (e->group) {
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK: launch_groups<NLSG_OBJ_ROSENBROCK>(e, grid, iter_begin, iter_end); break;
      case NLSG_OBJ_SPHERE: launch_groups<NLSG_OBJ_SPHERE>(e, grid, iter_begin, iter_end); break;
      case NLSG_OBJ_STYBLINSKI_TANG:
        launch_groups<NLSG_OBJ_STYBLINSKI_TANG>(e, grid, iter_begin, iter_end);
        break;
      default: launch_groups<NLSG_OBJ_RASTRIGIN>(e, grid, iter_begin, iter_end); break;
    }
    return;
  }
#define CALL(OBJ, C)                                                                            \
  if (vec)                                                                                      \
    hipLaunchKernelGGL((sann_anneal_kernel<OBJ, C, true>), grid, block, 0, e->stream, e->p,     \
                       iter_begin, iter_end);                                                   \
  else                                                                                          \
    hipLaunchKernelGGL((sann_anneal_kernel<OBJ, C, false>), grid, block, 0, e->stream, e->p,    \
                       iter_begin, iter_end)
  SANN_FOR_OBJ(e->cfg.objective, e->chunks, CALL)
#undef CALL
}

// The whole schedule; long schedules are cut into launches of a bounded number of trial points so
// that no single kernel runs for seconds (the chain's state waits in HBM between launches).
void launch_solve(nlsg_sann *e) {
  const uint64_t max_iter = e->cfg.max_iter;
  const uint64_t per_iter = e->p.inner ? e->p.inner : 1;
  uint64_t span = (1u << 16) / per_iter;
  if (span == 0) span = 1;
  uint64_t it = 0;
  do {  // max_iter == 0 still scores the start (:2781)
    const uint64_t end = max_iter - it < span ? max_iter : it + span;
    launch_anneal(e, it, end);
    it = end;
  } while (it < max_iter);
}

}  // namespace

static int sann_create(const nlsg_sann_config *cfg, const nlsg_custom_objective *custom,
                       nlsg_sann **out);

extern "C" {

int nlsg_sann_create(const nlsg_sann_config *cfg, nlsg_sann **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_sann_create_custom");
  return sann_create(cfg, nullptr, out);
}

int nlsg_sann_create_custom(const nlsg_sann_config *cfg, const nlsg_custom_objective *obj,
                            nlsg_sann **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  return sann_create(cfg, obj, out);
}

}  // extern "C"

static int sann_create(const nlsg_sann_config *cfg, const nlsg_custom_objective *custom,
                       nlsg_sann **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_sann_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_sann_config size mismatch (%u vs %zu)",
                cfg->struct_size, sizeof(nlsg_sann_config));
  if (!custom && (cfg->objective < 0 || cfg->objective > NLSG_OBJ_RASTRIGIN))
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (cfg->dim < 1 || cfg->batch < 1) return fail(NLSG_ERR_INVALID_ARG, "dim and batch must be >= 1");
  if (cfg->dim > 1024 && custom && custom->chain == NLSG_CUSTOM_VECTOR)
    return fail(NLSG_ERR_UNSUPPORTED,
                "dim %llu > 1024: a whole-vector objective needs the point in the wave's registers",
                (unsigned long long)cfg->dim);
  if (cfg->dim > 0xffffffffull) return fail(NLSG_ERR_UNSUPPORTED, "dim beyond 2^32");
  if (cfg->batch > 0x7fffffffull) return fail(NLSG_ERR_UNSUPPORTED, "batch too large");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));
  nlsg_sann *e = new (std::nothrow) nlsg_sann();
  if (!e) return fail(NLSG_ERR_OOM, "host allocation failed");
  e->cfg = *cfg;
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
  const uint64_t B = cfg->batch, D = cfg->dim;
  e->chunks = D <= 128 ? 1 : D <= 256 ? 2 : D <= 512 ? 4 : 8;
  e->group = D <= 8 ? 4 : D <= 16 ? 8 : D <= 32 ? 16 : D <= 64 ? 32 : 0;
  SannParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  hipError_t he = hipSuccess;
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.x), B * D * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.p), B * D * 8);
  if (he == hipSuccess && D > 1024) he = pool_malloc(reinterpret_cast<void **>(&p.trial), B * D * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.prob), B * sizeof(SannProblem));
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he != hipSuccess) {
    nlsg_sann_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device setup failed: %s", hipGetErrorString(he));
  }
  if (custom) {
    const int rc2 = rtc_build_sann(custom, D > 1024 ? 0 : e->chunks, D % 2 == 0, e->group, &e->rtc);
    if (rc2) {
      nlsg_sann_destroy(e);
      return rc2;
    }
  }
  p.zero = e->zero_dev;
  p.batch = B;
  p.D = D;
  p.seed = cfg->seed;
  p.chain_lo = cfg->chain_lo;
  p.inner = cfg->temperature_iter ? cfg->temperature_iter - 1 : 0;  // for (j = 1; j < t_iter; j++)
  p.temp_max = cfg->temperature_max;
  p.fmul = cfg->minimize ? 1.0 : -1.0;
  *out = e;
  return NLSG_OK;
}

extern "C" {

int nlsg_sann_destroy(nlsg_sann *e) {
  if (!e) return NLSG_OK;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  rtc_release(&e->rtc);
  pool_free(e->p.x);
  pool_free(e->p.p);
  pool_free(e->p.trial);
  pool_free(e->p.prob);
  pool_free(e->zero_dev);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  return NLSG_OK;
}

int nlsg_sann_minimize(nlsg_sann *e, double *x_inout_host, nlsg_status *status_host) {
  if (!e || !x_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  const uint64_t B = e->p.batch, D = e->p.D;
  NLSG_HIP(hipMemcpy(e->p.x, x_inout_host, B * D * 8, hipMemcpyHostToDevice));
  launch_solve(e);
  NLSG_HIP(launches_status());
  NLSG_HIP(hipStreamSynchronize(e->stream));
  NLSG_HIP(hipMemcpy(x_inout_host, e->p.x, B * D * 8, hipMemcpyDeviceToHost));
  if (status_host) {
    std::vector<SannProblem> pr(B);
    NLSG_HIP(hipMemcpy(pr.data(), e->p.prob, B * sizeof(SannProblem), hipMemcpyDeviceToHost));
    for (uint64_t b = 0; b < B; b++) {
      nlsg_status &st = status_host[b];
      st.f_value = pr[b].best;
      st.iteration = pr[b].iter;
      st.function_calls_used = pr[b].fcalls;
      st.gradient_evals_used = 0;
      st.hessian_evals_used = 0;
      st.best_index = b;
      st.val_no_change = 0;
      st.std_err = 0.0;
      st.done = 1;
      st.reserved = 0;
    }
  }
  return NLSG_OK;
}

int nlsg_sann_time_solve(nlsg_sann *e, const double *x0_host, uint32_t repeats, float *ms_total) {
  if (!e || !x0_host || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  float total = 0.f;
  for (uint32_t r = 0; r < repeats; r++) {
    NLSG_HIP(hipMemcpy(e->p.x, x0_host, e->p.batch * e->p.D * 8, hipMemcpyHostToDevice));
    NLSG_HIP(hipEventRecord(e->ev0, e->stream));
    launch_solve(e);
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
