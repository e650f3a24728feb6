// nlsolver_amd/csrc/nlsg_nmpso.hip — host side of the batched Nelder-Mead / PSO hybrid + C-ABI.
#include <cstdlib>
#include <new>
#include <vector>

#include "nlsg_nmpso_kernels.h"
#include "nlsg_rtc.h"

using namespace nlsg;

struct nlsg_nmpso {
  nlsg_nmpso_config cfg;
  HybParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *upper_dev = nullptr, *lower_dev = nullptr, *zero_dev = nullptr;
  HybRtcKernels rtc;  // objective == NLSG_OBJ_CUSTOM: the kernel hiprtc built for it
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

int wide_chunks(uint64_t n) { return n <= kHybMaxN ? 0 : n <= 256 ? 2 : n <= 512 ? 4 : 8; }

template <int OBJ>
void launch_wide(nlsg_nmpso *e, dim3 grid) {
  const dim3 block(hyb_wide_threads(wide_chunks(e->p.n)));
  const unsigned lds = static_cast<unsigned>(hyb_view_bytes(e->p.n));
  switch (wide_chunks(e->p.n)) {
    case 2: hipLaunchKernelGGL((nmpso_solve_wide_kernel<OBJ, 2>), grid, block, lds, e->stream, e->p); break;
    case 4: hipLaunchKernelGGL((nmpso_solve_wide_kernel<OBJ, 4>), grid, block, lds, e->stream, e->p); break;
    default: hipLaunchKernelGGL((nmpso_solve_wide_kernel<OBJ, 8>), grid, block, lds, e->stream, e->p); break;
  }
}
template <int OBJ>
hipError_t allow_wide_lds(uint64_t n) {  // past 64 KiB the dynamic allocation has to be announced
  // the attribute belongs to the instantiation, shared by every engine of the chunk class: the class
  // maximum, so that an engine created later with a smaller n cannot lower an earlier one's limit
  const int bytes = static_cast<int>(hyb_view_bytes(128ull * wide_chunks(n)));
  const void *fn = wide_chunks(n) == 2   ? reinterpret_cast<const void *>(nmpso_solve_wide_kernel<OBJ, 2>)
                   : wide_chunks(n) == 4 ? reinterpret_cast<const void *>(nmpso_solve_wide_kernel<OBJ, 4>)
                                         : reinterpret_cast<const void *>(nmpso_solve_wide_kernel<OBJ, 8>);
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

void launch(nlsg_nmpso *e) {
  const bool wide = e->p.n > kHybMaxN;
  const dim3 grid(static_cast<unsigned>(e->p.batch)),
      block(wide ? hyb_wide_threads(wide_chunks(e->p.n)) : hyb_block_threads(e->p.n));
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p};
    launch_module_kernel(e->rtc.solve, grid.x, block.x,
                         wide ? static_cast<unsigned>(hyb_view_bytes(e->p.n)) : 0, e->stream, args);
    return;
  }
  if (wide) {
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK: launch_wide<NLSG_OBJ_ROSENBROCK>(e, grid); break;
      case NLSG_OBJ_SPHERE: launch_wide<NLSG_OBJ_SPHERE>(e, grid); break;
      case NLSG_OBJ_STYBLINSKI_TANG: launch_wide<NLSG_OBJ_STYBLINSKI_TANG>(e, grid); break;
      default: launch_wide<NLSG_OBJ_RASTRIGIN>(e, grid); break;
    }
    return;
  }
  switch (e->cfg.objective) {
    case NLSG_OBJ_ROSENBROCK:
      hipLaunchKernelGGL(nmpso_solve_kernel<NLSG_OBJ_ROSENBROCK>, grid, block, 0, e->stream, e->p);
      break;
    case NLSG_OBJ_SPHERE:
      hipLaunchKernelGGL(nmpso_solve_kernel<NLSG_OBJ_SPHERE>, grid, block, 0, e->stream, e->p);
      break;
    case NLSG_OBJ_STYBLINSKI_TANG:
      hipLaunchKernelGGL(nmpso_solve_kernel<NLSG_OBJ_STYBLINSKI_TANG>, grid, block, 0, e->stream, e->p);
      break;
    default:
      hipLaunchKernelGGL(nmpso_solve_kernel<NLSG_OBJ_RASTRIGIN>, grid, block, 0, e->stream, e->p);
      break;
  }
}

int upload_bounds(nlsg_nmpso *e, const double *lower_host, const double *upper_host) {
  if (!e->p.bounded) return NLSG_OK;
  if (!lower_host || !upper_host)
    return fail(NLSG_ERR_INVALID_ARG, "a bounded engine needs lower and upper");
  NLSG_HIP(hipMemcpy(e->lower_dev, lower_host, e->p.n * 8, hipMemcpyHostToDevice));
  NLSG_HIP(hipMemcpy(e->upper_dev, upper_host, e->p.n * 8, hipMemcpyHostToDevice));
  return NLSG_OK;
}

}  // namespace

static int hyb_create(const nlsg_nmpso_config *cfg, const nlsg_custom_objective *custom,
                      nlsg_nmpso **out);

extern "C" {

int nlsg_nmpso_create(const nlsg_nmpso_config *cfg, nlsg_nmpso **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_nmpso_create_custom");
  return hyb_create(cfg, nullptr, out);
}

int nlsg_nmpso_create_custom(const nlsg_nmpso_config *cfg, const nlsg_custom_objective *obj,
                             nlsg_nmpso **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  return hyb_create(cfg, obj, out);
}

}  // extern "C"

static int hyb_create(const nlsg_nmpso_config *cfg, const nlsg_custom_objective *custom,
                      nlsg_nmpso **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_nmpso_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_nmpso_config size mismatch (%u vs %zu)",
                cfg->struct_size, sizeof(nlsg_nmpso_config));
  if (!custom && (cfg->objective < 0 || cfg->objective > NLSG_OBJ_RASTRIGIN))
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (cfg->batch < 1) return fail(NLSG_ERR_INVALID_ARG, "batch must be >= 1");
  if (cfg->dim < 2)  // the reference refuses it too (3627-3637)
    return fail(NLSG_ERR_INVALID_ARG, "dim must be >= 2: the hybrid does not support one dimension");
  if (cfg->dim > kHybWideMaxN)
    return fail(NLSG_ERR_UNSUPPORTED, "dim %llu > %d (an instance's sort keys, orders and trial "
                "points live in one workgroup's shared memory)", (unsigned long long)cfg->dim,
                kHybWideMaxN);
  if (cfg->batch > 0x7fffffffull) return fail(NLSG_ERR_UNSUPPORTED, "batch too large");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));
  nlsg_nmpso *e = new (std::nothrow) nlsg_nmpso();
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
  HybParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  const uint64_t B = cfg->batch, n = cfg->dim, rows = B * (3 * n + 1);
  hipError_t he = hipSuccess;
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.x), B * n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.pos), rows * n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.vel), rows * n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.prob), B * sizeof(HybProblem));
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->upper_dev), n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->lower_dev), n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he == hipSuccess && n > kHybMaxN && !custom) {
    switch (cfg->objective) {
      case NLSG_OBJ_ROSENBROCK: he = allow_wide_lds<NLSG_OBJ_ROSENBROCK>(n); break;
      case NLSG_OBJ_SPHERE: he = allow_wide_lds<NLSG_OBJ_SPHERE>(n); break;
      case NLSG_OBJ_STYBLINSKI_TANG: he = allow_wide_lds<NLSG_OBJ_STYBLINSKI_TANG>(n); break;
      default: he = allow_wide_lds<NLSG_OBJ_RASTRIGIN>(n); break;
    }
  }
  if (he != hipSuccess) {
    nlsg_nmpso_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device setup failed: %s", hipGetErrorString(he));
  }
  if (custom) {
    const int rc2 = rtc_build_nmpso(custom, wide_chunks(n), &e->rtc);
    if (rc2) {
      nlsg_nmpso_destroy(e);
      return rc2;
    }
  }
  p.upper = e->upper_dev;
  p.lower = e->lower_dev;
  p.zero = e->zero_dev;
  p.batch = B;
  p.n = n;
  p.max_iter = cfg->max_iter;
  p.no_change_iter = cfg->no_change_best_iter;
  p.seed = cfg->seed;
  p.inst_lo = cfg->inst_lo;
  p.alpha = cfg->alpha;
  p.gamma = cfg->gamma;
  p.rho = cfg->rho;
  p.sigma = cfg->sigma;
  p.inertia = cfg->inertia;
  p.cog = cfg->cognitive;
  p.soc = cfg->social;
  p.eps = cfg->eps;
  p.fmul = cfg->minimize ? 1.0 : -1.0;
  p.bounded = cfg->bounded ? 1 : 0;
  *out = e;
  return NLSG_OK;
}

extern "C" {

int nlsg_nmpso_destroy(nlsg_nmpso *e) {
  if (!e) return NLSG_OK;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  rtc_release(&e->rtc);
  pool_free(e->p.x);
  pool_free(e->p.pos);
  pool_free(e->p.vel);
  pool_free(e->p.prob);
  pool_free(e->upper_dev);
  pool_free(e->lower_dev);
  pool_free(e->zero_dev);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  return NLSG_OK;
}

int nlsg_nmpso_minimize(nlsg_nmpso *e, double *x_inout_host, const double *lower_host,
                        const double *upper_host, nlsg_status *status_host) {
  if (!e || !x_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  int rc = upload_bounds(e, lower_host, upper_host);
  if (rc) return rc;
  const uint64_t B = e->p.batch, n = e->p.n;
  NLSG_HIP(hipMemcpy(e->p.x, x_inout_host, B * n * 8, hipMemcpyHostToDevice));
  launch(e);
  NLSG_HIP(launches_status());
  NLSG_HIP(hipStreamSynchronize(e->stream));
  NLSG_HIP(hipMemcpy(x_inout_host, e->p.x, B * n * 8, hipMemcpyDeviceToHost));
  if (status_host) {
    std::vector<HybProblem> pr(B);
    NLSG_HIP(hipMemcpy(pr.data(), e->p.prob, B * sizeof(HybProblem), hipMemcpyDeviceToHost));
    for (uint64_t b = 0; b < B; b++) {
      nlsg_status &st = status_host[b];
      st.f_value = pr[b].f;
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

int nlsg_nmpso_time_solve(nlsg_nmpso *e, const double *x0_host, uint32_t repeats, float *ms_total) {
  if (!e || !x0_host || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (e->p.bounded) return fail(NLSG_ERR_UNSUPPORTED, "timing aid of the unbounded overloads");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  float total = 0.f;
  for (uint32_t r = 0; r < repeats; r++) {
    NLSG_HIP(hipMemcpy(e->p.x, x0_host, e->p.batch * e->p.n * 8, hipMemcpyHostToDevice));
    NLSG_HIP(hipEventRecord(e->ev0, e->stream));
    launch(e);
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
