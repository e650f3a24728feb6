// nlsolver_amd/csrc/nlsg_nm.hip — host side of the batched Nelder-Mead engine + C-ABI.
#include <cstdlib>
#include <new>
#include <vector>

#include "nlsg_nm_kernels.h"
#include "nlsg_rtc.h"

using namespace nlsg;

struct nlsg_nm {
  NmRtcKernels rtc;  // objective == NLSG_OBJ_CUSTOM: the kernel hiprtc built for it
  nlsg_nm_config cfg;
  NmParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *upper_dev = nullptr, *lower_dev = nullptr;
  size_t lds = 0;
  bool driver = false;  // n <= 128: nm_solve_driver_kernel (NLSG_NM_DRIVER=0: the phase-per-barrier kernel)
  unsigned long long *phase_dev = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {
// The opt-in for more than 64 KiB of dynamic LDS belongs to the kernel instantiation <OBJ, CHUNKS>,
// which every engine of that chunk class shares: it is set to the class maximum (the largest n the
// instantiation serves), never to one engine's size — a later, smaller engine must not lower it.
template <int OBJ, int CHUNKS>
hipError_t prepare1() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(nm_solve_kernel<OBJ, CHUNKS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024);  // (reference order adds term buffers behind the image)
}
template <int OBJ>
hipError_t prepare(int chunks) {
  switch (chunks) {
    case 1: return prepare1<OBJ, 1>();
    case 2: return prepare1<OBJ, 2>();
    case 4: return prepare1<OBJ, 4>();
    default: return prepare1<OBJ, 8>();
  }
}
template <int OBJ>
hipError_t prepare_driver() {
  hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void *>(nm_solve_driver_kernel<OBJ>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(nm_lds_bytes(128)));
  if constexpr (OBJ != NLSG_OBJ_RASTRIGIN)  // (reference order: term buffers behind the image)
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void *>(nm_solve_driver_kernel<OBJ, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return he;
}
template <int OBJ>
void launch_obj(nlsg_nm *e, dim3 grid, dim3 block) {
  if (e->driver) {
    if constexpr (OBJ != NLSG_OBJ_RASTRIGIN) {
      if (e->p.seq) {
        hipLaunchKernelGGL((nm_solve_driver_kernel<OBJ, true>), grid, block, e->lds, e->stream, e->p);
        return;
      }
    }
    hipLaunchKernelGGL(nm_solve_driver_kernel<OBJ>, grid, block, e->lds, e->stream, e->p);
    return;
  }
  switch (nm_chunks(e->p.n)) {
    case 1: hipLaunchKernelGGL((nm_solve_kernel<OBJ, 1>), grid, block, e->lds, e->stream, e->p); break;
    case 2: hipLaunchKernelGGL((nm_solve_kernel<OBJ, 2>), grid, block, e->lds, e->stream, e->p); break;
    case 4: hipLaunchKernelGGL((nm_solve_kernel<OBJ, 4>), grid, block, e->lds, e->stream, e->p); break;
    default: hipLaunchKernelGGL((nm_solve_kernel<OBJ, 8>), grid, block, e->lds, e->stream, e->p); break;
  }
}

void launch(nlsg_nm *e) {
  const dim3 grid(static_cast<unsigned>(e->p.batch)), block(nm_block_threads(e->p.n));
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p};
    launch_module_kernel(e->rtc.solve, grid.x, block.x, static_cast<unsigned>(e->lds), e->stream, args);
    return;
  }
  switch (e->cfg.objective) {
    case NLSG_OBJ_ROSENBROCK: launch_obj<NLSG_OBJ_ROSENBROCK>(e, grid, block); break;
    case NLSG_OBJ_SPHERE: launch_obj<NLSG_OBJ_SPHERE>(e, grid, block); break;
    case NLSG_OBJ_STYBLINSKI_TANG: launch_obj<NLSG_OBJ_STYBLINSKI_TANG>(e, grid, block); break;
    default: launch_obj<NLSG_OBJ_RASTRIGIN>(e, grid, block); break;
  }
}

int upload_bounds(nlsg_nm *e, const double *upper_host, const double *lower_host) {
  if (!e->cfg.bounded) return NLSG_OK;
  if (!upper_host || !lower_host)
    return fail(NLSG_ERR_INVALID_ARG, "bounded solver needs upper and lower");
  NLSG_HIP(hipMemcpy(e->upper_dev, upper_host, e->p.n * 8, hipMemcpyHostToDevice));
  NLSG_HIP(hipMemcpy(e->lower_dev, lower_host, e->p.n * 8, hipMemcpyHostToDevice));
  return NLSG_OK;
}
}  // namespace

extern "C" {

static int nm_create(const nlsg_nm_config *cfg, const nlsg_custom_objective *custom, nlsg_nm **out);

int nlsg_nm_create(const nlsg_nm_config *cfg, nlsg_nm **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_nm_create_custom");
  return nm_create(cfg, nullptr, out);
}

int nlsg_nm_create_custom(const nlsg_nm_config *cfg, const nlsg_custom_objective *obj, nlsg_nm **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  return nm_create(cfg, obj, out);
}

static int nm_create(const nlsg_nm_config *cfg, const nlsg_custom_objective *custom, nlsg_nm **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_nm_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_nm_config size mismatch (%u vs %zu)", cfg->struct_size,
                sizeof(nlsg_nm_config));
  if (!custom && (cfg->objective < 0 || cfg->objective > NLSG_OBJ_RASTRIGIN))
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (cfg->dim < 1 || cfg->batch < 1) return fail(NLSG_ERR_INVALID_ARG, "dim and batch must be >= 1");
  if (cfg->dim > 1024)
    return fail(NLSG_ERR_UNSUPPORTED, "dim %llu > 1024 is not covered by the device path",
                (unsigned long long)cfg->dim);
  if (cfg->batch > 0x7fffffffull) return fail(NLSG_ERR_UNSUPPORTED, "batch too large");
  if (cfg->flags & ~NLSG_NM_REFERENCE_ORDER) return fail(NLSG_ERR_INVALID_ARG, "unknown flags 0x%x", cfg->flags);
  if ((cfg->flags & NLSG_NM_REFERENCE_ORDER) &&
      (cfg->objective == NLSG_OBJ_RASTRIGIN || (custom && custom->chain == NLSG_CUSTOM_VECTOR)))
    return fail(NLSG_ERR_UNSUPPORTED,
                "NLSG_NM_REFERENCE_ORDER needs an objective given by its terms whose arithmetic the device "
                "shares with the reference (not Rastrigin: its cosine is the device's own; not a whole-vector body)");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));
  nlsg_nm *e = new (std::nothrow) nlsg_nm();
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
  NmParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  const uint64_t B = cfg->batch, n = cfg->dim;
  const bool seq = (cfg->flags & NLSG_NM_REFERENCE_ORDER) != 0;
  e->lds = nm_launch_lds_bytes(n, nm_block_threads(n) / 64, seq);
  hipError_t he = hipSuccess;
  const int chunks = nm_chunks(n);
  if (he == hipSuccess && chunks > 1)  // the simplexes themselves: past what LDS holds
    he = pool_malloc(reinterpret_cast<void **>(&p.simplex), B * (n + 1) * n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.x), B * n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&p.prob), B * sizeof(NmProblem));
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->upper_dev), n * 8);
  if (he == hipSuccess) he = pool_malloc(reinterpret_cast<void **>(&e->lower_dev), n * 8);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  {
    const char *sw = std::getenv("NLSG_NM_DRIVER");
    e->driver = chunks == 1 && !(sw && sw[0] == '0');
  }
  if (he == hipSuccess) he = prepare_driver<NLSG_OBJ_ROSENBROCK>();
  if (he == hipSuccess) he = prepare_driver<NLSG_OBJ_SPHERE>();
  if (he == hipSuccess) he = prepare_driver<NLSG_OBJ_STYBLINSKI_TANG>();
  if (he == hipSuccess) he = prepare_driver<NLSG_OBJ_RASTRIGIN>();
  if (he == hipSuccess) he = prepare<NLSG_OBJ_ROSENBROCK>(chunks);
  if (he == hipSuccess) he = prepare<NLSG_OBJ_SPHERE>(chunks);
  if (he == hipSuccess) he = prepare<NLSG_OBJ_STYBLINSKI_TANG>(chunks);
  if (he == hipSuccess) he = prepare<NLSG_OBJ_RASTRIGIN>(chunks);
  if (he == hipSuccess && custom) {
    const int rc2 = rtc_build_nm(custom, e->driver ? 0 : chunks, seq, &e->rtc);
    if (rc2) {
      nlsg_nm_destroy(e);
      return rc2;
    }
    // the module API's counterpart of prepare<>(): allow the simplex-sized dynamic LDS
    he = hipFuncSetAttribute(reinterpret_cast<const void *>(e->rtc.solve),
                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(e->lds));
  }
  if (he != hipSuccess) {
    nlsg_nm_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device setup failed: %s", hipGetErrorString(he));
  }
  p.upper = e->upper_dev;
  p.lower = e->lower_dev;
  p.batch = B;
  p.n = n;
  p.max_iter = cfg->max_iter;
  p.no_change_tol = cfg->no_change_best_tol;
  p.restarts = cfg->restarts;
  p.step = cfg->step;
  p.alpha = cfg->alpha;
  p.gamma = cfg->gamma;
  p.rho = cfg->rho;
  p.sigma = cfg->sigma;
  p.eps = cfg->eps;
  p.fmul = cfg->minimize ? 1.0 : -1.0;
  p.bounded = cfg->bounded ? 1 : 0;
  p.seq = 0;
  if (seq) {
    const char *sw = std::getenv("NLSG_NM_SEQ_WAVES");
    const int w = sw ? std::atoi(sw) : 16;
    p.seq = w < 1 ? 1 : (w > 16 ? 16 : w);
  }
  *out = e;
  return NLSG_OK;
}

int nlsg_nm_destroy(nlsg_nm *e) {
  if (!e) return NLSG_OK;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  rtc_release(&e->rtc);
  pool_free(e->p.simplex);
  pool_free(e->p.x);
  pool_free(e->p.prob);
  pool_free(e->upper_dev);
  pool_free(e->lower_dev);
  pool_free(e->phase_dev);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  return NLSG_OK;
}

int nlsg_nm_minimize(nlsg_nm *e, double *x_inout_host, const double *upper_host,
                     const double *lower_host, nlsg_status *status_host, double *eps_out_host) {
  if (!e || !x_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  int rc = upload_bounds(e, upper_host, lower_host);
  if (rc) return rc;
  const uint64_t B = e->p.batch, n = e->p.n;
  NLSG_HIP(hipMemcpy(e->p.x, x_inout_host, B * n * 8, hipMemcpyHostToDevice));
  launch(e);
  NLSG_HIP(launches_status());
  NLSG_HIP(hipStreamSynchronize(e->stream));
  NLSG_HIP(hipMemcpy(x_inout_host, e->p.x, B * n * 8, hipMemcpyDeviceToHost));
  std::vector<NmProblem> pr(B);
  NLSG_HIP(hipMemcpy(pr.data(), e->p.prob, B * sizeof(NmProblem), hipMemcpyDeviceToHost));
  for (uint64_t b = 0; b < B; b++) {
    if (status_host) {
      nlsg_status &st = status_host[b];
      st.f_value = pr[b].f;
      st.iteration = pr[b].iter;
      st.function_calls_used = pr[b].fcalls;
      st.gradient_evals_used = 0;
      st.hessian_evals_used = 0;
      st.best_index = b;
      st.val_no_change = 0;
      st.std_err = pr[b].eps;
      st.done = 1;
      st.reserved = 0;
    }
    if (eps_out_host) eps_out_host[b] = pr[b].eps;
  }
  return NLSG_OK;
}

int nlsg_nm_phase_cycles(nlsg_nm *e, const double *x0_host, uint64_t *cycles_host) {
  if (!e || !x0_host || !cycles_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  const uint64_t B = e->p.batch;
  // a measurement aid: its buffer exists from its first use on, and it profiles the unbounded
  // solve only (a bounded engine would run on whatever bounds its last minimize left behind)
  if (e->cfg.bounded) return fail(NLSG_ERR_UNSUPPORTED, "nlsg_nm_phase_cycles profiles unbounded engines");
  if (!e->phase_dev) NLSG_HIP(pool_malloc(reinterpret_cast<void **>(&e->phase_dev), B * kNmPhases * 8));
  NLSG_HIP(hipMemcpy(e->p.x, x0_host, B * e->p.n * 8, hipMemcpyHostToDevice));
  NLSG_HIP(hipMemset(e->phase_dev, 0, B * kNmPhases * 8));
  e->p.phase = e->phase_dev;
  launch(e);
  e->p.phase = nullptr;
  NLSG_HIP(launches_status());
  NLSG_HIP(hipStreamSynchronize(e->stream));
  NLSG_HIP(hipMemcpy(cycles_host, e->phase_dev, B * kNmPhases * 8, hipMemcpyDeviceToHost));
  return NLSG_OK;
}

int nlsg_nm_time_solve(nlsg_nm *e, const double *x0_host, uint32_t repeats, float *ms_total) {
  if (!e || !x0_host || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
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
