// nlsolver_amd/csrc/nlsg_pso.hip — host side of the PSO engine + its C-ABI
// (include/nlsg_c_api.h). Same structure as nlsg_de.hip. No CPU fallback.
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "nlsg_comm.h"
#include "nlsg_pso_kernels.h"
#include "nlsg_rtc.h"

using namespace nlsg;

struct nlsg_pso {
  nlsg_pso_config cfg;
  PsoParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *lower_dev = nullptr, *upper_dev = nullptr, *zero_dev = nullptr, *tab_dev = nullptr;
  ShardLocal *loc = nullptr;
  ShardComm *comm = nullptr;  // set by nlsg_pso_comm_attach
  PsoRtcKernels rtc;          // objective == NLSG_OBJ_CUSTOM: the kernels hiprtc built for it
  double *rec = nullptr;
  int chunks = 0;
  int group = 0;  // lanes per particle when several particles share a wave (D <= 64), else 0
  bool long_rows = false;  // D > 1024: rows streamed in segments (pso_*_long_kernel)
  unsigned move_grid_cap = 2048;  // workgroups of the striding move kernel: 8 per CU (move_blocks_per_cu)
  bool initialised = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

#define PSO_FOR_CHUNKS(OBJ, chunks, CALL) \
  switch (chunks) {                       \
    case 1: CALL(OBJ, 1); break;          \
    case 2: CALL(OBJ, 2); break;          \
    case 4: CALL(OBJ, 4); break;          \
    case 8: CALL(OBJ, 8); break;          \
    default: break;                       \
  }
#define PSO_FOR_OBJ(obj, chunks, CALL)                                                  \
  switch (obj) {                                                                        \
    case NLSG_OBJ_ROSENBROCK: PSO_FOR_CHUNKS(NLSG_OBJ_ROSENBROCK, chunks, CALL); break; \
    case NLSG_OBJ_SPHERE: PSO_FOR_CHUNKS(NLSG_OBJ_SPHERE, chunks, CALL); break;         \
    case NLSG_OBJ_STYBLINSKI_TANG:                                                      \
      PSO_FOR_CHUNKS(NLSG_OBJ_STYBLINSKI_TANG, chunks, CALL);                           \
      break;                                                                            \
    case NLSG_OBJ_RASTRIGIN: PSO_FOR_CHUNKS(NLSG_OBJ_RASTRIGIN, chunks, CALL); break;   \
    default: break;                                                                     \
  }

// D > 1024: the segment-streaming kernels, per objective, row alignment and PSO type only
#define PSO_FOR_OBJ_LONG(obj, CALL)                                   \
  switch (obj) {                                                      \
    case NLSG_OBJ_ROSENBROCK: CALL(NLSG_OBJ_ROSENBROCK); break;       \
    case NLSG_OBJ_SPHERE: CALL(NLSG_OBJ_SPHERE); break;               \
    case NLSG_OBJ_STYBLINSKI_TANG: CALL(NLSG_OBJ_STYBLINSKI_TANG); break; \
    case NLSG_OBJ_RASTRIGIN: CALL(NLSG_OBJ_RASTRIGIN); break;         \
    default: break;                                                   \
  }

void launch_init(nlsg_pso *e) {
  const dim3 grid(static_cast<unsigned>((e->p.shard_n + 3) / 4)), block(256);
  const bool vec = e->p.D % 2 == 0;
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p};
    launch_module_kernel(e->rtc.init, grid.x, 256, 0, e->stream, args);
    return;
  }
  if (e->long_rows) {
#define CALL(OBJ)                                                                            \
  if (vec)                                                                                   \
    hipLaunchKernelGGL((pso_init_long_kernel<OBJ, true>), grid, block, 0, e->stream, e->p);  \
  else                                                                                       \
    hipLaunchKernelGGL((pso_init_long_kernel<OBJ, false>), grid, block, 0, e->stream, e->p)
    PSO_FOR_OBJ_LONG(e->cfg.objective, CALL)
#undef CALL
    return;
  }
#define CALL(OBJ, C)                                                                        \
  if (vec)                                                                                  \
    hipLaunchKernelGGL((pso_init_kernel<OBJ, C, true>), grid, block, 0, e->stream, e->p);   \
  else                                                                                      \
    hipLaunchKernelGGL((pso_init_kernel<OBJ, C, false>), grid, block, 0, e->stream, e->p)
  PSO_FOR_OBJ(e->cfg.objective, e->chunks, CALL)
#undef CALL
}

template <int OBJ, int TYPE>
void launch_move_groups(nlsg_pso *e, dim3 grid, int timing, uint64_t iter_ovr) {
  const dim3 block(256);
  switch (e->group) {
    case 4:
      hipLaunchKernelGGL((pso_move_groups_kernel<OBJ, 4, TYPE>), grid, block, 0, e->stream, e->p,
                         timing, iter_ovr);
      break;
    case 8:
      hipLaunchKernelGGL((pso_move_groups_kernel<OBJ, 8, TYPE>), grid, block, 0, e->stream, e->p,
                         timing, iter_ovr);
      break;
    case 16:
      hipLaunchKernelGGL((pso_move_groups_kernel<OBJ, 16, TYPE>), grid, block, 0, e->stream, e->p,
                         timing, iter_ovr);
      break;
    default:
      hipLaunchKernelGGL((pso_move_groups_kernel<OBJ, 32, TYPE>), grid, block, 0, e->stream, e->p,
                         timing, iter_ovr);
      break;
  }
}
template <int OBJ>
void launch_move_groups_type(nlsg_pso *e, dim3 grid, int timing, uint64_t iter_ovr) {
  if (e->cfg.type == NLSG_PSO_ACCELERATED)
    launch_move_groups<OBJ, NLSG_PSO_ACCELERATED>(e, grid, timing, iter_ovr);
  else
    launch_move_groups<OBJ, NLSG_PSO_VANILLA>(e, grid, timing, iter_ovr);
}

void launch_move(nlsg_pso *e, int timing, uint64_t iter_ovr) {
  // waves: one per particle, or one per 64 / group particles
  const uint64_t per_wave = e->group ? 64 / e->group : 1;
  const uint64_t waves = (e->p.shard_n + per_wave - 1) / per_wave;
  dim3 grid(static_cast<unsigned>((waves + 3) / 4)), block(256);
  // the one-particle-per-wave kernel strides over the shard: enough workgroups to fill every CU
  // (eight four-wave workgroups each), each wave then keeps the shared rows for all its particles
  // (the Accelerated move, bound by the vector unit; the Vanilla move is bound by HBM and keeps
  // one particle per wave: more rows in flight — striding measured 0.79 -> 0.69 of the roofline)
  const bool vec = e->p.D % 2 == 0;
  const bool accel = e->cfg.type == NLSG_PSO_ACCELERATED;
  if (accel && !e->group && !e->long_rows && grid.x > e->move_grid_cap) grid.x = e->move_grid_cap;
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &timing, &iter_ovr};
    launch_module_kernel(e->rtc.move, grid.x, 256, 0, e->stream, args);
    return;
  }
  if (e->long_rows) {
#define CALL(OBJ)                                                                                \
  if (vec && accel)                                                                              \
    hipLaunchKernelGGL((pso_move_long_kernel<OBJ, true, NLSG_PSO_ACCELERATED>), grid, block, 0,  \
                       e->stream, e->p, timing, iter_ovr);                                       \
  else if (vec)                                                                                  \
    hipLaunchKernelGGL((pso_move_long_kernel<OBJ, true, NLSG_PSO_VANILLA>), grid, block, 0,      \
                       e->stream, e->p, timing, iter_ovr);                                       \
  else if (accel)                                                                                \
    hipLaunchKernelGGL((pso_move_long_kernel<OBJ, false, NLSG_PSO_ACCELERATED>), grid, block, 0, \
                       e->stream, e->p, timing, iter_ovr);                                       \
  else                                                                                           \
    hipLaunchKernelGGL((pso_move_long_kernel<OBJ, false, NLSG_PSO_VANILLA>), grid, block, 0,     \
                       e->stream, e->p, timing, iter_ovr)
    PSO_FOR_OBJ_LONG(e->cfg.objective, CALL)
#undef CALL
    return;
  }
  if (e->group) {
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK: launch_move_groups_type<NLSG_OBJ_ROSENBROCK>(e, grid, timing, iter_ovr); break;
      case NLSG_OBJ_SPHERE: launch_move_groups_type<NLSG_OBJ_SPHERE>(e, grid, timing, iter_ovr); break;
      case NLSG_OBJ_STYBLINSKI_TANG:
        launch_move_groups_type<NLSG_OBJ_STYBLINSKI_TANG>(e, grid, timing, iter_ovr);
        break;
      default: launch_move_groups_type<NLSG_OBJ_RASTRIGIN>(e, grid, timing, iter_ovr); break;
    }
    return;
  }
#define CALL(OBJ, C)                                                                          \
  if (vec && accel)                                                                           \
    hipLaunchKernelGGL((pso_move_kernel<OBJ, C, true, NLSG_PSO_ACCELERATED>), grid, block, 0, \
                       e->stream, e->p, timing, iter_ovr);                                    \
  else if (vec)                                                                               \
    hipLaunchKernelGGL((pso_move_kernel<OBJ, C, true, NLSG_PSO_VANILLA>), grid, block, 0,     \
                       e->stream, e->p, timing, iter_ovr);                                    \
  else if (accel)                                                                             \
    hipLaunchKernelGGL((pso_move_kernel<OBJ, C, false, NLSG_PSO_ACCELERATED>), grid, block,   \
                       0, e->stream, e->p, timing, iter_ovr);                                 \
  else                                                                                        \
    hipLaunchKernelGGL((pso_move_kernel<OBJ, C, false, NLSG_PSO_VANILLA>), grid, block, 0,    \
                       e->stream, e->p, timing, iter_ovr)
  PSO_FOR_OBJ(e->cfg.objective, e->chunks, CALL)
#undef CALL
}

// Workgroups per CU of the striding Accelerated move: eight (two more than the six a CU holds at
// once with this kernel's 77 registers). Round 4 tried one exactly resident round instead (4 .. 8
// per CU through this switch: 152 / 151 / 149 / 148 / 147 us per launch) — no tail effect to remove;
// the kernel sits at the vector unit's issue rate (rocprofv3: the unit 91 % busy at the 2.0 GHz the
// chip sustains under fp64 load), so only fewer instructions make it faster.
unsigned move_blocks_per_cu() {
  if (const char *b = std::getenv("NLSG_PSO_MOVE_BLOCKS_PER_CU"))  // A/B switch
    if (std::atoi(b) > 0) return static_cast<unsigned>(std::atoi(b));
  return 8;
}

void launch_local_summary(nlsg_pso *e, double *rec_dev) {
  hipLaunchKernelGGL(pso_scan_partial_kernel, dim3(e->p.ntiles), dim3(256), 0, e->stream, e->p);
  if (!(e->cfg.eps > 0)) {
    hipLaunchKernelGGL(pso_local_kernel, dim3(1), dim3(256), 0, e->stream, e->p, e->loc, rec_dev);
    return;
  }
  hipLaunchKernelGGL(pso_local_kernel, dim3(1), dim3(256), 0, e->stream, e->p, e->loc,
                     static_cast<double *>(nullptr));
  hipLaunchKernelGGL(pso_var_partial_kernel, dim3(e->p.ntiles), dim3(256), 0, e->stream, e->p,
                     &e->loc->mean);
  hipLaunchKernelGGL(pso_var_local_kernel, dim3(1), dim3(256), 0, e->stream, e->p, e->loc);
  hipLaunchKernelGGL(pso_pack_record_kernel, dim3(1), dim3(256), 0, e->stream, e->p, e->loc,
                     rec_dev);
}

void launch_turn_single(nlsg_pso *e) {
  if (e->cfg.eps > 0) {
    launch_local_summary(e, e->rec);
    hipLaunchKernelGGL(pso_finalize_kernel, dim3(1), dim3(256), 0, e->stream, e->p, e->rec, 1,
                       static_cast<uint64_t>(kRecHeader) + e->p.D);
  } else {
    hipLaunchKernelGGL(pso_scan_head_kernel, dim3(e->p.ntiles), dim3(256), 0, e->stream, e->p);
  }
  launch_move(e, 0, 0);
}

int read_state(nlsg_pso *e, PsoState *host) {
  hipLaunchKernelGGL(pso_settle_kernel, dim3(1), dim3(1), 0, e->stream, e->p);
  NLSG_HIP(hipMemcpyAsync(host, e->p.state, sizeof(PsoState), hipMemcpyDeviceToHost, e->stream));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

void fill_status(const PsoState &s, nlsg_status *out) {
  out->f_value = s.gbest_val;
  out->iteration = s.iter;
  out->function_calls_used = s.fevals;
  out->gradient_evals_used = 0;
  out->hessian_evals_used = 0;
  out->best_index = s.gbest_idx;
  out->val_no_change = s.val_no_change;
  out->std_err = s.std_err;
  out->done = s.done;
  out->reserved = 0;
}

}  // namespace

extern "C" {

static int pso_create(const nlsg_pso_config *cfg, const nlsg_custom_objective *custom, nlsg_pso **out);

int nlsg_pso_create(const nlsg_pso_config *cfg, nlsg_pso **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_pso_create_custom");
  PhaseClock clk;
  const int rc = pso_create(cfg, nullptr, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

int nlsg_pso_create_custom(const nlsg_pso_config *cfg, const nlsg_custom_objective *obj,
                           nlsg_pso **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  PhaseClock clk;
  const int rc = pso_create(cfg, obj, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

static int pso_create(const nlsg_pso_config *cfg, const nlsg_custom_objective *custom, nlsg_pso **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_pso_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_pso_config size mismatch (%u vs %zu)",
                cfg->struct_size, sizeof(nlsg_pso_config));
  if (cfg->dim < 1) return fail(NLSG_ERR_INVALID_ARG, "dim must be >= 1");
  if (cfg->dim > 1024 && custom && custom->chain == NLSG_CUSTOM_VECTOR)
    return fail(NLSG_ERR_UNSUPPORTED,
                "dim %llu > 1024: a whole-vector objective needs the point in the wave's registers",
                (unsigned long long)cfg->dim);
  if (cfg->dim > 0xffffffffull) return fail(NLSG_ERR_UNSUPPORTED, "dim beyond 2^32");
  if (!custom && (cfg->objective < 0 || cfg->objective > NLSG_OBJ_RASTRIGIN))
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (cfg->type != NLSG_PSO_VANILLA && cfg->type != NLSG_PSO_ACCELERATED)
    return fail(NLSG_ERR_INVALID_ARG, "unknown PSO type %d", cfg->type);
  if (cfg->shard_n < 1 || cfg->shard_lo + cfg->shard_n > cfg->n_particles)
    return fail(NLSG_ERR_INVALID_ARG, "shard [%llu,+%llu) invalid for %llu particles",
                (unsigned long long)cfg->shard_lo, (unsigned long long)cfg->shard_n,
                (unsigned long long)cfg->n_particles);
  if (cfg->shard_n > (1ull << 32))
    return fail(NLSG_ERR_UNSUPPORTED, "shard_n > 2^32 particles per engine");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));

  nlsg_pso *e = new (std::nothrow) nlsg_pso();
  if (!e) return fail(NLSG_ERR_OOM, "host allocation failed");
  e->cfg = *cfg;
  const uint64_t D = cfg->dim, n = cfg->shard_n;
  e->chunks = D <= 128 ? 1 : D <= 256 ? 2 : D <= 512 ? 4 : 8;
  e->long_rows = D > 1024;  // the reference has no limit (nlsolver.h:2498-2742)
  e->group = D <= 8 ? 4 : D <= 16 ? 8 : D <= 32 ? 16 : D <= 64 ? 32 : 0;
  if (const char *g = std::getenv("NLSG_PSO_GROUPS"))  // A/B switch: 0 = one particle per wave at any D
    if (g[0] == '0') e->group = 0;
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
  PsoParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  auto alloc = [&](void **ptr, size_t bytes) { return pool_malloc(ptr, bytes ? bytes : 8); };
  const bool vanilla = cfg->type == NLSG_PSO_VANILLA;
  const size_t rows = n * D * sizeof(double);
  hipError_t he = hipSuccess;
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.pos), rows);
  if (he == hipSuccess && vanilla) he = alloc(reinterpret_cast<void **>(&p.vel), rows);
  if (he == hipSuccess && vanilla) he = alloc(reinterpret_cast<void **>(&p.pbest_pos), rows);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.pbest_val), n * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.cur_val), n * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.gbest_x), D * sizeof(double));
  if (he == hipSuccess) he = hipMemset(p.gbest_x, 0, D * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->lower_dev), D * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->upper_dev), D * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.state), sizeof(PsoState));
  p.ntiles = static_cast<uint32_t>((n + kTile - 1) / kTile);
  if (he == hipSuccess)
    he = alloc(reinterpret_cast<void **>(&p.part), p.ntiles * sizeof(TilePartial));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.ticket), 8);
  if (he == hipSuccess) he = hipMemset(p.ticket, 0, 8);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->loc), sizeof(ShardLocal));
  if (he == hipSuccess)
    he = alloc(reinterpret_cast<void **>(&e->rec), (kRecHeader + D) * sizeof(double));
  // inertia schedule pow(inertia, iter) (:2613) computed with the host libm, as the
  // reference does, so the device uses bit-identical values
  // (max_iter + 1 entries — guarded against wrapping — or up to the first fixed point of the
  // sequence: 0 after underflow, 1, inf; at most 2^22 entries)
  const uint64_t want = cfg->max_iter == ~0ull ? ~0ull : cfg->max_iter + 1;
  const uint64_t cap = std::min<uint64_t>(want, 1u << 22);
  std::vector<double> tab;
  tab.reserve(std::min<uint64_t>(cap, 4096));
  p.tab_fixed = 0;
  for (uint64_t k = 0; k < cap; k++) {
    tab.push_back(std::pow(cfg->inertia, static_cast<double>(k)));
    if (k >= 2 && std::memcmp(&tab[k], &tab[k - 1], 8) == 0 && std::memcmp(&tab[k], &tab[k - 2], 8) == 0 &&
        (tab[k] == 0.0 || tab[k] == 1.0 || std::isinf(tab[k]))) {
      p.tab_fixed = 1;
      break;
    }
  }
  p.tab_len = tab.size();
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->tab_dev), p.tab_len * sizeof(double));
  if (he == hipSuccess)
    he = hipMemcpy(e->tab_dev, tab.data(), p.tab_len * sizeof(double), hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he != hipSuccess) {
    nlsg_pso_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device allocation failed: %s", hipGetErrorString(he));
  }
  p.lower = e->lower_dev;
  p.upper = e->upper_dev;
  p.zero = e->zero_dev;
  p.inertia_tab = e->tab_dev;
  p.n = cfg->n_particles;
  p.D = D;
  p.shard_lo = cfg->shard_lo;
  p.shard_n = n;
  p.inertia = cfg->inertia;
  p.cog = cfg->cognitive;
  p.soc = cfg->social;
  p.eps = cfg->eps;
  p.fmul = cfg->minimize ? 1.0 : -1.0;
  p.max_iter = cfg->max_iter;
  p.best_val_no_change = cfg->best_val_no_change;
  p.seed = cfg->seed;
  p.type = cfg->type;
  p.bounded = cfg->bounded ? 1 : 0;
  if (custom) {
    const int rc2 = rtc_build_pso(custom, e->long_rows ? 0 : e->chunks, p.D % 2 == 0, cfg->type, e->group, &e->rtc);
    if (rc2) {
      nlsg_pso_destroy(e);
      return rc2;
    }
  }
  if (cfg->type == NLSG_PSO_ACCELERATED && !e->group && !e->long_rows) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0)
      e->move_grid_cap = move_blocks_per_cu() * static_cast<unsigned>(cus);
  }
  *out = e;
  return NLSG_OK;
}

int nlsg_pso_destroy(nlsg_pso *e) {
  if (!e) return NLSG_OK;
  PhaseClock clk;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  pool_free(e->p.pos);
  pool_free(e->p.vel);
  pool_free(e->p.pbest_pos);
  pool_free(e->p.pbest_val);
  pool_free(e->p.cur_val);
  pool_free(e->p.gbest_x);
  pool_free(e->p.state);
  pool_free(e->p.part);
  pool_free(e->p.ticket);
  pool_free(e->lower_dev);
  pool_free(e->upper_dev);
  pool_free(e->zero_dev);
  pool_free(e->tab_dev);
  pool_free(e->loc);
  comm_detach(e->comm);
  rtc_release(&e->rtc);
  pool_free(e->rec);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  call_timing().destroy_ms = clk.lap();
  return NLSG_OK;
}

int nlsg_pso_init(nlsg_pso *e, const double *lower_host, const double *upper_host) {
  if (!e || !lower_host || !upper_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  const size_t bytes = e->p.D * sizeof(double);
  NLSG_HIP(hipMemcpyAsync(e->lower_dev, lower_host, bytes, hipMemcpyHostToDevice, e->stream));
  NLSG_HIP(hipMemcpyAsync(e->upper_dev, upper_host, bytes, hipMemcpyHostToDevice, e->stream));
  NLSG_HIP(hipStreamSynchronize(e->stream));  // host buffers are borrowed for this call only
  hipLaunchKernelGGL(pso_reset_state_kernel, dim3(1), dim3(1), 0, e->stream, e->p);
  launch_init(e);
  NLSG_HIP(launches_status());
  e->initialised = true;
  return NLSG_OK;
}

int nlsg_pso_step(nlsg_pso *e, uint64_t turns) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  if (e->cfg.shard_n != e->cfg.n_particles)
    return fail(NLSG_ERR_STATE,
                "sharded engine: use nlsg_pso_turn_begin / nlsg_pso_turn_end around the exchange");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  for (uint64_t t = 0; t < turns; t++) launch_turn_single(e);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_pso_status(nlsg_pso *e, nlsg_status *out) {
  if (!e || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  PsoState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  fill_status(s, out);
  return NLSG_OK;
}

int nlsg_pso_best(nlsg_pso *e, double *x_host, double *f, uint64_t *index) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  PsoState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  if (x_host)
    NLSG_HIP(hipMemcpy(x_host, e->p.gbest_x, e->p.D * sizeof(double), hipMemcpyDeviceToHost));
  if (f) *f = s.gbest_val;
  if (index) *index = s.gbest_idx;
  return NLSG_OK;
}

int nlsg_pso_download(nlsg_pso *e, double *pos_host, double *vel_host, double *pbest_val_host,
                      double *cur_val_host) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  const uint64_t n = e->p.shard_n, D = e->p.D;
  if (pos_host) NLSG_HIP(hipMemcpy(pos_host, e->p.pos, n * D * sizeof(double), hipMemcpyDeviceToHost));
  if (vel_host) {
    if (!e->p.vel) return fail(NLSG_ERR_STATE, "velocities exist only for Vanilla PSO");
    NLSG_HIP(hipMemcpy(vel_host, e->p.vel, n * D * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (pbest_val_host)
    NLSG_HIP(hipMemcpy(pbest_val_host, e->p.pbest_val, n * sizeof(double), hipMemcpyDeviceToHost));
  if (cur_val_host)
    NLSG_HIP(hipMemcpy(cur_val_host, e->p.cur_val, n * sizeof(double), hipMemcpyDeviceToHost));
  return NLSG_OK;
}

int nlsg_pso_minimize(nlsg_pso *e, double *x_out_host, const double *lower_host,
                      const double *upper_host, uint64_t poll_every, nlsg_status *out) {
  if (!e || !x_out_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  PhaseClock clk;
  int rc = nlsg_pso_init(e, lower_host, upper_host);
  if (rc) return rc;
  call_timing().init_ms = clk.lap();
  if (poll_every == 0) poll_every = 32;
  PsoState s;
  for (;;) {
    rc = nlsg_pso_step(e, poll_every);
    if (rc) return rc;
    rc = read_state(e, &s);
    if (rc) return rc;
    if (s.done) break;
  }
  call_timing().iterate_ms = clk.lap();
  // x = swarm_best_position (nlsolver.h:2601)
  NLSG_HIP(hipMemcpy(x_out_host, e->p.gbest_x, e->p.D * sizeof(double), hipMemcpyDeviceToHost));
  if (out) fill_status(s, out);
  call_timing().readback_ms = clk.lap();
  return NLSG_OK;
}

int nlsg_pso_time_move_kernel(nlsg_pso *e, uint32_t launches, float *ms_total) {
  if (!e || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  PsoState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  NLSG_HIP(hipEventRecord(e->ev0, e->stream));
  for (uint32_t k = 0; k < launches; k++) launch_move(e, 1, s.iter + k);
  NLSG_HIP(hipEventRecord(e->ev1, e->stream));
  NLSG_HIP(hipEventSynchronize(e->ev1));
  NLSG_HIP(launches_status());
  NLSG_HIP(hipEventElapsedTime(ms_total, e->ev0, e->ev1));
  e->initialised = false;  // the swarm moved without best bookkeeping: re-init before solving
  return NLSG_OK;
}

uint64_t nlsg_pso_record_doubles(const nlsg_pso *e) {
  return e ? static_cast<uint64_t>(kRecHeader) + e->p.D : 0;
}

int nlsg_pso_turn_begin(nlsg_pso *e, double *send_dev) {
  if (!e || !send_dev) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  launch_local_summary(e, send_dev);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_pso_turn_end(nlsg_pso *e, const double *gathered_dev, int32_t world) {
  if (!e || !gathered_dev || world < 1) return fail(NLSG_ERR_INVALID_ARG, "bad argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(pso_finalize_kernel, dim3(1), dim3(256), 0, e->stream, e->p, gathered_dev,
                     world, static_cast<uint64_t>(kRecHeader) + e->p.D);
  launch_move(e, 0, 0);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_pso_comm_attach(nlsg_pso *e, const unsigned char *unique_id, int32_t world, int32_t rank) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (e->comm) return fail(NLSG_ERR_STATE, "a communicator is already attached");
  if (static_cast<uint64_t>(world) * e->p.shard_n != e->p.n ||
      static_cast<uint64_t>(rank) * e->p.shard_n != e->p.shard_lo)
    return fail(NLSG_ERR_INVALID_ARG, "shard [%llu, +%llu) of %llu does not match rank %d of %d",
                (unsigned long long)e->p.shard_lo, (unsigned long long)e->p.shard_n,
                (unsigned long long)e->p.n, rank, world);
  NLSG_HIP(hipSetDevice(e->cfg.device));
  return comm_attach(&e->comm, unique_id, world, rank, static_cast<uint64_t>(kRecHeader) + e->p.D);
}

int nlsg_pso_comm_ranks(nlsg_pso *e, int32_t *world_out, int32_t *rank_out) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  return comm_query(e->comm, world_out, rank_out);
}

// `turns` sharded turns without a host round trip. The move needs the exchanged swarm best, so
// nothing can run beside the collective: summary -> all-gather -> finaliser -> move, all on the
// engine's stream (no cross-stream dependency to pay for).
int nlsg_pso_step_sharded(nlsg_pso *e, uint64_t turns) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_pso_init has not been called");
  if (!e->comm) return fail(NLSG_ERR_STATE, "nlsg_pso_comm_attach has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  ShardComm *c = e->comm;
  const uint64_t stride = static_cast<uint64_t>(kRecHeader) + e->p.D;
  for (uint64_t t = 0; t < turns; t++) {
    launch_local_summary(e, e->rec);
    NLSG_RCCL(rccl_api().AllGather(e->rec, c->gathered, stride, ncclDouble, c->comm, e->stream));
    hipLaunchKernelGGL(pso_finalize_kernel, dim3(1), dim3(256), 0, e->stream, e->p, c->gathered,
                       c->world, stride);
    launch_move(e, 0, 0);
  }
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

}  // extern "C"
