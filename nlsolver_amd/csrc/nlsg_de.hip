// nlsolver_amd/csrc/nlsg_de.hip — host side of the DE engine + its C-ABI
// (include/nlsg_c_api.h). Owns the device buffers and the HIP stream; enqueues
// the kernels of nlsg_de_kernels.h. No CPU fallback: every entry point either
// runs on a gfx950 device or returns an error.
#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include "nlsg_comm.h"
#include "nlsg_de_kernels.h"
#include "nlsg_rtc.h"

using namespace nlsg;

struct nlsg_de {
  nlsg_de_config cfg;
  DeParams p;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *x0_dev = nullptr;
  double *zero_dev = nullptr;
  double *rec = nullptr;  // local record (single-GPU finaliser input)
  int chunks = 0;
  int group = 0;  // lanes per agent when several agents share a wave (D <= 64), else 0
  bool long_rows = false;  // D > 1024: rows streamed in segments (de_*_long_kernel)
  bool initialised = false;
  uint64_t k = 0;            // generations launched so far (= index of the next head)
  // strategy random: the head of turn k (scan, stop tests) does not feed generation k+1
  // except through the stop flag, so it runs on a side stream beside that generation
  // (the generation is non-destructive: it writes the other population / score buffers).
  hipStream_t side = nullptr;
  hipEvent_t ev_gen[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_head[4] = {nullptr, nullptr, nullptr, nullptr};
  ShardComm *comm = nullptr;  // set by nlsg_de_comm_attach
  DeRtcKernels rtc;           // objective == NLSG_OBJ_CUSTOM: the kernels hiprtc built for it
  bool overlap = false;
  bool fused = false;   // head k and generation k+1 share one launch (strategy random, one GPU)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

template <typename K>
struct Dispatch;  // OBJ x CHUNKS dispatch of a kernel family

#define NLSG_FOR_CHUNKS(OBJ, chunks, CALL) \
  switch (chunks) {                        \
    case 1: CALL(OBJ, 1); break;           \
    case 2: CALL(OBJ, 2); break;           \
    case 4: CALL(OBJ, 4); break;           \
    case 8: CALL(OBJ, 8); break;           \
    default: break;                        \
  }
#define NLSG_FOR_OBJ(obj, chunks, CALL)                                              \
  switch (obj) {                                                                     \
    case NLSG_OBJ_ROSENBROCK: NLSG_FOR_CHUNKS(NLSG_OBJ_ROSENBROCK, chunks, CALL); break; \
    case NLSG_OBJ_SPHERE: NLSG_FOR_CHUNKS(NLSG_OBJ_SPHERE, chunks, CALL); break;     \
    case NLSG_OBJ_STYBLINSKI_TANG:                                                   \
      NLSG_FOR_CHUNKS(NLSG_OBJ_STYBLINSKI_TANG, chunks, CALL);                       \
      break;                                                                         \
    case NLSG_OBJ_RASTRIGIN: NLSG_FOR_CHUNKS(NLSG_OBJ_RASTRIGIN, chunks, CALL); break; \
    default: break;                                                                  \
  }

// kernels of a run-time compiled objective take the same arguments through the module API
void launch_module(nlsg_de *e, hipFunction_t fn, unsigned grid, void **args) {
  launch_module_kernel(fn, grid, 256, 0, e->stream, args);
}

// D > 1024: the segment-streaming kernels, per objective and row alignment only
#define NLSG_FOR_OBJ_LONG(obj, CALL)                                  \
  switch (obj) {                                                      \
    case NLSG_OBJ_ROSENBROCK: CALL(NLSG_OBJ_ROSENBROCK); break;       \
    case NLSG_OBJ_SPHERE: CALL(NLSG_OBJ_SPHERE); break;               \
    case NLSG_OBJ_STYBLINSKI_TANG: CALL(NLSG_OBJ_STYBLINSKI_TANG); break; \
    case NLSG_OBJ_RASTRIGIN: CALL(NLSG_OBJ_RASTRIGIN); break;         \
    default: break;                                                   \
  }

void launch_init(nlsg_de *e) {
  const dim3 grid(static_cast<unsigned>((e->p.shard_n + 3) / 4)), block(256);
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &e->x0_dev};
    launch_module(e, e->rtc.init, grid.x, args);
    return;
  }
  if (e->long_rows) {
#define CALL(OBJ)                                                                              \
  if (e->p.vec)                                                                                \
    hipLaunchKernelGGL((de_init_long_kernel<OBJ, true>), grid, block, 0, e->stream, e->p, e->x0_dev); \
  else                                                                                         \
    hipLaunchKernelGGL((de_init_long_kernel<OBJ, false>), grid, block, 0, e->stream, e->p, e->x0_dev)
    NLSG_FOR_OBJ_LONG(e->cfg.objective, CALL)
#undef CALL
    return;
  }
#define CALL(OBJ, C)                                                                          \
  if (e->p.vec)                                                                               \
    hipLaunchKernelGGL((de_init_kernel<OBJ, C, true>), grid, block, 0, e->stream, e->p,       \
                       e->x0_dev);                                                            \
  else                                                                                        \
    hipLaunchKernelGGL((de_init_kernel<OBJ, C, false>), grid, block, 0, e->stream, e->p,      \
                       e->x0_dev)
  NLSG_FOR_OBJ(e->cfg.objective, e->chunks, CALL)
#undef CALL
}

template <int OBJ>
void launch_generation_groups(nlsg_de *e, dim3 grid, int par, uint64_t generation, int ignore_done) {
  const dim3 block(256);
  switch (e->group) {
    case 4:
      hipLaunchKernelGGL((de_generation_groups_kernel<OBJ, 4>), grid, block, 0, e->stream, e->p, par,
                         generation, ignore_done);
      break;
    case 8:
      hipLaunchKernelGGL((de_generation_groups_kernel<OBJ, 8>), grid, block, 0, e->stream, e->p, par,
                         generation, ignore_done);
      break;
    case 16:
      hipLaunchKernelGGL((de_generation_groups_kernel<OBJ, 16>), grid, block, 0, e->stream, e->p, par,
                         generation, ignore_done);
      break;
    default:
      hipLaunchKernelGGL((de_generation_groups_kernel<OBJ, 32>), grid, block, 0, e->stream, e->p, par,
                         generation, ignore_done);
      break;
  }
}

void launch_generation(nlsg_de *e, int par, uint64_t generation, int ignore_done = 0) {
  e->p.gen_key = ctr_key(e->p.seed, generation);
  // waves: one per agent, or one per 64 / group agents
  const uint64_t per_wave = e->group ? 64 / e->group : 1;
  const uint64_t waves = (e->p.shard_n + per_wave - 1) / per_wave;
  const dim3 grid(static_cast<unsigned>((waves + 3) / 4)), block(256);
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &par, &generation, &ignore_done};
    launch_module(e, e->rtc.generation, grid.x, args);
    return;
  }
  if (e->long_rows) {
#define CALL(OBJ)                                                                               \
  if (e->p.vec)                                                                                 \
    hipLaunchKernelGGL((de_generation_long_kernel<OBJ, true>), grid, block, 0, e->stream, e->p, \
                       par, generation, ignore_done);                                           \
  else                                                                                          \
    hipLaunchKernelGGL((de_generation_long_kernel<OBJ, false>), grid, block, 0, e->stream, e->p, \
                       par, generation, ignore_done)
    NLSG_FOR_OBJ_LONG(e->cfg.objective, CALL)
#undef CALL
    return;
  }
  if (e->group) {
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK:
        launch_generation_groups<NLSG_OBJ_ROSENBROCK>(e, grid, par, generation, ignore_done);
        break;
      case NLSG_OBJ_SPHERE:
        launch_generation_groups<NLSG_OBJ_SPHERE>(e, grid, par, generation, ignore_done);
        break;
      case NLSG_OBJ_STYBLINSKI_TANG:
        launch_generation_groups<NLSG_OBJ_STYBLINSKI_TANG>(e, grid, par, generation, ignore_done);
        break;
      default:
        launch_generation_groups<NLSG_OBJ_RASTRIGIN>(e, grid, par, generation, ignore_done);
        break;
    }
    return;
  }
#define CALL(OBJ, C)                                                                          \
  if (e->p.vec)                                                                               \
    hipLaunchKernelGGL((de_generation_kernel<OBJ, C, true>), grid, block, 0, e->stream, e->p, \
                       par, generation, ignore_done);                                         \
  else                                                                                        \
    hipLaunchKernelGGL((de_generation_kernel<OBJ, C, false>), grid, block, 0, e->stream,      \
                       e->p, par, generation, ignore_done)
  NLSG_FOR_OBJ(e->cfg.objective, e->chunks, CALL)
#undef CALL
}

template <int OBJ>
void launch_fused_turn_groups(nlsg_de *e, dim3 grid, int par, uint64_t generation) {
  const dim3 block(256);
  switch (e->group) {
    case 4:
      hipLaunchKernelGGL((de_turn_groups_kernel<OBJ, 4>), grid, block, 0, e->stream, e->p, par, generation);
      break;
    case 8:
      hipLaunchKernelGGL((de_turn_groups_kernel<OBJ, 8>), grid, block, 0, e->stream, e->p, par, generation);
      break;
    case 16:
      hipLaunchKernelGGL((de_turn_groups_kernel<OBJ, 16>), grid, block, 0, e->stream, e->p, par, generation);
      break;
    default:
      hipLaunchKernelGGL((de_turn_groups_kernel<OBJ, 32>), grid, block, 0, e->stream, e->p, par, generation);
      break;
  }
}

// head k and generation k+1 in one launch (de_turn_kernel)
void launch_fused_turn(nlsg_de *e, int par, uint64_t generation) {
  e->p.gen_key = ctr_key(e->p.seed, generation);
  const uint64_t per_wave = e->group ? 64 / e->group : 1;
  const uint64_t waves = (e->p.shard_n + per_wave - 1) / per_wave;
  const dim3 grid(static_cast<unsigned>((waves + 3) / 4 + e->p.ntiles)), block(256);
  if (e->cfg.objective == NLSG_OBJ_CUSTOM) {
    void *args[] = {&e->p, &par, &generation};
    launch_module(e, e->rtc.turn, grid.x, args);
    return;
  }
  if (e->group) {
    switch (e->cfg.objective) {
      case NLSG_OBJ_ROSENBROCK: launch_fused_turn_groups<NLSG_OBJ_ROSENBROCK>(e, grid, par, generation); break;
      case NLSG_OBJ_SPHERE: launch_fused_turn_groups<NLSG_OBJ_SPHERE>(e, grid, par, generation); break;
      case NLSG_OBJ_STYBLINSKI_TANG:
        launch_fused_turn_groups<NLSG_OBJ_STYBLINSKI_TANG>(e, grid, par, generation);
        break;
      default: launch_fused_turn_groups<NLSG_OBJ_RASTRIGIN>(e, grid, par, generation); break;
    }
    return;
  }
#define CALL(OBJ, C)                                                                        \
  if (e->p.vec)                                                                             \
    hipLaunchKernelGGL((de_turn_kernel<OBJ, C, true>), grid, block, 0, e->stream, e->p, par, \
                       generation);                                                         \
  else                                                                                      \
    hipLaunchKernelGGL((de_turn_kernel<OBJ, C, false>), grid, block, 0, e->stream, e->p,    \
                       par, generation)
  NLSG_FOR_OBJ(e->cfg.objective, e->chunks, CALL)
#undef CALL
}

// Head of turn k outside a fused turn, one launch. rec_dev == nullptr: one GPU, the head
// finishes the turn; else the shard's exchange record.
void launch_head(nlsg_de *e, double *rec_dev, hipStream_t st) {
  hipLaunchKernelGGL(de_scan_head_kernel, dim3(e->p.ntiles), dim3(256), 0, st, e->p, e->k, rec_dev);
}
void launch_local_summary(nlsg_de *e, double *rec_dev, hipStream_t st) { launch_head(e, rec_dev, st); }
void launch_head_single(nlsg_de *e, hipStream_t st) { launch_head(e, nullptr, st); }

// One turn on one GPU. Serial form: head k, then generation k+1. Overlapped form
// (strategy random): generation k+1 is launched on the main stream as soon as head k-1 is
// done, head k runs on the side stream next to it; if head k fires a stop test the
// generation's output (other buffers) is simply never adopted.
int launch_turn_single(nlsg_de *e) {
  const uint64_t k = e->k;
  if (e->fused) {
    launch_fused_turn(e, static_cast<int>(k & 1), k + 1);
  } else if (!e->overlap) {
    launch_head_single(e, e->stream);
    launch_generation(e, static_cast<int>(k & 1), k + 1);
  } else {
    NLSG_HIP(hipStreamWaitEvent(e->side, e->ev_gen[k & 3], 0));      // population k exists
    launch_head_single(e, e->side);
    NLSG_HIP(hipEventRecord(e->ev_head[k & 3], e->side));
    if (k > 0) NLSG_HIP(hipStreamWaitEvent(e->stream, e->ev_head[(k - 1) & 3], 0));
    launch_generation(e, static_cast<int>(k & 1), k + 1);
    NLSG_HIP(hipEventRecord(e->ev_gen[(k + 1) & 3], e->stream));
  }
  e->k = k + 1;
  return NLSG_OK;
}

// make the main stream wait for everything queued on the side stream
int join_side(nlsg_de *e) {
  if (e->overlap && e->k > 0) NLSG_HIP(hipStreamWaitEvent(e->stream, e->ev_head[(e->k - 1) & 3], 0));
  return NLSG_OK;
}

int read_state(nlsg_de *e, DeState *host) {
  int rc = join_side(e);
  if (rc) return rc;
  hipLaunchKernelGGL(de_settle_kernel, dim3(1), dim3(1), 0, e->stream, e->p, e->k);
  NLSG_HIP(hipMemcpyAsync(host, e->p.state, sizeof(DeState), hipMemcpyDeviceToHost, e->stream));
  NLSG_HIP(hipStreamSynchronize(e->stream));
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

void fill_status(const DeState &s, nlsg_status *out) {
  out->f_value = s.best_f;
  out->iteration = s.iter;
  out->function_calls_used = s.fcalls;
  out->gradient_evals_used = 0;
  out->hessian_evals_used = 0;
  out->best_index = s.best_id;
  out->val_no_change = s.val_no_change;
  out->std_err = s.std_err;
  out->done = s.done;
  out->reserved = 0;
}

}  // namespace

extern "C" {

const char *nlsg_last_error(void) { return err_buf(); }
int nlsg_abi_version(void) { return NLSG_ABI_VERSION; }
int nlsg_release_cached(void) {
  pool_release_all();
  return NLSG_OK;
}
uint64_t nlsg_cached_bytes(void) { return pool_idle_bytes(); }
int nlsg_call_timing(double *ms_out6) {
  if (!ms_out6) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  const CallTiming &t = call_timing();
  ms_out6[0] = t.create_ms;
  ms_out6[1] = t.upload_ms;
  ms_out6[2] = t.init_ms;
  ms_out6[3] = t.iterate_ms;
  ms_out6[4] = t.readback_ms;
  ms_out6[5] = t.destroy_ms;
  return NLSG_OK;
}

int nlsg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess &&
        std::strncmp(prop.gcnArchName, "gfx950", 6) == 0)
      ok++;
  }
  return ok;
}

static int de_create(const nlsg_de_config *cfg, const nlsg_custom_objective *custom, nlsg_de **out);

int nlsg_de_create(const nlsg_de_config *cfg, nlsg_de **out) {
  if (cfg && cfg->objective == NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "NLSG_OBJ_CUSTOM engines are made by nlsg_de_create_custom");
  PhaseClock clk;
  const int rc = de_create(cfg, nullptr, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

int nlsg_de_create_custom(const nlsg_de_config *cfg, const nlsg_custom_objective *obj, nlsg_de **out) {
  if (!cfg || !obj) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (cfg->objective != NLSG_OBJ_CUSTOM)
    return fail(NLSG_ERR_INVALID_ARG, "cfg.objective must be NLSG_OBJ_CUSTOM");
  PhaseClock clk;
  const int rc = de_create(cfg, obj, out);
  call_timing().create_ms = clk.lap();
  return rc;
}

static int de_create(const nlsg_de_config *cfg, const nlsg_custom_objective *custom, nlsg_de **out) {
  if (!cfg || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(nlsg_de_config))
    return fail(NLSG_ERR_INVALID_ARG, "nlsg_de_config size mismatch (%u vs %zu)",
                cfg->struct_size, sizeof(nlsg_de_config));
  if (cfg->dim < 1) return fail(NLSG_ERR_INVALID_ARG, "dim must be >= 1");
  if (cfg->dim > 1024 && custom && custom->chain == NLSG_CUSTOM_VECTOR)
    return fail(NLSG_ERR_UNSUPPORTED,
                "dim %llu > 1024: a whole-vector objective needs the point in the wave's registers",
                (unsigned long long)cfg->dim);
  if (cfg->dim > 0xffffffffull) return fail(NLSG_ERR_UNSUPPORTED, "dim beyond 2^32");
  if (!custom && (cfg->objective < 0 || cfg->objective > NLSG_OBJ_RASTRIGIN))
    return fail(NLSG_ERR_INVALID_ARG, "unknown objective %d", cfg->objective);
  if (cfg->strategy != NLSG_DE_BEST && cfg->strategy != NLSG_DE_RANDOM)
    return fail(NLSG_ERR_INVALID_ARG, "unknown strategy %d", cfg->strategy);
  if (cfg->shard_n < 4 || cfg->shard_lo + cfg->shard_n > cfg->pop)
    return fail(NLSG_ERR_INVALID_ARG,
                "shard [%llu,+%llu) invalid for pop %llu (a shard needs >= 4 agents: three "
                "distinct donors besides the target, nlsolver.h:2331-2355)",
                (unsigned long long)cfg->shard_lo, (unsigned long long)cfg->shard_n,
                (unsigned long long)cfg->pop);
  if (cfg->shard_n > 0xffffffffull)  // donor indices are drawn as 32-bit values inside a shard
    return fail(NLSG_ERR_UNSUPPORTED, "shard_n >= 2^32 agents per engine");
  int rc = check_device(cfg->device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(cfg->device));

  nlsg_de *e = new (std::nothrow) nlsg_de();
  if (!e) return fail(NLSG_ERR_OOM, "host allocation failed");
  e->cfg = *cfg;
  const uint64_t D = cfg->dim, n = cfg->shard_n;
  e->chunks = D <= 128 ? 1 : D <= 256 ? 2 : D <= 512 ? 4 : 8;
  e->long_rows = D > 1024;  // the reference has no limit (nlsolver.h:2302-2477)
  e->group = D <= 8 ? 4 : D <= 16 ? 8 : D <= 32 ? 16 : D <= 64 ? 32 : 0;
  if (const char *g = std::getenv("NLSG_DE_GROUPS"))  // A/B switch: 0 = one agent per wave at any D
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
  DeParams &p = e->p;
  std::memset(&p, 0, sizeof p);
  auto alloc = [&](void **ptr, size_t bytes) { return pool_malloc(ptr, bytes ? bytes : 8); };
  hipError_t he = hipSuccess;
  const size_t rows = n * D * sizeof(double);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.buf[0]), rows);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.buf[1]), rows);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.scores[0]), n * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.scores[1]), n * sizeof(double));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.best_x), D * sizeof(double));
  if (he == hipSuccess && cfg->trace)
    he = alloc(reinterpret_cast<void **>(&p.trace), n * kTraceWords * sizeof(uint64_t));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.state), sizeof(DeState));
  p.ntiles = static_cast<uint32_t>((n + kTile - 1) / kTile);
  if (he == hipSuccess)
    he = alloc(reinterpret_cast<void **>(&p.part), p.ntiles * sizeof(TilePartial));
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&p.ticket), 8);
  if (he == hipSuccess) he = hipMemset(p.ticket, 0, 8);
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->zero_dev), 16);
  if (he == hipSuccess) he = hipMemset(e->zero_dev, 0, 16);
  p.zero = e->zero_dev;
  if (he == hipSuccess) he = alloc(reinterpret_cast<void **>(&e->x0_dev), D * sizeof(double));
  if (he == hipSuccess)
    he = alloc(reinterpret_cast<void **>(&e->rec), (kRecHeader + D) * sizeof(double));
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  // Overlapped turns (one GPU, strategy random) are opt-in (NLSG_DE_OVERLAP=1): on MI355X /
  // ROCm 7.2 the two cross-stream event waits per turn cost more (~+3 us) than the 10 us
  // head they hide, measured at pop = 65536 (61.7 vs 58.7 us per turn).
  const char *ov = std::getenv("NLSG_DE_OVERLAP");
  e->overlap = cfg->strategy == NLSG_DE_RANDOM && cfg->shard_n == cfg->pop && ov && ov[0] == '1';
  const char *fu = std::getenv("NLSG_DE_FUSED_TURN");
  e->fused = cfg->strategy == NLSG_DE_RANDOM && cfg->shard_n == cfg->pop && !e->long_rows &&
             !e->overlap && !cfg->trace && !(fu && fu[0] == '0');  // the trace buffer is not double-buffered
  if (e->overlap) {
    if (he == hipSuccess) he = pool_stream_get(&e->side);
    for (int i = 0; i < 4; i++) {
      if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_gen[i], hipEventDisableTiming);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_head[i], hipEventDisableTiming);
    }
  }
  if (he != hipSuccess) {
    nlsg_de_destroy(e);
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP,
                "device allocation failed: %s", hipGetErrorString(he));
  }
  p.pop = cfg->pop;
  p.D = D;
  p.shard_lo = cfg->shard_lo;
  p.shard_n = n;
  p.CR = cfg->CR;
  {  // the crossover test u01(z) < CR on the draw itself: u01 is monotone in z, so there is a
     // smallest z whose uniform reaches CR (none: every draw passes)
    const double cr = cfg->CR;
    if (u01(~0ull) < cr) {
      p.cr_all = 1;
      p.cr_thresh = ~0ull;
    } else if (!(u01(0) < cr)) {  // CR <= 0 or NaN: no draw passes
      p.cr_thresh = 0;
    } else {
      uint64_t lo = 0, hi = ~0ull;  // u01(lo) < CR <= u01(hi)
      while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (u01(mid) < cr) lo = mid; else hi = mid;
      }
      p.cr_thresh = hi;
    }
  }
  p.F = cfg->F;
  p.eps = cfg->eps;
  p.fmul = cfg->minimize ? 1.0 : -1.0;  // f_multiplier, nlsolver.h:2418
  p.max_iter = cfg->max_iter;
  p.best_val_no_change = cfg->best_val_no_change;
  p.seed = cfg->seed;
  p.strategy = cfg->strategy;
  p.vec = (D % 2 == 0) ? 1 : 0;
  // measured at pop 2^20 (1 GiB per buffer): generation 0.892 -> 0.805 ms; at pop 65 536 (both
  // buffers inside the Infinity Cache): 47.8 -> 46.6 us — the streamed stores win at both sizes
  p.stream = 1;
  if (const char *sv = std::getenv("NLSG_DE_STREAM")) p.stream = sv[0] == '1' ? 1 : 0;  // A/B switch
  if (custom) {
    const int rc2 = rtc_build_de(custom, e->long_rows ? 0 : e->chunks, p.vec != 0, e->group, &e->rtc);
    if (rc2) {
      nlsg_de_destroy(e);
      return rc2;
    }
  }
  *out = e;
  return NLSG_OK;
}

int nlsg_de_destroy(nlsg_de *e) {
  if (!e) return NLSG_OK;
  PhaseClock clk;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  pool_free(e->p.buf[0]);
  pool_free(e->p.buf[1]);
  pool_free(e->p.scores[0]);
  pool_free(e->p.scores[1]);
  if (e->side) {
    hipStreamSynchronize(e->side);
    pool_stream_put(e->cfg.device, e->side);
  }
  for (int i = 0; i < 4; i++) {
    if (e->ev_gen[i]) hipEventDestroy(e->ev_gen[i]);
    if (e->ev_head[i]) hipEventDestroy(e->ev_head[i]);
  }
  pool_free(e->p.best_x);
  pool_free(e->p.trace);
  pool_free(e->p.state);
  pool_free(e->p.part);
  pool_free(e->p.ticket);
  comm_detach(e->comm);
  rtc_release(&e->rtc);
  pool_free(e->x0_dev);
  pool_free(e->zero_dev);
  pool_free(e->rec);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->own_stream && e->stream) pool_stream_put(e->cfg.device, e->stream);
  delete e;
  call_timing().destroy_ms = clk.lap();
  return NLSG_OK;
}

int nlsg_de_init(nlsg_de *e, const double *x0_host) {
  if (!e || !x0_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  NLSG_HIP(hipMemcpyAsync(e->x0_dev, x0_host, e->p.D * sizeof(double), hipMemcpyHostToDevice,
                          e->stream));
  // the host buffer is borrowed for this call only
  NLSG_HIP(hipStreamSynchronize(e->stream));
  int rcj = join_side(e);  // a previous run's last head may still be queued on the side stream
  if (rcj) return rcj;
  hipLaunchKernelGGL(de_reset_state_kernel, dim3(1), dim3(1), 0, e->stream, e->p);
  launch_init(e);
  e->k = 0;
  if (e->overlap) NLSG_HIP(hipEventRecord(e->ev_gen[0], e->stream));
  NLSG_HIP(launches_status());
  e->initialised = true;
  return NLSG_OK;
}

int nlsg_de_step(nlsg_de *e, uint64_t turns) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  if (e->cfg.shard_n != e->cfg.pop)
    return fail(NLSG_ERR_STATE,
                "sharded engine: use nlsg_de_turn_begin / nlsg_de_turn_end around the exchange");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  for (uint64_t t = 0; t < turns; t++) {
    int rc = launch_turn_single(e);
    if (rc) return rc;
  }
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_de_status(nlsg_de *e, nlsg_status *out) {
  if (!e || !out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  DeState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  fill_status(s, out);
  return NLSG_OK;
}

int nlsg_de_best(nlsg_de *e, double *x_host, double *f, uint64_t *index) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  DeState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  if (x_host)
    NLSG_HIP(hipMemcpy(x_host, e->p.best_x, e->p.D * sizeof(double), hipMemcpyDeviceToHost));
  if (f) *f = s.best_f;
  if (index) *index = s.best_id;
  return NLSG_OK;
}

int nlsg_de_download(nlsg_de *e, double *pop_host, double *scores_host, uint64_t *trace_host) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  DeState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  const uint64_t n = e->p.shard_n, D = e->p.D;
  if (pop_host)
    NLSG_HIP(hipMemcpy(pop_host, e->p.buf[s.parity], n * D * sizeof(double),
                       hipMemcpyDeviceToHost));
  if (scores_host)
    NLSG_HIP(hipMemcpy(scores_host, e->p.scores[s.parity], n * sizeof(double),
                       hipMemcpyDeviceToHost));
  if (trace_host) {
    if (!e->p.trace) return fail(NLSG_ERR_STATE, "engine was created without cfg.trace");
    NLSG_HIP(hipMemcpy(trace_host, e->p.trace, n * kTraceWords * sizeof(uint64_t),
                       hipMemcpyDeviceToHost));
  }
  return NLSG_OK;
}

int nlsg_de_upload(nlsg_de *e, const double *pop_host, const double *scores_host) {
  if (!e || !pop_host || !scores_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  DeState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  const uint64_t n = e->p.shard_n, D = e->p.D;
  NLSG_HIP(hipMemcpy(e->p.buf[s.parity], pop_host, n * D * sizeof(double), hipMemcpyHostToDevice));
  NLSG_HIP(hipMemcpy(e->p.scores[s.parity], scores_host, n * sizeof(double),
                     hipMemcpyHostToDevice));
  return NLSG_OK;
}

int nlsg_de_minimize(nlsg_de *e, double *x_inout_host, uint64_t poll_every, nlsg_status *out) {
  if (!e || !x_inout_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  PhaseClock clk;
  int rc = nlsg_de_init(e, x_inout_host);
  if (rc) return rc;
  call_timing().init_ms = clk.lap();
  if (poll_every == 0) poll_every = 32;
  DeState s;
  for (;;) {
    rc = nlsg_de_step(e, poll_every);
    if (rc) return rc;
    rc = read_state(e, &s);
    if (rc) return rc;
    if (s.done) break;
  }
  call_timing().iterate_ms = clk.lap();
  // x = agents[best_id] (nlsolver.h:2444)
  NLSG_HIP(hipMemcpy(x_inout_host, e->p.best_x, e->p.D * sizeof(double), hipMemcpyDeviceToHost));
  if (out) fill_status(s, out);
  call_timing().readback_ms = clk.lap();
  return NLSG_OK;
}

int nlsg_de_time_generation_kernel(nlsg_de *e, uint32_t launches, float *ms_total) {
  if (!e || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  DeState s;
  int rc = read_state(e, &s);
  if (rc) return rc;
  NLSG_HIP(hipEventRecord(e->ev0, e->stream));
  for (uint32_t k = 0; k < launches; k++)
    launch_generation(e, (s.parity + static_cast<int>(k)) & 1, s.iter + 1 + k, 1);
  NLSG_HIP(hipEventRecord(e->ev1, e->stream));
  NLSG_HIP(hipEventSynchronize(e->ev1));
  NLSG_HIP(launches_status());
  NLSG_HIP(hipEventElapsedTime(ms_total, e->ev0, e->ev1));
  // The population advanced `launches` generations without best scans; the
  // engine must be re-initialised before it is used for a solve again.
  e->initialised = false;
  return NLSG_OK;
}

int nlsg_de_time_turns(nlsg_de *e, uint64_t turns, float *ms_total) {
  if (!e || !ms_total) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  if (e->cfg.shard_n != e->cfg.pop) return fail(NLSG_ERR_STATE, "sharded engine");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  int rcj = join_side(e);
  if (rcj) return rcj;
  NLSG_HIP(hipEventRecord(e->ev0, e->stream));
  for (uint64_t t = 0; t < turns; t++) {
    int rc = launch_turn_single(e);
    if (rc) return rc;
  }
  rcj = join_side(e);
  if (rcj) return rcj;
  NLSG_HIP(hipEventRecord(e->ev1, e->stream));
  NLSG_HIP(hipEventSynchronize(e->ev1));
  NLSG_HIP(launches_status());
  NLSG_HIP(hipEventElapsedTime(ms_total, e->ev0, e->ev1));
  return NLSG_OK;
}

uint64_t nlsg_de_record_doubles(const nlsg_de *e) {
  return e ? static_cast<uint64_t>(kRecHeader) + e->p.D : 0;
}

int nlsg_de_turn_begin(nlsg_de *e, double *send_dev) {
  if (!e || !send_dev) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  launch_local_summary(e, send_dev, e->stream);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_de_turn_end(nlsg_de *e, const double *gathered_dev, int32_t world) {
  int rc = nlsg_de_turn_finalize(e, gathered_dev, world);
  if (rc) return rc;
  return nlsg_de_turn_generation(e);
}

int nlsg_de_turn_finalize(nlsg_de *e, const double *gathered_dev, int32_t world) {
  if (!e || !gathered_dev || world < 1) return fail(NLSG_ERR_INVALID_ARG, "bad argument");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  hipLaunchKernelGGL(de_finalize_kernel, dim3(1), dim3(256), 0, e->stream, e->p, gathered_dev,
                     world, static_cast<uint64_t>(kRecHeader) + e->p.D);
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_de_turn_generation(nlsg_de *e) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  NLSG_HIP(hipSetDevice(e->cfg.device));
  launch_generation(e, static_cast<int>(e->k & 1), e->k + 1);
  e->k += 1;
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

int nlsg_de_can_speculate(const nlsg_de *e) {
  return (e && e->cfg.strategy == NLSG_DE_RANDOM) ? 1 : 0;
}

int nlsg_de_comm_attach(nlsg_de *e, const unsigned char *unique_id, int32_t world, int32_t rank) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (e->comm) return fail(NLSG_ERR_STATE, "a communicator is already attached");
  if (static_cast<uint64_t>(world) * e->p.shard_n != e->p.pop ||
      static_cast<uint64_t>(rank) * e->p.shard_n != e->p.shard_lo)
    return fail(NLSG_ERR_INVALID_ARG, "shard [%llu, +%llu) of %llu does not match rank %d of %d",
                (unsigned long long)e->p.shard_lo, (unsigned long long)e->p.shard_n,
                (unsigned long long)e->p.pop, rank, world);
  NLSG_HIP(hipSetDevice(e->cfg.device));
  return comm_attach(&e->comm, unique_id, world, rank, static_cast<uint64_t>(kRecHeader) + e->p.D);
}

int nlsg_de_comm_ranks(nlsg_de *e, int32_t *world_out, int32_t *rank_out) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  return comm_query(e->comm, world_out, rank_out);
}

// `turns` sharded turns without a host round trip.
// Strategy best: head k -> all-gather -> finaliser -> generation k+1, all on the engine's
// stream (nothing can overlap, so no cross-stream dependency is paid for).
// Strategy random: the generation needs the exchange only for the stop flag and writes the
// other buffers, so the whole head of turn k (summary, all-gather, finaliser) runs on the
// collective's stream beside generation k+1; generation k+1 waits for finaliser k-1 only.
// If finaliser k fires a stop test, generation k+1 is never adopted and k+2 is a no-op.
int nlsg_de_step_sharded(nlsg_de *e, uint64_t turns) {
  if (!e) return fail(NLSG_ERR_INVALID_ARG, "null engine");
  if (!e->initialised) return fail(NLSG_ERR_STATE, "nlsg_de_init has not been called");
  if (!e->comm) return fail(NLSG_ERR_STATE, "nlsg_de_comm_attach has not been called");
  if (turns == 0) return NLSG_OK;
  NLSG_HIP(hipSetDevice(e->cfg.device));
  ShardComm *c = e->comm;
  const uint64_t stride = static_cast<uint64_t>(kRecHeader) + e->p.D;
  if (e->cfg.strategy != NLSG_DE_RANDOM) {
    for (uint64_t t = 0; t < turns; t++) {
      const uint64_t k = e->k;
      launch_local_summary(e, e->rec, e->stream);
      NLSG_RCCL(rccl_api().AllGather(e->rec, c->gathered, stride, ncclDouble, c->comm, e->stream));
      hipLaunchKernelGGL(de_finalize_kernel, dim3(1), dim3(256), 0, e->stream, e->p, c->gathered,
                         c->world, stride);
      launch_generation(e, static_cast<int>(k & 1), k + 1);
      e->k = k + 1;
    }
    NLSG_HIP(launches_status());
    return NLSG_OK;
  }
  hipStream_t S = e->stream, C = c->stream;  // S is never the hipStreamLegacy handle (borrowed_stream)
  NLSG_HIP(hipEventRecord(c->pop_ready[e->k & 1], S));  // population k exists
  for (uint64_t t = 0; t < turns; t++) {
    const uint64_t k = e->k;
    NLSG_HIP(hipStreamWaitEvent(C, c->pop_ready[k & 1], 0));
    launch_local_summary(e, e->rec, C);
    NLSG_RCCL(rccl_api().AllGather(e->rec, c->gathered, stride, ncclDouble, c->comm, C));
    hipLaunchKernelGGL(de_finalize_kernel, dim3(1), dim3(256), 0, C, e->p, c->gathered, c->world,
                       stride);
    NLSG_HIP(hipEventRecord(c->head_done[k & 1], C));
    if (t > 0) NLSG_HIP(hipStreamWaitEvent(S, c->head_done[(k - 1) & 1], 0));
    launch_generation(e, static_cast<int>(k & 1), k + 1);
    NLSG_HIP(hipEventRecord(c->pop_ready[(k + 1) & 1], S));
    e->k = k + 1;
  }
  // the engine's stream ends behind the last head (status / download / the next call)
  NLSG_HIP(hipStreamWaitEvent(S, c->head_done[(e->k - 1) & 1], 0));
  NLSG_HIP(launches_status());
  return NLSG_OK;
}

}  // extern "C"
