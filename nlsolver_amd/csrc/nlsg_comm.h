// nlsolver_amd/csrc/nlsg_comm.h — the per-turn exchange of a sharded population inside the
// library: an RCCL all-gather of every rank's best record, issued by the engine
// itself on a second HIP stream. A host-driven turn (Python -> torch.distributed) costs ~85 us
// of CPU per turn, more than the 50 us generation it orders; driven from here the host only
// enqueues and the collective runs beside the generation.
//
// RCCL is not linked: the process already holds one copy (PyTorch's, loaded for
// torch.distributed), and a second instance would be a second set of transports. The host
// passes that library's path to nlsg_comm_load(), which resolves the five entry points it needs.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "nlsg_common.h"

namespace nlsg {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};

RcclApi &rccl_api();  // nlsg_comm.hip

#define NLSG_RCCL(call)                                                              \
  do {                                                                               \
    ncclResult_t r_ = (call);                                                        \
    if (r_ != ncclSuccess)                                                           \
      return ::nlsg::fail(NLSG_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                          ::nlsg::rccl_api().GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

constexpr unsigned kStreamOrderEvent = hipEventDisableTiming | hipEventDisableSystemFence;

// One shard's end of the exchange.
struct ShardComm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;  // the collective's stream
  hipEvent_t pop_ready[2] = {nullptr, nullptr};  // generation k is complete (engine's stream)
  hipEvent_t head_done[2] = {nullptr, nullptr};  // head k is complete (collective's stream)
  double *gathered = nullptr;    // [world][rec_doubles]
  uint64_t rec_doubles = 0;
  int world = 0, rank = 0;
};

inline void comm_detach(ShardComm *c) {
  if (!c) return;
  if (c->comm && rccl_api().CommDestroy) rccl_api().CommDestroy(c->comm);
  for (int i = 0; i < 2; i++) {
    if (c->pop_ready[i]) hipEventDestroy(c->pop_ready[i]);
    if (c->head_done[i]) hipEventDestroy(c->head_done[i]);
  }
  if (c->stream) hipStreamDestroy(c->stream);
  hipFree(c->gathered);
  delete c;
}

// what the communicator itself says about its size and this rank (bench.py's `rccl_ranks`)
inline int comm_query(const ShardComm *c, int32_t *world, int32_t *rank) {
  if (!c || !c->comm) return fail(NLSG_ERR_STATE, "no communicator is attached");
  int w = 0, r = 0;
  NLSG_RCCL(rccl_api().CommCount(c->comm, &w));
  NLSG_RCCL(rccl_api().CommUserRank(c->comm, &r));
  if (world) *world = w;
  if (rank) *rank = r;
  return NLSG_OK;
}

// Collective call: every rank of the job attaches with the same id (nlsg_comm_unique_id on one
// rank, broadcast by the host).
inline int comm_attach(ShardComm **out, const unsigned char *id, int world, int rank,
                       uint64_t rec_doubles) {
  RcclApi &api = rccl_api();
  if (!api.lib) return fail(NLSG_ERR_STATE, "nlsg_comm_load has not been called");
  if (!id || world < 1 || rank < 0 || rank >= world)
    return fail(NLSG_ERR_INVALID_ARG, "bad communicator arguments (world %d, rank %d)", world, rank);
  ShardComm *c = new ShardComm;
  c->world = world;
  c->rank = rank;
  c->rec_doubles = rec_doubles;
  // highest priority: its own hardware queue (streams of equal priority share a small pool and
  // a queue runs its packets in order, which would put the collective in front of the
  // generation instead of beside it), and the small collective kernel is not starved by the
  // generation's grid
  int prio_low = 0, prio_high = 0;
  hipError_t he = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  if (he == hipSuccess)
    he = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high);
  // The events only order two streams of THIS device: no system-scope fence (cache write-back
  // and invalidate) when they are recorded — with it every generation kernel started ~9 us late.
  for (int i = 0; i < 2 && he == hipSuccess; i++) {
    he = hipEventCreateWithFlags(&c->pop_ready[i], kStreamOrderEvent);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&c->head_done[i], kStreamOrderEvent);
  }
  if (he == hipSuccess)
    he = hipMalloc(reinterpret_cast<void **>(&c->gathered), world * rec_doubles * sizeof(double));
  // A rank whose local resources failed still joins the collective initialisation — its peers
  // are inside ncclCommInitRank and would wait for it for ever — and reports its failure after.
  ncclUniqueId uid;
  static_assert(sizeof(uid) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(&uid, id, sizeof uid);
  const ncclResult_t r = api.CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) c->comm = nullptr;
  if (he != hipSuccess) {
    comm_detach(c);
    return fail(NLSG_ERR_HIP, "communicator resources: %s", hipGetErrorString(he));
  }
  if (r != ncclSuccess) {
    comm_detach(c);
    return fail(NLSG_ERR_HIP, "ncclCommInitRank failed: %s", api.GetErrorString(r));
  }
  *out = c;
  return NLSG_OK;
}

}  // namespace nlsg
