// nlsolver_amd/csrc/nlsg_comm.hip — RCCL entry points resolved at run time (see nlsg_comm.h).
#include <mutex>

#include "nlsg_comm.h"

namespace nlsg {
RcclApi &rccl_api() {
  static RcclApi api;
  return api;
}
}  // namespace nlsg

using namespace nlsg;

extern "C" {

int nlsg_comm_load(const char *rccl_path) {
  static std::mutex load_mutex;  // shards may be attached from several host threads
  std::lock_guard<std::mutex> hold(load_mutex);
  RcclApi &api = rccl_api();
  if (api.lib) return NLSG_OK;
  const char *path = (rccl_path && rccl_path[0]) ? rccl_path : "librccl.so";
  void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(NLSG_ERR_UNSUPPORTED, "cannot load RCCL (%s): %s", path, dlerror());
  RcclApi a;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(lib, "ncclAllGather"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(lib, "ncclCommCount"));
  a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString ||
      !a.CommCount || !a.CommUserRank) {
    dlclose(lib);
    return fail(NLSG_ERR_UNSUPPORTED, "%s does not export the RCCL entry points", path);
  }
  a.lib = lib;
  api = a;
  return NLSG_OK;
}

int nlsg_comm_unique_id(unsigned char *id_out) {
  if (!id_out) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  RcclApi &api = rccl_api();
  if (!api.lib) return fail(NLSG_ERR_STATE, "nlsg_comm_load has not been called");
  ncclUniqueId uid;
  NLSG_RCCL(api.GetUniqueId(&uid));
  std::memcpy(id_out, &uid, sizeof uid);
  return NLSG_OK;
}

}  // extern "C"
