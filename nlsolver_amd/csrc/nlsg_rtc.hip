// nlsolver_amd/csrc/nlsg_rtc.hip — run-time compilation of user objectives (see nlsg_rtc.h).
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <mutex>
#include <string>
#include <vector>

#include "nlsg_rtc.h"

#include "build/nlsg_embedded_sources.inc"

namespace nlsg {
namespace {

struct RtcApi {
  void *lib = nullptr;
  decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
  decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
  decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
  decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
  decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
  decltype(&hiprtcGetCode) GetCode = nullptr;
  decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
  decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
  decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
};

RtcApi &rtc_api() {
  static RtcApi api;
  return api;
}

// engines may be created from several host threads at once: the table is filled under a lock
// and published whole (api.lib is set last, by the assignment of the finished copy)
std::mutex &rtc_mutex() {
  static std::mutex m;
  return m;
}

int rtc_load(const char *path_in) {
  std::lock_guard<std::mutex> hold(rtc_mutex());
  RtcApi &api = rtc_api();
  if (api.lib) return NLSG_OK;
  const char *path = (path_in && path_in[0]) ? path_in : "libhiprtc.so";
  void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(NLSG_ERR_UNSUPPORTED, "cannot load hiprtc (%s): %s", path, dlerror());
  RtcApi a;
#define NLSG_RTC_SYM(field, name)                                            \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, #name));          \
  if (!a.field) {                                                            \
    dlclose(lib);                                                            \
    return fail(NLSG_ERR_UNSUPPORTED, "%s does not export " #name, path);    \
  }
  NLSG_RTC_SYM(CreateProgram, hiprtcCreateProgram)
  NLSG_RTC_SYM(CompileProgram, hiprtcCompileProgram)
  NLSG_RTC_SYM(GetProgramLogSize, hiprtcGetProgramLogSize)
  NLSG_RTC_SYM(GetProgramLog, hiprtcGetProgramLog)
  NLSG_RTC_SYM(GetCodeSize, hiprtcGetCodeSize)
  NLSG_RTC_SYM(GetCode, hiprtcGetCode)
  NLSG_RTC_SYM(DestroyProgram, hiprtcDestroyProgram)
  NLSG_RTC_SYM(AddNameExpression, hiprtcAddNameExpression)
  NLSG_RTC_SYM(GetLoweredName, hiprtcGetLoweredName)
#undef NLSG_RTC_SYM
  a.lib = lib;
  api = a;
  return NLSG_OK;
}

}  // namespace

// Compiles `kernel_header` around the user's Objective<NLSG_OBJ_CUSTOM> and returns the module
// with one function per name expression (kernel template-ids).
static int rtc_compile(const nlsg_custom_objective *obj, const char *kernel_header,
                       const std::vector<std::string> &name_exprs, hipModule_t *mod_out,
                       std::vector<hipFunction_t> *fns_out) {
  if (!obj || !obj->term_body || !obj->term_body[0])
    return fail(NLSG_ERR_INVALID_ARG, "a custom objective needs a term body");
  int rc = rtc_load(nullptr);
  if (rc) return rc;
  RtcApi &api = rtc_api();
  std::string src = std::string("#include \"") + kernel_header + "\"\n"
                    "namespace nlsg {\n"
                    "template <>\n"
                    "struct Objective<NLSG_OBJ_CUSTOM> {\n";
  if (obj->chain == NLSG_CUSTOM_VECTOR) {
    // whole-vector form: term_body is the body of  double f(const X &x, uint64_t D)
    src += "  static constexpr bool kChain = false;\n"
           "  static constexpr bool kWhole = true;\n"
           "  __device__ static inline double term(double, double) { return 0.0; }\n"
           "  __device__ static inline uint64_t n_terms(uint64_t) { return 0; }\n"
           "  __device__ static inline double finish(double s, uint64_t) { return s; }\n"
           "  template <typename X>\n"
           "  __device__ static inline double whole(const X &x, uint64_t D) {\n    (void)D;\n#line 1 "
           "\"vector_body\"\n";
    src += obj->term_body;
    src += "\n  }\n};\n}  // namespace nlsg\n";
  } else {
    src += "  static constexpr bool kWhole = false;\n"
           "  static constexpr bool kChain = ";
    src += obj->chain ? "true" : "false";
    src += ";\n  __device__ static inline double term(double xi, double xn) {\n    (void)xn;\n#line 1 "
           "\"term_body\"\n";
    src += obj->term_body;
    src += "\n  }\n"
           "  __device__ static inline uint64_t n_terms(uint64_t D) { return kChain ? (D ? D - 1 : 0) : D; }\n"
           "  __device__ static inline double finish(double s, uint64_t D) {\n    (void)D;\n#line 1 "
           "\"finish_body\"\n";
    src += (obj->finish_body && obj->finish_body[0]) ? obj->finish_body : "return s;";
    src += "\n  }\n"
           "  template <typename X>\n"
           "  __device__ static inline double whole(const X &, uint64_t) { return 0.0; }\n"
           "};\n}  // namespace nlsg\n";
  }

  const int nh = static_cast<int>(sizeof(kEmbedded) / sizeof(kEmbedded[0]));
  std::vector<const char *> names(nh), texts(nh);
  for (int i = 0; i < nh; i++) {
    names[i] = kEmbedded[i].name;
    texts[i] = kEmbedded[i].text;
  }
  hiprtcProgram prog = nullptr;
  if (api.CreateProgram(&prog, src.c_str(), "nlsg_custom_objective.hip", nh, texts.data(),
                        names.data()) != HIPRTC_SUCCESS)
    return fail(NLSG_ERR_HIP, "hiprtcCreateProgram failed");
  for (const std::string &n : name_exprs)
    if (api.AddNameExpression(prog, n.c_str()) != HIPRTC_SUCCESS) {
      api.DestroyProgram(&prog);
      return fail(NLSG_ERR_HIP, "hiprtcAddNameExpression(%s) failed", n.c_str());
    }
  // the flags of csrc/Makefile: device arithmetic must stay bit-reproducible
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-fast-math"};
  const hiprtcResult cr = api.CompileProgram(prog, 5, opts);
  if (cr != HIPRTC_SUCCESS) {
    size_t ls = 0;
    api.GetProgramLogSize(prog, &ls);
    std::string log(ls ? ls : 1, '\0');
    if (ls) api.GetProgramLog(prog, &log[0]);
    api.DestroyProgram(&prog);
    if (log.size() > 400) log.resize(400);
    return fail(NLSG_ERR_INVALID_ARG, "custom objective does not compile: %s", log.c_str());
  }
  size_t cs = 0;
  if (api.GetCodeSize(prog, &cs) != HIPRTC_SUCCESS || cs == 0) {
    api.DestroyProgram(&prog);
    return fail(NLSG_ERR_HIP, "hiprtcGetCodeSize failed");
  }
  std::vector<char> code(cs);
  if (api.GetCode(prog, code.data()) != HIPRTC_SUCCESS) {
    api.DestroyProgram(&prog);
    return fail(NLSG_ERR_HIP, "hiprtcGetCode failed");
  }
  hipModule_t mod = nullptr;
  hipError_t he = hipModuleLoadData(&mod, code.data());
  std::vector<hipFunction_t> fns(name_exprs.size(), nullptr);
  for (size_t i = 0; i < name_exprs.size() && he == hipSuccess; i++) {
    const char *lowered = nullptr;  // lives in the program: look the function up before destroying it
    if (api.GetLoweredName(prog, name_exprs[i].c_str(), &lowered) != HIPRTC_SUCCESS) {
      he = hipErrorNotFound;
      break;
    }
    he = hipModuleGetFunction(&fns[i], mod, lowered);
  }
  api.DestroyProgram(&prog);
  if (he != hipSuccess) {
    if (mod) hipModuleUnload(mod);
    return fail(NLSG_ERR_HIP, "loading the compiled objective failed: %s", hipGetErrorString(he));
  }
  *mod_out = mod;
  *fns_out = fns;
  return NLSG_OK;
}

static std::string targs(int chunks, bool vec) {
  return std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) + ", " + std::to_string(chunks) + ", " +
         (vec ? "true" : "false");
}

int rtc_build_de(const nlsg_custom_objective *obj, int chunks, bool vec, int group,
                 DeRtcKernels *out) {
  if (chunks == 0) {  // D > 1024: the segment-streaming kernels (no fused turn)
    const std::string t = std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) + ", " + (vec ? "true" : "false");
    std::vector<hipFunction_t> f;
    DeRtcKernels k;
    const int rc = rtc_compile(obj, "nlsg_de_kernels.h",
                               {"nlsg::de_init_long_kernel<" + t + ">", "nlsg::de_generation_long_kernel<" + t + ">"},
                               &k.mod, &f);
    if (rc) return rc;
    k.init = f[0];
    k.generation = f[1];
    k.turn = nullptr;
    *out = k;
    return NLSG_OK;
  }
  const std::string t = targs(chunks, vec);
  const std::string gen =
      group ? "nlsg::de_generation_groups_kernel<" + std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) +
                  ", " + std::to_string(group) + ">"
            : "nlsg::de_generation_kernel<" + t + ">";
  std::vector<hipFunction_t> f;
  DeRtcKernels k;
  const std::string turn =
      group ? "nlsg::de_turn_groups_kernel<" + std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) + ", " +
                  std::to_string(group) + ">"
            : "nlsg::de_turn_kernel<" + t + ">";
  const int rc = rtc_compile(obj, "nlsg_de_kernels.h", {"nlsg::de_init_kernel<" + t + ">", gen, turn},
                             &k.mod, &f);
  if (rc) return rc;
  k.init = f[0];
  k.generation = f[1];
  k.turn = f[2];
  *out = k;
  return NLSG_OK;
}

int rtc_build_pso(const nlsg_custom_objective *obj, int chunks, bool vec, int type, int group,
                  PsoRtcKernels *out) {
  if (chunks == 0) {  // D > 1024: the segment-streaming kernels
    const std::string t = std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) + ", " + (vec ? "true" : "false");
    std::vector<hipFunction_t> f;
    PsoRtcKernels k;
    const int rc = rtc_compile(obj, "nlsg_pso_kernels.h",
                               {"nlsg::pso_init_long_kernel<" + t + ">",
                                "nlsg::pso_move_long_kernel<" + t + ", " + std::to_string(type) + ">"},
                               &k.mod, &f);
    if (rc) return rc;
    k.init = f[0];
    k.move = f[1];
    *out = k;
    return NLSG_OK;
  }
  const std::string t = targs(chunks, vec);
  const std::string move =
      group ? "nlsg::pso_move_groups_kernel<" + std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) +
                  ", " + std::to_string(group) + ", " + std::to_string(type) + ">"
            : "nlsg::pso_move_kernel<" + t + ", " + std::to_string(type) + ">";
  std::vector<hipFunction_t> f;
  PsoRtcKernels k;
  const int rc = rtc_compile(obj, "nlsg_pso_kernels.h", {"nlsg::pso_init_kernel<" + t + ">", move},
                             &k.mod, &f);
  if (rc) return rc;
  k.init = f[0];
  k.move = f[1];
  *out = k;
  return NLSG_OK;
}

int rtc_build_bfgs(const nlsg_custom_objective *obj, int chunks, bool vec, BfgsRtcKernels *out) {
  // <CHUNKS, VEC, MODEL>: MODEL = the objective id selects the finite-difference BfgsModel
  const std::string t = std::to_string(chunks) + ", " + (vec ? "true" : "false") + ", " +
                        std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM));
  std::vector<hipFunction_t> f;
  BfgsRtcKernels k;
  const int rc = rtc_compile(obj, "nlsg_bfgs_kernels.h",
                             {"nlsg::bfgs_init_kernel<" + t + ">", "nlsg::bfgs_search_kernel<" + t + ">"},
                             &k.mod, &f);
  if (rc) return rc;
  k.init = f[0];
  k.search = f[1];
  *out = k;
  return NLSG_OK;
}
void rtc_release(BfgsRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = BfgsRtcKernels();
}

int rtc_build_nm(const nlsg_custom_objective *obj, int chunks, bool reference_order, NmRtcKernels *out) {
  std::vector<hipFunction_t> f;
  NmRtcKernels k;
  const std::string id = std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM));
  // chunks == 0: the driver-wave kernel (n <= 128)
  const int rc = rtc_compile(obj, "nlsg_nm_kernels.h",
                             {chunks == 0 ? "nlsg::nm_solve_driver_kernel<" + id + (reference_order ? ", true>" : ">")
                                          : "nlsg::nm_solve_kernel<" + id + ", " + std::to_string(chunks) + ">"},
                             &k.mod, &f);
  if (rc) return rc;
  k.solve = f[0];
  *out = k;
  return NLSG_OK;
}
void rtc_release(NmRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = NmRtcKernels();
}

int rtc_build_nmpso(const nlsg_custom_objective *obj, int wide_chunks, HybRtcKernels *out) {
  std::vector<hipFunction_t> f;
  HybRtcKernels k;
  const std::string id = std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM));
  const std::string name = wide_chunks ? "nlsg::nmpso_solve_wide_kernel<" + id + ", " +
                                             std::to_string(wide_chunks) + ">"
                                       : "nlsg::nmpso_solve_kernel<" + id + ">";
  const int rc = rtc_compile(obj, "nlsg_nmpso_kernels.h", {name}, &k.mod, &f);
  if (rc) return rc;
  k.solve = f[0];
  *out = k;
  return NLSG_OK;
}
void rtc_release(HybRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = HybRtcKernels();
}

int rtc_build_sann(const nlsg_custom_objective *obj, int chunks, bool vec, int group,
                   SannRtcKernels *out) {
  std::vector<hipFunction_t> f;
  SannRtcKernels k;
  const std::string name =
      chunks == 0 ? "nlsg::sann_anneal_long_kernel<" + std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) +
                        ", " + (vec ? "true" : "false") + ">"
      : group ? "nlsg::sann_anneal_groups_kernel<" + std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM)) +
                  ", " + std::to_string(group) + ">"
            : "nlsg::sann_anneal_kernel<" + targs(chunks, vec) + ">";
  const int rc = rtc_compile(obj, "nlsg_sann_kernels.h", {name}, &k.mod, &f);
  if (rc) return rc;
  k.anneal = f[0];
  *out = k;
  return NLSG_OK;
}
void rtc_release(SannRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = SannRtcKernels();
}

int rtc_build_lm(const nlsg_custom_objective *obj, int wide_chunks, bool reference_order, LmRtcKernels *out) {
  std::vector<hipFunction_t> f;
  LmRtcKernels k;
  const std::string id = std::to_string(static_cast<int>(NLSG_OBJ_CUSTOM));
  const std::string name =
      wide_chunks ? (reference_order ? "nlsg::lm_wide_fd_lanes_kernel<" + id + ">"
                                     : "nlsg::lm_wide_fd_eval_kernel<" + id + ", " + std::to_string(wide_chunks) + ">")
                  : "nlsg::lm_fd_iter_kernel<" + id + (reference_order ? ", true>" : ">");
  const int rc = rtc_compile(obj, "nlsg_lm_kernels.h", {name}, &k.mod, &f);
  if (rc) return rc;
  k.iter = f[0];
  *out = k;
  return NLSG_OK;
}
void rtc_release(LmRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = LmRtcKernels();
}

void rtc_release(DeRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = DeRtcKernels();
}
void rtc_release(PsoRtcKernels *k) {
  if (k && k->mod) hipModuleUnload(k->mod);
  if (k) *k = PsoRtcKernels();
}

}  // namespace nlsg

extern "C" int nlsg_rtc_load(const char *hiprtc_path) { return nlsg::rtc_load(hiprtc_path); }
