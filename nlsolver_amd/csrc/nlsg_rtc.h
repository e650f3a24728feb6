// nlsolver_amd/csrc/nlsg_rtc.h — user objectives compiled for the device at engine creation
// (SURVEY.md §8f N3). The kernels are the SAME templates the built-in objectives use
// (nlsg_de_kernels.h, embedded as text at build time) instantiated by hiprtc around a
// user-written Objective<NLSG_OBJ_CUSTOM>; hiprtc is resolved at run time (nlsg_rtc_load).
#pragma once

#include "nlsg_common.h"

namespace nlsg {

struct DeRtcKernels {
  hipModule_t mod = nullptr;
  hipFunction_t init = nullptr, generation = nullptr, turn = nullptr;
};
struct PsoRtcKernels {
  hipModule_t mod = nullptr;
  hipFunction_t init = nullptr, move = nullptr;
};

// Compiles de_init / de_generation / de_turn kernels for the objective, CHUNKS = chunks, VEC = vec.
// group != 0: generation and fused turn are the packed kernels (`group` lanes per agent)
int rtc_build_de(const nlsg_custom_objective *obj, int chunks, bool vec, int group,
                 DeRtcKernels *out);
void rtc_release(DeRtcKernels *k);
struct BfgsRtcKernels {  // finite-difference model around the user's objective
  hipModule_t mod = nullptr;
  hipFunction_t init = nullptr, search = nullptr;
};
int rtc_build_bfgs(const nlsg_custom_objective *obj, int chunks, bool vec, BfgsRtcKernels *out);
void rtc_release(BfgsRtcKernels *k);
struct NmRtcKernels {
  hipModule_t mod = nullptr;
  hipFunction_t solve = nullptr;
};
int rtc_build_nm(const nlsg_custom_objective *obj, int chunks, bool reference_order, NmRtcKernels *out);
void rtc_release(NmRtcKernels *k);
struct LmRtcKernels {  // finite-difference model (default functors) around the user's objective
  hipModule_t mod = nullptr;
  hipFunction_t iter = nullptr;
};
// wide_chunks != 0: the evaluation kernel of the n > 64 path (a probe point of wide_chunks x 128
// coordinates per wave) instead of the one-wave iteration kernel
int rtc_build_lm(const nlsg_custom_objective *obj, int wide_chunks, bool reference_order, LmRtcKernels *out);
void rtc_release(LmRtcKernels *k);
struct HybRtcKernels {
  hipModule_t mod = nullptr;
  hipFunction_t solve = nullptr;
};
// wide_chunks != 0: the n > 128 kernel with that many 128-coordinate chunks per lane
int rtc_build_nmpso(const nlsg_custom_objective *obj, int wide_chunks, HybRtcKernels *out);
void rtc_release(HybRtcKernels *k);
struct SannRtcKernels {
  hipModule_t mod = nullptr;
  hipFunction_t anneal = nullptr;
};
// group != 0: the packed kernel (several chains per wave) with `group` lanes per chain
int rtc_build_sann(const nlsg_custom_objective *obj, int chunks, bool vec, int group,
                   SannRtcKernels *out);
void rtc_release(SannRtcKernels *k);
// pso_init / pso_move kernels; type = nlsg_pso_type.
// group != 0: the move is the packed kernel (`group` lanes per particle)
int rtc_build_pso(const nlsg_custom_objective *obj, int chunks, bool vec, int type, int group,
                  PsoRtcKernels *out);
void rtc_release(PsoRtcKernels *k);

}  // namespace nlsg
