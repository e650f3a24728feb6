// nlsolver_amd/csrc/nlsg_probe.hip — the device's deterministic math primitives on caller-chosen
// arguments (C-ABI nlsg_probe_math): what the engines' bit-for-bit agreement with the CPU
// restatement rests on, checkable directly on millions of inputs instead of through solver runs.
#include <vector>

#include "nlsg_common.h"
#include "nlsg_math.h"

using namespace nlsg;

namespace {

__global__ __launch_bounds__(256) void probe_kernel(int fn, const uint64_t *in, uint64_t *out,
                                                    uint64_t n) {
  __shared__ double rn_tab[kRnormTabDoubles];
  rnorm_table_to_lds(rn_tab);
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t b = in[i];
  const double x = __longlong_as_double(static_cast<long long>(b));
  double r;
  switch (fn) {
    case NLSG_PROBE_LOG: r = det_log(x); break;
    case NLSG_PROBE_COS: r = det_cos(x); break;
    case NLSG_PROBE_EXP: r = det_exp(x); break;
    case NLSG_PROBE_TANH: r = det_tanh(x); break;
    case NLSG_PROBE_COS_2PI: r = det_cos_2pi(x); break;
    case NLSG_PROBE_U01: r = u01(b); break;
    default: r = det_rnorm(b, rn_tab); break;  // NLSG_PROBE_RNORM: the input is the 64-bit draw
  }
  out[i] = static_cast<uint64_t>(__double_as_longlong(r));
}

}  // namespace

extern "C" int nlsg_probe_math(int32_t fn, const uint64_t *in_host, uint64_t *out_host, uint64_t n,
                               int32_t device) {
  if (!in_host || !out_host) return fail(NLSG_ERR_INVALID_ARG, "null argument");
  if (fn < NLSG_PROBE_LOG || fn > NLSG_PROBE_RNORM)
    return fail(NLSG_ERR_INVALID_ARG, "unknown probe function %d", fn);
  if (n == 0) return NLSG_OK;
  if (n > (1ull << 31)) return fail(NLSG_ERR_UNSUPPORTED, "more than 2^31 arguments per call");
  const int rc = check_device(device);
  if (rc) return rc;
  NLSG_HIP(hipSetDevice(device));
  uint64_t *in = nullptr, *out = nullptr;
  hipError_t he = hipMalloc(reinterpret_cast<void **>(&in), n * 8);
  if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&out), n * 8);
  if (he == hipSuccess) he = hipMemcpy(in, in_host, n * 8, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(probe_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       nullptr, static_cast<int>(fn), in, out, n);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipMemcpy(out_host, out, n * 8, hipMemcpyDeviceToHost);
  hipFree(in);
  hipFree(out);
  if (he != hipSuccess)
    return fail(he == hipErrorOutOfMemory ? NLSG_ERR_OOM : NLSG_ERR_HIP, "probe failed: %s",
                hipGetErrorString(he));
  return NLSG_OK;
}
