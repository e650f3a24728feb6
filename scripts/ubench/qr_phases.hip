// scripts/ubench/qr_phases.hip — what the two sides of the LM QR step cost (measurement aid, not
// product code): lm_solve_qr<512> on synthetic LDS images, 1 .. 4 workgroups per CU (the launch
// is sized to the CU count times the residency), whole / Givens side only / apply side only.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../nlsolver_amd/csrc/nlsg_common.h"
#include "../../nlsolver_amd/csrc/nlsg_lm_kernels.h"

using namespace nlsg;

template <int PROBE>
__global__ __launch_bounds__(512) void k_qr(double *out, int n, int pad_bytes) {
  extern __shared__ __align__(16) unsigned char smem[];
  LmQrShared &qs = *reinterpret_cast<LmQrShared *>(smem);
  const int t = threadIdx.x;
  for (int e = t; e < 64 * 64; e += 512) {
    const int i = e >> 6, j = e & 63;
    const double v = 1.0 / (1.0 + (i > j ? i - j : j - i)) + (i == j ? 10.0 : 0.0) + 1e-3 * (blockIdx.x & 7);
    qs.R[i * kLmQrStride + j] = v;
  }
  if (t < 64) qs.R[t * kLmQrStride + 64] = 1.0 + 0.01 * t;
  __syncthreads();
  lm_solve_qr<512, PROBE>(qs, n);
  if (t < n) out[blockIdx.x * 64 + t] = qs.upd[t];
  (void)pad_bytes;
}

template <int PROBE>
float run(double *out, int blocks, int lds) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_qr<PROBE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k_qr<PROBE><<<blocks, 512, lds>>>(out, 64, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) k_qr<PROBE><<<blocks, 512, lds>>>(out, 64, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

int main(int argc, char **argv) {
  double *out;
  hipMalloc(&out, 8192 * 64 * 8);
  const int base = static_cast<int>(sizeof(LmQrShared));
  if (argc > 1) {  // under rocprofv3 --pmc: the three variants at the product's launch shape
    printf("whole %.1f, Givens side only %.1f, apply side only %.1f us\n", run<0>(out, 8192, base),
           run<1>(out, 8192, base), run<2>(out, 8192, base));
    return 0;
  }
  // residency r per CU is forced through the LDS request: 160 KB / r
  for (int r : {1, 2, 3, 4}) {
    const int lds = r == 4 ? base : (160 * 1024) / r - 1024;
    const int blocks = 256 * r;  // one round
    printf("resident %d/CU (%d blocks, one round): whole %.1f us, Givens side only %.1f us, apply side only %.1f us\n", r,
           blocks, run<0>(out, blocks, lds), run<1>(out, blocks, lds), run<2>(out, blocks, lds));
  }
  printf("8192 problems at 4/CU: whole %.1f us, Givens side only %.1f us, apply side only %.1f us\n",
         run<0>(out, 8192, base), run<1>(out, 8192, base), run<2>(out, 8192, base));
  return 0;
}
