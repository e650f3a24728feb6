// scripts/ubench/mfma_f64_peak.hip — what the fp64 matrix pipe of gfx950 sustains (measurement
// aid, not product code): back-to-back v_mfma_f64_16x16x4_f64 on independent accumulators, and
// the same with fp64 vector fmas interleaved (do the two share the pipe?), at 1, 2, 4 and 8 waves
// per SIMD, against the 78.6 TFLOP/s the data sheet gives at 2.4 GHz.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

template <int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void k_mfma(double *out) {
  const int t = threadIdx.x;
  v4d acc[8];
  for (int i = 0; i < 8; i++) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + t * 1e-6, b = 1.0 - t * 1e-6;
  double v[4] = {1.0, 1.1, 1.2, 1.3};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < VALU_PER_MFMA; q++)
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[q & 3]) : "v"(a));
    }
  }
  double s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + t] = s;
}

template <int V>
void run(double *out, int waves_per_simd) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256 * waves_per_simd;
  k_mfma<V><<<blocks, 256>>>(out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_mfma<V><<<blocks, 256>>>(out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double mfma = double(blocks) * 4 * ITER * 8;           // wave-level instructions
  const double tf = mfma * 2048.0 / (ms * 1e-3) * 1e-12;       // 16 x 16 x 4 x 2 flop each
  const double vtf = mfma * V * 128.0 / (ms * 1e-3) * 1e-12;   // 64 lanes x 2 flop
  std::printf("%d fp64 vector fma per MFMA, %d waves/SIMD: %7.3f ms  matrix %6.2f TFLOP/s + vector %6.2f "
              "TFLOP/s = %6.2f; %5.1f cycles per MFMA slot per SIMD at 2.4 GHz\n",
              V, waves_per_simd, ms, tf, vtf, tf + vtf,
              ms * 1e-3 * 2.4e9 / (double(ITER) * 8 * waves_per_simd));
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  for (int w : {1, 2, 4, 8}) run<0>(out, w);
  for (int w : {2, 8}) run<4>(out, w);
  for (int w : {2, 8}) run<16>(out, w);
  hipFree(out);
  return 0;
}
