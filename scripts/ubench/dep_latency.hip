// scripts/ubench/dep_latency.hip — what a SINGLE wave pays for dependent fp64 instructions on
// gfx950 (measurement aid, not product code): chains of dependent v_add_f64 / v_fma_f64, one, two
// and four independent chains interleaved, timed with s_memtime inside the wave. The latency-bound
// kernels (Nelder-Mead's driver wave, the Givens wave of the QR solvers, the Cholesky step) run
// exactly such chains with nothing else resident on their SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int N = 4096;

template <int CHAINS, bool FMA>
__global__ void k_chain(double *out, unsigned long long *cyc) {
  double v[CHAINS];
  for (int c = 0; c < CHAINS; c++) v[c] = 1.0 + threadIdx.x * 1e-9 + c;
  const double a = 1.0000001;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (FMA)
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[c]) : "v"(a));
      else
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(a));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CHAINS; c++) s += v[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, bool FMA>
void run(double *out, unsigned long long *cyc, int blocks) {
  k_chain<CHAINS, FMA><<<blocks, 64>>>(out, cyc);
  hipDeviceSynchronize();
  k_chain<CHAINS, FMA><<<blocks, 64>>>(out, cyc);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%s chains=%d blocks=%d: %.2f cycles per instruction, %.2f per chain link\n", FMA ? "fma" : "add",
         CHAINS, blocks, double(c) / (double(N) * CHAINS), double(c) / N);
}

int main() {
  double *out;
  unsigned long long *cyc;
  hipMalloc(&out, 8 * 64 * 8192);  // one double per thread of the largest launch
  hipMalloc(&cyc, 8);
  for (int blocks : {1, 1024, 8192}) {  // one wave on the chip; one per SIMD; eight per SIMD
    run<1, false>(out, cyc, blocks);
    run<2, false>(out, cyc, blocks);
    run<4, false>(out, cyc, blocks);
    run<1, true>(out, cyc, blocks);
    run<2, true>(out, cyc, blocks);
    run<4, true>(out, cyc, blocks);
  }
  return 0;
}
