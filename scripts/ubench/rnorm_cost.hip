// scripts/ubench/rnorm_cost.hip — what the pieces of one normal draw cost on gfx950
// (measurement aid, not product code): each kernel runs ITER dependent-free repetitions of one
// piece per lane, 8 waves per SIMD, and reports ns per wave-instruction-group.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "nlsg_common.h"
#include "nlsg_math.h"

using namespace nlsg;

constexpr int ITER = 4096;

template <int WHAT>
__global__ __launch_bounds__(256) void piece(double *out, uint64_t seed) {
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  uint64_t k = seed + tid * kGolden;
  double acc = 0.0;
  for (int i = 0; i < ITER; i++) {
    if (WHAT == 0) {  // two counter draws -> two uniforms
      const double u1 = u01(mix64(k + kGolden * static_cast<uint64_t>(2 * i)));
      const double u2 = u01(mix64(k + kGolden * static_cast<uint64_t>(2 * i + 1)));
      acc = acc + u1 * u2;
    } else {
      // cheap uniform in (0,1) so that only the piece under test costs
      const double u = (static_cast<double>((static_cast<uint32_t>(k) + 40503u * i) | 1u)) * 0x1p-32;
      if (WHAT == 1) acc = acc + det_log(u);
      if (WHAT == 2) acc = acc + det_cos(2 * 3.141593 * u);
      if (WHAT == 3) acc = acc + sqrt(u + 1.0);
      if (WHAT == 4) acc = acc + 1.0 / (u + 1.0);
      if (WHAT == 5) acc = acc + sqrt(-2 * det_log(u)) * det_cos(2 * 3.141593 * (1.0 - u));
      if (WHAT == 6) acc = acc + u * 1.5;  // the loop and the cheap uniform alone
    }
  }
  out[tid] = acc;
}

template <int WHAT>
float run(double *out, const char *name) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256 * 8;  // 8 waves per SIMD
  piece<WHAT><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  piece<WHAT><<<blocks, 256>>>(out, 2);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double per = static_cast<double>(blocks) * 256 * ITER;
  std::printf("%-28s %8.3f ms  %7.2f ps per lane-op  (%.1f cycles per wave-op per SIMD at 2.4 GHz)\n",
              name, ms, ms * 1e9 / per, ms * 1e-3 * 2.4e9 / (ITER * 8.0));
  return ms;
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  run<6>(out, "loop + cheap uniform");
  run<0>(out, "2 x (mix64 + u01)");
  run<1>(out, "det_log");
  run<2>(out, "det_cos");
  run<3>(out, "sqrt");
  run<4>(out, "division");
  run<5>(out, "sqrt(-2 log) * cos");
  hipFree(out);
  return 0;
}
