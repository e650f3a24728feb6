// scripts/ubench/wave_placement.hip — where the waves of co-resident workgroups land (measurement
// aid, not product code). Launch shape of lm_qr_step_kernel<512>: 512 threads, 34 KiB of dynamic
// LDS, four workgroups per CU. Every wave records HW_ID (wave slot, SIMD, CU, SE) and XCC_ID; the
// host prints, per CU, which SIMD each workgroup's wave 7 (the Givens wave) sits on.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) void k_place(unsigned *out, int spin) {
  extern __shared__ unsigned char smem[];
  const int wid = threadIdx.x >> 6;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + wid) * 2] = hw;
    out[(blockIdx.x * 8 + wid) * 2 + 1] = xcc;
  }
  // stay resident so that the CU fills up with four workgroups
  volatile unsigned char *s = smem;
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < static_cast<unsigned long long>(spin)) s[threadIdx.x] = 1;
  __syncthreads();
}

int main() {
  const int blocks = 8192;
  unsigned *out;
  hipMalloc(&out, blocks * 8 * 2 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_place), hipFuncAttributeMaxDynamicSharedMemorySize, 34 * 1024);
  k_place<<<blocks, 512, 34 * 1024>>>(out, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 8 * 2);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  // (xcc, se, cu) -> list of (block, simd of wave 7, simds of waves 0..7)
  std::map<unsigned, std::vector<int>> per_cu;
  long same_simd_pairs = 0, pairs = 0;
  int hist[4] = {0, 0, 0, 0};
  for (int b = 0; b < blocks; b++) {
    const unsigned hw = h[(b * 8 + 7) * 2], xcc = h[(b * 8 + 7) * 2 + 1] & 0xf;
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7;
    hist[simd]++;
    per_cu[(xcc << 8) | (se << 4) | cu].push_back(static_cast<int>(simd) | (b << 2));
  }
  printf("wave 7's SIMD over %d workgroups: %d %d %d %d\n", blocks, hist[0], hist[1], hist[2], hist[3]);
  for (int b = 0; b < 4; b++) {
    printf("block %d: waves on SIMDs", b);
    for (int w = 0; w < 8; w++) printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
    printf("  wave slots");
    for (int w = 0; w < 8; w++) printf(" %u", h[(b * 8 + w) * 2] & 15);
    printf("  tg_id %u cu %u se %u xcc %u\n", (h[b * 16] >> 16) & 15, (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7,
           h[b * 16 + 1] & 15);
  }
  int shown = 0;
  for (auto &kv : per_cu) {
    if (shown++ < 6) {
      printf("cu key %03x: %zu workgroups, (block:simd of wave 7)", kv.first, kv.second.size());
      for (size_t i = 0; i < kv.second.size() && i < 12; i++) printf(" %d:%d", kv.second[i] >> 2, kv.second[i] & 3);
      printf("\n");
    }
  }
  printf("distinct CUs seen: %zu\n", per_cu.size());
  return 0;
}
