// EXPERIMENT: four MFMA butterflies on four register-resident elements (the register group of an LDS-resident tile), timed at 1,
// 2 and 4 waves per SIMD -- what the matrix-core butterflies would cost at the occupancy of the VALU passes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef int shk_v16i __attribute__((ext_vector_type(16)));
#include "group4.inc"

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k(const shk_x8* in, shk_x8* out, const void* table, unsigned long long* cyc, int iters) {
  shk_x8 x[4];
  for (int m = 0; m < 4; ++m) x[m] = in[(blockIdx.x * 256 + threadIdx.x) * 4 + m];
  shk_v16i offs;
  for (int r = 0; r < 16; ++r) offs[r] = (1 << 20) + ((threadIdx.x & 32) ? 239 : 17 + r);
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t mlo = (uint32_t)reinterpret_cast<uintptr_t>(table), mhi = (uint32_t)(reinterpret_cast<uintptr_t>(table) >> 32);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) shk_group4_asm(x, offs, lane * 16u, mlo, mhi, 2u * wave);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int m = 0; m < 4; ++m) out[(blockIdx.x * 256 + threadIdx.x) * 4 + m] = x[m];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  const int maxb = 2048;
  shk_x8 *in, *out; void* table; unsigned long long* cyc;
  hipMalloc(&in, (size_t)maxb * 256 * 4 * 32); hipMalloc(&out, (size_t)maxb * 256 * 4 * 32);
  hipMemset(in, 0x5a, (size_t)maxb * 256 * 4 * 32);
  hipMalloc(&table, 1 << 20); hipMemset(table, 1, 1 << 20);
  hipMalloc(&cyc, 8 * 4 * maxb);
  std::vector<unsigned long long> h(4 * maxb);
  const int iters = 200;
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(k), 256, 0);
  printf("occupancy API: %d workgroups of 256 threads per CU\n", occ);
  for (int w = 1; w <= 8; w *= 2) {
    const int blocks = 256 * w;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, table, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 8 * 4 * blocks, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + 4 * blocks);
    const double per = (double)h[2 * blocks] / iters;
    printf("%d workgroup(s) per CU (= waves per SIMD if all resident): %.0f cycles per group of 4 butterflies per wave, %.1f per butterfly of SIMD time\n",
           w, per, per / 4 / (w > 4 ? 4 : w));
  }
  return 0;
}
