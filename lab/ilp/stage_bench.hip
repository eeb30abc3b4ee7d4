// Times the generated butterfly stages (mfma_bfly.inc) in isolation: registers only, the operand-image table resident in
// L2, 1 and 2 waves per SIMD.  cycles per stage execution, median over waves (s_memtime).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef int shk_v16i __attribute__((ext_vector_type(16)));
#include "mfma_bfly.inc"

template <int STAGE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k(const shk_x8* in, shk_x8* out, const void* table, unsigned long long* cyc, int iters) {
  shk_x8 x[16];
  for (int m = 0; m < 16; ++m) x[m] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + m];
  shk_v16i offs;
  for (int r = 0; r < 16; ++r) offs[r] = (threadIdx.x & 32) ? SHK_OFFS[1][r] : SHK_OFFS[0][r];
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t mlo = (uint32_t)reinterpret_cast<uintptr_t>(table), mhi = (uint32_t)(reinterpret_cast<uintptr_t>(table) >> 32);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (STAGE == 1) shk_stage1_asm_7(x, offs, lane * 16u, mlo, mhi, 2u * wave);
    else shk_stage2_asm_7(x, offs, lane * 16u, mlo, mhi);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int m = 0; m < 16; ++m) out[(blockIdx.x * 256 + threadIdx.x) * 16 + m] = x[m];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  const int maxb = 1024;
  shk_x8 *in, *out; void* table; unsigned long long* cyc;
  hipMalloc(&in, (size_t)maxb * 256 * 16 * 32); hipMalloc(&out, (size_t)maxb * 256 * 16 * 32);
  hipMemset(in, 0x5a, (size_t)maxb * 256 * 16 * 32);
  hipMalloc(&table, 4096 * 256); hipMemset(table, 1, 4096 * 256);
  hipMalloc(&cyc, 8 * 4 * maxb);
  std::vector<unsigned long long> h(4 * maxb);
  const int iters = 20;
  for (int stage = 1; stage <= 2; ++stage)
    for (int w = 1; w <= 2; ++w) {
      const int blocks = 256 * w;
      for (int rep = 0; rep < 2; ++rep) {
        if (stage == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, table, cyc, iters);
        else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, in, out, table, cyc, iters);
      }
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, 8 * 4 * blocks, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.begin() + 4 * blocks);
      printf("stage %d (radix 2^7), %d wave(s) per SIMD: %.0f cycles per stage execution (median wave)\n", stage, w, (double)h[2 * blocks] / iters);
    }
  return 0;
}
