// EXPERIMENT: do matrix-core butterfly blocks and dense VALU code share a SIMD without loss?  One workgroup of 1024 threads per
// CU = 4 waves per SIMD; waves 0-3 and 8-11 play role A, waves 4-7 and 12-15 role B (wave w sits on SIMD w % 4, so every SIMD
// holds two waves of each role).  Roles: M = the four-butterfly MFMA register group (group4.inc), V = a dense VALU loop of
// v_mad_u64_u32 in four independent chains (what the VALU butterflies issue).  Modes MM / VV / MV: ticks per iteration and role.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef int shk_v16i __attribute__((ext_vector_type(16)));
#include "group4.inc"

#define VALU_PER_ITER 400
__device__ __forceinline__ void valu_iter(uint64_t (&acc)[4], uint32_t a, uint32_t b) {
#pragma unroll
  for (int i = 0; i < VALU_PER_ITER / 4; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
  }
}

template <int MODE>  // 0 = all M, 1 = all V, 2 = mixed
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k(const shk_x8* in, shk_x8* out, const void* table,
                                                                                      unsigned long long* cyc, int iters) {
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool role_m = MODE == 0 || (MODE == 2 && ((wave >> 2) & 1) == 0);
  const size_t idx = (size_t)(blockIdx.x * 1024 + threadIdx.x) * 4;
  unsigned long long t0, t1;
  if (role_m) {
    shk_x8 x[4];
    for (int m = 0; m < 4; ++m) x[m] = in[idx + m];
    shk_v16i offs;
    for (int r = 0; r < 16; ++r) offs[r] = (1 << 20) + ((threadIdx.x & 32) ? 239 : 17 + r);
    const uint32_t mlo = (uint32_t)reinterpret_cast<uintptr_t>(table), mhi = (uint32_t)(reinterpret_cast<uintptr_t>(table) >> 32);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) shk_group4_asm(x, offs, lane * 16u, mlo, mhi, 2u * (wave & 3));
    t1 = __builtin_amdgcn_s_memtime();
    for (int m = 0; m < 4; ++m) out[idx + m] = x[m];
  } else {
    uint64_t acc[4] = {lane, lane + 1, lane + 2, lane + 3};
    const uint32_t a = in[idx][0], b = in[idx][1];
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) valu_iter(acc, a, b);
    t1 = __builtin_amdgcn_s_memtime();
    shk_x8 r = {(uint32_t)acc[0], (uint32_t)acc[1], (uint32_t)acc[2], (uint32_t)acc[3], 0, 0, 0, 0};
    out[idx] = r;
  }
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, const shk_x8* in, shk_x8* out, const void* table, unsigned long long* cyc, int iters) {
  const int blocks = 256;
  std::vector<unsigned long long> h(16 * blocks);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, in, out, table, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, 8 * 16 * blocks, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> m, v;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 16; ++w) {
      const bool role_m = MODE == 0 || (MODE == 2 && ((w >> 2) & 1) == 0);
      (role_m ? m : v).push_back(h[b * 16 + w]);
    }
  auto med = [&](std::vector<unsigned long long>& a) { std::sort(a.begin(), a.end()); return a.empty() ? 0.0 : (double)a[a.size() / 2] / iters; };
  printf("%-28s", name);
  if (!m.empty()) printf("  M waves: %7.0f ticks per group of 4 butterflies", med(m));
  if (!v.empty()) printf("  V waves: %7.0f ticks per %d v_mad_u64_u32 (%.2f per instruction and wave)", med(v), VALU_PER_ITER, med(v) / VALU_PER_ITER);
  printf("\n");
}

int main() {
  const size_t n = (size_t)256 * 1024 * 4;
  shk_x8 *in, *out; void* table; unsigned long long* cyc;
  hipMalloc(&in, n * 32); hipMalloc(&out, n * 32);
  hipMemset(in, 0x5a, n * 32);
  hipMalloc(&table, 1 << 20); hipMemset(table, 1, 1 << 20);
  hipMalloc(&cyc, 8 * 16 * 256);
  const int iters = 200;
  run<0>("all 16 waves M", in, out, table, cyc, iters);
  run<1>("all 16 waves V", in, out, table, cyc, iters);
  run<2>("8 waves M + 8 waves V", in, out, table, cyc, iters);
  printf("(work-conserving sharing would give, per SIMD: 2 x [M alone / 4] + 2 x [V alone / 4] of issue time per iteration pair)\n");
  return 0;
}
