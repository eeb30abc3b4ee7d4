// tools/r04/lds_occupancy.hip -- how many 256-thread workgroups with L KiB of LDS are resident per CU?  Every workgroup spins a fixed number of
// clock ticks; with G = 256 * k workgroups the launch takes one spin if k workgroups fit a CU and two if they do not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void __launch_bounds__(256) spin(uint32_t* out, long long ticks) {
  extern __shared__ uint4 pad[];
  if (out == nullptr) pad[threadIdx.x] = make_uint4(1, 2, 3, 4);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  uint32_t* d; CK(hipMalloc(&d, 4 * 256 * 16));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long ticks = 100000 * 2;  // 100 MHz wall clock: 2 ms
  for (int kib : {32, 31, 30, 28, 26, 24, 20, 16}) {
    printf("LDS %2d KiB per workgroup:", kib);
    for (int k = 3; k <= 10; ++k) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(spin, dim3(256 * k), dim3(256), (size_t)kib * 1024, 0, d, ticks);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  k=%d %.1f ms", k, ms);
    }
    printf("\n");
  }
  return 0;
}
