// tools/blake_occ.hip -- BLAKE2s pair-hash throughput against resident waves per SIMD (dynamic LDS limits the workgroups per CU):
// how much of the 40 G hashes/s of the bare loop do the Merkle kernels lose to their LDS staging buffers (5 waves per SIMD)?
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc tools/blake_occ.hip -o tools/blake_occ && tools/blake_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "blake2s.cuh"
#include "fp256.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#ifndef OCC_UNROLL
#define OCC_UNROLL 1
#endif
template <int ILP>
__global__ void __launch_bounds__(256) k_blake(const fp* in, fp* out) {
  extern __shared__ uint4 pad[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (in == nullptr) pad[threadIdx.x] = make_uint4(1, 2, 3, 4);  // keep the allocation
  b2digest d[ILP];
  fp b = in[gid * 2 + 1];
#pragma unroll
  for (int j = 0; j < ILP; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) d[j].h[k] = in[gid * 2].v[k] + j;
#pragma unroll OCC_UNROLL
  for (int i = 0; i < 64 / ILP; ++i)
#pragma unroll
    for (int j = 0; j < ILP; ++j) d[j] = b2_hash_pair(d[j].h, b.v);
  fp r;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    r.v[k] = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) r.v[k] ^= d[j].h[k];
  }
  out[gid] = r;
}

template <int ILP>
void run(const fp* din, fp* dout, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto k = k_blake<ILP>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int wg_per_cu = 2; wg_per_cu <= 8; ++wg_per_cu) {
    const size_t lds = wg_per_cu == 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu) & ~(size_t)1023;  // 256 threads = 1 wave per SIMD per WG
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, din, dout);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("ILP %d, %d waves per SIMD (LDS %3zu KiB per workgroup): %.3f ms  %.1f G pair-hashes/s\n", ILP, wg_per_cu, lds / 1024, best,
           (double)blocks * 256 * 64 / best / 1e6);
  }
}

int main() {
  const int blocks = 256 * 40;
  fp *din, *dout;
  CK(hipMalloc(&din, sizeof(fp) * blocks * 256 * 2));
  CK(hipMalloc(&dout, sizeof(fp) * blocks * 256));
  CK(hipMemset(din, 0x5a, sizeof(fp) * blocks * 256 * 2));
  run<1>(din, dout, blocks);
  run<2>(din, dout, blocks);
  return 0;
}
