// tools/r04/blake_x2.hip -- bare pair-hash throughput with TWO compressions in lock-step in one asm block (gen_b2x2.py), checked
// against the C++ compression.   hipcc -O3 --offload-arch=gfx950 -DB2_NO_ASM -DB2X2_INC='"/tmp/b2x2_add3e64.inc"' -I starks_amd/csrc ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "blake2s.cuh"
#include B2X2_INC
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// two independent single-block hashes of (l0 || r), (l1 || r)
__device__ __forceinline__ void hash_pair_x2(uint32_t (&d0)[8], uint32_t (&d1)[8], const uint32_t (&r)[8]) {
  b2x16 va, vb, ma, mb;
  uint32_t h[8];
  b2_init(h);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    va[i] = h[i]; vb[i] = h[i];
    ma[i] = d0[i]; mb[i] = d1[i];
    ma[8 + i] = r[i]; mb[8 + i] = r[i];
  }
  const uint32_t iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu ^ 64u, 0x9B05688Cu, ~0x1F83D9ABu, 0x5BE0CD19u};
#pragma unroll
  for (int i = 0; i < 8; ++i) { va[8 + i] = iv[i]; vb[8 + i] = iv[i]; }
  b2_rounds_x2(va, vb, ma, mb);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d0[i] = h[i] ^ va[i] ^ va[8 + i];
    d1[i] = h[i] ^ vb[i] ^ vb[8 + i];
  }
}

template <bool X2>
__global__ void __launch_bounds__(256) k_blake(const uint32_t* in, uint32_t* out, int iters) {
  extern __shared__ uint4 pad[];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (in == nullptr) pad[threadIdx.x] = make_uint4(1, 2, 3, 4);
  uint32_t d0[8], d1[8], r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { d0[k] = in[gid * 16 + k]; d1[k] = d0[k] + 1; r[k] = in[gid * 16 + 8 + k]; }
  for (int i = 0; i < iters; ++i) {
    if (X2) {
      hash_pair_x2(d0, d1, r);
    } else {
      b2digest a = b2_hash_pair(d0, r), b = b2_hash_pair(d1, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) { d0[k] = a.h[k]; d1[k] = b.h[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) out[gid * 8 + k] = d0[k] ^ d1[k];
}

int main() {
  const int blocks = 256 * 40, iters = 32;
  uint32_t *din, *dout, *dref;
  const size_t n = (size_t)blocks * 256;
  CK(hipMalloc(&din, n * 64)); CK(hipMalloc(&dout, n * 32)); CK(hipMalloc(&dref, n * 32));
  uint32_t* h = (uint32_t*)malloc(n * 64);
  for (size_t i = 0; i < n * 16; ++i) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
  CK(hipMemcpy(din, h, n * 64, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int x2 = 0; x2 < 2; ++x2) {
    auto k = x2 ? k_blake<true> : k_blake<false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int w : {4, 5, 8}) {
      const size_t lds = w == 8 ? 0 : (size_t)(160 * 1024 / w) & ~(size_t)1023;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, din, x2 ? dout : dref, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%s, up to %d waves per SIMD: %.3f ms  %.1f G pair-hashes/s\n", x2 ? "two hashes in lock-step (asm)" : "C++ compression", w, best,
             (double)n * iters * 2 / best / 1e6);
    }
  }
  uint32_t *a = (uint32_t*)malloc(n * 32), *b = (uint32_t*)malloc(n * 32);
  CK(hipMemcpy(a, dout, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(b, dref, n * 32, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n * 8; ++i) bad += a[i] != b[i];
  printf("%zu words compared, %zu mismatches\n", n * 8, bad);
  return bad != 0;
}
