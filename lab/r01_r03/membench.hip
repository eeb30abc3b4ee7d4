// membench.hip -- HBM copy patterns for 32-byte elements on gfx950 (sizing the Merkle / conversion kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// (a) one 32-byte element per lane: two dwordx4 at +0 / +16, lane stride 32 B
__global__ void __launch_bounds__(256) copy_elem32(const uint4* in, uint4* out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint4 a = in[2 * i], b = in[2 * i + 1];
  out[2 * i] = a; out[2 * i + 1] = b;
}
// (b) 16 bytes per lane, contiguous across the wave
__global__ void __launch_bounds__(256) copy_chunk16(const uint4* in, uint4* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  out[i] = in[i];
}
// (c) permute4 gather: thread i reads 4 elements at stride q and writes them adjacent (the Merkle leaf pattern)
__global__ void __launch_bounds__(256) gather4_elem32(const uint4* in, uint4* out, size_t n) {
  size_t q = n / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= q) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint4 a = in[2 * (i + j * q)], b = in[2 * (i + j * q) + 1];
    out[2 * (4 * i + j)] = a; out[2 * (4 * i + j) + 1] = b;
  }
}
// (d) same gather, but the 128 B a thread produces are transposed through LDS so that stores are lane-contiguous
__global__ void __launch_bounds__(256) gather4_lds(const uint4* in, uint4* out, size_t n) {
  __shared__ uint4 buf[256 * 8];
  size_t q = n / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int t = threadIdx.x;
  if (i < q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 a = in[2 * (i + j * q)], b = in[2 * (i + j * q) + 1];
      // chunk index inside the block's 32 KiB output: (4 t + j) * 2 + h ; XOR-swizzle the low bits by t to spread banks
      int c0 = (4 * t + j) * 2;
      buf[(c0) ^ ((t >> 1) & 6)] = a;
      buf[(c0 + 1) ^ ((t >> 1) & 6)] = b;
    }
  }
  __syncthreads();
  size_t obase = (size_t)blockIdx.x * 256 * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int c = k * 256 + t;            // linear chunk
    int src_t = c >> 3;             // thread that produced it
    uint4 v = buf[c ^ ((src_t >> 1) & 6)];
    if (obase + c < 2 * n) out[obase + c] = v;
  }
}
int main() {
  const size_t n = (size_t)1 << 24;  // elements of 32 B = 512 MiB
  uint4 *a, *b;
  CK(hipMalloc(&a, n * 32)); CK(hipMalloc(&b, n * 32));
  CK(hipMemset(a, 1, n * 32));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 4; ++variant) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      if (variant == 0) hipLaunchKernelGGL(copy_elem32, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
      if (variant == 1) hipLaunchKernelGGL(copy_chunk16, dim3((2 * n + 255) / 256), dim3(256), 0, 0, a, b, 2 * n);
      if (variant == 2) hipLaunchKernelGGL(gather4_elem32, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, a, b, n);
      if (variant == 3) hipLaunchKernelGGL(gather4_lds, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, a, b, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    const char* names[] = {"elem32 copy (2 x dwordx4 per lane)", "chunk16 copy (lane-contiguous)", "permute4 gather, 128 B per thread", "permute4 gather, LDS-transposed stores"};
    printf("%-42s %.3f ms  %.2f TB/s (read+write)\n", names[variant], best, 2.0 * n * 32 / best / 1e9);
  }
  return 0;
}
