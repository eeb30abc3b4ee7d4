// tools/r04/merkle_lab.hip -- ablations of the Merkle leaf kernel (csrc/kernels.hip:merkle_leaves_kernel): which part of its time is
// hashing, which is memory, and how well they overlap.  FLAGS: bit 0 = real loads, bit 1 = leaf-level store, bit 2 = pair-level
// store, bit 3 = hashing.  A disabled part is replaced by the cheapest thing that keeps the rest alive.
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc tools/r04/merkle_lab.hip -o tools/r04/merkle_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "blake2s.cuh"
#include "fp256.cuh"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int TPB = 256;
__device__ __forceinline__ void store8(uint32_t* p, const uint32_t w[8]) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ uint32_t chunk_swz(uint32_t c) { return (c & ~15u) | ((c & 15u) ^ ((c >> 4) & 15u)); }
template <int CH>
__device__ __forceinline__ void block_store_chunks(uint4* lds, uint4* gdst, const uint4 (&v)[CH], uint32_t t) {
#pragma unroll
  for (int c = 0; c < CH; ++c) lds[chunk_swz(CH * t + c)] = v[c];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const uint32_t c = k * TPB + t;
    gdst[c] = lds[chunk_swz(c)];
  }
  __syncthreads();
}
// wave-local: a wave parks the CH chunks of its 64 rows in its own LDS slice and streams them out lane-contiguously; the LDS unit
// serves a wave's requests in order, so no workgroup barrier is involved
template <int CH>
__device__ __forceinline__ void wave_store_chunks(uint4* lds, uint4* gdst, const uint4 (&v)[CH], uint32_t t) {
  const uint32_t lane = t & 63u, wave = t >> 6;
  uint4* mine = lds + wave * (64 * CH);
  uint4* gmine = gdst + wave * (64 * CH);
#pragma unroll
  for (int c = 0; c < CH; ++c) mine[chunk_swz(CH * lane + c)] = v[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const uint32_t c = k * 64 + lane;
    gmine[c] = mine[chunk_swz(c)];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint4 pack4(const uint32_t* w) { return make_uint4(w[0], w[1], w[2], w[3]); }

template <int FLAGS, int LDSK, int ROWS = 1>
__global__ void __launch_bounds__(TPB) leaves(const fp* leaves, uint64_t n, uint32_t* tree) {
  __shared__ uint4 lds[TPB * LDSK];
  const uint64_t q = n >> 2;
  const uint32_t t = threadIdx.x;
#pragma unroll 1
  for (int rr = 0; rr < ROWS; ++rr) {
  const uint64_t row0 = ((uint64_t)blockIdx.x * ROWS + rr) * TPB, i = row0 + t;
  uint32_t w[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (FLAGS & 1) {
      fp v = fp_canon(fp_load(leaves + i + j * q));
      fp_to_wire_words(v, w[j]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) w[j][k] = (uint32_t)i * (2 * k + 1) + j;
    }
  }
  if (FLAGS & 2) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = pack4(w[j]);
      v[2 * j + 1] = pack4(w[j] + 4);
    }
    if (FLAGS & 16) wave_store_chunks<8>(lds, reinterpret_cast<uint4*>(tree + (n + 4 * row0) * 8), v, t);
    else block_store_chunks<8>(lds, reinterpret_cast<uint4*>(tree + (n + 4 * row0) * 8), v, t);
  }
  b2digest d0, d1, d2;
  if (FLAGS & 8) {
    d0 = b2_hash_pair(w[0], w[1]);
    d1 = b2_hash_pair(w[2], w[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) { d0.h[k] = w[0][k] ^ w[1][k]; d1.h[k] = w[2][k] ^ w[3][k]; }
  }
  if (FLAGS & 4) {
    uint4 v[4] = {pack4(d0.h), pack4(d0.h + 4), pack4(d1.h), pack4(d1.h + 4)};
    if (FLAGS & 16) wave_store_chunks<4>(lds, reinterpret_cast<uint4*>(tree + (n / 2 + 2 * row0) * 8), v, t);
    else block_store_chunks<4>(lds, reinterpret_cast<uint4*>(tree + (n / 2 + 2 * row0) * 8), v, t);
  }
  if (FLAGS & 8) {
    d2 = b2_hash_pair(d0.h, d1.h);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) d2.h[k] = d0.h[k] + d1.h[k];
  }
  if ((FLAGS & 6) || d2.h[0] == 0x12345678u) store8(tree + (n / 4 + i) * 8, d2.h);
  }
}

template <int FLAGS, int LDSK, int ROWS = 1>
float run(const fp* dx, uint32_t* dt, uint64_t n, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((leaves<FLAGS, LDSK, ROWS>), dim3((unsigned)(n / 4 / TPB / ROWS)), dim3(TPB), 0, 0, dx, n, dt);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-72s LDS %2d KiB  %.1f us\n", what, LDSK * 4, best * 1e3);
  return best;
}
int main() {
  const uint64_t n = 1ull << 24;
  fp* dx; uint32_t* dt;
  CK(hipMalloc(&dx, n * 32)); CK(hipMalloc(&dt, n * 64));
  {  // pseudo-random values below p (constant data lets the chip clock higher: not what the library sees)
    uint32_t* h = (uint32_t*)malloc(n * 32);
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < n * 8; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      h[i] = (uint32_t)(x >> 16);
      if ((i & 7) == 7) h[i] &= 0x7fffffffu;
    }
    CK(hipMemcpy(dx, h, n * 32, hipMemcpyHostToDevice));
    free(h);
  }
  run<15, 8>(dx, dt, n, "everything (the library's kernel, leaves stored)");
  run<13, 8>(dx, dt, n, "no leaf-level store (the FRI / STARK variant)");
  run<13, 4>(dx, dt, n, "no leaf-level store, 16 KiB of LDS");
  run<8, 1>(dx, dt, n, "hashing only (inputs synthesised, nothing stored)");
  run<8, 8>(dx, dt, n, "hashing only, 32 KiB of LDS allocated");
  run<8, 2>(dx, dt, n, "hashing only (8 KiB: up to 6 waves per SIMD by registers)");
  run<8, 5>(dx, dt, n, "hashing only (20 KiB: 8 workgroups fit, registers allow 6)");
  run<8, 6>(dx, dt, n, "hashing only (24 KiB: 6 workgroups per CU)");
  run<8, 7>(dx, dt, n, "hashing only (28 KiB: 5 workgroups per CU)");
  run<8, 10>(dx, dt, n, "hashing only (40 KiB: 4 workgroups per CU)");
  run<8, 12>(dx, dt, n, "hashing only (48 KiB: 3 workgroups per CU)");
  run<8, 16>(dx, dt, n, "hashing only (64 KiB: 2 workgroups per CU)");
  run<15, 10>(dx, dt, n, "everything (40 KiB: 4 workgroups per CU)");
  run<15, 12>(dx, dt, n, "everything (48 KiB: 3 workgroups per CU)");
  run<15, 8>(dx, dt, n, "everything (32 KiB) again, after the others");
  run<15, 9>(dx, dt, n, "everything (36 KiB: 4 workgroups per CU)");
  run<15, 10>(dx, dt, n, "everything (40 KiB: 4 workgroups per CU) again");
  run<15, 8>(dx, dt, n, "everything (32 KiB) a third time");
  run<9, 1>(dx, dt, n, "loads + hashing, nothing stored");
  run<7, 8>(dx, dt, n, "loads + both stores through LDS, no hashing (memory path alone)");
  run<5, 4>(dx, dt, n, "loads + pair-level store, no hashing");
  run<14, 8>(dx, dt, n, "hashing + stores, inputs synthesised");
  run<31, 8>(dx, dt, n, "everything, wave-local transposition");
  run<31, 8, 2>(dx, dt, n, "everything, wave-local transposition, 2 rows per thread");
  run<31, 8, 4>(dx, dt, n, "everything, wave-local transposition, 4 rows per thread");
  run<29, 4>(dx, dt, n, "no leaf-level store, wave-local transposition, 16 KiB");
  run<29, 4, 2>(dx, dt, n, "no leaf-level store, wave-local transposition, 16 KiB, 2 rows per thread");
  run<29, 4, 4>(dx, dt, n, "no leaf-level store, wave-local transposition, 16 KiB, 4 rows per thread");
  run<8, 8, 2>(dx, dt, n, "hashing only, 2 rows per thread");
  run<8, 8, 4>(dx, dt, n, "hashing only, 4 rows per thread");
  run<8, 8, 8>(dx, dt, n, "hashing only, 8 rows per thread");
  run<8, 1, 8>(dx, dt, n, "hashing only, 8 rows per thread");
  run<15, 8, 2>(dx, dt, n, "everything, 2 rows per thread");
  run<15, 8, 4>(dx, dt, n, "everything, 4 rows per thread");
  run<15, 8, 8>(dx, dt, n, "everything, 8 rows per thread");
  run<13, 4, 4>(dx, dt, n, "no leaf-level store, 4 rows per thread");
  run<8, 8>(dx, dt, n * 4, "hashing only, 4 x the rows (2^26 leaves' worth), 32 KiB LDS");
  run<8, 8>(dx, dt, n / 4, "hashing only, 1/4 of the rows (2^22 leaves' worth), 32 KiB LDS");
  return 0;
}
