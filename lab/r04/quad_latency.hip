// tools/r04/quad_latency.hip -- latency of ONE quad-lane BLAKE2s compression chain (csrc/blake2s.cuh: b2q_compress, what the Merkle top
// kernel, the FRI tail kernel and the index sampling pay per tree level) for a wave that is ALONE on its SIMD, and with a companion
// wave on the same SIMD that (a) waits at a barrier, (b) spins on s_nop, (c) spins on v_mov, (d) runs its own compression chain.
// A lone wave issues a VALU instruction every 3-5 ns whatever the dependencies (profiles/r04_blake2s_issue_rate_study.txt); does a
// second wave on the SIMD change what the first one gets?
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc tools/r04/quad_latency.hip -o tools/r04/quad_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "blake2s.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// MODE 0: 4 waves (one per SIMD), all hashing.  MODE 1..4: 8 waves; waves 0-3 hash, waves 4-7 = companions:
// 1 = exit at once, 2 = s_nop spin, 3 = v_mov spin, 4 = hash too (their own slots).  MODE 5: 8 waves, companions s_sleep spin.
template <int MODE>
__global__ void __launch_bounds__(512) k(uint32_t* out, int iters, volatile uint32_t* flag) {
  __shared__ __attribute__((aligned(16))) uint32_t slots[128 * 16];
  __shared__ uint32_t done;
  const uint32_t tid = threadIdx.x, quad = tid >> 2, q = tid & 3, wave = tid >> 6;
  if (tid == 0) done = 0;
  for (int i = tid; i < 128 * 16; i += blockDim.x) slots[i] = i * 2654435761u + blockIdx.x;
  __syncthreads();
  const bool worker = wave < 4 || MODE == 4;
  if (worker) {
    b2q_addr ad;
    b2q_addr_init(ad, quad * 64, q);
    uint32_t h_lo = 0, h_hi = 0;
    for (int i = 0; i < iters; ++i) {
      b2q_compress(ad, slots, q, 64, h_lo, h_hi);
      uint32_t* dst = slots + quad * 16;  // the digest becomes part of the next message: a serial chain
      dst[q] = h_lo;
      dst[4 + q] = h_hi;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    out[blockIdx.x * blockDim.x + tid] = h_lo ^ h_hi;
    if (wave < 4 && (tid & 63) == 0) atomicAdd(&done, 1u);
  } else {
    uint32_t x = tid;
    if (MODE == 2) {
      while (*(volatile uint32_t*)&done < 4) asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7");
    } else if (MODE == 3) {
      while (*(volatile uint32_t*)&done < 4) asm volatile("v_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0" : "+v"(x));
    } else if (MODE == 5) {
      while (*(volatile uint32_t*)&done < 4) asm volatile("s_sleep 1");
    }
    if (x == 0xffffffffu) out[0] = x;
  }
}

template <int MODE>
void run(uint32_t* dout, const char* name, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 2000;
  const int threads = MODE == 0 ? 256 : 512;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, dout, iters, nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-70s %3d workgroups: %.3f us per compression\n", name, blocks, best * 1e3 / iters);
}

int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 4u * 512 * 1024));
  for (int blocks : {1, 256}) {
    run<0>(dout, "4 waves, one per SIMD, all hashing (the top kernel's situation)", blocks);
    run<1>(dout, "8 waves, companions exit at once", blocks);
    run<2>(dout, "8 waves, companions spin on s_nop", blocks);
    run<5>(dout, "8 waves, companions spin on s_sleep", blocks);
    run<3>(dout, "8 waves, companions spin on v_mov", blocks);
    run<4>(dout, "8 waves, two hashing waves per SIMD", blocks);
  }
  return 0;
}
