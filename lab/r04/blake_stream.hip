// tools/r04/blake_stream.hip -- one BLAKE2s round (two half-rounds of 4 x G on 16 state + 8 + 8 message registers) as instruction
// streams in different orders / instruction choices, timed per instruction (gen_blake_stream.py writes the variants).
//   python3 tools/r04/gen_blake_stream.py && hipcc -O3 --offload-arch=gfx950 tools/r04/blake_stream.hip -o tools/r04/blake_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "blake_stream.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define STATE "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), \
              "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
#define MSG "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7])

template <int MODE>
__global__ void __launch_bounds__(256) kern(uint32_t* out, int iters) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v[16], m[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = gid * (2 * i + 3);
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = gid ^ (0x9e3779b9u * (i + 1));
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(BS_ASM_0 : STATE : MSG);
    if (MODE == 1) asm volatile(BS_ASM_1 : STATE : MSG);
    if (MODE == 2) asm volatile(BS_ASM_2 : STATE : MSG);
    if (MODE == 3) asm volatile(BS_ASM_3 : STATE : MSG);
    if (MODE == 4) asm volatile(BS_ASM_4 : STATE : MSG);
    if (MODE == 5) asm volatile(BS_ASM_5 : STATE : MSG);
    if (MODE == 6) asm volatile(BS_ASM_6 : STATE : MSG);
    if (MODE == 7) asm volatile(BS_ASM_7 : STATE : MSG);
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r ^= v[i];
  out[gid] = r;
}
template <int MODE>
void run(uint32_t* dout, const char* name, int count) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 3000;
  printf("%-58s %3d instr", name, count);
  for (int w : {2, 4, 8}) {
    const int blocks = 256 * w * 4;
    auto k = kern<MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = w == 8 ? 0 : (size_t)(160 * 1024 / w) - 1024;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, dout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double rounds_per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("  w%d: %6.1f ns/round %5.2f ns/instr", w, best * 1e6 / rounds_per_simd, best * 1e6 / rounds_per_simd / count);
  }
  printf("\n");
}
int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 4u * 256 * 256 * 8 * 4));
  run<0>(dout, BS_NAME_0, BS_COUNT_0); run<1>(dout, BS_NAME_1, BS_COUNT_1); run<2>(dout, BS_NAME_2, BS_COUNT_2); run<3>(dout, BS_NAME_3, BS_COUNT_3);
  run<4>(dout, BS_NAME_4, BS_COUNT_4); run<5>(dout, BS_NAME_5, BS_COUNT_5); run<6>(dout, BS_NAME_6, BS_COUNT_6); run<7>(dout, BS_NAME_7, BS_COUNT_7);
  return 0;
}
