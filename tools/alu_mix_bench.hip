// tools/alu_mix_bench.hip -- wall-clock integer-VALU peak for the instruction mix of the NTT tile pass (gen_alu_mix.py) at 4 and 5
// resident waves per SIMD, the occupancies the pass runs at.  bench.py runs this binary next to its timed region and puts the
// result on the JSON line as roofline_alu.peak (HIP events, no in-kernel tick counters).
//   python3 tools/gen_alu_mix.py && hipcc -O3 --offload-arch=gfx950 tools/alu_mix_bench.hip -o tools/alu_mix_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "alu_mix.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v56","v57","v58","v59","v60","v61","v62","v63","s20","s21","s22","s23","s24","s25","s26","s27","s30","s31"
__global__ void __launch_bounds__(256) kern(uint32_t* out, int iters) {
  extern __shared__ uint4 pad[];
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (out == nullptr) pad[threadIdx.x] = make_uint4(1, 2, 3, 4);
  asm volatile("s_mov_b64 s[30:31], 0x5555\n"
               "v_mov_b32 v48, %0\n v_add_u32 v49, 3, v48\n v_add_u32 v50, 5, v49\n v_add_u32 v51, 7, v50\n v_add_u32 v52, 9, v51\n v_add_u32 v53, 11, v52\n v_add_u32 v54, 13, v53\n v_add_u32 v55, 17, v54\n"
               "v_add_u32 v56, 19, v55\n v_add_u32 v57, 23, v56\n v_add_u32 v58, 29, v57\n v_add_u32 v59, 31, v58\n v_add_u32 v60, 37, v59\n v_add_u32 v61, 41, v60\n v_add_u32 v62, 43, v61\n v_add_u32 v63, 47, v62\n"
               :: "v"(gid) : CLOB);
  for (int i = 0; i < iters; ++i) asm volatile(ALU_MIX_ASM ::: CLOB);
  uint32_t r;
  asm volatile("v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v20\n v_xor_b32 %0, %0, v33\n v_xor_b32 %0, %0, v47" : "=v"(r) :: CLOB);
  out[gid] = r;
}
int main() {
  uint32_t* dout;
  const int iters = 6000;
  CK(hipMalloc(&dout, 4u * 256 * 256 * 5 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double rate[2];
  int k = 0;
  for (int w : {4, 5}) {
    const size_t lds = (size_t)(160 * 1024 / w) - 1024;  // w workgroups of 4 waves per CU = w waves per SIMD
    const int blocks = 256 * w * 6;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, dout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    rate[k++] = (double)blocks * 256 * iters * ALU_MIX_COUNT / (best * 1e-3) / 1e12;
  }
  printf("{\"unit\": \"T lane-ops/s\", \"waves4\": %.3f, \"waves5\": %.3f, \"instructions_per_loop\": %d, \"mix\": \"%s\", "
         "\"method\": \"HIP-event wall clock, independent instructions, %d iterations x %d instructions per lane, 256 CUs x 4 SIMDs\"}\n",
         rate[0], rate[1], ALU_MIX_COUNT, ALU_MIX_DESC, iters, ALU_MIX_COUNT);
  return 0;
}
