// tools/r04/blake_stream2.hip -- BLAKE2s round streams on explicit registers, incl. two hashes in lock-step (gen_blake_stream2.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "mix3.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
template <int MODE>
__global__ void __launch_bounds__(256) kern(uint32_t* out, int iters) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  asm volatile("v_mov_b32 v8, %0\n v_add_u32 v9, 1, v8\n v_add_u32 v10, 3, v9\n v_add_u32 v11, 5, v10\n v_add_u32 v12, 7, v11\n v_add_u32 v13, 9, v12\n v_add_u32 v14, 11, v13\n v_add_u32 v15, 1, v14\n"
               "v_add_u32 v16, 1, v15\n v_add_u32 v17, 1, v16\n v_add_u32 v18, 3, v17\n v_add_u32 v19, 5, v18\n v_add_u32 v20, 7, v19\n v_add_u32 v21, 9, v20\n v_add_u32 v22, 11, v21\n v_add_u32 v23, 1, v22\n"
               "v_add_u32 v24, 1, v23\n v_add_u32 v25, 1, v24\n v_add_u32 v26, 3, v25\n v_add_u32 v27, 5, v26\n v_add_u32 v28, 7, v27\n v_add_u32 v29, 9, v28\n v_add_u32 v30, 11, v29\n v_add_u32 v31, 1, v30\n"
               "v_add_u32 v32, 1, v31\n v_add_u32 v33, 1, v32\n v_add_u32 v34, 3, v33\n v_add_u32 v35, 5, v34\n v_add_u32 v36, 7, v35\n v_add_u32 v37, 9, v36\n v_add_u32 v38, 11, v37\n v_add_u32 v39, 1, v38\n"
               "v_add_u32 v40, 1, v39\n v_add_u32 v41, 1, v40\n v_add_u32 v42, 3, v41\n v_add_u32 v43, 5, v42\n v_add_u32 v44, 7, v43\n v_add_u32 v45, 9, v44\n v_add_u32 v46, 11, v45\n v_add_u32 v47, 1, v46\n"
               "v_add_u32 v48, 1, v47\n v_add_u32 v49, 1, v48\n v_add_u32 v50, 3, v49\n v_add_u32 v51, 5, v50\n v_add_u32 v52, 7, v51\n v_add_u32 v53, 9, v52\n v_add_u32 v54, 11, v53\n v_add_u32 v55, 1, v54\n"
               "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n" :: "v"(gid) : CLOB);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(M3_ASM_0 ::: CLOB);
    if (MODE == 1) asm volatile(M3_ASM_1 ::: CLOB);
    if (MODE == 2) asm volatile(M3_ASM_2 ::: CLOB);
    if (MODE == 3) asm volatile(M3_ASM_3 ::: CLOB);
    if (MODE == 4) asm volatile(M3_ASM_4 ::: CLOB);
    if (MODE == 5) asm volatile(M3_ASM_5 ::: CLOB);
    if (MODE == 6) asm volatile(M3_ASM_6 ::: CLOB);
    if (MODE == 7) asm volatile(M3_ASM_7 ::: CLOB);
  }
  uint32_t r;
  asm volatile("v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v16\n v_xor_b32 %0, %0, v23\n v_xor_b32 %0, %0, v24\n v_xor_b32 %0, %0, v39\n v_xor_b32 %0, %0, v56" : "=v"(r) :: CLOB);
  out[gid] = r;
}
template <int MODE>
void run(uint32_t* dout, const char* name, int count, int hashes = 1) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 3000;
  printf("%-72s %3d instr", name, count);
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
    const double rounds_per_simd = (double)blocks * 4 * iters * hashes / 1024.0;
    printf("  w%d: %6.1f ns/block %5.2f ns/instr", w, best * 1e6 / rounds_per_simd, best * 1e6 / rounds_per_simd / count * hashes);
  }
  printf("\n");
}
int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 4u * 256 * 256 * 8 * 4));
  run<0>(dout, M3_NAME_0, M3_COUNT_0); run<1>(dout, M3_NAME_1, M3_COUNT_1); run<2>(dout, M3_NAME_2, M3_COUNT_2);
  run<3>(dout, M3_NAME_3, M3_COUNT_3); run<4>(dout, M3_NAME_4, M3_COUNT_4); run<5>(dout, M3_NAME_5, M3_COUNT_5);
  run<6>(dout, M3_NAME_6, M3_COUNT_6); run<7>(dout, M3_NAME_7, M3_COUNT_7);
  return 0;
}
