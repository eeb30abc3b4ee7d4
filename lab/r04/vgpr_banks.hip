// tools/r04/vgpr_banks.hip -- does the fast rate of v_xor_b32 / v_add_u32 on gfx950 depend on WHICH registers an instruction names?
// 16 independent instructions per block on explicit registers v8..v39; variants differ in (src1 - dst) mod 4 and in dst == src0.
//   hipcc -O3 --offload-arch=gfx950 tools/r04/vgpr_banks.hip -o tools/r04/vgpr_banks && tools/r04/vgpr_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39"
// I(d, a, b): v_xor_b32 v[d], v[a], v[b]
#define I(op, d, a, b) op " v" #d ", v" #a ", v" #b "\n"

template <int MODE>
__global__ void __launch_bounds__(256) kern(uint32_t* out, int iters) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
               "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
               "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
               "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n" :: "v"(gid) : CLOB);
  for (int i = 0; i < iters; ++i) {
    // in place (dst == src0), src1 at distance 16 (same bank)
    if (MODE == 0) asm volatile(I("v_xor_b32",8,8,24) I("v_xor_b32",9,9,25) I("v_xor_b32",10,10,26) I("v_xor_b32",11,11,27) I("v_xor_b32",12,12,28) I("v_xor_b32",13,13,29) I("v_xor_b32",14,14,30) I("v_xor_b32",15,15,31)
                                I("v_xor_b32",16,16,32) I("v_xor_b32",17,17,33) I("v_xor_b32",18,18,34) I("v_xor_b32",19,19,35) I("v_xor_b32",20,20,36) I("v_xor_b32",21,21,37) I("v_xor_b32",22,22,38) I("v_xor_b32",23,23,39) ::: CLOB);
    // in place, src1 at distance 17 (next bank)
    if (MODE == 1) asm volatile(I("v_xor_b32",8,8,25) I("v_xor_b32",9,9,26) I("v_xor_b32",10,10,27) I("v_xor_b32",11,11,28) I("v_xor_b32",12,12,29) I("v_xor_b32",13,13,30) I("v_xor_b32",14,14,31) I("v_xor_b32",15,15,32)
                                I("v_xor_b32",16,16,33) I("v_xor_b32",17,17,34) I("v_xor_b32",18,18,35) I("v_xor_b32",19,19,36) I("v_xor_b32",20,20,37) I("v_xor_b32",21,21,38) I("v_xor_b32",22,22,39) I("v_xor_b32",23,23,24) ::: CLOB);
    // in place, distance 18
    if (MODE == 2) asm volatile(I("v_xor_b32",8,8,26) I("v_xor_b32",9,9,27) I("v_xor_b32",10,10,28) I("v_xor_b32",11,11,29) I("v_xor_b32",12,12,30) I("v_xor_b32",13,13,31) I("v_xor_b32",14,14,32) I("v_xor_b32",15,15,33)
                                I("v_xor_b32",16,16,34) I("v_xor_b32",17,17,35) I("v_xor_b32",18,18,36) I("v_xor_b32",19,19,37) I("v_xor_b32",20,20,38) I("v_xor_b32",21,21,39) I("v_xor_b32",22,22,24) I("v_xor_b32",23,23,25) ::: CLOB);
    // src0 == src1 register (one read)
    if (MODE == 3) asm volatile(I("v_xor_b32",8,24,24) I("v_xor_b32",9,25,25) I("v_xor_b32",10,26,26) I("v_xor_b32",11,27,27) I("v_xor_b32",12,28,28) I("v_xor_b32",13,29,29) I("v_xor_b32",14,30,30) I("v_xor_b32",15,31,31)
                                I("v_xor_b32",16,32,32) I("v_xor_b32",17,33,33) I("v_xor_b32",18,34,34) I("v_xor_b32",19,35,35) I("v_xor_b32",20,36,36) I("v_xor_b32",21,37,37) I("v_xor_b32",22,38,38) I("v_xor_b32",23,39,39) ::: CLOB);
    // three different registers, all in different banks: d = bank b, a = bank b+1, b = bank b+2
    if (MODE == 4) asm volatile(I("v_xor_b32",8,25,30) I("v_xor_b32",9,26,31) I("v_xor_b32",10,27,32) I("v_xor_b32",11,28,33) I("v_xor_b32",12,29,34) I("v_xor_b32",13,30,35) I("v_xor_b32",14,31,36) I("v_xor_b32",15,32,37)
                                I("v_xor_b32",16,33,38) I("v_xor_b32",17,34,39) I("v_xor_b32",18,35,24) I("v_xor_b32",19,36,25) I("v_xor_b32",20,37,26) I("v_xor_b32",21,38,27) I("v_xor_b32",22,39,28) I("v_xor_b32",23,24,29) ::: CLOB);
    // three different registers, all in the SAME bank
    if (MODE == 5) asm volatile(I("v_xor_b32",8,24,32) I("v_xor_b32",9,25,33) I("v_xor_b32",10,26,34) I("v_xor_b32",11,27,35) I("v_xor_b32",12,28,36) I("v_xor_b32",13,29,37) I("v_xor_b32",14,30,38) I("v_xor_b32",15,31,39)
                                I("v_xor_b32",16,32,24) I("v_xor_b32",17,33,25) I("v_xor_b32",18,34,26) I("v_xor_b32",19,35,27) I("v_xor_b32",20,36,28) I("v_xor_b32",21,37,29) I("v_xor_b32",22,38,30) I("v_xor_b32",23,39,31) ::: CLOB);
    // dependent chain of distance 4 (as in lock-step BLAKE2s): each instruction reads the result written 4 instructions earlier
    if (MODE == 6) asm volatile(I("v_xor_b32",8,8,25) I("v_xor_b32",9,9,26) I("v_xor_b32",10,10,27) I("v_xor_b32",11,11,28) I("v_xor_b32",12,12,8) I("v_xor_b32",13,13,9) I("v_xor_b32",14,14,10) I("v_xor_b32",15,15,11)
                                I("v_xor_b32",8,8,12) I("v_xor_b32",9,9,13) I("v_xor_b32",10,10,14) I("v_xor_b32",11,11,15) I("v_xor_b32",12,12,8) I("v_xor_b32",13,13,9) I("v_xor_b32",14,14,10) I("v_xor_b32",15,15,11) ::: CLOB);
    // dependent chain of distance 1
    if (MODE == 7) asm volatile(I("v_xor_b32",8,8,25) I("v_xor_b32",8,8,26) I("v_xor_b32",8,8,27) I("v_xor_b32",8,8,28) I("v_xor_b32",8,8,29) I("v_xor_b32",8,8,30) I("v_xor_b32",8,8,31) I("v_xor_b32",8,8,32)
                                I("v_xor_b32",8,8,33) I("v_xor_b32",8,8,34) I("v_xor_b32",8,8,35) I("v_xor_b32",8,8,36) I("v_xor_b32",8,8,37) I("v_xor_b32",8,8,38) I("v_xor_b32",8,8,39) I("v_xor_b32",8,8,24) ::: CLOB);
    // dependent chain of distance 2
    if (MODE == 8) asm volatile(I("v_xor_b32",8,8,25) I("v_xor_b32",9,9,26) I("v_xor_b32",8,8,27) I("v_xor_b32",9,9,28) I("v_xor_b32",8,8,29) I("v_xor_b32",9,9,30) I("v_xor_b32",8,8,31) I("v_xor_b32",9,9,32)
                                I("v_xor_b32",8,8,33) I("v_xor_b32",9,9,34) I("v_xor_b32",8,8,35) I("v_xor_b32",9,9,36) I("v_xor_b32",8,8,37) I("v_xor_b32",9,9,38) I("v_xor_b32",8,8,39) I("v_xor_b32",9,9,24) ::: CLOB);
  }
  uint32_t r;
  asm volatile("v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v16\n v_xor_b32 %0, %0, v23" : "=v"(r) :: CLOB);
  out[gid] = r;
}
static const char* NAME[9] = {"in place, src1 same bank (dist 16)", "in place, src1 next bank (dist 17)", "in place, src1 dist 18", "src0 == src1",
                              "3 registers, 3 banks", "3 registers, one bank", "dependent, distance 4", "dependent, distance 1", "dependent, distance 2"};
template <int MODE>
void run(uint32_t* dout) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  printf("%-40s", NAME[MODE]);
  for (int w : {1, 2, 4, 8}) {
    const int blocks = 256 * w * 4;
    auto k = kern<MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = w == 8 ? 0 : (size_t)(160 * 1024 / w) - (w == 1 ? 4096 : 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, dout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)blocks * 4 * iters * 16 / 1024.0;
    printf("  w%d: %5.2f ns/instr", w, best * 1e6 / instr_per_simd);
  }
  printf("\n");
}
int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 4u * 256 * 256 * 8 * 4));
  run<0>(dout); run<1>(dout); run<2>(dout); run<3>(dout); run<4>(dout); run<5>(dout); run<6>(dout); run<7>(dout); run<8>(dout);
  return 0;
}
