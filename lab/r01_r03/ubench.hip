// ubench.hip -- gfx950 integer-ALU probes that size the 256-bit modmul (not product code).
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc tools/ubench.hip -o tools/ubench
// Prints instruction issue rates for the candidate multiply instructions and the throughput of the
// fp256 modmul/add/sub from starks_amd/csrc/fp256.cuh, after checking them against a host computation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "fp256.cuh"
#include "fp29.cuh"  // tools/fp29.cuh (the experiment is not part of the library)
#include "blake2s.cuh"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int ITERS = 4096;

// ---- raw instruction probes: 8 independent chains per lane ---------------------------------------
#define PROBE(NAME, ASM8)                                                                   \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {               \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;    \
    uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;        \
    uint32_t b = a0 | 0x80000001u;                                                          \
    for (int i = 0; i < ITERS; ++i) {                                                       \
      asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),       \
                   "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","s20","s21","s22","s23","s24","s25","s26","s27"); \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
  }

PROBE(k_mul_lo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
PROBE(k_mul_hi, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n")
PROBE(k_mad24, "v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n"
               "v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n")
PROBE(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
                 "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n")
PROBE(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
PROBE(k_addc, "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
              "v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n")
PROBE(k_dot4, "v_dot4_u32_u8 %0, %0, %8, %1\n v_dot4_u32_u8 %1, %1, %8, %2\n v_dot4_u32_u8 %2, %2, %8, %3\n v_dot4_u32_u8 %3, %3, %8, %4\n"
              "v_dot4_u32_u8 %4, %4, %8, %5\n v_dot4_u32_u8 %5, %5, %8, %6\n v_dot4_u32_u8 %6, %6, %8, %7\n v_dot4_u32_u8 %7, %7, %8, %0\n")


PROBE(k_addco, "v_add_co_u32 %0, s[10:11], %0, %8\n v_add_co_u32 %1, s[12:13], %1, %8\n v_add_co_u32 %2, s[14:15], %2, %8\n v_add_co_u32 %3, s[16:17], %3, %8\n"
               "v_add_co_u32 %4, s[18:19], %4, %8\n v_add_co_u32 %5, s[20:21], %5, %8\n v_add_co_u32 %6, s[22:23], %6, %8\n v_add_co_u32 %7, s[24:25], %7, %8\n")
// addc with independent carry registers written far earlier (no VALU->SGPR->VALU hazard inside the loop)
PROBE(k_addc_indep, "v_addc_co_u32 %0, s[10:11], %0, %8, s[26:27]\n v_addc_co_u32 %1, s[12:13], %1, %8, s[26:27]\n v_addc_co_u32 %2, s[14:15], %2, %8, s[26:27]\n v_addc_co_u32 %3, s[16:17], %3, %8, s[26:27]\n"
                    "v_addc_co_u32 %4, s[18:19], %4, %8, s[26:27]\n v_addc_co_u32 %5, s[20:21], %5, %8, s[26:27]\n v_addc_co_u32 %6, s[22:23], %6, %8, s[26:27]\n v_addc_co_u32 %7, s[24:25], %7, %8, s[26:27]\n")
PROBE(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n")
PROBE(k_cndmask, "v_cndmask_b32 %0, %0, %8, s[26:27]\n v_cndmask_b32 %1, %1, %8, s[26:27]\n v_cndmask_b32 %2, %2, %8, s[26:27]\n v_cndmask_b32 %3, %3, %8, s[26:27]\n"
                 "v_cndmask_b32 %4, %4, %8, s[26:27]\n v_cndmask_b32 %5, %5, %8, s[26:27]\n v_cndmask_b32 %6, %6, %8, s[26:27]\n v_cndmask_b32 %7, %7, %8, s[26:27]\n")
PROBE(k_add3, "v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n"
              "v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0\n")
PROBE(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
PROBE(k_alignbit, "v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 7\n v_alignbit_b32 %2, %2, %2, 7\n v_alignbit_b32 %3, %3, %3, 7\n v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n v_alignbit_b32 %7, %7, %7, 7\n")
PROBE(k_perm, "v_perm_b32 %0, %0, %0, %8\n v_perm_b32 %1, %1, %1, %8\n v_perm_b32 %2, %2, %2, %8\n v_perm_b32 %3, %3, %3, %8\n v_perm_b32 %4, %4, %4, %8\n v_perm_b32 %5, %5, %5, %8\n v_perm_b32 %6, %6, %6, %8\n v_perm_b32 %7, %7, %7, %8\n")
PROBE(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %1, %1, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %2, %2, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %3, %3, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n"
                  "v_xor_b32_sdwa %4, %4, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %5, %5, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %6, %6, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %7, %7, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n")
PROBE(k_xad, "v_xad_u32 %0, %0, %8, %1\n v_xad_u32 %1, %1, %8, %2\n v_xad_u32 %2, %2, %8, %3\n v_xad_u32 %3, %3, %8, %4\n v_xad_u32 %4, %4, %8, %5\n v_xad_u32 %5, %5, %8, %6\n v_xad_u32 %6, %6, %8, %7\n v_xad_u32 %7, %7, %8, %0\n")
PROBE(k_and_or, "v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0\n")
PROBE(k_lshlrev, "v_lshlrev_b32 %0, 7, %0\n v_lshlrev_b32 %1, 7, %1\n v_lshlrev_b32 %2, 7, %2\n v_lshlrev_b32 %3, 7, %3\n v_lshlrev_b32 %4, 7, %4\n v_lshlrev_b32 %5, 7, %5\n v_lshlrev_b32 %6, 7, %6\n v_lshlrev_b32 %7, 7, %7\n")
// one dependent chain of v_mad_u64_u32 per lane: latency
__global__ void __launch_bounds__(256) k_mad64_dep(uint32_t* out, uint32_t seed) {
  uint64_t a0 = threadIdx.x + seed;
  uint32_t b = (uint32_t)a0 | 0x80000001u, c = b * 77 + 5;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n"
        "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n"
        "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n"
        "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n"
        : "+v"(a0) : "v"(b), "v"(c) : "s10", "s11");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32);
}
__global__ void __launch_bounds__(256) k_lshladd64(uint32_t* out, uint32_t seed) {
  uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b = a0 * 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
        "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
  }
  uint64_t r = a0 ^ a1 ^ a2 ^ a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

// v_mad_u64_u32: 4 independent 64-bit accumulator chains per lane (8 VGPRs)
__global__ void __launch_bounds__(256) k_mad64(uint32_t* out, uint32_t seed) {
  uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
  uint32_t b = (uint32_t)a0 | 0x80000001u, c = b * 77 + 5;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[12:13], %4, %5, %1\n"
        "v_mad_u64_u32 %2, s[14:15], %4, %5, %2\n v_mad_u64_u32 %3, s[16:17], %4, %5, %3\n"
        "v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[12:13], %4, %5, %1\n"
        "v_mad_u64_u32 %2, s[14:15], %4, %5, %2\n v_mad_u64_u32 %3, s[16:17], %4, %5, %3\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(b), "v"(c)
        : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
  }
  uint64_t r = a0 ^ a1 ^ a2 ^ a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

// f64 fma, 4 chains
__global__ void __launch_bounds__(256) k_fma64(uint32_t* out, uint32_t seed) {
  double a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
  double b = 1.0000001, c = 0.5;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile(
        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(b), "v"(c));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}

// ---- fp256 op throughput ---------------------------------------------------------------------------
constexpr int FITERS = 512;

template <int CHAINS>
__global__ void __launch_bounds__(256) k_fpmul(const fp* in, fp* out) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  fp a[CHAINS];
  fp b = in[gid * 2 + 1];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    a[c] = in[gid * 2];
    a[c].v[0] += c;
  }
  for (int i = 0; i < FITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = fp_mul(a[c], b);
  }
  fp r = a[0];
#pragma unroll
  for (int c = 1; c < CHAINS; ++c) r = fp_add(r, a[c]);
  out[gid] = r;
}

__global__ void __launch_bounds__(256) k_fpaddsub(const fp* in, fp* out) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  fp a = in[gid * 2], b = in[gid * 2 + 1];
  for (int i = 0; i < FITERS; ++i) {
    fp s = fp_add(a, b);
    fp d = fp_sub(a, b);
    a = s;
    b = d;
  }
  out[gid] = fp_add(a, b);
}

// butterfly chain: (x, y) -> (x + w*y, x - w*y)
__global__ void __launch_bounds__(256) k_fpbfly(const fp* in, fp* out) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  fp x = in[gid * 2], y = in[gid * 2 + 1], w = in[(gid * 2 + 3) % 1024];
  for (int i = 0; i < FITERS; ++i) {
    fp t = fp_mul(y, w);
    fp s = fp_add(x, t);
    fp d = fp_sub(x, t);
    x = s;
    y = d;
  }
  out[gid] = fp_add(x, y);
}


// ---- unsaturated 9 x 29 arithmetic (fp29.cuh) -------------------------------------------------------------
__global__ void __launch_bounds__(256) k_f29mul(const fp* in, fp* out) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  fp29 a = f29_from_fp(in[gid * 2]);
  fp29 w = f29_balanced_from_fp(fp_canon(in[gid * 2 + 1]));
  for (int i = 0; i < FITERS; ++i) a = f29_mul(a, w);
  out[gid] = f29_to_fp(a);
}
// DIT butterfly chain: t = w*y ; (x, y) <- (x + t, x - t); digits renormalised every 4 levels
__global__ void __launch_bounds__(256) k_f29bfly(const fp* in, fp* out) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  fp29 x = f29_from_fp(in[gid * 2]), y = f29_from_fp(in[gid * 2 + 1]);
  fp29 w = f29_balanced_from_fp(fp_canon(in[(gid * 2 + 3) % 1024]));
  for (int i = 0; i < FITERS; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fp29 t = f29_mul(y, w);
      fp29 s = f29_add(x, t);
      y = f29_sub(x, t);
      x = s;
    }
    x = f29_normalize(x);
    y = f29_normalize(y);
  }
  out[gid] = f29_to_fp(f29_add(x, y));
}
__global__ void k_f29check(const fp* in, fp* out, int n) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  fp a = in[gid * 2], b = fp_canon(in[gid * 2 + 1]);
  fp29 xa = f29_from_fp(a), wb = f29_balanced_from_fp(b);
  fp29 m = f29_mul(xa, wb);
  out[gid * 4 + 0] = fp_canon(f29_to_fp(m));                                 // a*b
  out[gid * 4 + 1] = fp_canon(f29_to_fp(f29_add(xa, m)));                    // a + a*b
  out[gid * 4 + 2] = fp_canon(f29_to_fp(f29_sub(xa, m)));                    // a - a*b
  fp29 d = f29_sub(f29_sub(f29_sub(xa, m), m), m);                           // a - 3ab (lazy, negative limbs)
  out[gid * 4 + 3] = fp_canon(f29_to_fp(f29_mul(f29_normalize(d), wb)));     // (a - 3ab) * b
}

// BLAKE2s throughput: a chain of 64 pair-hashes per lane
__global__ void __launch_bounds__(256) k_blake(const fp* in, fp* out) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  fp a = in[gid * 2], b = in[gid * 2 + 1];
  b2digest d;
#pragma unroll
  for (int k = 0; k < 8; ++k) d.h[k] = a.v[k];
  for (int i = 0; i < 64; ++i) d = b2_hash_pair(d.h, b.v);
  fp r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = d.h[k];
  out[gid] = r;
}

__global__ void k_fpcheck(const fp* in, fp* out, int n) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  fp a = in[gid * 2], b = in[gid * 2 + 1];
  out[gid * 4 + 0] = fp_canon(fp_mul(a, b));
  out[gid * 4 + 1] = fp_canon(fp_add(a, b));
  out[gid * 4 + 2] = fp_canon(fp_sub(a, b));
  out[gid * 4 + 3] = fp_canon(fp_mul(fp_add(a, b), fp_sub(a, b)));
}

// ---- host reference with unsigned __int128 (independent of fp256.cuh) -----------------------------
typedef unsigned __int128 u128;
struct H256 {
  uint64_t w[4];
};
static const H256 HP = {{0xfffffea100000001ull, ~0ull, ~0ull, ~0ull}};
static int hcmp(const H256& a, const H256& b) {
  for (int i = 3; i >= 0; --i)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}
static H256 hsubraw(const H256& a, const H256& b, int* borrow) {
  H256 r;
  u128 bw = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a.w[i] - b.w[i] - bw;
    r.w[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
  *borrow = (int)bw;
  return r;
}
static H256 hmod_generic(const uint64_t t[8]) {
  // bit-serial reduction of a 512-bit number mod p (slow, obviously correct)
  H256 r = {{0, 0, 0, 0}};
  for (int bit = 511; bit >= 0; --bit) {
    int top = (int)(r.w[3] >> 63);
    for (int i = 3; i > 0; --i) r.w[i] = (r.w[i] << 1) | (r.w[i - 1] >> 63);
    r.w[0] = (r.w[0] << 1) | ((t[bit / 64] >> (bit % 64)) & 1);
    int bw;
    H256 s = hsubraw(r, HP, &bw);
    if (top || !bw) r = s;
  }
  return r;
}
static H256 hmul(const H256& a, const H256& b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u128 cy = 0;
    for (int j = 0; j < 4; ++j) {
      u128 m = (u128)a.w[i] * b.w[j] + t[i + j] + cy;
      t[i + j] = (uint64_t)m;
      cy = m >> 64;
    }
    t[i + 4] = (uint64_t)cy;
  }
  return hmod_generic(t);
}
static H256 hadd(const H256& a, const H256& b) {
  uint64_t t[8] = {0};
  u128 cy = 0;
  for (int i = 0; i < 4; ++i) {
    u128 m = (u128)a.w[i] + b.w[i] + cy;
    t[i] = (uint64_t)m;
    cy = m >> 64;
  }
  t[4] = (uint64_t)cy;
  return hmod_generic(t);
}
static H256 hcanon(const H256& a) {
  uint64_t t[8] = {a.w[0], a.w[1], a.w[2], a.w[3], 0, 0, 0, 0};
  return hmod_generic(t);
}
static H256 hsub(const H256& a, const H256& b) {
  H256 bc = hcanon(b);
  int bw;
  H256 nb = hsubraw(HP, bc, &bw);  // p - b
  return hadd(hcanon(a), nb);
}
static H256 from_fp(const fp& x) {
  H256 r;
  for (int i = 0; i < 4; ++i) r.w[i] = (uint64_t)x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32);
  return r;
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

template <typename F>
static double time_kernel(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e-3;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  const int blocks = prop.multiProcessorCount * 8, threads = 256;
  const size_t nthreads = (size_t)blocks * threads;
  uint32_t* dout;
  CK(hipMalloc(&dout, nthreads * 4));

  struct {
    const char* name;
    void (*k)(uint32_t*, uint32_t);
    int per_iter;
  } probes[] = {{"v_mul_lo_u32", k_mul_lo, 8},       {"v_mul_hi_u32", k_mul_hi, 8}, {"v_mad_u32_u24", k_mad24, 8},
                {"v_mul_hi_u32_u24", k_mulhi24, 8}, {"v_add_u32", k_add, 8},       {"v_add(c)_co_u32", k_addc, 8},
                {"v_dot4_u32_u8", k_dot4, 8},        {"v_mad_u64_u32", k_mad64, 8}, {"v_fma_f64", k_fma64, 8},
                {"v_add_co_u32 (indep)", k_addco, 8}, {"v_addc_co_u32 (indep)", k_addc_indep, 8}, {"v_mov_b32", k_mov, 8},
                {"v_cndmask_b32", k_cndmask, 8}, {"v_add3_u32", k_add3, 8}, {"v_xor_b32", k_xor, 8}, {"v_alignbit_b32", k_alignbit, 8},
                {"v_mad_u64_u32 dep chain", k_mad64_dep, 8}, {"v_lshl_add_u64", k_lshladd64, 8},
                {"v_perm_b32", k_perm, 8}, {"v_xor_b32 sdwa", k_xor_sdwa, 8}, {"v_xad_u32", k_xad, 8},
                {"v_and_or_b32", k_and_or, 8}, {"v_lshlrev_b32", k_lshlrev, 8}};
  for (auto& p : probes) {
    double s = time_kernel([&] { hipLaunchKernelGGL(p.k, dim3(blocks), dim3(threads), 0, 0, dout, 1u); });
    double lane_ops = (double)nthreads * ITERS * p.per_iter;
    double per_cu_clk = lane_ops / s / prop.multiProcessorCount / (prop.clockRate * 1e3);
    printf("%-18s %8.3f ms  %7.2f T lane-ops/s  %6.1f lanes/clk/CU (at nominal clock)\n", p.name, s * 1e3,
           lane_ops / s * 1e-12, per_cu_clk);
  }

  // ---- fp256 correctness on device vs host ----
  const int NCHK = 4096;
  std::vector<fp> hin(NCHK * 2);
  for (int i = 0; i < NCHK * 2; ++i)
    for (int j = 0; j < 8; ++j) hin[i].v[j] = (uint32_t)rng();
  // edge cases
  auto setall = [&](fp& x, uint32_t v) { for (int j = 0; j < 8; ++j) x.v[j] = v; };
  setall(hin[0], 0xffffffffu); setall(hin[1], 0xffffffffu);
  setall(hin[2], 0); setall(hin[3], 0xffffffffu);
  setall(hin[4], 0xffffffffu); hin[4].v[0] = 0; hin[4].v[1] = 0xfffffea1u;  // p - 1
  hin[5] = hin[4];
  setall(hin[6], 0); hin[6].v[0] = 1; setall(hin[7], 0xffffffffu); hin[7].v[0] = 0xfffffffeu;
  setall(hin[8], 0xffffffffu); hin[8].v[1] = 0xfffffea1u; hin[8].v[0] = 1; setall(hin[9], 0xffffffffu);  // p, 2^256-1
  setall(hin[10], 0); setall(hin[11], 0); hin[11].v[0] = 1;                                            // 0 - 1
  setall(hin[12], 0); hin[12].v[0] = 5; setall(hin[13], 0xffffffffu); hin[13].v[0] = 0xfffffff0u;   // small - huge
  // rare wave-uniform branches of fp_add / fp_sub / fp_reduce_wide (vectors crafted with a Python model)
  // add: a = 2^256-1, b = 2^256-2^32+6 -> carry; the fold's carry leaves limb 1; second wrap
  setall(hin[14], 0xffffffffu);
  setall(hin[15], 0xffffffffu); hin[15].v[0] = 6;
  // sub: 0 - (2^256-5): borrow, the fold's borrow leaves limb 1, second borrow
  setall(hin[16], 0); setall(hin[17], 0xffffffffu); hin[17].v[0] = 0xfffffffbu;
  // sub: wrapped difference 7*2^64+3: borrow leaves limb 1, no second borrow (a = r - 1, b = 2^256 - 1)
  setall(hin[18], 0); hin[18].v[0] = 2; hin[18].v[2] = 7; setall(hin[19], 0xffffffffu);
  // mul: x = 2*floor((2^257-1)/c), y = 2^255: the second fold carries out of limb 2 and wraps once more
  setall(hin[20], 0);
  hin[20].v[0] = 0x87417e26u; hin[20].v[1] = 0x72cbf66fu; hin[20].v[2] = 0x65a6e2eau; hin[20].v[3] = 0x5fd11f73u;
  hin[20].v[4] = 0x5fba1f38u; hin[20].v[5] = 0x4030ce4bu; hin[20].v[6] = 0x02ead958u;
  setall(hin[21], 0); hin[21].v[7] = 0x80000000u;
  fp *din, *dres;
  CK(hipMalloc(&din, sizeof(fp) * nthreads * 2));
  CK(hipMalloc(&dres, sizeof(fp) * nthreads * 4));
  CK(hipMemcpy(din, hin.data(), sizeof(fp) * NCHK * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_fpcheck, dim3((NCHK + 255) / 256), dim3(256), 0, 0, din, dres, NCHK);
  std::vector<fp> hres(NCHK * 4);
  CK(hipMemcpy(hres.data(), dres, sizeof(fp) * NCHK * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < NCHK; ++i) {
    H256 a = from_fp(hin[2 * i]), b = from_fp(hin[2 * i + 1]);
    H256 e[4] = {hmul(a, b), hadd(a, b), hsub(a, b), hmul(hadd(a, b), hsub(a, b))};
    for (int k = 0; k < 4; ++k) {
      H256 g = from_fp(hres[4 * i + k]);
      if (hcmp(g, e[k]) != 0) {
        if (bad < 10) printf("MISMATCH case %d op %d\n", i, k);
        ++bad;
      }
    }
  }
  // also the host build of the same fp256.cuh functions
  int hbad = 0;
  for (int i = 0; i < NCHK; ++i) {
    fp a = hin[2 * i], b = hin[2 * i + 1];
    fp m = fp_canon(fp_mul(a, b));
    if (hcmp(from_fp(m), hmul(from_fp(a), from_fp(b))) != 0) ++hbad;
  }
  printf("fp256 device check: %d mismatches of %d; host-build check: %d mismatches\n", bad, NCHK * 4, hbad);


  // ---- fp29 correctness on device vs host reference ----
  {
    hipLaunchKernelGGL(k_f29check, dim3((NCHK + 255) / 256), dim3(256), 0, 0, din, dres, NCHK);
    CK(hipMemcpy(hres.data(), dres, sizeof(fp) * NCHK * 4, hipMemcpyDeviceToHost));
    int bad29 = 0;
    for (int i = 0; i < NCHK; ++i) {
      H256 a = hcanon(from_fp(hin[2 * i])), b = hcanon(from_fp(hin[2 * i + 1]));
      H256 ab = hmul(a, b);
      H256 a3 = hsub(hsub(hsub(a, ab), ab), ab);
      H256 e[4] = {ab, hadd(a, ab), hsub(a, ab), hmul(a3, b)};
      for (int k = 0; k < 4; ++k)
        if (hcmp(from_fp(hres[4 * i + k]), e[k]) != 0) {
          if (bad29 < 10) printf("F29 MISMATCH case %d op %d\n", i, k);
          ++bad29;
        }
    }
    printf("fp29 device check: %d mismatches of %d\n", bad29, NCHK * 4);
    bad += bad29;
  }

  // ---- fp256 throughput ----
  std::vector<fp> big(nthreads * 2);
  for (size_t i = 0; i < nthreads * 2; ++i)
    for (int j = 0; j < 8; ++j) big[i].v[j] = (uint32_t)rng();
  CK(hipMemcpy(din, big.data(), sizeof(fp) * nthreads * 2, hipMemcpyHostToDevice));
  for (int wpb = 1; wpb <= 8; wpb *= 2) {
    int bl = prop.multiProcessorCount * wpb;
    size_t nt = (size_t)bl * threads;
    double s1 = time_kernel([&] { hipLaunchKernelGGL(k_fpmul<1>, dim3(bl), dim3(threads), 0, 0, din, dres); });
    double s2 = time_kernel([&] { hipLaunchKernelGGL(k_fpmul<2>, dim3(bl), dim3(threads), 0, 0, din, dres); });
    double s3 = time_kernel([&] { hipLaunchKernelGGL(k_fpbfly, dim3(bl), dim3(threads), 0, 0, din, dres); });
    double s4 = time_kernel([&] { hipLaunchKernelGGL(k_fpaddsub, dim3(bl), dim3(threads), 0, 0, din, dres); });
    double s7 = time_kernel([&] { hipLaunchKernelGGL(k_blake, dim3(bl), dim3(threads), 0, 0, din, dres); });
    printf("blocks/CU=%d  BLAKE2s pair hashes: %7.2f G/s\n", wpb, (double)nt * 64 / s7 * 1e-9);
    double s5 = time_kernel([&] { hipLaunchKernelGGL(k_f29mul, dim3(bl), dim3(threads), 0, 0, din, dres); });
    double s6 = time_kernel([&] { hipLaunchKernelGGL(k_f29bfly, dim3(bl), dim3(threads), 0, 0, din, dres); });
    printf("blocks/CU=%d  modmul x1: %7.2f G/s   x2: %7.2f G/s   butterfly: %7.2f G/s   add+sub pair: %7.2f G/s | f29 mul: %7.2f G/s  f29 DIT butterfly: %7.2f G/s\n", wpb,
           (double)nt * FITERS / s1 * 1e-9, (double)nt * FITERS * 2 / s2 * 1e-9, (double)nt * FITERS / s3 * 1e-9,
           (double)nt * FITERS / s4 * 1e-9, (double)nt * FITERS / s5 * 1e-9, (double)nt * FITERS / s6 * 1e-9);
  }
  return bad || hbad;
}
