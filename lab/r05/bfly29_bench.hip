// lab/r05/bfly29_bench.hip -- round-5 experiment: radix-4 DIF register groups (two butterfly levels on four elements: four products by
// table constants, four adds, four subs) in the library's saturated 8 x 32-bit arithmetic (fp_add / fp_sub / fp_mul2) against the
// unsaturated 9 x 29-bit prototype (fp29b.cuh: f29_add / f29_sub / f29_norm / f29_mul2).  Both kernels run the same chain of groups on
// the same inputs; the final values are compared (as canonical residues) and the sustained rate is reported.
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc -I lab/r05 lab/r05/bfly29_bench.hip -o lab/r05/bfly29_bench && lab/r05/bfly29_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "fp29b.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int TW = 1024;  // twiddles per table (L1 / L2 resident, as in the tile passes)

__device__ unsigned long long* g_stamps;  // per wave: shader cycles spent in the loop (s_memtime), shader clock / real time ratio
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k32(const fp* in, const fp2* tw, fp* out, int iters) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  fp x0 = fp_load(in + 4 * g), x1 = fp_load(in + 4 * g + 1), x2 = fp_load(in + 4 * g + 2), x3 = fp_load(in + 4 * g + 3);
  uint32_t ti = (uint32_t)(g >> 6) * 2654435761u;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const uint32_t l = threadIdx.x & 63u;  // lanes take consecutive table entries (as the early levels of a tile pass do)
    const fp2 w1 = fp2_load(tw + ((ti + l) & (TW - 1))), w2 = fp2_load(tw + (((ti >> 10) + l) & (TW - 1)));
    // level 1: (x0, x2), (x1, x3)
    const fp s1 = fp_add(x0, x2), d1 = fp_mul2(fp_sub(x0, x2), w1);
    const fp s2 = fp_add(x1, x3), d2 = fp_mul2(fp_sub(x1, x3), w2);
    const fp2 w3 = fp2_load(tw + (((ti >> 5) + l) & (TW - 1))), w4 = fp2_load(tw + (((ti >> 15) + l) & (TW - 1)));
    // level 2: (s1, s2), (d1, d2)
    x0 = fp_add(s1, s2);
    x1 = fp_mul2(fp_sub(s1, s2), w3);
    x2 = fp_add(d1, d2);
    x3 = fp_mul2(fp_sub(d1, d2), w4);
    ti = ti * 1664525u + 1013904223u;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0 && g_stamps) { g_stamps[2 * (g >> 6)] = t1 - t0; g_stamps[2 * (g >> 6) + 1] = r1 - r0; }
  fp_store(out + 4 * g, x0); fp_store(out + 4 * g + 1, x1); fp_store(out + 4 * g + 2, x2); fp_store(out + 4 * g + 3, x3);
}

__device__ __forceinline__ f29w f29w_load(const f29w* p) {
  f29w r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
  r.w0.v[0] = a.x; r.w0.v[1] = a.y; r.w0.v[2] = a.z; r.w0.v[3] = a.w; r.w0.v[4] = b.x; r.w0.v[5] = b.y; r.w0.v[6] = b.z; r.w0.v[7] = b.w;
  r.w0.v[8] = c.x; r.w1.v[0] = c.y; r.w1.v[1] = c.z; r.w1.v[2] = c.w; r.w1.v[3] = d.x; r.w1.v[4] = d.y; r.w1.v[5] = d.z; r.w1.v[6] = d.w;
  r.w1.v[7] = e.x; r.w1.v[8] = e.y; r.pad[0] = r.pad[1] = 0;
  return r;
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k29(const fp* in, const f29w* tw, fp* out, f29c C1, f29c C2, int iters) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f29 x0 = f29_from_fp(fp_load(in + 4 * g)), x1 = f29_from_fp(fp_load(in + 4 * g + 1)), x2 = f29_from_fp(fp_load(in + 4 * g + 2)),
      x3 = f29_from_fp(fp_load(in + 4 * g + 3));
  uint32_t ti = (uint32_t)(g >> 6) * 2654435761u;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const uint32_t l = threadIdx.x & 63u;
    const f29w w1 = f29w_load(tw + ((ti + l) & (TW - 1))), w2 = f29w_load(tw + (((ti >> 10) + l) & (TW - 1)));
    // inputs fresh (limbs < 2^29 + 2^8)
    const f29 s1 = f29_add(x0, x2), d1 = f29_mul2(f29_sub(x0, x2, C1), w1);
    const f29 s2 = f29_add(x1, x3), d2 = f29_mul2(f29_sub(x1, x3, C1), w2);
    const f29w w3 = f29w_load(tw + (((ti >> 5) + l) & (TW - 1))), w4 = f29w_load(tw + (((ti >> 15) + l) & (TW - 1)));
    x0 = f29_norm(f29_add(s1, s2));            // < 2^31 + ..: back to fresh before it meets another sum
    x1 = f29_mul2(f29_sub(s1, s2, C2), w3);    // s2 limbs < 2^30 + 2^9
    x2 = f29_norm(f29_add(d1, d2));
    x3 = f29_mul2(f29_sub(d1, d2, C1), w4);
    ti = ti * 1664525u + 1013904223u;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0 && g_stamps) { g_stamps[2 * (g >> 6)] = t1 - t0; g_stamps[2 * (g >> 6) + 1] = r1 - r0; }
  fp_store(out + 4 * g, f29_to_fp(x0)); fp_store(out + 4 * g + 1, f29_to_fp(x1)); fp_store(out + 4 * g + 2, f29_to_fp(x2));
  fp_store(out + 4 * g + 3, f29_to_fp(x3));
}

static uint64_t s_ = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (uint32_t)(s_ >> 16); }

// lds: dynamic LDS per workgroup, only to set the resident workgroups per CU (36 KiB -> 4 = 4 waves per SIMD, 31 KiB -> 5), as the
// tile passes' own LDS image does
template <class K, class... A>
static double run(K k, size_t lds, int blocks, int iters, A... a) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, a..., iters);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, a..., iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // the last launch's stamps: median cycles per wave in the loop, and the shader clock it ran at (s_memrealtime ticks at 100 MHz)
  std::vector<unsigned long long> st(2 * (size_t)blocks * 4);
  unsigned long long* dptr; CK(hipMemcpyFromSymbol(&dptr, HIP_SYMBOL(g_stamps), sizeof dptr));
  CK(hipMemcpy(st.data(), dptr, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, ghz;
  double wave_seconds = 0;
  for (size_t i = 0; i < st.size() / 2; ++i) wave_seconds += (double)st[2 * i + 1] * 1e-8;
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, lds));
  printf(" [resident waves per SIMD: %.2f measured, %d by the occupancy query]", wave_seconds / (ms / 10 * 1e-3 * 1024), occ);
  for (size_t i = 0; i < st.size() / 2; ++i) { cyc.push_back((double)st[2 * i]); if (st[2 * i + 1]) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  printf(" [median %.0f cycles per wave per group, shader clock %.2f GHz]", cyc[cyc.size() / 2] / iters, ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
  return ms / 10;
}

int main(int argc, char** argv) {
  const int blocks = 256 * 20, N = blocks * 256 * 4, iters = argc > 1 ? atoi(argv[1]) : 300;
  std::vector<fp> x(N), o32(N), o29(N);
  std::vector<fp2> w(TW);
  std::vector<f29w> w29(TW);
  fp two128 = fp_zero(); two128.v[4] = 1;
  for (int i = 0; i < N; ++i) { for (int k = 0; k < 8; ++k) x[i].v[k] = rnd(); if ((i & 63) == 1) for (int k = 0; k < 8; ++k) x[i].v[k] = 0xffffffffu; if ((i & 63) == 2) x[i] = fp_zero(); }
  for (int i = 0; i < TW; ++i) {
    fp t; for (int k = 0; k < 8; ++k) t.v[k] = rnd();
    if (i == 1) for (int k = 0; k < 8; ++k) t.v[k] = 0xffffffffu;
    if (i == 2) t = fp_one();
    if (i == 3) t = fp_canon(fp_neg(fp_one()));
    w[i].w = fp_canon(t); w[i].w128 = fp_canon(fp_mul(w[i].w, two128));
    w29[i] = f29w_from_fp(w[i].w);
  }
  const f29c C1 = f29_sub_const(1u << 30), C2 = f29_sub_const((1u << 30) + (1u << 12));
  fp *dx, *d32, *d29; fp2* dw; f29w* dw29;
  CK(hipMalloc(&dx, sizeof(fp) * N)); CK(hipMalloc(&d32, sizeof(fp) * N)); CK(hipMalloc(&d29, sizeof(fp) * N));
  CK(hipMalloc(&dw, sizeof(fp2) * TW)); CK(hipMalloc(&dw29, sizeof(f29w) * TW));
  CK(hipMemcpy(dx, x.data(), sizeof(fp) * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, w.data(), sizeof(fp2) * TW, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw29, w29.data(), sizeof(f29w) * TW, hipMemcpyHostToDevice));
  unsigned long long* dst; CK(hipMalloc(&dst, 16 * (size_t)blocks * 4)); CK(hipMemset(dst, 0, 16 * (size_t)blocks * 4));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof dst));
  // correctness: a short chain, every output as a canonical residue
  hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, dx, dw, d32, 7);
  hipLaunchKernelGGL(k29<4>, dim3(blocks), dim3(256), 0, 0, dx, dw29, d29, C1, C2, 7);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o32.data(), d32, sizeof(fp) * N, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o29.data(), d29, sizeof(fp) * N, hipMemcpyDeviceToHost));
  long bad = 0;
  for (int i = 0; i < N; ++i) if (!fp_eq_canon(fp_canon(o32[i]), fp_canon(o29[i]))) { if (!bad) printf("first mismatch at %d\n", i); ++bad; }
  printf("{\"check_values\": %d, \"mismatches\": %ld,\n", N, bad);
  {
    hipFuncAttributes fa;
    const void* ks[4] = {reinterpret_cast<const void*>(k32<4>), reinterpret_cast<const void*>(k29<4>), reinterpret_cast<const void*>(k29<5>), reinterpret_cast<const void*>(k32<5>)};
    const char* names[4] = {"k32<4>", "k29<4>", "k29<5>", "k32<5>"};
    for (int i = 0; i < 4; ++i) {
      CK(hipFuncGetAttributes(&fa, ks[i]));
      printf(" \"%s\": {\"numRegs\": %d, \"static_lds\": %zu, \"scratch\": %zu, \"blocks_per_cu_by_lds_kib\": {", names[i], fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
      for (int kib : {0, 16, 31, 36}) {
        int occ = 0;
        CK(hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 40 << 10));
        if (i == 0) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k32<4>, 256, (size_t)kib << 10));
        if (i == 1) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k29<4>, 256, (size_t)kib << 10));
        if (i == 2) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k29<5>, 256, (size_t)kib << 10));
        if (i == 3) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k32<5>, 256, (size_t)kib << 10));
        printf("\"%d\": %d%s", kib, occ, kib == 36 ? "}},\n" : ", ");
      }
    }
  }
  const double groups = (double)blocks * 256 * iters;
  double m;
  m = run(k32<4>, 36 << 10, blocks, iters, dx, dw, d32);  printf(" \"fp32_4waves_G_groups_per_s\": %.2f,\n", groups / m / 1e6);
  m = run(k29<4>, 36 << 10, blocks, iters, dx, dw29, d29, C1, C2); printf(" \"fp29_4waves_G_groups_per_s\": %.2f,\n", groups / m / 1e6);
  m = run(k29<5>, 31 << 10, blocks, iters, dx, dw29, d29, C1, C2); printf(" \"fp29_5waves_G_groups_per_s\": %.2f,\n", groups / m / 1e6);
  m = run(k32<5>, 31 << 10, blocks, iters, dx, dw, d32);  printf(" \"fp32_5waves_G_groups_per_s\": %.2f}\n", groups / m / 1e6);
  return bad ? 1 : 0;
}
