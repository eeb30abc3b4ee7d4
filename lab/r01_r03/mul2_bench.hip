// tools/mul2_bench.hip -- the product by a table constant kept as a pair (w, w 2^128 mod p) (fp256.cuh: fp_mul2) against
// fp_mul on the GPU: bit-exact check (canonical forms) and butterfly throughput.
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc tools/mul2_bench.hip -o tools/mul2_bench && tools/mul2_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "fp256.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_check(const fp* x, const fp2* w, fp* o1, fp* o2, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const fp a = fp_load(x + g);
  fp2 ww;
  ww.w = fp_load(&w[g].w);
  ww.w128 = fp_load(&w[g].w128);
  fp_store(o1 + g, fp_canon(fp_mul(a, ww.w)));
  fp_store(o2 + g, fp_canon(fp_mul2(a, ww)));
}
constexpr int ITERS = 512;
__global__ void __launch_bounds__(256) k_bfly1(const fp* in, const fp2* w, fp* out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const fp tw = fp_load(&w[g & 1023].w);
  fp x = fp_load(in + g), y = fp_load(in + g + 1);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) {
    const fp s = fp_add(x, y);
    const fp d = fp_mul(fp_sub(x, y), tw);
    x = s;
    y = d;
  }
  fp_store(out + g, fp_add(x, y));
}
__global__ void __launch_bounds__(256) k_bfly2(const fp* in, const fp2* w, fp* out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  fp2 tw;
  tw.w = fp_load(&w[g & 1023].w);
  tw.w128 = fp_load(&w[g & 1023].w128);
  fp x = fp_load(in + g), y = fp_load(in + g + 1);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) {
    const fp s = fp_add(x, y);
    const fp d = fp_mul2(fp_sub(x, y), tw);
    x = s;
    y = d;
  }
  fp_store(out + g, fp_add(x, y));
}

static uint64_t s_ = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (uint32_t)(s_ >> 16); }

int main() {
  const int N = 1 << 22;
  std::vector<fp> x(N), o1(N), o2(N);
  std::vector<fp2> w(N);
  fp two128 = fp_zero();
  two128.v[4] = 1;
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 8; ++k) { x[i].v[k] = rnd(); w[i].w.v[k] = rnd(); }
    const int m = i & 31;
    if (m == 1) for (int k = 0; k < 8; ++k) x[i].v[k] = 0xffffffffu;
    if (m == 2) for (int k = 0; k < 8; ++k) w[i].w.v[k] = 0xffffffffu;
    if (m == 3) for (int k = 0; k < 8; ++k) { x[i].v[k] = 0xffffffffu; w[i].w.v[k] = 0xffffffffu; }
    if (m == 4) x[i] = fp_zero();
    if (m == 5) for (int k = 0; k < 7; ++k) x[i].v[k] = 0xffffffffu;
    if (m == 6) for (int k = 4; k < 8; ++k) x[i].v[k] = 0xffffffffu;
    if (m == 7) { w[i].w = fp_zero(); w[i].w.v[0] = 1; }
    w[i].w128 = (m & 8) ? fp_mul(w[i].w, two128) : fp_canon(fp_mul(w[i].w, two128));
  }
  fp *dx, *d1, *d2;
  fp2* dw;
  CK(hipMalloc(&dx, sizeof(fp) * (N + 1)));
  CK(hipMalloc(&d1, sizeof(fp) * N));
  CK(hipMalloc(&d2, sizeof(fp) * N));
  CK(hipMalloc(&dw, sizeof(fp2) * N));
  CK(hipMemset(dx, 0, sizeof(fp) * (N + 1)));
  CK(hipMemcpy(dx, x.data(), sizeof(fp) * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, w.data(), sizeof(fp2) * N, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check, dim3(N / 256), dim3(256), 0, 0, dx, dw, d1, d2, N);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o1.data(), d1, sizeof(fp) * N, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o2.data(), d2, sizeof(fp) * N, hipMemcpyDeviceToHost));
  long bad = 0, badh = 0;
  for (int i = 0; i < N; ++i) {
    if (memcmp(&o1[i], &o2[i], sizeof(fp))) ++bad;
    if ((i & 63) == 0) {  // and against the host's portable product
      const fp h = fp_canon(fp_mul(x[i], w[i].w));
      if (memcmp(&h, &o2[i], sizeof(fp))) ++badh;
    }
  }
  printf("check: %d products, fp_mul2 vs fp_mul on the device: %ld mismatches; vs the host product (1 in 64): %ld\n", N, bad, badh);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = 256 * 20;
  for (int v = 0; v < 2; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      if (v == 0) hipLaunchKernelGGL(k_bfly1, dim3(blocks), dim3(256), 0, 0, dx, dw, d1);
      else hipLaunchKernelGGL(k_bfly2, dim3(blocks), dim3(256), 0, 0, dx, dw, d1);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("butterfly with %s: %.3f ms  %.2f G/s\n", v == 0 ? "fp_mul          " : "fp_mul2 (pair w)", best,
           (double)blocks * 256 * ITERS / best / 1e6);
  }
  return bad || badh;
}
