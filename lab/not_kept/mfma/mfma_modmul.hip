// mfma_modmul.hip -- prototype: (a - b) * w mod p for a twiddle w SHARED by the 64 lanes of a wave, with the 256x256-bit
// product on the matrix cores (v_mfma_i32_32x32x32_i8) instead of 64 v_mad_u64_u32 + 62 v_addc per lane.  Not product
// code: it sizes the idea against fp_mul (starks_amd/csrc/fp256.cuh) and checks it bit for bit.
//
//   hipcc -O3 --offload-arch=gfx950 -I starks_amd/csrc tools/mfma_modmul.hip -o tools/mfma_modmul && tools/mfma_modmul
//
// Idea.  x * w mod p = sum_k x_k * (w * 2^(8k) mod p) over the 32 bytes x_k of x: a 32 x 32 byte-matrix (one per twiddle,
// precomputed) times the byte vector of x.  One MFMA multiplies that matrix by 32 such vectors.  Details:
//   * i8 operands are signed.  The matrix holds signed digits in [-128, 127] of a representative of w * 2^(8k) mod p in
//     (-2^255 - 2^247, 2^255 - 2^247]; the data bytes are offset by 128 (x ^ 0x80..80 read as signed bytes is x - E,
//     E = 0x8080..80).  The butterfly needs (a - b) * w, computed as [W | -W] x [a - E ; b - E]: the offsets cancel, and the
//     subtraction is free (two accumulating MFMAs, the second with the digits of -w * 2^(8k)).
//   * the accumulators start from constants O_j >= 2^20 with sum_j O_j 2^(8j) == 0 (mod p), so every partial sum is
//     non-negative and the carry chain is unsigned.
//   * lane l owns element l.  An MFMA column is spread over lanes (c, c + 32): v_permlane32_swap moves the upper 16 bytes of
//     the lower lanes' elements up and the lower 16 bytes of the upper lanes' elements down (4 swaps per operand), two MFMA
//     groups cover the 64 elements, each lane normalises two half-results (16 partial sums -> 4 limbs + carry) and 5 swaps
//     bring the halves of its own element back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fp256.cuh"
#include "mfma_split.cuh"

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// v.upper32lanes <-> u.lower32lanes
__device__ __forceinline__ void swap32(uint32_t& v, uint32_t& u) {
  auto r = __builtin_amdgcn_permlane32_swap(v, u, false, false);
  v = r[0];
  u = r[1];
}

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t k, uint64_t c) {
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c) : "vcc");
  return d;
}

// 16 non-negative partial sums at byte spacing -> 4 limbs + carry
__device__ __forceinline__ void norm16(const v16i& s, uint32_t out[5]) {
  uint32_t cin = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const uint32_t e = (uint32_t)s[4 * m] + cin;
    uint64_t t = mad64((uint32_t)s[4 * m + 1], 1u << 8, (uint64_t)e);
    t = mad64((uint32_t)s[4 * m + 2], 1u << 16, t);
    t = mad64((uint32_t)s[4 * m + 3], 1u << 24, t);
    out[m] = (uint32_t)t;
    cin = (uint32_t)(t >> 32);
  }
  out[4] = cin;
}

// (a - b) * w, lazily reduced.  wf / nwf: this lane's 16 bytes of the matrices of w and -w; cinit: its 16 offsets.
__device__ __forceinline__ fp mfma_submul(const fp& a, const fp& b, const v4i wf, const v4i nwf, const v16i cinit) {
  uint32_t A[8], B[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    A[i] = a.v[i] ^ 0x80808080u;
    B[i] = b.v[i] ^ 0x80808080u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    swap32(A[i], A[4 + i]);
    swap32(B[i], B[4 + i]);
  }
  const v4i a1 = {(int)A[0], (int)A[1], (int)A[2], (int)A[3]}, a2 = {(int)A[4], (int)A[5], (int)A[6], (int)A[7]};
  const v4i b1 = {(int)B[0], (int)B[1], (int)B[2], (int)B[3]}, b2 = {(int)B[4], (int)B[5], (int)B[6], (int)B[7]};
  v16i acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, a1, cinit, 0, 0, 0);
  v16i acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, a2, cinit, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(nwf, b1, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(nwf, b2, acc2, 0, 0, 0);
  uint32_t r1[5], r2[5];
  norm16(acc1, r1);
  norm16(acc2, r2);
#pragma unroll
  for (int i = 0; i < 5; ++i) swap32(r1[i], r2[i]);
  // r1 = limbs 0..3 + carry into limb 4, r2 = limbs 4..7 + carry out (weight 2^256), all of this lane's element
  fp r;
  uint32_t cy;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = r1[i];
  r.v[4] = fp_addc(r2[0], r1[4], 0, &cy);
#pragma unroll
  for (int i = 1; i < 4; ++i) r.v[4 + i] = fp_addc(r2[i], 0u, cy, &cy);
  const uint32_t T = r2[4] + cy;  // < 2^15
  // fold T * 2^256 == T * c,  T c = (T * 351 << 32) - T =: D (2 limbs)
  uint32_t bd, c2;
  const uint32_t D0 = fp_subb(0u, T, 0, &bd);
  const uint32_t D1 = T * 351u - bd;
  r.v[0] = fp_addc(r.v[0], D0, 0, &c2);
  r.v[1] = fp_addc(r.v[1], D1, c2, &c2);
  r.v[2] = fp_addc(r.v[2], 0u, c2, &c2);
  if (FP_ANY(c2)) {
#pragma unroll
    for (int i = 3; i < 8; ++i) r.v[i] = fp_addc(r.v[i], 0u, c2, &c2);
    fp_add_c_masked_low(r, c2);
  }
  return r;
}

struct Frags {
  v4i w[64], nw[64];
  v16i cinit[64];
};

__global__ void __launch_bounds__(64) k_check(const fp* a, const fp* b, const Frags* f, fp* out, fp* ref, fp w, int n) {
  const int l = threadIdx.x;
  const v4i wf = f->w[l], nwf = f->nw[l];
  const v16i ci = f->cinit[l];
  for (int base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
    const fp x = fp_load(a + base + l), y = fp_load(b + base + l);
    fp_store(out + base + l, fp_canon(mfma_submul(x, y, wf, nwf, ci)));
    fp_store(ref + base + l, fp_canon(fp_mul(fp_sub(x, y), w)));
  }
}

// ---- split form (mfma_split.cuh): lane c holds limbs 0..3, lane c + 32 limbs 4..7 of element c ---------------------------
__device__ __forceinline__ hf hf_load(const fp* p, bool hb) {
  const uint4 q = reinterpret_cast<const uint4*>(p)[hb ? 1 : 0];
  hf r;
  r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  return r;
}
__device__ __forceinline__ void hf_store(fp* p, const hf& r, bool hb) {
  reinterpret_cast<uint4*>(p)[hb ? 1 : 0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
// out[3n .. 3n+2] = a + b, a - b, (a - b) w in split arithmetic (lazily reduced; the host canonicalises)
__global__ void __launch_bounds__(64) k_check_split(const fp* a, const fp* b, const Frags* f, fp* out, int n) {
  const int l = threadIdx.x, c = l & 31;
  const bool hb = l >= 32;
  const v4i wf = f->w[l], nwf = f->nw[l];
  const shk_kinit ci = shk_mfma_kinit(l);
  for (int base = blockIdx.x * 32; base < n; base += gridDim.x * 32) {
    const hf x = hf_load(a + base + c, hb), y = hf_load(b + base + c, hb);
    hf_store(out + 3 * (size_t)(base + c), hf_add(x, y, hb), hb);
    hf_store(out + 3 * (size_t)(base + c) + 1, hf_sub(x, y, hb), hb);
    hf_store(out + 3 * (size_t)(base + c) + 2, hf_submul(x, y, wf, nwf, ci, hb), hb);
  }
}
// split <-> whole round trip: out[n] = the element as the lower / upper lane sees it after hf_to_whole(A = a, B = b)
__global__ void __launch_bounds__(64) k_check_convert(const fp* a, const fp* b, fp* out, int n) {
  const int l = threadIdx.x, c = l & 31;
  const bool hb = l >= 32;
  for (int base = blockIdx.x * 32; base < n; base += gridDim.x * 32) {
    const hf x = hf_load(a + base + c, hb), y = hf_load(b + base + c, hb);
    const fp w = hf_to_whole(x, y);  // lower lane: a whole, upper lane: b whole
    fp_store(out + 2 * (size_t)(base + c) + (hb ? 1 : 0), w);
  }
}
__global__ void __launch_bounds__(256) k_bfly_split(const fp* in, const Frags* f, fp* out) {
  const int l = threadIdx.x & 63, g = blockIdx.x * 256 + threadIdx.x;
  const bool hb = l >= 32;
  const v4i wf = f->w[l], nwf = f->nw[l];
  const shk_kinit ci = shk_mfma_kinit(l);
  hf x = hf_load(in + (g >> 1), hb), y = hf_load(in + (g >> 1) + 1, hb);
#pragma unroll 1
  for (int i = 0; i < 512; ++i) {
    const hf s = hf_add(x, y, hb);
    const hf d = hf_submul(x, y, wf, nwf, ci, hb);
    x = s;
    y = d;
  }
  hf_store(out + (g >> 1), hf_add(x, y, hb), hb);
}

constexpr int ITERS = 512;
__global__ void __launch_bounds__(256) k_tput_mfma(const fp* in, const Frags* f, fp* out) {
  const int l = threadIdx.x & 63, g = blockIdx.x * 256 + threadIdx.x;
  const v4i wf = f->w[l], nwf = f->nw[l];
  const v16i ci = f->cinit[l];
  fp x = fp_load(in + g), y = fp_load(in + g + 1);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) {
    fp z = mfma_submul(x, y, wf, nwf, ci);
    y = x;
    x = z;
  }
  fp_store(out + g, x);
}
__global__ void __launch_bounds__(256) k_tput_valu(const fp* in, fp* out, fp w) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  fp x = fp_load(in + g), y = fp_load(in + g + 1);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) {
    fp z = fp_mul(fp_sub(x, y), w);
    y = x;
    x = z;
  }
  fp_store(out + g, x);
}
// the whole radix-2 DIF butterfly: (a, b) -> (a + b, (a - b) w)
__global__ void __launch_bounds__(256) k_bfly_mfma(const fp* in, const Frags* f, fp* out) {
  const int l = threadIdx.x & 63, g = blockIdx.x * 256 + threadIdx.x;
  const v4i wf = f->w[l], nwf = f->nw[l];
  const v16i ci = f->cinit[l];
  fp x = fp_load(in + g), y = fp_load(in + g + 1);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) {
    fp s = fp_add(x, y);
    fp d = mfma_submul(x, y, wf, nwf, ci);
    x = s;
    y = d;
  }
  fp_store(out + g, fp_add(x, y));
}
__global__ void __launch_bounds__(256) k_bfly_valu(const fp* in, fp* out, fp w) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  fp x = fp_load(in + g), y = fp_load(in + g + 1);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) {
    fp s = fp_add(x, y);
    fp d = fp_mul(fp_sub(x, y), w);
    x = s;
    y = d;
  }
  fp_store(out + g, fp_add(x, y));
}

// ---- host: matrices ---------------------------------------------------------------------------------------------
static void fp_to_bytes_le(const fp& a, uint8_t b[32]) {
  for (int i = 0; i < 8; ++i)
    for (int k = 0; k < 4; ++k) b[4 * i + k] = (uint8_t)(a.v[i] >> (8 * k));
}
// signed digits d[0..31] in [-128, 127] with sum d_m 256^m == v (mod p), v canonical in [0, p)
static void signed_digits(const fp& v, int8_t d[32]) {
  uint8_t u[33];
  fp_to_bytes_le(v, u);
  u[32] = 0;
  bool small = true;  // v <= 0x7f7f..7f ?
  for (int m = 31; m >= 0; --m) {
    if (u[m] != 0x7f) {
      small = u[m] < 0x7f;
      break;
    }
  }
  if (!small) {  // use v - p = v + c - 2^256: 33-byte two's complement
    fp c = fp_zero();
    c.v[0] = FP_C0;
    c.v[1] = FP_C1;
    uint32_t cy = 0;
    fp t;
    for (int i = 0; i < 8; ++i) t.v[i] = fp_addc(v.v[i], c.v[i], cy, &cy);
    if (cy) {
      fprintf(stderr, "signed_digits: v + c overflowed (v not canonical?)\n");
      exit(1);
    }
    fp_to_bytes_le(t, u);
    u[32] = 0xff;
  }
  int carry = 0;
  for (int m = 0; m < 32; ++m) {
    int t = u[m] + carry;
    if (t >= 128) {
      d[m] = (int8_t)(t - 256);
      carry = 1;
    } else {
      d[m] = (int8_t)t;
      carry = 0;
    }
  }
  if ((int)(int8_t)u[32] + carry != 0) {
    fprintf(stderr, "signed_digits: no 32-digit form\n");
    exit(1);
  }
}
static int rho(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }  // byte position of MFMA output row i

static void build_frags(const fp& w, Frags* f) {
  const fp wc = fp_canon(w), nwc = fp_canon(fp_neg(w));
  fp t = wc, nt = nwc;
  const fp k256 = fp_from_u32(256u);
  int8_t dw[32][32], dn[32][32];  // [kappa][digit]
  for (int kappa = 0; kappa < 32; ++kappa) {
    signed_digits(t, dw[kappa]);
    signed_digits(nt, dn[kappa]);
    t = fp_canon(fp_mul(t, k256));
    nt = fp_canon(fp_mul(nt, k256));
  }
  for (int lane = 0; lane < 64; ++lane) {
    const int i = lane & 31, h = lane >> 5;
    uint8_t bw[16], bn[16];
    for (int j = 0; j < 16; ++j) {
      bw[j] = (uint8_t)dw[16 * h + j][rho(i)];
      bn[j] = (uint8_t)dn[16 * h + j][rho(i)];
    }
    memcpy(&f->w[lane], bw, 16);
    memcpy(&f->nw[lane], bn, 16);
  }
  // offsets: O_j = 2^20 + byte_j(delta), delta == -(2^20 * sum_j 256^j) (mod p)
  fp base = fp_zero(), pw = fp_one();
  const fp k20 = fp_from_u32(1u << 20);
  for (int j = 0; j < 32; ++j) {
    base = fp_add(base, fp_mul(k20, pw));
    pw = fp_canon(fp_mul(pw, k256));
  }
  uint8_t delta[32];
  fp_to_bytes_le(fp_canon(fp_neg(base)), delta);
  for (int lane = 0; lane < 64; ++lane) {
    const int h = lane >> 5;
    int32_t c[16];
    for (int r = 0; r < 16; ++r) c[r] = (1 << 20) + delta[16 * h + r];
    memcpy(&f->cinit[lane], c, 64);
  }
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static fp rnd_fp() {
  fp r;
  for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)rng();
  return r;
}

template <class F>
static double time_kernel(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int N = 1 << 20;
  std::vector<fp> ha(N), hb(N);
  for (int i = 0; i < N; ++i) {
    ha[i] = rnd_fp();
    hb[i] = rnd_fp();
  }
  // crafted operands: extremes, values around p, zero, equal operands, byte patterns that stress the signed offset
  const uint32_t pat[] = {0u, 1u, 0xffffffffu, 0x80808080u, 0x7f7f7f7fu, 0x80000000u, 0x00000080u, 0xfffffea1u, 0xfffffea0u, 0x0000015eu};
  int ix = 0;
  for (uint32_t pa : pat)
    for (uint32_t pb : pat) {
      for (int k = 0; k < 8; ++k) {
        ha[ix].v[k] = pa;
        hb[ix].v[k] = pb;
      }
      ++ix;
      for (int k = 0; k < 8; ++k) {
        ha[ix].v[k] = k == 1 ? pa : 0xffffffffu;
        hb[ix].v[k] = k == 0 ? pb : 0u;
      }
      ++ix;
    }
  for (int i = ix; i < ix + 64; ++i) hb[i] = fp_zero();  // plain x * w
  fp *da, *db, *dout, *dref;
  Frags* df;
  CK(hipMalloc(&da, (N + 64) * sizeof(fp)));
  CK(hipMalloc(&db, (N + 64) * sizeof(fp)));
  CK(hipMalloc(&dout, (N + 64) * sizeof(fp)));
  CK(hipMalloc(&dref, (N + 64) * sizeof(fp)));
  CK(hipMalloc(&df, sizeof(Frags)));
  CK(hipMemcpy(da, ha.data(), N * sizeof(fp), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * sizeof(fp), hipMemcpyHostToDevice));
  std::vector<fp> out(N), ref(N);
  long bad = 0, total = 0;
  // twiddles: random ones, 1, -1, 2, the 2^32-th root of unity powers, values with extreme digits
  std::vector<fp> ws;
  ws.push_back(fp_one());
  ws.push_back(fp_canon(fp_neg(fp_one())));
  ws.push_back(fp_from_u32(2u));
  ws.push_back(fp_from_u32(0x80u));
  {
    fp t;
    for (int k = 0; k < 8; ++k) t.v[k] = 0x7f7f7f7fu;
    ws.push_back(t);
    for (int k = 0; k < 8; ++k) t.v[k] = 0x80808080u;
    ws.push_back(fp_canon(t));
  }
  for (int i = 0; i < 10; ++i) ws.push_back(fp_canon(rnd_fp()));
  Frags hf;
  for (const fp& w : ws) {
    build_frags(w, &hf);
    CK(hipMemcpy(df, &hf, sizeof hf, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1024), dim3(64), 0, 0, da, db, df, dout, dref, w, N);
    CK(hipGetLastError());
    CK(hipMemcpy(out.data(), dout, N * sizeof(fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref.data(), dref, N * sizeof(fp), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) {
      ++total;
      if (memcmp(&out[i], &ref[i], sizeof(fp))) {
        if (bad < 5) {
          printf("MISMATCH i=%d\n  got ", i);
          for (int k = 7; k >= 0; --k) printf("%08x", out[i].v[k]);
          printf("\n  ref ");
          for (int k = 7; k >= 0; --k) printf("%08x", ref[i].v[k]);
          printf("\n");
        }
        ++bad;
      }
    }
  }
  printf("check: %ld (a, b, w) triples, %ld mismatches against fp_mul(fp_sub(a, b), w)\n", total, bad);
  if (bad) return 1;

  // ---- split form: a + b, a - b, (a - b) w against the whole-form functions, and the split <-> whole conversion -----------
  {
    // rare paths: carries that ripple through a whole half and across the top
    fp ones;
    for (int k = 0; k < 8; ++k) ones.v[k] = 0xffffffffu;
    fp lowones = fp_zero();
    for (int k = 0; k < 4; ++k) lowones.v[k] = 0xffffffffu;
    ha[ix] = ones; hb[ix] = fp_one(); ++ix;
    ha[ix] = lowones; hb[ix] = fp_one(); ++ix;
    ha[ix] = fp_zero(); hb[ix] = fp_one(); ++ix;          // 0 - 1: borrows through everything
    ha[ix] = fp_one(); hb[ix] = ones; ++ix;
    { fp t = fp_zero(); t.v[4] = 1; ha[ix] = t; hb[ix] = fp_one(); ++ix; }   // 2^128 - 1: borrow stops at limb 4
    { fp t = ones; t.v[0] = 0xfffffea0u; ha[ix] = t; hb[ix] = t; ++ix; }
    CK(hipMemcpy(da, ha.data(), N * sizeof(fp), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * sizeof(fp), hipMemcpyHostToDevice));
    fp* d3;
    CK(hipMalloc(&d3, (size_t)3 * N * sizeof(fp)));
    std::vector<fp> o3((size_t)3 * N);
    long sbad = 0, stot = 0;
    for (size_t wi = 0; wi < ws.size(); wi += 5) {
      const fp w = ws[wi];
      build_frags(w, &hf);
      CK(hipMemcpy(df, &hf, sizeof hf, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_check_split, dim3(1024), dim3(64), 0, 0, da, db, df, d3, N);
      CK(hipGetLastError());
      CK(hipMemcpy(o3.data(), d3, (size_t)3 * N * sizeof(fp), hipMemcpyDeviceToHost));
      for (int i = 0; i < N; ++i) {
        const fp want[3] = {fp_canon(fp_add(ha[i], hb[i])), fp_canon(fp_sub(ha[i], hb[i])), fp_canon(fp_mul(fp_sub(ha[i], hb[i]), w))};
        for (int k = 0; k < 3; ++k) {
          ++stot;
          const fp got = fp_canon(o3[(size_t)3 * i + k]);
          if (memcmp(&got, &want[k], sizeof(fp))) {
            if (sbad < 5) printf("SPLIT MISMATCH i=%d op=%d\n", i, k);
            ++sbad;
          }
        }
      }
    }
    hipLaunchKernelGGL(k_check_convert, dim3(1024), dim3(64), 0, 0, da, db, d3, N);
    CK(hipMemcpy(o3.data(), d3, (size_t)2 * N * sizeof(fp), hipMemcpyDeviceToHost));
    long cbad = 0;
    for (int i = 0; i < N; ++i)
      if (memcmp(&o3[(size_t)2 * i], &ha[i], sizeof(fp)) || memcmp(&o3[(size_t)2 * i + 1], &hb[i], sizeof(fp))) ++cbad;
    printf("split form: %ld results (a + b, a - b, (a - b) w), %ld mismatches; split <-> whole: %ld mismatches of %d\n", stot, sbad,
           cbad, N);
    if (sbad || cbad) return 1;
    CK(hipFree(d3));
  }

  // throughput
  const int blocks = 256 * 8;
  const double ops = (double)blocks * 256 * ITERS;
  build_frags(ws.back(), &hf);
  CK(hipMemcpy(df, &hf, sizeof hf, hipMemcpyHostToDevice));
  const fp w = ws.back();
  double t;
  t = time_kernel([&] { hipLaunchKernelGGL(k_tput_valu, dim3(blocks), dim3(256), 0, 0, da, dout, w); });
  printf("(a-b)*w   VALU  fp_mul(fp_sub):   %8.3f ms  %7.2f G/s\n", t, ops / t / 1e6);
  t = time_kernel([&] { hipLaunchKernelGGL(k_tput_mfma, dim3(blocks), dim3(256), 0, 0, da, df, dout); });
  printf("(a-b)*w   MFMA i8 shared twiddle: %8.3f ms  %7.2f G/s\n", t, ops / t / 1e6);
  t = time_kernel([&] { hipLaunchKernelGGL(k_bfly_valu, dim3(blocks), dim3(256), 0, 0, da, dout, w); });
  printf("butterfly VALU:                   %8.3f ms  %7.2f G/s\n", t, ops / t / 1e6);
  t = time_kernel([&] { hipLaunchKernelGGL(k_bfly_mfma, dim3(blocks), dim3(256), 0, 0, da, df, dout); });
  printf("butterfly MFMA i8 shared twiddle: %8.3f ms  %7.2f G/s\n", t, ops / t / 1e6);
  t = time_kernel([&] { hipLaunchKernelGGL(k_bfly_split, dim3(blocks), dim3(256), 0, 0, da, df, dout); });
  printf("butterfly MFMA, split form:       %8.3f ms  %7.2f G/s  (32 butterflies per wave instruction group)\n", t, ops / 2 / t / 1e6);
  return 0;
}
