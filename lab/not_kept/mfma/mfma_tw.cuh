// mfma_tw.cuh -- (a - b) * w mod p for a twiddle w shared by 32 (or 64) lanes, with the 256 x 256-bit product on the
// matrix cores (v_mfma_i32_32x32x32_i8) instead of 64 v_mad_u64_u32 + 62 v_addc per lane (fp256.cuh:fp_mul).
// Same field as fp256.cuh (starks/modp.py:25-106); results are lazily reduced and congruent to fp_mul(fp_sub(a, b), w),
// so every canonical residue -- and therefore every byte the library returns -- is unchanged.
//
// x * w mod p = sum_k x_k * (w * 2^(8k) mod p) over the 32 bytes x_k of x: a 32 x 32 byte matrix (one per twiddle,
// precomputed on the host: shk_build_twmat) times the byte vector of x.  One MFMA applies it to 32 vectors.
//   * i8 operands are signed.  The matrix holds the signed digits in [-128, 127] of a representative of w * 2^(8k) mod p;
//     the data bytes are offset by 128: x ^ 0x80..80 read as signed bytes is x - E, E = 0x8080..80.  The butterfly needs
//     (a - b) * w = [W | -W] x [a - E ; b - E]: the offsets cancel and the subtraction costs nothing (two accumulating
//     MFMAs, the second with the digits of -w * 2^(8k)).
//   * the accumulators start from constants O_j = 2^20 + delta_j with sum_j O_j 2^(8j) == 0 (mod p): every partial sum
//     is non-negative (|sum| <= 64 * 128 * 128 = 2^20), so the carry chain is unsigned.
//   * lane l owns element l.  An MFMA column is spread over lanes (c, c + 32): v_permlane32_swap moves the upper 16 bytes
//     of the lower lanes' elements up and the lower 16 bytes of the upper lanes' elements down (4 swaps per operand);
//     MFMA group 1 covers the elements of lanes 0..31, group 2 those of lanes 32..63 (each group may use its own
//     twiddle); every lane normalises two half-results (16 partial sums at byte spacing -> 4 limbs + carry) and 5 swaps
//     bring the two halves of its own element back.
// Measured (tools/mfma_modmul.hip, profiles/r02_mfma_modmul_prototype.txt): bit-exact on 16.7 M (a, b, w) triples,
// 317 G butterflies/s against 174 G/s for the VALU butterfly.
#pragma once
#include "fp256.cuh"

typedef int shk_v4i __attribute__((ext_vector_type(4)));
typedef int shk_v16i __attribute__((ext_vector_type(16)));

// Per-twiddle operand image: lane l's 16 bytes of the matrix of w and of -w (A operand of the MFMA).
// Row i of the MFMA output (register r of lane-half h: i = (r & 3) + 8 (r >> 2) + 4 h) is byte position 16 h + r.
struct TwMat {
  uint32_t w[64][4];
  uint32_t nw[64][4];
};
static_assert(sizeof(TwMat) == 2048, "TwMat layout");

// v.upper32lanes <-> u.lower32lanes
__device__ __forceinline__ void shk_swap32(uint32_t& v, uint32_t& u) {
  auto r = __builtin_amdgcn_permlane32_swap(v, u, false, false);
  v = r[0];
  u = r[1];
}
// Accumulator offsets.  Row j of a product is a signed sum (|s_j| < 2^23); adding O_j = 2^20 + delta_j, where
// delta = -(2^20 * (2^256 - 1) / 255) mod p (little-endian bytes 17, 16, 240, 239, 160, 232, 217, then 239 for the remaining
// 25) makes every row non-negative without changing the residue: sum_j O_j 2^(8j) == 0 (mod p).  Rather than starting the
// MFMA chain from a 16-register block of offsets, the chain starts from zero and the offsets of a limb's four rows,
// K_m = sum_k O_(4m+k) 2^(8k), enter as the 64-bit addend of the first v_mad_i64_i32 of that limb (6 registers).
struct shk_kinit {
  uint64_t k0, k1, k23;  // limbs 0, 1 (differ between the lane halves) and limbs 2, 3
};
__device__ __forceinline__ shk_kinit shk_mfma_kinit(uint32_t lane) {
  const bool up = lane >= 32;
  const uint64_t K = (1ull << 20) * 0x01010101ull;
  shk_kinit k;
  k.k0 = K + (up ? 0xefefefefull : 0xeff01011ull);
  k.k1 = K + (up ? 0xefefefefull : 0xefd9e8a0ull);
  k.k23 = K + 0xefefefefull;
  return k;
}
__device__ __forceinline__ uint64_t shk_mad64s(uint32_t a, uint32_t k, uint64_t c) {
  uint64_t d;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c) : "vcc");
  return d;
}
// `asm volatile` keeps the order: two independent carry chains alternate, so neither waits on its own previous result
__device__ __forceinline__ uint64_t shk_mad64sv(uint32_t a, uint32_t k, uint64_t c) {
  uint64_t d;
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c) : "vcc");
  return d;
}
// 16 signed row sums at byte spacing (+ offsets) -> 4 limbs + carry (< 2^16)
__device__ __forceinline__ void shk_norm16k(const shk_v16i& s, const shk_kinit& kc, uint32_t out[5]) {
  uint32_t cin = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    // limb 0 starts with plain arithmetic on purpose: the hazard recogniser does not look inside inline asm, and the
    // first reader of an MFMA result must be an instruction it sees (it then inserts the wait states for the whole block)
    const uint32_t e = (uint32_t)s[4 * m] + cin;  // |s| < 2^23, cin < 2^24
    uint64_t t = m == 0 ? kc.k0 + (uint64_t)(int64_t)(int32_t)e : shk_mad64s(e, 1u, m == 1 ? kc.k1 : kc.k23);
    t = shk_mad64s((uint32_t)s[4 * m + 1], 1u << 8, t);
    t = shk_mad64s((uint32_t)s[4 * m + 2], 1u << 16, t);
    t = shk_mad64s((uint32_t)s[4 * m + 3], 1u << 24, t);
    out[m] = (uint32_t)t;
    cin = (uint32_t)(t >> 32);
  }
  out[4] = cin;
}
// the same for the two accumulators of a whole-form butterfly at once
__device__ __forceinline__ void shk_norm16x2k(const shk_v16i& s1, const shk_v16i& s2, const shk_kinit& kc, uint32_t o1[5],
                                              uint32_t o2[5]) {
  uint32_t c1 = 0, c2 = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const uint64_t k = m == 0 ? kc.k0 : m == 1 ? kc.k1 : kc.k23;
    const uint32_t e1 = (uint32_t)s1[4 * m] + c1, e2 = (uint32_t)s2[4 * m] + c2;
    uint64_t t1 = m == 0 ? k + (uint64_t)(int64_t)(int32_t)e1 : shk_mad64sv(e1, 1u, k);  // see shk_norm16k
    uint64_t t2 = m == 0 ? k + (uint64_t)(int64_t)(int32_t)e2 : shk_mad64sv(e2, 1u, k);
    t1 = shk_mad64sv((uint32_t)s1[4 * m + 1], 1u << 8, t1);
    t2 = shk_mad64sv((uint32_t)s2[4 * m + 1], 1u << 8, t2);
    t1 = shk_mad64sv((uint32_t)s1[4 * m + 2], 1u << 16, t1);
    t2 = shk_mad64sv((uint32_t)s2[4 * m + 2], 1u << 16, t2);
    t1 = shk_mad64sv((uint32_t)s1[4 * m + 3], 1u << 24, t1);
    t2 = shk_mad64sv((uint32_t)s2[4 * m + 3], 1u << 24, t2);
    o1[m] = (uint32_t)t1;
    o2[m] = (uint32_t)t2;
    c1 = (uint32_t)(t1 >> 32);
    c2 = (uint32_t)(t2 >> 32);
  }
  o1[4] = c1;
  o2[4] = c2;
}

// d = (a - b) * w.  MFMA group 1 (the elements of lanes 0..31) uses the matrices (w1, nw1), group 2 (lanes 32..63)
// uses (w2, nw2); all 64 lanes pass their own 16-byte fragment of both.  Every lane of the wave must be active.
__device__ __forceinline__ fp shk_mfma_submul2(const fp& a, const fp& b, const shk_v4i w1, const shk_v4i nw1, const shk_v4i w2,
                                               const shk_v4i nw2, const shk_kinit& kc) {
  uint32_t A[8], B[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    A[i] = a.v[i] ^ 0x80808080u;
    B[i] = b.v[i] ^ 0x80808080u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    shk_swap32(A[i], A[4 + i]);
    shk_swap32(B[i], B[4 + i]);
  }
  const shk_v4i a1 = {(int)A[0], (int)A[1], (int)A[2], (int)A[3]}, a2 = {(int)A[4], (int)A[5], (int)A[6], (int)A[7]};
  const shk_v4i b1 = {(int)B[0], (int)B[1], (int)B[2], (int)B[3]}, b2 = {(int)B[4], (int)B[5], (int)B[6], (int)B[7]};
  const shk_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  shk_v16i acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, a1, zero, 0, 0, 0);
  shk_v16i acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, a2, zero, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(nw1, b1, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(nw2, b2, acc2, 0, 0, 0);
  uint32_t r1[5], r2[5];
  shk_norm16x2k(acc1, acc2, kc, r1, r2);
#pragma unroll
  for (int i = 0; i < 5; ++i) shk_swap32(r1[i], r2[i]);
  // r1 = limbs 0..3 + carry into limb 4, r2 = limbs 4..7 + carry out (weight 2^256), all of this lane's own element
  fp r;
  uint32_t cy;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = r1[i];
  r.v[4] = fp_addc(r2[0], r1[4], 0, &cy);
#pragma unroll
  for (int i = 1; i < 4; ++i) r.v[4 + i] = fp_addc(r2[i], 0u, cy, &cy);
  const uint32_t T = r2[4] + cy;  // < 2^15
  // fold T * 2^256 == T * c,  T c = (T * 351 << 32) - T =: D (2 limbs, >= 0)
  uint32_t bd, c2;
  const uint32_t D0 = fp_subb(0u, T, 0, &bd);
  const uint32_t D1 = T * 351u - bd;
  r.v[0] = fp_addc(r.v[0], D0, 0, &c2);
  r.v[1] = fp_addc(r.v[1], D1, c2, &c2);
  r.v[2] = fp_addc(r.v[2], 0u, c2, &c2);
  if (FP_ANY(c2)) {  // carry out of limb 2: ~2^-41 per lane
#pragma unroll
    for (int i = 3; i < 8; ++i) r.v[i] = fp_addc(r.v[i], 0u, c2, &c2);
    fp_add_c_masked_low(r, c2);
  }
  return r;
}
__device__ __forceinline__ fp shk_mfma_submul(const fp& a, const fp& b, const shk_v4i w, const shk_v4i nw, const shk_kinit& kc) {
  return shk_mfma_submul2(a, b, w, nw, w, nw, kc);
}
__device__ __forceinline__ shk_v4i shk_ld_frag(const uint32_t (*rows)[4], uint32_t lane) {
  const uint4 q = *reinterpret_cast<const uint4*>(rows[lane]);
  const shk_v4i r = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
  return r;
}

// ---- host: build the operand image of one twiddle --------------------------------------------------------------------
#include <string.h>
namespace shk_twmat_detail {
inline void to_bytes_le(const fp& a, uint8_t b[32]) {
  for (int i = 0; i < 8; ++i)
    for (int k = 0; k < 4; ++k) b[4 * i + k] = (uint8_t)(a.v[i] >> (8 * k));
}
// signed digits d[0..31] in [-128, 127] with sum d_m 256^m == v (mod p), v canonical in [0, p).  Values above
// 0x7f7f..7f are represented through v - p (every residue has a representative in the 32-digit range, whose length is
// 2^256 - 1 > p).  Returns false if neither fits (cannot happen for canonical v).
inline bool signed_digits(const fp& v, int8_t d[32]) {
  uint8_t u[33];
  to_bytes_le(v, u);
  u[32] = 0;
  bool small = true;
  for (int m = 31; m >= 0; --m) {
    if (u[m] != 0x7f) {
      small = u[m] < 0x7f;
      break;
    }
  }
  if (!small) {  // v - p = v + c - 2^256: 33-byte two's complement with sign byte 0xff
    uint32_t cy = 0;
    fp t;
    t.v[0] = fp_addc(v.v[0], FP_C0, 0, &cy);
    t.v[1] = fp_addc(v.v[1], FP_C1, cy, &cy);
    for (int i = 2; i < 8; ++i) t.v[i] = fp_addc(v.v[i], 0u, cy, &cy);
    if (cy) return false;
    to_bytes_le(t, u);
    u[32] = 0xff;
  }
  int carry = 0;
  for (int m = 0; m < 32; ++m) {
    const int t = u[m] + carry;
    if (t >= 128) {
      d[m] = (int8_t)(t - 256);
      carry = 1;
    } else {
      d[m] = (int8_t)t;
      carry = 0;
    }
  }
  return (int)(int8_t)u[32] + carry == 0;
}
inline int rho(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }  // byte position of MFMA output row i
}  // namespace shk_twmat_detail

inline bool shk_build_twmat(const fp& w, TwMat* out) {
  using namespace shk_twmat_detail;
  fp t = fp_canon(w), nt = fp_canon(fp_neg(w));
  const fp k256 = fp_from_u32(256u);
  int8_t dw[32][32], dn[32][32];  // [kappa][digit]
  for (int kappa = 0; kappa < 32; ++kappa) {
    if (!signed_digits(t, dw[kappa]) || !signed_digits(nt, dn[kappa])) return false;
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
    memcpy(out->w[lane], bw, 16);
    memcpy(out->nw[lane], bn, 16);
  }
  return true;
}
