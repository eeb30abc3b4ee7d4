// fp29.cuh -- unsaturated arithmetic in Z/p for the inside of the NTT tile kernels: 9 signed limbs of 29 bits.
//
// Why (measured on gfx950, tools/ubench.hip): v_add_co/v_addc_co_u32 issue at HALF rate and a VALU read of a
// carry written by the previous VALU needs 2 wait states, so carry chains are the expensive part of saturated
// 8 x 32-bit arithmetic (a 256-bit add + fold = ~23 half-rate instructions, a product needs one v_addc per
// v_mad_u64_u32).  With 29-bit limbs in 32-bit registers
//   * add / sub are 9 full-rate v_add_u32 / v_sub_u32, no carries, no reduction (limbs just grow);
//   * the product is 81 v_mad_i64_i32 into 64-bit columns that cannot overflow (9 * 2^31 * 2^28 < 2^63),
//     no carry catch at all; carries are resolved once per column by shift / mask;
//   * in a decimation-in-TIME butterfly (a, b) -> (a + w b, a - w b) one operand of every add is a freshly
//     normalised product, so limb magnitudes grow by 2^28 per level (not x2): a whole 8-level tile transform
//     needs no intermediate normalisation.
// Reduction: 2^261 == 32 (2^256) == 32 c = 89856 * 2^29 - 32 (mod p), c = 2^256 - p = 351 * 2^32 - 1, so a
// digit h of weight 2^(261 + 29 j) folds into columns j (x -32) and j + 1 (x 89856).
//
// Bounds are validated by the bit-accurate model tools/fp29_model.py (asserts every intermediate).
// Element invariants:  "lazy"  : |limb| < 2^31, value == x (mod p)
//                      "fresh" : limbs 0..7 in [-2^28 - 2^18, 2^28 + 2^18], |limb 8| <= 2^28  (mul output)
// Twiddles are stored as BALANCED digits w_j in [-2^28, 2^28) (9 x i32, padded to 12 words = 48 B).
#pragma once
#include "fp256.cuh"

struct fp29 {
  int32_t v[9];
};

#define F29_MASK 0x1fffffff
#define F29_BIAS 0x10000000 /* 2^28 */
#define F29_K1 89856
#define F29_K0 (-32)

// saturated 8 x 32 (value < 2^256) -> 9 unsigned 29-bit limbs
FP_HD fp29 f29_from_fp(const fp& a) {
  fp29 r;
  r.v[0] = (int32_t)(a.v[0] & F29_MASK);
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    // bits [29 i, 29 i + 29) straddle words (29 i) / 32 and the next one
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)a.v[w] | ((uint64_t)(w + 1 < 8 ? a.v[w + 1] : 0u) << 32);
    r.v[i] = (int32_t)((uint32_t)(two >> s) & F29_MASK);
  }
  r.v[8] = (int32_t)(a.v[7] >> 8);  // bits 232..255
  return r;
}

// canonical residue (< p) -> balanced digits in [-2^28, 2^28)   (host: twiddle tables)
FP_HD fp29 f29_balanced_from_fp(const fp& a) {
  fp29 u = f29_from_fp(a);
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int32_t d = u.v[i] + c;
    c = 0;
    if (d >= F29_BIAS) {
      d -= (F29_MASK + 1);
      c = 1;
    }
    u.v[i] = d;
  }
  return u;  // c == 0: the top digit of a value < 2^256 is < 2^24
}

FP_HD fp29 f29_add(const fp29& a, const fp29& b) {
  fp29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
FP_HD fp29 f29_sub(const fp29& a, const fp29& b) {
  fp29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] - b.v[i];
  return r;
}

// acc += a * b (signed 32 x 32 -> 64): one v_mad_i64_i32.  hipcc otherwise splits part of these into
// v_mad_u64_u32 + v_mul_lo_u32 sign corrections.
FP_HD int64_t f29_mad(int64_t acc, int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t junk;
  int64_t out;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(out), "=s"(junk) : "v"(a), "v"(b), "v"(acc));
  return out;
#else
  return acc + (int64_t)a * (int64_t)b;
#endif
}
// a * b with no addend (first product of a high column)
FP_HD int64_t f29_mul32(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t junk;
  int64_t out;
  asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(out), "=s"(junk) : "v"(a), "v"(b));
  return out;
#else
  return (int64_t)a * (int64_t)b;
#endif
}

// x lazy (|x_i| < 2^31), w balanced twiddle -> fresh product x * w (mod p)
FP_HD fp29 f29_mul(const fp29& x, const fp29& w) {
  int64_t C[18];
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    int64_t acc = (int64_t)F29_BIAS;
    bool first = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = k - i;
      if (j >= 0 && j < 9) {
        acc = (first && k >= 9) ? f29_mul32(x.v[i], w.v[j]) : f29_mad(acc, x.v[i], w.v[j]);
        first = false;
      }
    }
    C[k] = acc;
  }
  C[17] = 0;
  // high half -> digits h_0..h_8 in [0, 2^29), h_9 small signed
  int32_t h[10];
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    h[k - 9] = (int32_t)((uint32_t)C[k] & F29_MASK);
    C[k + 1] += C[k] >> 29;
  }
  h[8] = (int32_t)((uint32_t)C[17] & F29_MASK);
  h[9] = (int32_t)(C[17] >> 29);
  // fold 2^261 * H
  int64_t L[11];
#pragma unroll
  for (int k = 0; k < 9; ++k) L[k] = C[k];
  L[9] = F29_BIAS;
  L[10] = F29_BIAS;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    L[j] = f29_mad(L[j], h[j], F29_K0);
    L[j + 1] = f29_mad(L[j + 1], h[j], F29_K1);
  }
  // low pass to balanced digits
  int32_t r[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    r[k] = (int32_t)((uint32_t)L[k] & F29_MASK) - F29_BIAS;
    L[k + 1] += L[k] >> 29;
  }
  const int32_t o0 = r[9];
  const int32_t o1 = (int32_t)(L[10] - F29_BIAS);
  // second fold of O = o0 + 2^29 o1 (weight 2^261) into limbs 0..2, ripple into limb 3
  // (r[k] + BIAS is the masked, non-negative digit: it zero-extends to 64 bits for free)
  int64_t t0 = f29_mad((int64_t)(uint32_t)(r[0] + F29_BIAS), o0, F29_K0);
  int64_t t1 = f29_mad(f29_mad((int64_t)(uint32_t)(r[1] + F29_BIAS), o0, F29_K1), o1, F29_K0);
  int64_t t2 = f29_mad((int64_t)(uint32_t)(r[2] + F29_BIAS), o1, F29_K1);
  fp29 out;
  out.v[0] = (int32_t)((uint32_t)t0 & F29_MASK) - F29_BIAS;
  t1 += t0 >> 29;
  out.v[1] = (int32_t)((uint32_t)t1 & F29_MASK) - F29_BIAS;
  t2 += t1 >> 29;
  out.v[2] = (int32_t)((uint32_t)t2 & F29_MASK) - F29_BIAS;
  out.v[3] = r[3] + (int32_t)(t2 >> 29);
#pragma unroll
  for (int k = 4; k < 9; ++k) out.v[k] = r[k];
  return out;
}

// digits back to balanced form without changing the value (limb 8 absorbs the carry)
FP_HD fp29 f29_normalize(const fp29& a) {
  fp29 r;
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int32_t t = a.v[k] + c + F29_BIAS;
    r.v[k] = (int32_t)((uint32_t)t & F29_MASK) - F29_BIAS;
    c = t >> 29;
  }
  r.v[8] = a.v[8] + c;
  return r;
}

// lazy element (|value| < 2^264) -> saturated 8 x 32 limbs in [0, 2^256), congruent mod p
FP_HD fp f29_to_fp(const fp29& a) {
  // + 512 p so that the value is positive; 512 p = 2^265 - 512 c + ... in 29-bit digits:
  //   512 p = 2^265 - 351 * 2^41 + 512
  // digits (weight 2^(29 i)): i = 0: 512; i = 1: -(351 * 2^41 >> 29 = 351 * 2^12 = 1437696) ; top (i = 8): 2^33
  int64_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = a.v[i];
  t[0] += 512;
  t[1] -= 1437696;
  t[8] += (int64_t)1 << 33;
  uint32_t r[9];
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t v = t[k] + c;
    r[k] = (uint32_t)v & F29_MASK;
    c = v >> 29;
  }
  const uint64_t top = (uint64_t)(t[8] + c);  // < 2^34
  const uint32_t hi = (uint32_t)(top >> 24);  // bits >= 256  (< 2^10)
  r[8] = (uint32_t)top & 0xffffffu;
  // repack 9 x 29 -> 8 x 32
  fp s;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    // word w holds bits [32 w, 32 w + 32): digits floor(32 w / 29) and the next
    const int bit = 32 * w, d = bit / 29, sh = bit - 29 * d;
    uint64_t two = (uint64_t)r[d] | ((uint64_t)(d + 1 < 9 ? r[d + 1] : 0u) << 29) |
                   ((uint64_t)(d + 2 < 9 ? r[d + 2] : 0u) << 58);
    s.v[w] = (uint32_t)(two >> sh);
  }
  // + hi * c   (c = 351 * 2^32 - 1):  s += (hi * 351) << 32 ; s -= hi  -- via the saturated helpers
  fp add = fp_zero();
  const uint64_t hc = (uint64_t)hi * 351u;  // < 2^19
  add.v[1] = (uint32_t)hc;
  fp subt = fp_zero();
  subt.v[0] = hi;
  return fp_sub(fp_add(s, add), subt);
}
