// fp29b.cuh -- round-5 prototype: unsaturated arithmetic in Z/p with 9 UNSIGNED limbs of 29 bits, for the inside of the NTT tile passes.
//
// Why: with saturated 8 x 32-bit limbs every partial product costs a v_mad_u64_u32 AND a carry catch (v_addc_co_u32), both slow-rate
// instructions, and add / sub are 8-link carry chains with two wait states per link.  With 29-bit limbs in 32-bit registers a column of
// up to 9 products of (< 2^31.5) x (< 2^29) fits a 64-bit accumulator: NO carry catch at all, the inter-column carry is the next
// column's initial addend (one 64-bit shift + one mask per column), add / sub are 9 full-rate v_add_u32 / v_sub_u32.
// Round 1 tried 9 x 29 signed digits with a full 17-column product and an 18-digit fold (lab/r01_r03/fp29.cuh: level with the 32-bit
// form).  New here: the table constant comes as a PAIR of images (w, w 2^145 mod p) -- x = x_lo + 2^145 x_hi, five + four limbs -- so the
// product has 13 columns instead of 17 and only FIVE high digits to fold, and the fold constant 2^261 mod p = 89855 * 2^29 + (2^29 - 32)
// has two positive 29-bit digits: 81 + 10 multiplies, 13 shifts, no carry instruction.
//
// value(x) = sum x.v[i] 2^(29 i)  (mod p);  "fresh": every limb < 2^29 + 2^8;  product inputs: limbs < 2^31.5 (F29_XMAX).
#pragma once
#include "fp256.cuh"

struct f29 {
  uint32_t v[9];
};
struct f29w {  // table constant: canonical 29-bit digits of w and of w * 2^145 mod p; 80 bytes (five 16-byte loads)
  f29 w0, w1;
  uint32_t pad[2];
};
#define F29_M 0x1fffffffu
#define F29_CLO 0x1fffffe0u /* 2^29 - 32 */
#define F29_CHI 89855u      /* 2^261 mod p = F29_CHI * 2^29 + F29_CLO */

// 8 x 32 (any value < 2^256) -> 9 x 29 digits (exact, canonical digits): digit i = bits [29 i, 29 i + 29), spelled out with constant
// shifts (a loop over the word index leaves the compiler indexing the limbs dynamically, through scratch / LDS)
FP_HD uint32_t f29_funnel(uint32_t hi, uint32_t lo, int s) {  // bits [s, s + 32) of hi:lo, 0 < s < 32
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
FP_HD f29 f29_from_fp(const fp& a) {
  f29 r;
  r.v[0] = a.v[0] & F29_M;
  r.v[1] = f29_funnel(a.v[1], a.v[0], 29) & F29_M;
  r.v[2] = f29_funnel(a.v[2], a.v[1], 26) & F29_M;
  r.v[3] = f29_funnel(a.v[3], a.v[2], 23) & F29_M;
  r.v[4] = f29_funnel(a.v[4], a.v[3], 20) & F29_M;
  r.v[5] = f29_funnel(a.v[5], a.v[4], 17) & F29_M;
  r.v[6] = f29_funnel(a.v[6], a.v[5], 14) & F29_M;
  r.v[7] = f29_funnel(a.v[7], a.v[6], 11) & F29_M;
  r.v[8] = a.v[7] >> 8;  // bits 232..255
  return r;
}
// digits (limbs < 2^32, value < 2^266) -> 8 x 32 lazily reduced (< 2^256)
FP_HD fp f29_to_fp(const f29& x) {
  // exact 29-bit digits d0..d8 and what is left above 2^261
  uint32_t d0, d1, d2, d3, d4, d5, d6, d7, d8, c;
  d0 = x.v[0] & F29_M; c = x.v[0] >> 29;
  uint32_t t;
  t = x.v[1] + c; d1 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[2] + c; d2 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[3] + c; d3 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[4] + c; d4 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[5] + c; d5 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[6] + c; d6 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[7] + c; d7 = t & F29_M; c = (t >> 29) + (t < c ? 8u : 0u);
  t = x.v[8] + c; d8 = t & 0xffffffu;
  const uint32_t hi = (t >> 24) + (t < c ? 256u : 0u);  // multiples of 2^256 (t wrapped: 2^32 >> 24)
  fp r;
  r.v[0] = d0 | (d1 << 29);
  r.v[1] = (d1 >> 3) | (d2 << 26);
  r.v[2] = (d2 >> 6) | (d3 << 23);
  r.v[3] = (d3 >> 9) | (d4 << 20);
  r.v[4] = (d4 >> 12) | (d5 << 17);
  r.v[5] = (d5 >> 15) | (d6 << 14);
  r.v[6] = (d6 >> 18) | (d7 << 11);
  r.v[7] = (d7 >> 21) | (d8 << 8);
  // + hi * c256, c256 = 2^256 mod p = 351 * 2^32 - 1
  fp tt = fp_zero();
  const uint64_t m = (uint64_t)hi * 351u;
  tt.v[1] = (uint32_t)m;
  tt.v[2] = (uint32_t)(m >> 32);
  fp h = fp_zero();
  h.v[0] = hi;
  return fp_sub(fp_add(r, tt), h);
}

FP_HD f29 f29_add(const f29& a, const f29& b) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a - b + C, C == 0 (mod p) with every limb >= the bound of b's limbs: K = 0: b limbs < 2^30; K = 1: b limbs < 2^30 + 2^12
// C_i = base + e_i, sum e_i 2^(29 i) = -base * sum 2^(29 i) mod p (canonical digits; tools: f29_sub_const below builds them on the host)
struct f29c {
  uint32_t c[9];
};
FP_HD f29 f29_sub(const f29& a, const f29& b, const f29c& C) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] - b.v[i] + C.c[i];
  return r;
}
// limbs < 2^32 -> fresh (limbs 0..7 < 2^29, limb 8 < 2^29 + 8), same residue: what the top limb holds above 2^261 is folded back
// first (2^261 == F29_CHI 2^29 + F29_CLO), then one carry pass
FP_HD f29 f29_norm(const f29& a) {
  f29 r;
  const uint32_t h = a.v[8] >> 29;  // < 8
  uint64_t c = (uint64_t)a.v[0] + (uint64_t)h * F29_CLO;
  r.v[0] = (uint32_t)c & F29_M;
  c = (c >> 29) + a.v[1] + (uint64_t)h * F29_CHI;
  r.v[1] = (uint32_t)c & F29_M;
  uint32_t cc = (uint32_t)(c >> 29);  // < 2^4
#pragma unroll
  for (int i = 2; i < 8; ++i) {
    const uint32_t t = a.v[i] + cc;   // a.v[i] < 2^32 - 16 by every caller's bound
    r.v[i] = t & F29_M;
    cc = t >> 29;
  }
  r.v[8] = (a.v[8] & F29_M) + cc;
  return r;
}

// x (limbs < 2^31.5) times the table constant w -> fresh.  Columns as templates on the column index: every limb index is a compile-time
// constant (a plain unrolled loop left the digit array dynamically indexed, which the compiler then parks in LDS).
template <int K>
FP_HD uint64_t f29_col(uint64_t acc, const f29& x, const f29w& w) {  // acc += sum of the products of column K
#define F29_T0(I) if constexpr (K - (I) >= 0 && K - (I) < 9) acc += (uint64_t)x.v[I] * w.w0.v[K - (I) >= 0 && K - (I) < 9 ? K - (I) : 0];
#define F29_T1(I) if constexpr (K - (I) >= 0 && K - (I) < 9) acc += (uint64_t)x.v[5 + (I)] * w.w1.v[K - (I) >= 0 && K - (I) < 9 ? K - (I) : 0];
  F29_T0(0) F29_T1(0) F29_T0(1) F29_T1(1) F29_T0(2) F29_T1(2) F29_T0(3) F29_T1(3) F29_T0(4)
#undef F29_T0
#undef F29_T1
  return acc;
}
FP_HD f29 f29_mul2(const f29& x, const f29w& w) {
  uint64_t acc = 0;
  // columns 6..12 first (their carry chain starts at column 6), so that the digits 9..13 are known when the low columns fold them in
  acc = f29_col<6>(acc, x, w);  const uint32_t r6 = (uint32_t)acc & F29_M;  acc >>= 29;
  acc = f29_col<7>(acc, x, w);  const uint32_t r7 = (uint32_t)acc & F29_M;  acc >>= 29;
  acc = f29_col<8>(acc, x, w);  const uint32_t r8 = (uint32_t)acc & F29_M;  acc >>= 29;
  acc = f29_col<9>(acc, x, w);  const uint32_t r9 = (uint32_t)acc & F29_M;  acc >>= 29;
  acc = f29_col<10>(acc, x, w); const uint32_t r10 = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<11>(acc, x, w); const uint32_t r11 = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<12>(acc, x, w); const uint32_t r12 = (uint32_t)acc & F29_M; acc >>= 29;
  const uint32_t r13 = (uint32_t)acc;  // < 2^32
  f29 o;
  acc = f29_col<0>(0, x, w) + (uint64_t)r9 * F29_CLO;
  o.v[0] = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<1>(acc, x, w) + (uint64_t)r10 * F29_CLO + (uint64_t)r9 * F29_CHI;
  o.v[1] = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<2>(acc, x, w) + (uint64_t)r11 * F29_CLO + (uint64_t)r10 * F29_CHI;
  o.v[2] = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<3>(acc, x, w) + (uint64_t)r12 * F29_CLO + (uint64_t)r11 * F29_CHI;
  o.v[3] = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<4>(acc, x, w) + (uint64_t)r13 * F29_CLO + (uint64_t)r12 * F29_CHI;
  o.v[4] = (uint32_t)acc & F29_M; acc >>= 29;
  acc = f29_col<5>(acc, x, w) + (uint64_t)r13 * F29_CHI;
  o.v[5] = (uint32_t)acc & F29_M; acc >>= 29;
  // what is left of the low chain (< 2^36) goes into limbs 6 and 7
  const uint32_t t = r6 + ((uint32_t)acc & F29_M);
  o.v[6] = t & F29_M;
  o.v[7] = r7 + (uint32_t)(acc >> 29) + (t >> 29);
  o.v[8] = r8;
  return o;
}

// ---- host helpers (tables) ----------------------------------------------------------------------------------------------------
inline f29w f29w_from_fp(const fp& w) {
  // 2^145 = 2^128 * 2^17
  fp t = fp_zero();
  t.v[4] = 1u << 17;
  f29w r;
  r.pad[0] = r.pad[1] = 0;
  r.w0 = f29_from_fp(fp_canon(w));
  r.w1 = f29_from_fp(fp_canon(fp_mul(fp_canon(w), t)));
  return r;
}
inline f29c f29_sub_const(uint32_t base) {
  // E = -base * sum_i 2^(29 i) mod p
  fp s = fp_zero(), b29 = fp_zero(), pw = fp_one();
  b29.v[0] = 1u << 29;
  for (int i = 0; i < 9; ++i) {
    s = fp_add(s, pw);
    pw = fp_mul(pw, b29);
  }
  const fp E = fp_canon(fp_neg(fp_mul(s, fp_from_u32(base))));
  const f29 e = f29_from_fp(E);
  f29c r;
  for (int i = 0; i < 9; ++i) r.c[i] = base + e.v[i];
  return r;
}
