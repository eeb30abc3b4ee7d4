// fp256.cuh -- arithmetic in Z/p, p = 2^256 - 351*2^32 + 1 (the "MiMC prime" of the reference,
// starks/utils.py:22, starks/modp.py:25-106), written for gfx950 integer VALUs.
//
// Representation: 8 x u32 little-endian limbs ("limb form").  Values are kept LAZILY reduced in
// [0, 2^256); because 2^256 == c (mod p) with c = 2^256 - p = 351*2^32 - 1 (41 bits), a carry out of
// limb 7 is folded back by adding c, and a 512-bit product hi*2^256 + lo folds to lo + hi*c.
// `fp_canon` maps to the unique residue in [0, p) -- done once where a value leaves the device, which
// makes results bit-identical to the reference's `int(n) % p` after every op (modp.py:36).
//
// Every function is total on [0, 2^256) inputs (double folds are handled), so unreduced inputs such as
// field(b'\xff'*32) (modp.py:33-34) are safe.
//
// All functions are __host__ __device__: the same code builds the twiddle tables on the host and is
// exercised on the CPU through the table builders; the device code is pinned by tests/test_gpu_parity.py
// (test_rare_carry_branches, the NTT / FRI / STARK parity tests) against the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FP_HD __host__ __device__ __forceinline__

// "does any lane of the wave see this rare condition?" -- a wave-uniform branch (s_cbranch on the ballot), so the
// common path skips the rare carry-propagation chains entirely.
#if defined(__HIP_DEVICE_COMPILE__)
#define FP_ANY(x) (__builtin_amdgcn_ballot_w64((x) != 0) != 0)
#else
#define FP_ANY(x) ((x) != 0)
#endif

struct fp {
  uint32_t v[8];
};

#define FP_P0 0x00000001u
#define FP_P1 0xfffffea1u
#define FP_PX 0xffffffffu /* limbs 2..7 of p */
#define FP_C0 0xffffffffu /* c = 2^256 - p, limb 0 */
#define FP_C1 0x0000015eu /* limb 1 (350) */

FP_HD uint32_t fp_addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t* cout) {
  return __builtin_addc(a, b, cin, cout);
}
FP_HD uint32_t fp_subb(uint32_t a, uint32_t b, uint32_t bin, uint32_t* bout) {
  return __builtin_subc(a, b, bin, bout);
}

FP_HD fp fp_zero() {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
FP_HD fp fp_one() {
  fp r = fp_zero();
  r.v[0] = 1;
  return r;
}
FP_HD fp fp_from_u32(uint32_t x) {
  fp r = fp_zero();
  r.v[0] = x;
  return r;
}

// r = a + (m ? c : 0) on all 8 limbs, returns carry out.  m is 0 or 1.
FP_HD uint32_t fp_add_c_masked(fp& a, uint32_t m) {
  uint32_t mask = 0u - m, cy;
  a.v[0] = fp_addc(a.v[0], FP_C0 & mask, 0, &cy);
  a.v[1] = fp_addc(a.v[1], FP_C1 & mask, cy, &cy);
#pragma unroll
  for (int i = 2; i < 8; ++i) a.v[i] = fp_addc(a.v[i], 0, cy, &cy);
  return cy;
}
// a += (m ? c : 0) when a is known to be < 2^96 - 2^42 (after a second wrap): 3 limbs suffice.
FP_HD void fp_add_c_masked_low(fp& a, uint32_t m) {
  uint32_t mask = 0u - m, cy;
  a.v[0] = fp_addc(a.v[0], FP_C0 & mask, 0, &cy);
  a.v[1] = fp_addc(a.v[1], FP_C1 & mask, cy, &cy);
  a.v[2] = fp_addc(a.v[2], 0, cy, &cy);
}
FP_HD uint32_t fp_sub_c_masked(fp& a, uint32_t m) {
  uint32_t mask = 0u - m, bw;
  a.v[0] = fp_subb(a.v[0], FP_C0 & mask, 0, &bw);
  a.v[1] = fp_subb(a.v[1], FP_C1 & mask, bw, &bw);
#pragma unroll
  for (int i = 2; i < 8; ++i) a.v[i] = fp_subb(a.v[i], 0, bw, &bw);
  return bw;
}

// (a + b) mod p, lazily reduced.  modp.py:43-45.
// A carry out of limb 7 is folded as + c (2^256 == c).  c has two limbs, so the fold touches limbs 0..1; a carry
// out of limb 1 needs limb 1 >= 2^32 - 351 (probability ~1e-7 per lane): the propagation through limbs 2..7 and
// the possible second wrap live behind a wave-uniform branch that is almost never taken.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SHK_NO_ADD_ASM)
#include "fp256_addasm.inc"  // fp_add_asm / fp_sub_asm: the common paths below as one asm statement each (gen_addasm.py)
__device__ __forceinline__ uint32_t fp_lane_bit(uint64_t mask) {
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  return (uint32_t)(mask >> lane) & 1u;
}
#define FP_ADD_ASM 1
#endif

FP_HD fp fp_add(const fp& a, const fp& b) {
  fp r;
#if defined(FP_ADD_ASM)
  const uint64_t any = fp_add_asm(a, b, r);
  if (any != 0) {  // scalar test: no VALU instruction on the common path
    uint32_t c1 = fp_lane_bit(any);
#pragma unroll
    for (int i = 2; i < 8; ++i) r.v[i] = fp_addc(r.v[i], 0, c1, &c1);
    fp_add_c_masked_low(r, c1);
  }
  return r;
#else
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = fp_addc(a.v[i], b.v[i], cy, &cy);
  const uint32_t m = 0u - cy;
  uint32_t c1;
  r.v[0] = fp_addc(r.v[0], FP_C0 & m, 0, &c1);
  r.v[1] = fp_addc(r.v[1], FP_C1 & m, c1, &c1);
  if (FP_ANY(c1)) {
#pragma unroll
    for (int i = 2; i < 8; ++i) r.v[i] = fp_addc(r.v[i], 0, c1, &c1);
    fp_add_c_masked_low(r, c1);  // second wrap leaves r < 2^42, cannot wrap again
  }
  return r;
#endif
}

// (a - b) mod p, lazily reduced.  modp.py:47-49.
// A borrow means the 256-bit difference stands for r - 2^256 == r - c: subtract c from limbs 0..1; a borrow out of
// limb 1 (limb 1 < 351, ~1e-7 per lane) is propagated behind a wave-uniform branch, where a second borrow (r was < c)
// means the value now is 2^256 - d with d < 2^41 and c is subtracted once more from limbs 0..1.
FP_HD fp fp_sub(const fp& a, const fp& b) {
  fp r;
#if defined(FP_ADD_ASM)
  const uint64_t any = fp_sub_asm(a, b, r);
  if (any != 0) {
    uint32_t b1 = fp_lane_bit(any);
#pragma unroll
    for (int i = 2; i < 8; ++i) r.v[i] = fp_subb(r.v[i], 0, b1, &b1);
    const uint32_t m2 = 0u - b1;
    uint32_t b3;
    r.v[0] = fp_subb(r.v[0], FP_C0 & m2, 0, &b3);
    r.v[1] = fp_subb(r.v[1], FP_C1 & m2, b3, &b3);
  }
  return r;
#else
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = fp_subb(a.v[i], b.v[i], bw, &bw);
  const uint32_t m = 0u - bw;
  uint32_t b1;
  r.v[0] = fp_subb(r.v[0], FP_C0 & m, 0, &b1);
  r.v[1] = fp_subb(r.v[1], FP_C1 & m, b1, &b1);
  if (FP_ANY(b1)) {
#pragma unroll
    for (int i = 2; i < 8; ++i) r.v[i] = fp_subb(r.v[i], 0, b1, &b1);
    const uint32_t m2 = 0u - b1;
    uint32_t b3;
    r.v[0] = fp_subb(r.v[0], FP_C0 & m2, 0, &b3);
    r.v[1] = fp_subb(r.v[1], FP_C1 & m2, b3, &b3);
  }
  return r;
#endif
}

// unique residue in [0, p): subtract p once if a >= p (a < 2^256 < 2p).
FP_HD fp fp_canon(const fp& a) {
  fp t;
  uint32_t bw;
  t.v[0] = fp_subb(a.v[0], FP_P0, 0, &bw);
  t.v[1] = fp_subb(a.v[1], FP_P1, bw, &bw);
#pragma unroll
  for (int i = 2; i < 8; ++i) t.v[i] = fp_subb(a.v[i], FP_PX, bw, &bw);
  fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = bw ? a.v[i] : t.v[i];
  return r;
}

FP_HD fp fp_neg(const fp& a) { return fp_sub(fp_zero(), a); }

// a / 4 mod p without a product: p == 1 (mod 4), so a + k p with k = (-a) mod 4 is divisible by 4, and (a + k p) >> 2 < 2^256 for
// any a in [0, 2^256) (a + 3 p < 2^258).  Lazily reduced in and out.  (The FRI fold's final 1/4: ~20 instructions against a
// general product's ~150.)
FP_HD fp fp_div4(const fp& a) {
  const uint32_t k = (0u - a.v[0]) & 3u;
  uint32_t t[9];
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t pi = i == 0 ? FP_P0 : i == 1 ? FP_P1 : FP_PX;
    acc += (uint64_t)a.v[i] + (uint64_t)k * pi;
    t[i] = (uint32_t)acc;
    acc >>= 32;
  }
  t[8] = (uint32_t)acc;
  fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (t[i] >> 2) | (t[i + 1] << 30);
  return r;
}

FP_HD bool fp_eq_canon(const fp& a, const fp& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
  return d == 0;
}

// 256x256 -> 512 bit schoolbook product, column (product-scanning) order: one 32x32+64 multiply-add
// (v_mad_u64_u32) plus one carry add per partial product, 96-bit running column accumulator.
FP_HD void fp_mul_wide(const uint32_t a[8], const uint32_t b[8], uint32_t t[16]) {
  uint64_t acc = 0;
  uint32_t top = 0;
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    const int lo = k < 8 ? 0 : k - 7;
    const int hi = k < 8 ? k : 7;
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      uint64_t pr = (uint64_t)a[i] * b[k - i];
      acc += pr;
      top += (acc < pr) ? 1u : 0u;
    }
    t[k] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)top << 32);
    top = 0;
  }
  t[15] = (uint32_t)acc;
}

// fold a 512-bit value t = hi*2^256 + lo into [0, 2^256):  t == lo + hi*c,  hi*c = (hi*351 << 32) - hi.
FP_HD fp fp_reduce_wide(const uint32_t t[16]) {
  // A = hi * 351  (9 limbs)
  uint32_t A[9];
  uint64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t m = (uint64_t)t[8 + i] * 351u + cy;
    A[i] = (uint32_t)m;
    cy = m >> 32;
  }
  A[8] = (uint32_t)cy;
  // R = lo + (A << 32) - hi        (10 limbs, non-negative because A<<32 >= hi)
  uint32_t R[10];
  uint32_t c1 = 0, b1 = 0;
  // limb 0: lo[0] + 0 - hi[0]
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t lo = i < 8 ? t[i] : 0u;
    uint32_t as = (i >= 1 && i <= 9) ? A[i - 1] : 0u;
    uint32_t hi = i < 8 ? t[8 + i] : 0u;
    uint32_t s = fp_addc(lo, as, c1, &c1);
    R[i] = fp_subb(s, hi, b1, &b1);
  }
  // (c1 and b1 cancel at the top: R < 2^298 fits in 10 limbs, so final c1 == b1.)
  // second fold: h2 = R[8] + R[9]*2^32 (< 2^42);  r = R[0..7] + h2*c,  h2*c = (h2*351 << 32) - h2 =: D (3 limbs, >= 0).
  // D is formed first so that the long chain is an addition (adding zero limbs with carry is one v_addc each;
  // subtracting zero limbs with borrow is not, see fp_sub).
  uint64_t h2 = (uint64_t)R[8] | ((uint64_t)R[9] << 32);
  uint64_t B = h2 * 351u;  // < 2^51
  uint32_t D[3], bd = 0;
  D[0] = fp_subb(0u, R[8], 0, &bd);
  D[1] = fp_subb((uint32_t)B, R[9], bd, &bd);
  D[2] = (uint32_t)(B >> 32) - bd;
  fp r;
  uint32_t c2 = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) r.v[i] = fp_addc(R[i], D[i], c2, &c2);
#pragma unroll
  for (int i = 3; i < 8; ++i) r.v[i] = R[i];
  if (FP_ANY(c2)) {  // carry out of limb 2: ~1e-4 per lane (D[2] < 2^19)
#pragma unroll
    for (int i = 3; i < 8; ++i) r.v[i] = fp_addc(r.v[i], 0u, c2, &c2);
    fp_add_c_masked_low(r, c2);  // after a wrap r < 2^84: three limbs suffice
  }
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
#ifndef FP_MULASM_INC
#define FP_MULASM_INC "fp256_mulasm.inc"
#endif
#include FP_MULASM_INC  // fp_mul_wide_asm: the same product as fp_mul_wide, hand-scheduled for gfx950
#endif

// (a * b) mod p, lazily reduced.  modp.py:51-53.
FP_HD fp fp_mul(const fp& a, const fp& b) {
  uint32_t t[16];
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul_wide_asm(a.v, b.v, t);
#else
  fp_mul_wide(a.v, b.v, t);
#endif
  return fp_reduce_wide(t);
}

FP_HD fp fp_sqr(const fp& a) { return fp_mul(a, a); }

// ---- product by a table constant kept as a PAIR (w, w * 2^128 mod p) ---------------------------------------------------------
// x * w == x_lo * w + x_hi * (w 2^128 mod p) with x = x_lo + 2^128 x_hi: the same 64 partial products as fp_mul, in 11
// columns instead of 15, and a 385-bit result, so that only 129 bits have to be folded instead of 256: 5 multiplications by 351
// and 7-limb chains instead of 8 and 10-limb ones (about 150 instead of 190 instructions per product).  Every twiddle of a
// butterfly comes from a table, so the second image costs 32 bytes of (cache-resident) table per twiddle and nothing else.
struct fp2 {
  fp w, w128;  // w (any representative below 2^256) and w * 2^128 mod p
};
FP_HD void fp_mul2_wide(const uint32_t x[8], const uint32_t w0[8], const uint32_t w1[8], uint32_t t[13]) {
  uint64_t acc = 0;
  uint32_t top = 0;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = k - i;
      if (j >= 0 && j <= 7) {
        uint64_t pr = (uint64_t)x[i] * w0[j];
        acc += pr;
        top += (acc < pr) ? 1u : 0u;
        pr = (uint64_t)x[4 + i] * w1[j];
        acc += pr;
        top += (acc < pr) ? 1u : 0u;
      }
    }
    t[k] = (uint32_t)acc;
    if (k < 10) {
      acc = (acc >> 32) | ((uint64_t)top << 32);
      top = 0;
    }
  }
  t[11] = (uint32_t)(acc >> 32);
  t[12] = top;
}
// fold t = hi * 2^256 + lo, hi = t[8..12] < 2^129, into [0, 2^256): lo + hi c, hi c = (hi * 351 << 32) - hi =: D >= 0.
// hi * 351 < 2^138 has five limbs, so D has six (D[5] < 2^10) and the carry of lo + D leaves limb 5 with probability ~2^-22.
FP_HD fp fp_reduce_13(const uint32_t t[13]) {
  uint32_t A[5];
  uint64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint64_t m = (uint64_t)t[8 + i] * 351u + cy;
    A[i] = (uint32_t)m;
    cy = m >> 32;  // 0 after the last limb
  }
  uint32_t D[6], bd = 0;
  D[0] = fp_subb(0u, t[8], 0, &bd);
#pragma unroll
  for (int i = 1; i < 5; ++i) D[i] = fp_subb(A[i - 1], t[8 + i], bd, &bd);
  D[5] = A[4] - bd;
  fp r;
  uint32_t c = 0;
#if defined(FP_ADD_ASM)
  const uint64_t any = fp_add6_asm(t, D, r.v);
  r.v[6] = t[6];
  r.v[7] = t[7];
  if (any != 0) {
    c = fp_lane_bit(any);
#else
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = fp_addc(t[i], D[i], c, &c);
  r.v[6] = t[6];
  r.v[7] = t[7];
  if (FP_ANY(c)) {
#endif
    r.v[6] = fp_addc(r.v[6], 0u, c, &c);
    r.v[7] = fp_addc(r.v[7], 0u, c, &c);
    // c = 1: the sum passed 2^256 == c; what is left is below 2^171, so adding c cannot wrap again
    const uint32_t mask = 0u - c;
    uint32_t c2;
    r.v[0] = fp_addc(r.v[0], FP_C0 & mask, 0, &c2);
    r.v[1] = fp_addc(r.v[1], FP_C1 & mask, c2, &c2);
#pragma unroll
    for (int i = 2; i < 8; ++i) r.v[i] = fp_addc(r.v[i], 0u, c2, &c2);
  }
  return r;
}
// x * w mod p, lazily reduced, w given as a pair
FP_HD fp fp_mul2(const fp& x, const fp2& w) {
  uint32_t t[13];
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul2_wide_asm(x.v, w.w.v, w.w128.v, t);
#else
  fp_mul2_wide(x.v, w.w.v, w.w128.v, t);
#endif
  return fp_reduce_13(t);
}

// a^e for a 64-bit exponent (square-and-multiply, numbertype.py:68-84 computes the same value).
FP_HD fp fp_pow_u64(fp a, uint64_t e) {
  fp r = fp_one();
  while (e) {
    if (e & 1) r = fp_mul(r, a);
    a = fp_sqr(a);
    e >>= 1;
  }
  return r;
}

// a^-1 = a^(p-2) (modp.py:71-79 uses extended Euclid; the residue is the same).  0 -> 0.
FP_HD fp fp_inv(const fp& a) {
  fp r = fp_one(), b = a;
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    // p - 2 = 2^256 - 351 * 2^32 - 1: limb 0 = 0xffffffff, limb 1 = 0xfffffea0, limbs 2..7 = 0xffffffff
    const uint32_t limb = (i >> 5) == 1 ? 0xfffffea0u : 0xffffffffu;
    if ((limb >> (i & 31)) & 1u) r = fp_mul(r, b);
    b = fp_sqr(b);
  }
  return r;
}

// ---- wire form <-> limb form ------------------------------------------------------------------
// Wire form = 32 bytes big-endian (modp.py:94-95).  Word k of the wire form read as a little-endian
// u32 is bswap(limb[7-k]).
FP_HD uint32_t fp_bswap32(uint32_t x) { return __builtin_bswap32(x); }

FP_HD fp fp_from_wire_words(const uint32_t w[8]) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = fp_bswap32(w[7 - i]);
  return r;
}
FP_HD void fp_to_wire_words(const fp& a, uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = fp_bswap32(a.v[7 - i]);
}

// ---- 32-byte element loads/stores (two dwordx4 per lane) ------------------------------------------
__device__ __forceinline__ fp fp_load(const fp* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ fp2 fp2_load(const fp2* p) {
  fp2 r;
  r.w = fp_load(&p->w);
  r.w128 = fp_load(&p->w128);
  return r;
}
__device__ __forceinline__ void fp_store(fp* p, const fp& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

