// mfma_split.cuh -- field arithmetic on elements split over a lane pair: lane c (< 32) holds limbs 0..3 of the element of
// column c, lane c + 32 holds limbs 4..7 ("split form").  This is the MFMA's own operand layout (mfma_tw.cuh: an MFMA column
// is spread over lanes (c, c + 32)), so a butterfly needs no lane swaps and one accumulator; a lane carries 4 registers per
// element instead of 8, which lets a tile pass run four waves per SIMD instead of two (tools/not_kept/ntt_mfma2.hip:
// measured 25-30 % slower than the VALU passes, DESIGN.md section 5 -- this header is prototype material, not library code).  The price is that
// every carry between the halves crosses lanes (hf_fix / hf_fix_sub).
// Values are lazily reduced exactly as in fp256.cuh: a split element stands for the same residue as the whole-form one.
#pragma once
#include "mfma_tw.cuh"

struct hf {
  uint32_t v[4];
};

// the partner lane's (l ^ 32) value of x
__device__ __forceinline__ uint32_t hf_partner(uint32_t x, bool hb) {
  uint32_t a = x, b = x;
  shk_swap32(a, b);  // a.upper <- b.lower (old), b.lower <- a.upper (old)
  return hb ? a : b;
}

// Pending carries.  A lower lane's p has weight 2^128: its partner adds it at limb 4.  An upper lane's p has weight
// 2^256 == c = 351 * 2^32 - 1: its partner adds p * c at limbs 0..1.  One round resolves all but ~2^-60 of the cases;
// the loop is wave-uniform.  p < 2^23.
__device__ __forceinline__ void hf_fix(hf& r, uint32_t p, bool hb) {
  do {
    const uint32_t in = hf_partner(p, hb);
    uint32_t bd, cy;
    const uint32_t neg = fp_subb(0u, in, 0, &bd);  // -in, bd = (in != 0)
    const uint32_t d0 = hb ? in : neg;             // upper: + in at limb 4 ; lower: + (in * 351 << 32) - in
    const uint32_t d1 = hb ? 0u : in * 351u - bd;
    r.v[0] = fp_addc(r.v[0], d0, 0, &cy);
    r.v[1] = fp_addc(r.v[1], d1, cy, &cy);
    r.v[2] = fp_addc(r.v[2], 0u, cy, &cy);
    r.v[3] = fp_addc(r.v[3], 0u, cy, &cy);
    p = cy;
  } while (FP_ANY(p));
}
// Pending borrows (0 / 1): a lower lane's borrow takes 1 from limb 4 of the partner; an upper lane's borrow stands for
// -2^256 == -c, taken from limbs 0..1 of the partner.
__device__ __forceinline__ void hf_fix_sub(hf& r, uint32_t p, bool hb) {
  do {
    const uint32_t in = hf_partner(p, hb);
    const uint32_t m = 0u - in;
    const uint32_t d0 = hb ? in : m;               // lower: c0 = 0xffffffff when in == 1
    const uint32_t d1 = hb ? 0u : (m & FP_C1);
    uint32_t bw;
    r.v[0] = fp_subb(r.v[0], d0, 0, &bw);
    r.v[1] = fp_subb(r.v[1], d1, bw, &bw);
    r.v[2] = fp_subb(r.v[2], 0u, bw, &bw);
    r.v[3] = fp_subb(r.v[3], 0u, bw, &bw);
    p = bw;
  } while (FP_ANY(p));
}

__device__ __forceinline__ hf hf_add(const hf& a, const hf& b, bool hb) {
  hf r;
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = fp_addc(a.v[i], b.v[i], cy, &cy);
  hf_fix(r, cy, hb);
  return r;
}
__device__ __forceinline__ hf hf_sub(const hf& a, const hf& b, bool hb) {
  hf r;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = fp_subb(a.v[i], b.v[i], bw, &bw);
  hf_fix_sub(r, bw, hb);
  return r;
}

// d = (a - b) * w: the lane's four registers ARE its share of the MFMA operand; w / nw = its fragments of the twiddle's
// matrices (TwMat), kc = shk_mfma_kinit(lane).  Every lane of the wave must be active.
__device__ __forceinline__ hf hf_submul(const hf& a, const hf& b, const shk_v4i w, const shk_v4i nw, const shk_kinit& kc, bool hb) {
  const shk_v4i av = {(int)(a.v[0] ^ 0x80808080u), (int)(a.v[1] ^ 0x80808080u), (int)(a.v[2] ^ 0x80808080u),
                      (int)(a.v[3] ^ 0x80808080u)};
  const shk_v4i bv = {(int)(b.v[0] ^ 0x80808080u), (int)(b.v[1] ^ 0x80808080u), (int)(b.v[2] ^ 0x80808080u),
                      (int)(b.v[3] ^ 0x80808080u)};
  const shk_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  shk_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, av, zero, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(nw, bv, acc, 0, 0, 0);
  uint32_t r5[5];
  shk_norm16k(acc, kc, r5);
  hf r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = r5[i];
  hf_fix(r, r5[4], hb);
  return r;
}

// split <-> whole for a pair of rows held by one lane pair: after hf_to_whole the lower lane holds the whole element A, the
// upper lane the whole element B (4 swaps); hf_from_whole is its inverse.
__device__ __forceinline__ fp hf_to_whole(hf A, hf B) {
  fp x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    shk_swap32(A.v[k], B.v[k]);  // A.v[k]: (lower: A limb k, upper: B limb k) ; B.v[k]: (lower: A limb 4+k, upper: B limb 4+k)
    x.v[k] = A.v[k];
    x.v[4 + k] = B.v[k];
  }
  return x;
}
__device__ __forceinline__ void hf_from_whole(const fp& x, hf& A, hf& B) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    A.v[k] = x.v[k];
    B.v[k] = x.v[4 + k];
    shk_swap32(A.v[k], B.v[k]);
  }
}
