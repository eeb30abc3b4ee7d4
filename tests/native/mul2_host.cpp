// host check of fp_mul2 against fp_mul (both through the portable C paths of fp256.cuh)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "fp256.cuh"
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
int main() {
  fp two128 = fp_zero();
  two128.v[4] = 1;
  long bad = 0, n = 0;
  for (int it = 0; it < 400000; ++it) {
    fp x, w;
    for (int i = 0; i < 8; ++i) { x.v[i] = rnd(); w.v[i] = rnd(); }
    const int m = it % 16;
    if (m == 1) for (int i = 0; i < 8; ++i) x.v[i] = 0xffffffffu;
    if (m == 2) for (int i = 0; i < 8; ++i) w.v[i] = 0xffffffffu;
    if (m == 3) { for (int i = 0; i < 8; ++i) { x.v[i] = 0xffffffffu; w.v[i] = 0xffffffffu; } }
    if (m == 4) x = fp_zero();
    if (m == 5) for (int i = 0; i < 7; ++i) x.v[i] = 0xffffffffu;
    if (m == 6) { for (int i = 4; i < 8; ++i) x.v[i] = 0xffffffffu; for (int i = 0; i < 8; ++i) w.v[i] = 0xffffffffu; }
    fp2 ww;
    ww.w = w;
    ww.w128 = fp_canon(fp_mul(w, two128));
    fp a = fp_canon(fp_mul(x, w)), b = fp_canon(fp_mul2(x, ww));
    if (memcmp(&a, &b, sizeof a)) ++bad;
    // also with a lazily reduced second image
    ww.w128 = fp_mul(w, two128);
    b = fp_canon(fp_mul2(x, ww));
    if (memcmp(&a, &b, sizeof a)) ++bad;
    ++n;
  }
  // operands found by search that take fp_reduce_13's rare branch (carry out of limb 5) with w = the 4th root of unity
  {
    const fp w4 = {{0xfa5aa3a4u, 0x377de0dfu, 0x4d7e179du, 0x2fad473cu, 0x455ace10u, 0x8a4103c1u, 0x19cb5303u, 0x4ed93a77u}};
    const fp xs[2] = {{{0x863f1f72u, 0x3495ef6au, 0x44877dfcu, 0x8013931du, 0x5fa97b87u, 0xb549b1eeu, 0xabf9806du, 0xff6a2ad1u}}, {{0x0edcc14au, 0xe336370cu, 0x0568b5b6u, 0xb9c050b3u, 0x4c9ae08du, 0x2b504d59u, 0x1e477486u, 0x73733a36u}}};
    fp2 ww;
    ww.w = w4;
    ww.w128 = fp_canon(fp_mul(w4, two128));
    for (int k = 0; k < 2; ++k) {
      uint32_t t[13];
      fp_mul2_wide(xs[k].v, ww.w.v, ww.w128.v, t);
      // the rare branch is taken iff lo[0..5] + D[0..5] carries: recompute the condition the slow way
      unsigned __int128 acc = 0;
      uint32_t A[5], D[6];
      uint64_t cy = 0;
      for (int i = 0; i < 5; ++i) { uint64_t m = (uint64_t)t[8 + i] * 351u + cy; A[i] = (uint32_t)m; cy = m >> 32; }
      uint32_t bd = 0;
      D[0] = fp_subb(0u, t[8], 0, &bd);
      for (int i = 1; i < 5; ++i) D[i] = fp_subb(A[i - 1], t[8 + i], bd, &bd);
      D[5] = A[4] - bd;
      uint32_t c = 0;
      for (int i = 0; i < 6; ++i) (void)fp_addc(t[i], D[i], c, &c);
      (void)acc;
      if (!c) { printf("vector %d does not take the rare branch\n", k); ++bad; }
      const fp a = fp_canon(fp_mul(xs[k], w4)), b = fp_canon(fp_mul2(xs[k], ww));
      if (memcmp(&a, &b, sizeof a)) ++bad;
    }
  }
  // fp_div4 (the fold's 1/4 by shifting): 4 * (x / 4) == x for random, unreduced and edge operands, and x / 4 == x * 4^-1
  {
    const fp four = fp_from_u32(4u);
    fp inv4 = fp_one();  // 4^-1 = ((p + 1) / 2)^2: build it as 2^-1 squared, 2^-1 = (p + 1) / 2
    fp half;
    {  // (p + 1) >> 1, exactly
      uint32_t q[8] = {FP_P0 + 1u, FP_P1, FP_PX, FP_PX, FP_PX, FP_PX, FP_PX, FP_PX};
      for (int i = 0; i < 8; ++i) half.v[i] = (q[i] >> 1) | (i < 7 ? q[i + 1] << 31 : 0u);
    }
    inv4 = fp_mul(half, half);
    const fp chk = fp_canon(fp_mul(inv4, four)), one = fp_one();
    if (memcmp(&chk, &one, sizeof chk)) { printf("4^-1 is wrong\n"); ++bad; }
    for (int it = 0; it < 200000; ++it) {
      fp x;
      for (int i = 0; i < 8; ++i) x.v[i] = rnd();
      const int m = it % 16;
      if (m == 1) for (int i = 0; i < 8; ++i) x.v[i] = 0xffffffffu;
      if (m == 2) x = fp_zero();
      if (m == 3) { x = fp_zero(); x.v[0] = it & 7; }
      if (m == 4) { for (int i = 2; i < 8; ++i) x.v[i] = FP_PX; x.v[1] = FP_P1; x.v[0] = (uint32_t)(it & 7); }  // p .. p + 7
      if (m == 5) { for (int i = 2; i < 8; ++i) x.v[i] = FP_PX; x.v[1] = FP_P1 - 1u; x.v[0] = 0xfffffffcu + (it & 3); }  // just below p
      if (m == 6) { for (int i = 1; i < 8; ++i) x.v[i] = 0xffffffffu; x.v[0] = 0xfffffffcu + (it & 3); }
      const fp q = fp_div4(x);
      const fp back = fp_canon(fp_mul(q, four)), want = fp_canon(x), viamul = fp_canon(fp_mul(x, inv4)), qc = fp_canon(q);
      if (memcmp(&back, &want, sizeof back) || memcmp(&viamul, &qc, sizeof qc)) ++bad;
    }
  }
  printf("%ld products, %ld mismatches\n", n, bad);
  return bad != 0;
}
