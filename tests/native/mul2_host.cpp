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
  printf("%ld products, %ld mismatches\n", n, bad);
  return bad != 0;
}
