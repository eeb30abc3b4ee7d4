/* oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by or called from the product
 * path (starks_amd/, libstarkhip.so); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg load it.
 *
 * A scalar C restatement of the reference's (computablelabs/starks, pure Python) hot path over the
 * MiMC prime p = 2^256 - 351*2^32 + 1, following the reference ALGORITHM function by function --
 * recursive radix-2 DIT with the naive <=4-point base case, Lagrange-interpolation fold with one batched
 * inversion, sequential BLAKE2s tree, the FRI loop including its redundant iNTT->NTT per round -- so that
 * timing it is a fair "the reference's algorithm in compiled code, one core" CPU baseline.
 * Parity is pinned by tests/test_oracle_golden.py against fixtures generated from the live reference.
 *
 * All buffers are in wire form: 32-byte big-endian field elements (starks/modp.py:94-95).
 * Build: gcc -O2 -shared -fPIC oracle/oracle.c -o oracle/liboracle.so   (see oracle/Makefile)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t w[4]; } fe; /* little-endian 64-bit limbs, always canonical (< p) */

static const fe FE_P = {{0xfffffea100000001ull, ~0ull, ~0ull, ~0ull}};
static const uint64_t FE_C = 0x15effffffffull; /* 2^256 - p = 351*2^32 - 1 */

/* ---- starks/modp.py:25-106 --------------------------------------------------------------------- */
static int fe_geq_p(const fe* a) {
  for (int i = 3; i >= 0; --i) {
    if (a->w[i] != FE_P.w[i]) return a->w[i] > FE_P.w[i];
  }
  return 1;
}
static void fe_sub_p(fe* a) {
  u128 bw = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a->w[i] - FE_P.w[i] - bw;
    a->w[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
}
static fe fe_from_wire(const uint8_t* b) { /* modp.py:33-34 then reduced as the first arithmetic op would */
  fe r;
  for (int i = 0; i < 4; ++i) {
    uint64_t v = 0;
    for (int j = 0; j < 8; ++j) v = (v << 8) | b[(3 - i) * 8 + j];
    r.w[i] = v;
  }
  if (fe_geq_p(&r)) fe_sub_p(&r);
  return r;
}
static void fe_to_wire(const fe* a, uint8_t* b) { /* modp.py:94-95 */
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) b[(3 - i) * 8 + j] = (uint8_t)(a->w[i] >> (56 - 8 * j));
}
static fe fe_u64(uint64_t x) {
  fe r = {{x, 0, 0, 0}};
  return r;
}
static int fe_is_zero(const fe* a) { return (a->w[0] | a->w[1] | a->w[2] | a->w[3]) == 0; }
static int fe_eq(const fe* a, const fe* b) {
  return a->w[0] == b->w[0] && a->w[1] == b->w[1] && a->w[2] == b->w[2] && a->w[3] == b->w[3];
}
static fe fe_add(const fe* a, const fe* b) { /* modp.py:43-45 */
  fe r;
  u128 cy = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a->w[i] + b->w[i] + cy;
    r.w[i] = (uint64_t)t;
    cy = t >> 64;
  }
  if (cy) { /* + 2^256 == + c */
    u128 t = (u128)r.w[0] + FE_C;
    r.w[0] = (uint64_t)t;
    cy = t >> 64;
    for (int i = 1; i < 4 && cy; ++i) {
      t = (u128)r.w[i] + cy;
      r.w[i] = (uint64_t)t;
      cy = t >> 64;
    }
  }
  if (fe_geq_p(&r)) fe_sub_p(&r);
  return r;
}
static fe fe_neg(const fe* a) { /* modp.py:55-56 */
  if (fe_is_zero(a)) return *a;
  fe r;
  u128 bw = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)FE_P.w[i] - a->w[i] - bw;
    r.w[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
  return r;
}
static fe fe_sub(const fe* a, const fe* b) { /* modp.py:47-49 */
  fe nb = fe_neg(b);
  return fe_add(a, &nb);
}
static fe fe_mul(const fe* a, const fe* b) { /* modp.py:51-53: (a*b) % p */
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u128 cy = 0;
    for (int j = 0; j < 4; ++j) {
      u128 m = (u128)a->w[i] * b->w[j] + t[i + j] + cy;
      t[i + j] = (uint64_t)m;
      cy = m >> 64;
    }
    t[i + 4] = (uint64_t)cy;
  }
  /* hi*2^256 + lo == hi*c + lo */
  uint64_t r[5];
  u128 cy = 0;
  for (int i = 0; i < 4; ++i) {
    u128 m = (u128)t[4 + i] * FE_C + t[i] + cy;
    r[i] = (uint64_t)m;
    cy = m >> 64;
  }
  r[4] = (uint64_t)cy; /* < 2^42 */
  u128 m = (u128)r[4] * FE_C + r[0];
  fe out;
  out.w[0] = (uint64_t)m;
  cy = m >> 64;
  for (int i = 1; i < 4; ++i) {
    m = (u128)r[i] + cy;
    out.w[i] = (uint64_t)m;
    cy = m >> 64;
  }
  if (cy) { /* wrapped once more: value is tiny now, + c cannot wrap */
    m = (u128)out.w[0] + FE_C;
    out.w[0] = (uint64_t)m;
    cy = m >> 64;
    for (int i = 1; i < 4 && cy; ++i) {
      m = (u128)out.w[i] + cy;
      out.w[i] = (uint64_t)m;
      cy = m >> 64;
    }
  }
  if (fe_geq_p(&out)) fe_sub_p(&out);
  return out;
}
static fe fe_pow(fe a, const uint64_t e[4]) { /* numbertype.py:68-84 (same value) */
  fe r = fe_u64(1);
  for (int i = 0; i < 256; ++i) {
    if ((e[i / 64] >> (i % 64)) & 1) r = fe_mul(&r, &a);
    a = fe_mul(&a, &a);
  }
  return r;
}
static fe fe_inv(const fe* a) { /* modp.py:71-79: extended Euclid in the reference; a^(p-2) is the same residue */
  uint64_t e[4] = {FE_P.w[0] - 2, FE_P.w[1], FE_P.w[2], FE_P.w[3]};
  return fe_pow(*a, e);
}

/* ---- BLAKE2s-256, unkeyed (RFC 7693) = hashlib.blake2s as used at merkle_tree.py:1-5 ------------- */
static const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                   0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define B2S_G(a, b, c, d, x, y) \
  do {                          \
    a = a + b + (x);            \
    d = rotr32(d ^ a, 16);      \
    c = c + d;                  \
    b = rotr32(b ^ c, 12);      \
    a = a + b + (y);            \
    d = rotr32(d ^ a, 8);       \
    c = c + d;                  \
    b = rotr32(b ^ c, 7);       \
  } while (0)
static void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, int last) {
  uint32_t m[16], v[16];
  for (int i = 0; i < 16; ++i)
    m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) |
           ((uint32_t)block[4 * i + 3] << 24);
  for (int i = 0; i < 8; ++i) {
    v[i] = h[i];
    v[i + 8] = B2S_IV[i];
  }
  v[12] ^= (uint32_t)t;
  v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] = ~v[14];
  for (int r = 0; r < 10; ++r) {
    const uint8_t* s = B2S_SIGMA[r];
    B2S_G(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
    B2S_G(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
    B2S_G(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
    B2S_G(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
    B2S_G(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
    B2S_G(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
    B2S_G(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
    B2S_G(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
void or_blake2s(const uint8_t* msg, uint64_t len, uint8_t out[32]) {
  uint32_t h[8];
  for (int i = 0; i < 8; ++i) h[i] = B2S_IV[i];
  h[0] ^= 0x01010020u; /* digest 32, key 0, fanout 1, depth 1 */
  uint64_t off = 0;
  uint8_t block[64];
  while (len - off > 64) {
    b2s_compress(h, msg + off, off + 64, 0);
    off += 64;
  }
  memset(block, 0, 64);
  memcpy(block, msg + off, (size_t)(len - off));
  b2s_compress(h, block, len, 1);
  for (int i = 0; i < 8; ++i) {
    out[4 * i] = (uint8_t)h[i];
    out[4 * i + 1] = (uint8_t)(h[i] >> 8);
    out[4 * i + 2] = (uint8_t)(h[i] >> 16);
    out[4 * i + 3] = (uint8_t)(h[i] >> 24);
  }
}

/* ---- starks/fft.py:287-331 ---------------------------------------------------------------------- */
/* roots are addressed as roots[k*stride], k < L (the reference's roots_of_unity[::2] slicing) */
static void simple_ft(const fe* vals, const fe* roots, size_t stride, size_t L, fe* out) { /* fft.py:287-300 */
  for (size_t i = 0; i < L; ++i) {
    fe last = fe_u64(0);
    for (size_t j = 0; j < L; ++j) {
      fe t = fe_mul(&vals[j], &roots[((i * j) % L) * stride]);
      last = fe_add(&last, &t);
    }
    out[i] = last;
  }
}
static void fft_rec(const fe* vals, size_t n, const fe* roots, size_t stride, fe* out) { /* fft.py:303-314 */
  if (n <= 4) {
    simple_ft(vals, roots, stride, n, out);
    return;
  }
  size_t h = n / 2;
  fe* ev = (fe*)malloc(sizeof(fe) * n * 2);
  fe* od = ev + h;
  fe* L = ev + n;
  fe* R = L + h;
  for (size_t i = 0; i < h; ++i) { /* vals[::2], vals[1::2] */
    ev[i] = vals[2 * i];
    od[i] = vals[2 * i + 1];
  }
  fft_rec(ev, h, roots, stride * 2, L);
  fft_rec(od, h, roots, stride * 2, R);
  for (size_t i = 0; i < h; ++i) {
    fe yr = fe_mul(&R[i], &roots[i * stride]);
    out[i] = fe_add(&L[i], &yr);
    out[i + h] = fe_sub(&L[i], &yr);
  }
  free(ev);
}
/* fft.py:316-331.  n must be the multiplicative order of w (the reference finds it by walking the
 * powers); returns 0 on success, -1 if w^n != 1 or an earlier power is 1, -2 if n_in > n. */
static int fft_1d(const fe* in, size_t n_in, const fe* w, int inverse, fe* out, size_t n) {
  if (n_in > n) return -2;
  fe* rootz = (fe*)malloc(sizeof(fe) * (n + 1));
  rootz[0] = fe_u64(1);
  for (size_t i = 1; i <= n; ++i) { /* fft.py:319-321 */
    rootz[i] = fe_mul(&rootz[i - 1], w);
    if (i < n && fe_eq(&rootz[i], &rootz[0])) {
      free(rootz);
      return -1;
    }
  }
  if (!fe_eq(&rootz[n], &rootz[0])) {
    free(rootz);
    return -1;
  }
  fe* vals = (fe*)calloc(n, sizeof(fe)); /* zero padding, fft.py:323-324 */
  memcpy(vals, in, sizeof(fe) * n_in);
  if (inverse) { /* fft.py:325-328: roots reversed, then scale by n^(p-2) */
    fe* rev = (fe*)malloc(sizeof(fe) * n);
    for (size_t i = 0; i < n; ++i) rev[i] = rootz[n - i]; /* rootz[:0:-1] */
    fft_rec(vals, n, rev, 1, out);
    fe nn = fe_u64((uint64_t)n);
    fe invlen = fe_inv(&nn);
    for (size_t i = 0; i < n; ++i) out[i] = fe_mul(&out[i], &invlen);
    free(rev);
  } else {
    fft_rec(vals, n, rootz, 1, out); /* rootz[:-1] */
  }
  free(vals);
  free(rootz);
  return 0;
}
int or_fft(const uint8_t* in, uint64_t n_in, const uint8_t w[32], int inverse, uint8_t* out, uint64_t n) {
  fe* a = (fe*)malloc(sizeof(fe) * (n_in ? n_in : 1));
  fe* o = (fe*)malloc(sizeof(fe) * n);
  for (uint64_t i = 0; i < n_in; ++i) a[i] = fe_from_wire(in + 32 * i);
  fe ww = fe_from_wire(w);
  int rc = fft_1d(a, n_in, &ww, inverse, o, n);
  if (rc == 0)
    for (uint64_t i = 0; i < n; ++i) fe_to_wire(&o[i], out + 32 * i);
  free(a);
  free(o);
  return rc;
}

/* ---- starks/utils.py:30-38 ---------------------------------------------------------------------- */
int or_power_cycle(const uint8_t w[32], uint64_t n, uint8_t* out) {
  fe ww = fe_from_wire(w), cur = fe_u64(1);
  for (uint64_t i = 0; i < n; ++i) {
    fe_to_wire(&cur, out + 32 * i);
    cur = fe_mul(&cur, &ww);
  }
  fe one = fe_u64(1);
  return fe_eq(&cur, &one) ? 0 : -1;
}

/* ---- starks/merkle_tree.py:11-68 ---------------------------------------------------------------- */
/* leaves: n x 32 B (already in wire form, natural order); nodes: 2n x 32 B, nodes[0] = zeros (the
 * reference stores b'' there), nodes[1] = root. */
void or_merkelize(const uint8_t* leaves, uint64_t n, uint8_t* nodes) {
  uint64_t q = n / 4;
  memset(nodes, 0, 32);
  for (uint64_t i = 0; i < q; ++i) /* permute4, merkle_tree.py:11-23 */
    for (int j = 0; j < 4; ++j) memcpy(nodes + 32 * (n + 4 * i + j), leaves + 32 * (i + j * q), 32);
  if (n < 4) memcpy(nodes + 32 * n, leaves, 32 * n); /* (n//4 == 0 gives an empty tree in the reference; not used) */
  for (uint64_t i = n - 1; i >= 1; --i) or_blake2s(nodes + 64 * i, 64, nodes + 32 * i); /* merkle_tree.py:54-55 */
}
static uint64_t index_in_permuted(uint64_t x, uint64_t L) { /* merkle_tree.py:26-33 */
  uint64_t q = L / 4;
  return x / q + 4 * (x % q);
}
/* merkle_tree.py:59-68; writes log2(n)+1 nodes, returns the count */
uint64_t or_mk_branch(const uint8_t* nodes, uint64_t n, uint64_t index, uint8_t* out) {
  uint64_t idx = index_in_permuted(index, n) + n, k = 0;
  memcpy(out, nodes + 32 * idx, 32);
  k = 1;
  while (idx > 1) {
    memcpy(out + 32 * k, nodes + 32 * (idx ^ 1), 32);
    ++k;
    idx /= 2;
  }
  return k;
}

/* ---- starks/utils.py:60-90 ---------------------------------------------------------------------- */
int or_pseudorandom_indices(const uint8_t entropy[32], uint32_t modulus, uint32_t count, uint32_t exclude,
                            uint32_t* out) {
  if (modulus >= (1u << 24)) return -1; /* utils.py:69 */
  size_t cap = 32 + 4 * (size_t)count + 64, len = 32;
  uint8_t* data = (uint8_t*)malloc(cap);
  memcpy(data, entropy, 32);
  while (len < 4 * (size_t)count) { /* utils.py:74-75 */
    or_blake2s(data + len - 32, 32, data + len);
    len += 32;
  }
  uint32_t real = exclude ? (uint32_t)((uint64_t)modulus * (exclude - 1) / exclude) : modulus;
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t v = ((uint32_t)data[4 * i] << 24) | ((uint32_t)data[4 * i + 1] << 16) | ((uint32_t)data[4 * i + 2] << 8) |
                 data[4 * i + 3];
    uint32_t x = v % real;
    out[i] = exclude ? x + 1 + x / (exclude - 1) : x; /* utils.py:90 */
  }
  free(data);
  return 0;
}

/* ---- starks/poly_utils.py:301-320, 412-440 + polynomial.py:158-164 as used at fri.py:235-242 ------ */
static void multi_inv(const fe* v, size_t n, fe* out) { /* poly_utils.py:301-320 (zero element -> 1, see pyoracle) */
  fe* partials = (fe*)malloc(sizeof(fe) * (n + 1));
  partials[0] = fe_u64(1);
  fe one = fe_u64(1);
  for (size_t i = 0; i < n; ++i) partials[i + 1] = fe_mul(&partials[i], fe_is_zero(&v[i]) ? &one : &v[i]);
  fe inv = fe_inv(&partials[n]);
  for (size_t i = n; i > 0; --i) {
    out[i - 1] = fe_mul(&partials[i - 1], &inv);
    if (!fe_is_zero(&v[i - 1])) inv = fe_mul(&inv, &v[i - 1]);
  }
  free(partials);
}
static fe poly4_eval(const fe c[4], const fe* x) { /* polynomial.py:158-164 */
  fe y = fe_u64(0), pw = fe_u64(1);
  for (int i = 0; i < 4; ++i) {
    fe t = fe_mul(&pw, &c[i]);
    y = fe_add(&y, &t);
    pw = fe_mul(&pw, x);
  }
  return y;
}
static void eq_poly(const fe* xa, const fe* xb, const fe* xc, fe out[4]) { /* (X-xa)(X-xb)(X-xc), poly_utils.py:420-423 */
  fe ab = fe_mul(xa, xb), ac = fe_mul(xa, xc), bc = fe_mul(xb, xc);
  fe abc = fe_mul(&ab, xc);
  out[0] = fe_neg(&abc);
  fe s = fe_add(&ab, &ac);
  out[1] = fe_add(&s, &bc);
  fe t = fe_add(xa, xb);
  t = fe_add(&t, xc);
  out[2] = fe_neg(&t);
  out[3] = fe_u64(1);
}
/* column[i] = P_i(special_x), P_i the cubic through (xs[i+jq], values[i+jq]), j<4 (fri.py:235-242) */
static void fri_fold(const fe* values, const fe* xs, size_t n, const fe* special_x, fe* column) {
  size_t q = n / 4;
  fe* eqs = (fe*)malloc(sizeof(fe) * 16 * q);
  fe* targets = (fe*)malloc(sizeof(fe) * n);
  fe* invs = (fe*)malloc(sizeof(fe) * n);
  for (size_t i = 0; i < q; ++i) {
    const fe* x0 = &xs[i];
    const fe* x1 = &xs[i + q];
    const fe* x2 = &xs[i + 2 * q];
    const fe* x3 = &xs[i + 3 * q];
    fe* e = eqs + 16 * i;
    eq_poly(x1, x2, x3, e);
    eq_poly(x0, x2, x3, e + 4);
    eq_poly(x0, x1, x3, e + 8);
    eq_poly(x0, x1, x2, e + 12);
    targets[4 * i + 0] = poly4_eval(e, x0);
    targets[4 * i + 1] = poly4_eval(e + 4, x1);
    targets[4 * i + 2] = poly4_eval(e + 8, x2);
    targets[4 * i + 3] = poly4_eval(e + 12, x3);
  }
  multi_inv(targets, n, invs); /* poly_utils.py:430 */
  for (size_t i = 0; i < q; ++i) {
    fe invy[4], c[4];
    for (int j = 0; j < 4; ++j) invy[j] = fe_mul(&values[i + j * q], &invs[4 * i + j]);
    for (int k = 0; k < 4; ++k) {
      fe acc = fe_u64(0);
      for (int j = 0; j < 4; ++j) {
        fe t = fe_mul(&eqs[16 * i + 4 * j + k], &invy[j]);
        acc = fe_add(&acc, &t);
      }
      c[k] = acc;
    }
    column[i] = poly4_eval(c, special_x);
  }
  free(eqs);
  free(targets);
  free(invs);
}
int or_fold(const uint8_t* values, uint64_t n, const uint8_t w[32], const uint8_t special_x[32], uint8_t* out) {
  fe* v = (fe*)malloc(sizeof(fe) * n);
  fe* xs = (fe*)malloc(sizeof(fe) * n);
  fe* col = (fe*)malloc(sizeof(fe) * (n / 4));
  fe ww = fe_from_wire(w), cur = fe_u64(1);
  for (uint64_t i = 0; i < n; ++i) {
    v[i] = fe_from_wire(values + 32 * i);
    xs[i] = cur;
    cur = fe_mul(&cur, &ww);
  }
  fe sx = fe_from_wire(special_x); /* unreduced bytes ctor, then every use reduces (modp.py:33-36) */
  fri_fold(v, xs, n, &sx, col);
  for (uint64_t i = 0; i < n / 4; ++i) fe_to_wire(&col[i], out + 32 * i);
  free(v);
  free(xs);
  free(col);
  return 0;
}

/* ---- starks/fri.py:189-266 (commented SmoothSubgroupFRI.generate_proximity_proof) ---------------- */
/* Flat proof layout (shared with the product's sh_fri_prove):
 *   per round:  root2 (32) | for each sampled y: branch(m2, y) then branch(m, y + q*j), j = 0..3
 *   last:       the final layer's values (32 B each).
 * Returns the number of bytes written, or a negative error code. */
static int64_t fri_rec(const fe* coeffs, size_t n_coeffs, fe w, uint64_t maxdeg_plus_1, uint32_t exclude,
                       uint32_t samples, uint8_t* out, uint64_t cap) {
  /* order of w = transform length (fft.py:319-321) */
  size_t n = 1;
  fe t = w, one = fe_u64(1);
  while (!fe_eq(&t, &one)) {
    t = fe_mul(&t, &t);
    n *= 2;
    if (n > ((size_t)1 << 32)) return -1;
  }
  fe* values = (fe*)malloc(sizeof(fe) * n);
  if (fft_1d(coeffs, n_coeffs, &w, 0, values, n) != 0) { /* fri.py:207-208 */
    free(values);
    return -1;
  }
  if (maxdeg_plus_1 <= 16) { /* fri.py:212-214 */
    if (cap < 32 * n) {
      free(values);
      return -3;
    }
    for (size_t i = 0; i < n; ++i) fe_to_wire(&values[i], out + 32 * i);
    free(values);
    return (int64_t)(32 * n);
  }
  fe* xs = (fe*)malloc(sizeof(fe) * n); /* fri.py:217 get_power_cycle */
  xs[0] = one;
  for (size_t i = 1; i < n; ++i) xs[i] = fe_mul(&xs[i - 1], &w);
  uint8_t* wire = (uint8_t*)malloc(32 * n);
  uint8_t* m = (uint8_t*)malloc(64 * n);
  for (size_t i = 0; i < n; ++i) fe_to_wire(&values[i], wire + 32 * i);
  or_merkelize(wire, n, m);            /* fri.py:224 */
  fe special_x = fe_from_wire(m + 32); /* fri.py:229 */
  size_t q = n / 4;
  fe* column = (fe*)malloc(sizeof(fe) * q);
  fri_fold(values, xs, n, &special_x, column); /* fri.py:235-242 */
  uint8_t* m2 = (uint8_t*)malloc(64 * q);
  for (size_t i = 0; i < q; ++i) fe_to_wire(&column[i], wire + 32 * i);
  or_merkelize(wire, q, m2); /* fri.py:243 */
  uint32_t* ys = (uint32_t*)malloc(4 * samples);
  int64_t rc = -1;
  uint64_t pos = 0;
  if (or_pseudorandom_indices(m2 + 32, (uint32_t)q, samples, exclude, ys) == 0) { /* fri.py:246-247 */
    unsigned lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    uint64_t need = 32 + (uint64_t)samples * 32 * ((lg - 1) + 4 * (lg + 1));
    if (cap >= need) {
      memcpy(out, m2 + 32, 32);
      pos = 32;
      for (uint32_t s = 0; s < samples; ++s) { /* fri.py:251-254 */
        pos += 32 * or_mk_branch(m2, q, ys[s], out + pos);
        for (int j = 0; j < 4; ++j) pos += 32 * or_mk_branch(m, n, ys[s] + q * j, out + pos);
      }
      /* fri.py:260-266: inverse FFT of the column over w^4, strip trailing zeros, recurse (samples -> 40) */
      fe w2 = fe_mul(&w, &w), w4 = fe_mul(&w2, &w2);
      fe* cpoly = (fe*)malloc(sizeof(fe) * q);
      if (fft_1d(column, q, &w4, 1, cpoly, q) == 0) {
        size_t deg = q;
        while (deg > 0 && fe_is_zero(&cpoly[deg - 1])) --deg; /* polynomial.py:58 */
        int64_t sub = fri_rec(cpoly, deg, w4, maxdeg_plus_1 / 4, exclude, 40, out + pos, cap - pos);
        rc = sub < 0 ? sub : (int64_t)pos + sub;
      }
      free(cpoly);
    } else {
      rc = -3;
    }
  }
  free(ys);
  free(m2);
  free(column);
  free(m);
  free(wire);
  free(xs);
  free(values);
  return rc;
}
int64_t or_fri_prove(const uint8_t* coeffs, uint64_t n_coeffs, const uint8_t w[32], uint64_t maxdeg_plus_1,
                     uint32_t exclude, uint32_t samples, uint8_t* out, uint64_t cap) {
  fe* c = (fe*)malloc(sizeof(fe) * (n_coeffs ? n_coeffs : 1));
  for (uint64_t i = 0; i < n_coeffs; ++i) c[i] = fe_from_wire(coeffs + 32 * i);
  while (n_coeffs > 0 && fe_is_zero(&c[n_coeffs - 1])) --n_coeffs; /* the reference takes a Poly (polynomial.py:58) */
  int64_t rc = fri_rec(c, n_coeffs, fe_from_wire(w), maxdeg_plus_1, exclude, samples, out, cap);
  free(c);
  return rc;
}

/* ---- starks/stark.py:27-36, 253-256: low-degree extension --------------------------------------- */
int or_lde(const uint8_t* trace, uint64_t steps, uint32_t ext, const uint8_t g2[32], uint8_t* out) {
  fe g = fe_from_wire(g2);
  uint64_t e[4] = {ext, 0, 0, 0};
  fe g1 = fe_pow(g, e);
  fe* tr = (fe*)malloc(sizeof(fe) * steps);
  fe* co = (fe*)malloc(sizeof(fe) * steps);
  fe* ev = (fe*)malloc(sizeof(fe) * steps * ext);
  for (uint64_t i = 0; i < steps; ++i) tr[i] = fe_from_wire(trace + 32 * i);
  int rc = fft_1d(tr, steps, &g1, 1, co, steps);
  if (rc == 0) rc = fft_1d(co, steps, &g, 0, ev, steps * ext);
  if (rc == 0)
    for (uint64_t i = 0; i < steps * ext; ++i) fe_to_wire(&ev[i], out + 32 * i);
  free(tr);
  free(co);
  free(ev);
  return rc;
}

/* element-wise field ops on wire-form arrays (for pinning the arithmetic against field.json) */
void or_field_op(int op, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    fe x = fe_from_wire(a + 32 * i), y = fe_from_wire(b + 32 * i), r;
    switch (op) {
      case 0: r = fe_add(&x, &y); break;
      case 1: r = fe_sub(&x, &y); break;
      case 2: r = fe_mul(&x, &y); break;
      default: r = fe_inv(&x); break;
    }
    fe_to_wire(&r, out + 32 * i);
  }
}
