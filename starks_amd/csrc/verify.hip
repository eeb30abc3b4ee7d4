// verify.hip -- the verifiers behind the C ABI (host code only; no kernel, no context, no GPU needed):
//   sh_fri_verify    = SmoothSubgroupFRI.verify_proximity_proof (starks/fri.py:268-366, commented in the reference)
//   sh_stark_verify  = STARK.verify_proof / verify_proof_at_position (starks/stark.py:281-388)
// on the FLAT proofs sh_fri_prove / sh_stark_prove write (layouts: include/starkhip.h).  A verifier does ~10^4 hashes and ~10^4
// field operations per proof: latency-bound scalar work, which the reference also runs on the CPU; it uses the same field code as
// the device (fp256.cuh is __host__ __device__) and the C++ BLAKE2s rounds of blake2s.cuh.  Where the reference interpolates (a cubic
// per sampled row, poly_utils.py:412-440) the unique polynomial is evaluated in the closed form the device's fold uses
// (csrc/kernels.hip:fri_fold_row): residues are unique, so accept / reject decisions are the reference's.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/starkhip.h"
#include "blake2s.cuh"
#include "fp256.cuh"
#include "internal.hpp"  // SHK_STARK_MAX_WIDTH, SHK_STARK_MAX_TERMS: the prover's limits are the verifier's

namespace {

// ---- BLAKE2s-256 of a byte string (hashlib.blake2s(x).digest(), merkle_tree.py:5) ------------------------------------------------
void h_blake(const uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t h[8];
  b2_init(h);
  size_t off = 0;
  uint32_t m[16];
  while (len - off > 64) {
    memcpy(m, msg + off, 64);
    off += 64;
    b2_compress_cpp(h, m, (uint32_t)off, false);
  }
  uint8_t last[64] = {0};
  memcpy(last, msg + off, len - off);
  memcpy(m, last, 64);
  b2_compress_cpp(h, m, (uint32_t)len, true);
  memcpy(out, h, 32);
}
void h_blake2(const uint8_t* a, size_t la, const uint8_t* b, size_t lb, uint8_t out[32]) {
  std::vector<uint8_t> buf(la + lb);
  memcpy(buf.data(), a, la);
  memcpy(buf.data() + la, b, lb);
  h_blake(buf.data(), buf.size(), out);
}

// ---- field ------------------------------------------------------------------------------------------------------------------------
fp f_from_wire(const uint8_t b[32]) {  // int.from_bytes(b, 'big') % p
  uint32_t w[8];
  memcpy(w, b, 32);
  return fp_canon(fp_from_wire_words(w));
}
bool f_eq(const fp& a, const fp& b) { return fp_eq_canon(fp_canon(a), fp_canon(b)); }
bool f_is_zero(const fp& a) { return fp_eq_canon(fp_canon(a), fp_zero()); }
fp f_pow(fp a, uint64_t e) { return fp_pow_u64(a, e); }
int ilog2u(uint64_t n) {
  int k = 0;
  while ((1ull << k) < n) ++k;
  return k;
}

// ---- get_pseudorandom_indices (utils.py:60-90) -------------------------------------------------------------------------------------
bool sample_indices(const uint8_t entropy[32], uint64_t modulus, uint32_t count, uint32_t exclude, std::vector<uint32_t>* out) {
  if (modulus >= (1ull << 24) || exclude == 1) return false;  // assert modulus < 2**24; division by zero in the reference
  std::vector<uint8_t> data(entropy, entropy + 32);
  while (data.size() < 4ull * count) {
    uint8_t d[32];
    h_blake(data.data() + data.size() - 32, 32, d);
    data.insert(data.end(), d, d + 32);
  }
  const uint64_t real = exclude ? modulus * (exclude - 1) / exclude : modulus;
  if (real == 0) return false;
  out->clear();
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t w = ((uint32_t)data[4 * i] << 24) | ((uint32_t)data[4 * i + 1] << 16) | ((uint32_t)data[4 * i + 2] << 8) | data[4 * i + 3];
    const uint32_t x = (uint32_t)(w % real);
    out->push_back(exclude ? x + 1 + x / (exclude - 1) : x);
  }
  return true;
}

// ---- verify_branch (merkle_tree.py:71-86): entries of `width0` bytes (the leaf and, for packed trees, its sibling leaf) then 32-byte nodes
// proof = leaf (leaf_bytes) | entry 1 (leaf_bytes for packed trees, else 32) | 32-byte nodes; `entries` in all.
bool verify_branch(const uint8_t root[32], uint64_t index, const uint8_t* proof, uint32_t entries, size_t leaf_bytes, bool packed) {
  if (entries < 2) return false;
  const uint64_t half = 1ull << (entries - 1);  // 2**len(proof) // 2
  const uint64_t q = half / 4;
  if (q == 0 || index >= half) return false;
  uint64_t idx = index / q + 4 * (index % q) + half;  // get_index_in_permuted + half
  std::vector<uint8_t> v(proof, proof + leaf_bytes);
  size_t off = leaf_bytes;
  for (uint32_t e = 1; e < entries; ++e) {
    const size_t len = (e == 1 && packed) ? leaf_bytes : 32;
    uint8_t d[32];
    if (idx & 1)
      h_blake2(proof + off, len, v.data(), v.size(), d);
    else
      h_blake2(v.data(), v.size(), proof + off, len, d);
    v.assign(d, d + 32);
    off += len;
    idx >>= 1;
  }
  return memcmp(v.data(), root, 32) == 0;
}

// value at x of the cubic through (x1 I^j, row[j]), I = the primitive 4th root w^(n/4): the closed form of csrc/kernels.hip:fri_fold_row
fp cubic_at(const fp row[4], const fp& x1_inv, const fp& inv_i, const fp& x) {
  const fp t = fp_mul(x, x1_inv);
  const fp u0 = fp_add(row[0], row[2]), u1 = fp_sub(row[0], row[2]), u2 = fp_add(row[1], row[3]);
  const fp u3 = fp_mul(fp_sub(row[1], row[3]), inv_i);
  const fp G0 = fp_add(u0, u2), G2 = fp_sub(u0, u2), G1 = fp_add(u1, u3), G3 = fp_sub(u1, u3);
  fp acc = fp_add(fp_mul(G3, t), G2);
  acc = fp_add(fp_mul(acc, t), G1);
  acc = fp_add(fp_mul(acc, t), G0);
  return fp_div4(acc);
}

struct Cursor {
  const uint8_t* p;
  uint64_t left;
  const uint8_t* take(uint64_t n) {
    if (n > left) return nullptr;
    const uint8_t* r = p;
    p += n;
    left -= n;
    return r;
  }
};

// fri.py:268-366 on the flat layout.  SH_OK = accepted, SH_ERR_REJECTED = some check failed, SH_ERR_INVALID = malformed arguments / length.
int fri_verify(Cursor cur, const uint8_t merkle_root_in[32], uint64_t n, const fp& root, uint64_t maxdeg_plus_1, uint32_t exclude,
               uint32_t samples) {
  uint8_t merkle_root[32];
  memcpy(merkle_root, merkle_root_in, 32);
  fp w = root;
  uint64_t roudeg = n, md = maxdeg_plus_1;
  bool first = true;
  std::vector<uint32_t> ys;
  while (md > 16) {
    if (roudeg < 16) return SH_ERR_INVALID;
    const uint32_t s = first ? samples : 40;  // the prover's recursion falls back to 40 (fri.py:262-266)
    const uint64_t q = roudeg / 4;
    const uint32_t lg = (uint32_t)ilog2u(roudeg), l2 = lg - 1, l1 = lg + 1;
    const uint8_t* root2 = cur.take(32);
    if (!root2) return SH_ERR_INVALID;
    // the round's branches must all be there before anything is derived from `s` (a hostile count must cost nothing: the sampling
    // below hashes 4 s bytes of entropy)
    if (cur.left / (32ull * (l2 + 4ull * l1)) < s) return SH_ERR_INVALID;
    const fp special_x = f_from_wire(merkle_root);  // field(m[1]) (fri.py:229)
    if (!sample_indices(root2, q, s, exclude, &ys)) return SH_ERR_INVALID;
    const fp inv_i = f_pow(w, 3 * q);  // I^-1 = I^3
    for (uint32_t i = 0; i < s; ++i) {
      const uint64_t y = ys[i];
      const uint8_t* b0 = cur.take(32ull * l2);
      if (!b0) return SH_ERR_INVALID;
      fp row[4];
      for (int j = 0; j < 4; ++j) {
        const uint8_t* bj = cur.take(32ull * l1);
        if (!bj) return SH_ERR_INVALID;
        if (!verify_branch(merkle_root, y + q * j, bj, l1, 32, false)) return SH_ERR_REJECTED;
        row[j] = f_from_wire(bj);
      }
      if (!verify_branch(root2, y, b0, l2, 32, false)) return SH_ERR_REJECTED;
      const fp colval = f_from_wire(b0);
      const fp x1_inv = f_pow(w, (roudeg - y) % roudeg);  // w^-y
      if (!f_eq(cubic_at(row, x1_inv, inv_i, special_x), colval)) return SH_ERR_REJECTED;
    }
    memcpy(merkle_root, root2, 32);
    w = f_pow(w, 4);
    md /= 4;
    roudeg /= 4;
    first = false;
  }
  // the final layer (fri.py:340-366): its Merkle root is the last committed root, and the values off the first maxdeg_plus_1 points
  // lie on the interpolant through those
  const uint64_t len = roudeg;
  if (len < 4 || len > cur.left / 32) return SH_ERR_INVALID;  // (before 32 * len: no wrap-around)
  const uint8_t* data = cur.take(32 * len);
  if (!data || cur.left != 0) return SH_ERR_INVALID;
  {
    std::vector<uint8_t> nodes(64 * len, 0);
    const uint64_t q = len / 4;
    for (uint64_t i = 0; i < q; ++i)
      for (uint64_t j = 0; j < 4; ++j) memcpy(&nodes[32 * (len + 4 * i + j)], data + 32 * (i + j * q), 32);  // permute4
    for (uint64_t i = len - 1; i >= 1; --i) h_blake(&nodes[64 * i], 64, &nodes[32 * i]);
    if (memcmp(&nodes[32], merkle_root, 32) != 0) return SH_ERR_REJECTED;
  }
  std::vector<uint64_t> pts;
  for (uint64_t x = 0; x < len; ++x)
    if (!exclude || x % exclude) pts.push_back(x);
  const uint64_t k = md < pts.size() ? md : pts.size();
  std::vector<fp> xs(len), vals(len);
  xs[0] = fp_one();
  for (uint64_t i = 1; i < len; ++i) xs[i] = fp_mul(xs[i - 1], w);
  for (uint64_t i = 0; i < len; ++i) vals[i] = f_from_wire(data + 32 * i);
  // barycentric form of the interpolant through the first k retained points
  std::vector<fp> wgt(k);
  for (uint64_t a = 0; a < k; ++a) {
    fp den = fp_one();
    for (uint64_t b = 0; b < k; ++b)
      if (b != a) den = fp_mul(den, fp_sub(xs[pts[a]], xs[pts[b]]));
    wgt[a] = fp_mul(vals[pts[a]], fp_inv(den));
  }
  for (uint64_t t = k; t < pts.size(); ++t) {
    const fp x = xs[pts[t]];
    fp total = fp_zero();
    for (uint64_t a = 0; a < k; ++a) {
      fp num = wgt[a];
      for (uint64_t b = 0; b < k; ++b)
        if (b != a) num = fp_mul(num, fp_sub(x, xs[pts[b]]));
      total = fp_add(total, num);
    }
    if (!f_eq(total, vals[pts[t]])) return SH_ERR_REJECTED;
  }
  return SH_OK;
}

fp h_root_pow2(int lg) {  // 7^((p - 1) / 2^lg)
  const uint32_t pm1[8] = {0u, 0xfffffea1u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  uint32_t e[8];
  for (int i = 0; i < 8; ++i) {
    const int lo = i + lg / 32, sh = lg % 32;
    uint64_t v = lo < 8 ? pm1[lo] : 0;
    if (sh) v = (v >> sh) | ((uint64_t)(lo + 1 < 8 ? pm1[lo + 1] : 0) << (32 - sh));
    e[i] = (uint32_t)v;
  }
  fp r = fp_one(), b = fp_from_u32(7u);
  for (int i = 0; i < 256; ++i) {
    if ((e[i / 32] >> (i % 32)) & 1) r = fp_mul(r, b);
    b = fp_sqr(b);
  }
  return r;
}

}  // namespace

extern "C" {

int sh_fri_verify(const uint8_t* proof, uint64_t proof_len, const uint8_t merkle_root[32], uint64_t n, const uint8_t root[32],
                  uint64_t maxdeg_plus_1, uint32_t exclude_multiples_of, uint32_t samples) {
  if (!proof || !merkle_root || !root || n < 4 || (n & (n - 1)) || samples == 0) return SH_ERR_INVALID;
  if (n > (1ull << 32)) return SH_ERR_UNSUPPORTED;  // the field has no root of a larger power-of-two order
  const fp w = f_from_wire(root);
  if (!f_eq(f_pow(w, n / 2), fp_neg(fp_one()))) return SH_ERR_ROOT_ORDER;
  return fri_verify(Cursor{proof, proof_len}, merkle_root, n, w, maxdeg_plus_1, exclude_multiples_of, samples);
}

int sh_stark_verify(const uint8_t* proof, uint64_t proof_len, const uint8_t* inputs, const uint8_t* outputs, uint64_t steps, uint32_t ext,
                    uint32_t width, const uint8_t* term_coefs, const uint8_t* term_exps, const uint32_t* term_counts, uint32_t samples) {
  if (!proof || !inputs || !outputs || !term_coefs || !term_exps || !term_counts || width == 0 || samples == 0) return SH_ERR_INVALID;
  if (steps < 2 || (steps & (steps - 1)) || ext < 2 || (ext & (ext - 1))) return SH_ERR_INVALID;
  if (steps >= (1ull << 24) || ext >= (1u << 24) || steps * ext >= (1ull << 24)) return SH_ERR_INVALID;  // utils.py:69 (no wrap-around)
  if (width > SHK_STARK_MAX_WIDTH) return SH_ERR_UNSUPPORTED;  // as sh_stark_prove (capi.hip:stark_terms)
  const uint64_t n = steps * ext;
  const uint32_t lg = (uint32_t)ilog2u(n), k = 3 * width;
  uint32_t degree = 0, begin = 0;
  std::vector<uint32_t> tbegin(width + 1, 0);
  for (uint32_t d = 0; d < width; ++d) {
    tbegin[d] = begin;
    if (term_counts[d] > SHK_STARK_MAX_TERMS) return SH_ERR_UNSUPPORTED;
    begin += term_counts[d];
  }
  tbegin[width] = begin;
  if (begin == 0 || begin > SHK_STARK_MAX_TERMS) return begin ? SH_ERR_UNSUPPORTED : SH_ERR_INVALID;
  std::vector<fp> coef(begin);
  for (uint32_t t = 0; t < begin; ++t) {
    coef[t] = f_from_wire(term_coefs + 32 * t);
    uint32_t sum = 0;
    for (uint32_t v = 0; v < width; ++v) sum += term_exps[(size_t)t * width + v];
    if (sum > degree) degree = sum;  // as sh_stark_prove takes it (capi.hip:stark_terms): every listed term counts
  }
  Cursor cur{proof, proof_len};
  const uint8_t* m_root = cur.take(32);
  const uint8_t* l_root = cur.take(32);
  if (!m_root || !l_root) return SH_ERR_INVALID;
  const uint64_t pb = 32ull * (2 * k + (lg - 1)), lb = 32ull * (lg + 1);
  if (cur.left / (2 * pb + lb) < samples) return SH_ERR_INVALID;  // (before the product: no wrap-around)
  const uint8_t* branches = cur.take((2 * pb + lb) * samples);
  if (!branches) return SH_ERR_INVALID;
  const fp g2 = h_root_pow2((int)lg);
  // the low-degree proof of the linear combination (stark.py:287-289)
  const int rc = fri_verify(cur, l_root, n, g2, steps * (uint64_t)degree, ext, 40);
  if (rc != SH_OK) return rc;
  std::vector<uint32_t> pos;
  if (!sample_indices(l_root, n, samples, ext, &pos)) return SH_ERR_INVALID;
  const fp last = f_pow(g2, (steps - 1) * ext);
  const fp inv_last_m1 = fp_inv(fp_sub(last, fp_one()));
  std::vector<fp> px(width), dx(width), bx(width), pg(width);
  for (uint32_t i = 0; i < samples; ++i) {
    const uint8_t* b1 = branches + (2 * pb + lb) * i;
    const uint8_t* b2 = b1 + pb;
    const uint8_t* b3 = b2 + pb;
    const uint64_t x_i = pos[i];
    if (!verify_branch(m_root, x_i, b1, lg + 1, 32 * k, true)) return SH_ERR_REJECTED;
    if (!verify_branch(m_root, (x_i + ext) % n, b2, lg + 1, 32 * k, true)) return SH_ERR_REJECTED;
    if (!verify_branch(l_root, x_i, b3, lg + 1, 32, false)) return SH_ERR_REJECTED;
    for (uint32_t d = 0; d < width; ++d) {
      px[d] = f_from_wire(b1 + 32 * d);
      dx[d] = f_from_wire(b1 + 32 * (width + d));
      bx[d] = f_from_wire(b1 + 32 * (2 * width + d));
      pg[d] = f_from_wire(b2 + 32 * d);
    }
    const fp x = f_pow(g2, x_i);
    const fp xm = fp_sub(x, last);
    if (f_is_zero(xm)) return SH_ERR_REJECTED;  // the sampling excludes the trace points; a proof that lands there is malformed
    const fp zvalue = fp_mul(fp_sub(f_pow(x, steps), fp_one()), fp_inv(xm));  // Z(x) = (x^steps - 1) / (x - x_last)
    const fp z2 = fp_mul(fp_sub(x, fp_one()), xm);
    for (uint32_t d = 0; d < width; ++d) {
      // transition constraint: P_d(g1 x) - step_d(P(x)) = Z(x) D_d(x) (stark.py:355-365)
      fp acc = fp_zero();
      for (uint32_t t = tbegin[d]; t < tbegin[d + 1]; ++t) {
        fp prod = coef[t];
        for (uint32_t v = 0; v < width; ++v)
          for (uint32_t e = 0; e < term_exps[(size_t)t * width + v]; ++e) prod = fp_mul(prod, px[v]);
        acc = fp_add(acc, prod);
      }
      if (!f_eq(fp_sub(pg[d], acc), fp_mul(zvalue, dx[d]))) return SH_ERR_REJECTED;
      // boundary constraint: B_d(x) Z2(x) + I_d(x) = P_d(x), I_d through (1, input), (x_last, output) (stark.py:367-374)
      const fp in = f_from_wire(inputs + 32 * d), out = f_from_wire(outputs + 32 * d);
      const fp slope = fp_mul(fp_sub(out, in), inv_last_m1);
      const fp interp = fp_add(fp_sub(in, slope), fp_mul(slope, x));
      if (!f_eq(fp_sub(px[d], fp_mul(bx[d], z2)), interp)) return SH_ERR_REJECTED;
    }
  }
  return SH_OK;
}

}  // extern "C"
