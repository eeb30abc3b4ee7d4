// stark.hip -- the kernels between the low-degree extension and the FRI commit in STARK.mk_proof
// (starks/stark.py:233-279): transition-constraint quotient D, boundary quotient B, the packed Merkle tree over
// (P, D, B), the pseudorandom linear combination and the Merkle spot checks.
//
// The reference builds C = P(g1 X) - step(P), D = C / Z and B = (P - I) / Z2 in COEFFICIENT form with schoolbook
// polynomial arithmetic (O(n^2), stark.py:38-104) and then evaluates them on the N = steps * ext points x_i = G2^i.
// Here D and B are evaluated directly, exactly, on that domain (s = steps, r = x_last = g1^-1, G1 = {x^s = 1}):
//   Z  = (X^s - 1) / (X - r)         D(x) = C(x) (x - r) / (x^s - 1)
//   Z2 = (X - 1)(X - r)              B(x) = (P(x) - I(x)) / ((x - 1)(x - r))
// * off the trace points (i % ext != 0): x^s - 1 = omega^(i % ext) - 1 takes ext - 1 values (omega = G2^s), whose
//   inverses are constants; 1 / ((x_i - 1)(x_i - r)) is a table cached per (steps, ext).
// * on the trace points x in G1 the quotients are 0/0; the polynomial's value there is the formal-derivative limit
//   (l'Hopital; exact in any field, s is invertible):  D(x) = C'(x) (x - r) x / s  for x != r,  D(r) = C(r) r / s,
//   B(1) = (P'(1) - b) / (1 - r),  B(r) = (P'(r) - b) / (r - 1)   (I = a + b X).  With Q = X P'(X) (coefficients k p_k,
//   one s-point NTT per column)  x C'(x) = Q_c(g1 x) - sum_v d step_c / d X_v (P(x)) Q_v(x).
// C(x) = 0 on G1 minus {r} is the statement "the witness is a valid trace" (the reference asserts cp % z == 0,
// stark.py:76); it is checked there directly.  Every value is the exact residue of the reference's polynomial at that
// point, so every Merkle root and proof byte is identical.
#include <stdlib.h>

#include "blake2s.cuh"
#include "internal.hpp"

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(uint64_t work, int tpb = TPB) { return (unsigned)((work + tpb - 1) / tpb); }

__device__ __forceinline__ void load8(const uint32_t* p, uint32_t w[8]) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store8(uint32_t* p, const uint32_t w[8]) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// g^e from a two-level power table
__device__ __forceinline__ fp pow_lookup(const fp* lo, const fp* hi, uint32_t lb, uint64_t e) {
  fp v = fp_load(lo + (e & ((1ull << lb) - 1)));
  if (hi) v = fp_mul(v, fp_load(hi + (e >> lb)));
  return v;
}

// ---- boundary interpolant (stark.py:81-96, poly_utils.py:397-410) -----------------------------------------
// I_c(X) = a + b X through (1, input_c) and (x_last, output_c), output_c = witness[c][steps - 1]:
// b = (output - input) / (x_last - 1), a = input - b.  One thread per column; run BEFORE the trace is transformed.
__global__ void __launch_bounds__(TPB) stark_interp_kernel(const fp* trace, const fp* inputs, uint64_t steps, uint32_t cols,
                                                           fp inv_last_m1, fp* iab) {
  const uint32_t c = blockIdx.x * TPB + threadIdx.x;
  if (c >= cols) return;
  const fp in = fp_load(inputs + c);
  const fp out = fp_load(trace + (uint64_t)c * steps + (steps - 1));
  const fp b = fp_canon(fp_mul(fp_sub(out, in), inv_last_m1));
  fp two128 = fp_zero();
  two128.v[4] = 1;
  fp_store(iab + 3 * c, fp_sub(in, b));
  fp_store(iab + 3 * c + 1, b);
  fp_store(iab + 3 * c + 2, fp_mul(b, two128));  // the slope as an fp_mul2 pair
}

// ---- step polynomials: sparse terms coef * prod_v X_v^exps[v] (multivariate_polynomial.py:329-338) -----------
template <int W>
__device__ __forceinline__ fp eval_terms(const fp* coef, const uint8_t* exps, uint32_t t0, uint32_t t1, const fp (&P)[W]) {
  fp acc = fp_zero();
  for (uint32_t t = t0; t < t1; ++t) {
    const uint8_t* ex = exps + (uint64_t)t * (W + 1);
    const bool unit = ex[W] != 0;  // coefficient 1 (the common case): the first factor starts the product
    fp prod = fp_load(coef + t);
    bool started = !unit;
#pragma unroll
    for (int v = 0; v < W; ++v) {
      uint32_t e = ex[v];
      if (e && !started) {
        prod = P[v];
        started = true;
        --e;
      }
      for (uint32_t k = 0; k < e; ++k) prod = fp_mul(prod, P[v]);
    }
    acc = fp_add(acc, prod);
  }
  return acc;
}

// q[c][k] = k * p[c][k]: coefficients of Q_c = X P_c'(X)
__global__ void __launch_bounds__(TPB) stark_qprep_kernel(const fp* pcoef, fp* q, uint64_t steps, uint64_t cols) {
  const uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (g >= cols * steps) return;
  const uint64_t k = g % steps;
  fp_store(q + g, fp_mul(fp_load(pcoef + g), fp_from_u32((uint32_t)k)));
}

// The domain tables, once per (steps, ext) (cached by the context): out[0][i] = 1 / ((x_i - 1)(x_i - r)) (0 at x = 1 and x = r),
// out[1][i] = x_i, out[2][i] = (x_i - r) / (x_i^s - 1) (0 on the trace points).
__global__ void __launch_bounds__(TPB) stark_domain_tables_kernel(fp* out, uint64_t n, uint32_t ext, const fp* tw_lo, const fp* tw_hi,
                                                                  uint32_t tw_lb, fp x_last, const fp* inv_omega) {
  const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const fp x = pow_lookup(tw_lo, tw_hi, tw_lb, i);
  fp_store(out + i, fp_inv(fp_mul(fp_sub(x, fp_one()), fp_sub(x, x_last))));  // fp_inv(0) = 0
  fp_store(out + n + i, fp_canon(x));
  const uint32_t j = (uint32_t)i & (ext - 1);
  fp_store(out + 2 * n + i, j ? fp_mul(fp_sub(x, x_last), fp_load(inv_omega + j)) : fp_zero());
}

// ---- the trace points x_k = g1^k (domain index k * ext): D everywhere, B at x = 1 and x = x_last, and the check ----
template <int W>
__global__ void __launch_bounds__(TPB) stark_trace_points_kernel(StarkArgs a) {
  const uint64_t N = a.n, s = a.steps;
  const uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (g >= s * a.batch) return;
  const uint64_t b = g / s, k = g - b * s;
  const uint64_t i = k * a.ext, knext = (k + 1) & (s - 1);
  const fp* we = a.wit + b * W * s;  // P_v(x_k) = witness[v][k]
  const fp* qe = a.q_evals + b * W * s;
  fp P[W];
#pragma unroll
  for (int v = 0; v < W; ++v) P[v] = fp_load(we + (uint64_t)v * s + k);
  const fp x = fp_load(a.xpow + i);
  const bool last = knext == 0;
  const fp scale = fp_mul(last ? a.x_last : fp_sub(x, a.x_last), a.inv_steps);
#pragma unroll 1
  for (int c = 0; c < W; ++c) {
    const uint64_t col = (b * W + c) * N;
    const fp acc = eval_terms<W>(a.term_coef, a.term_exps, a.term_begin[c], a.term_begin[c + 1], P);
    const fp cval = fp_sub(fp_load(we + (uint64_t)c * s + knext), acc);  // witness[c][k+1] - step_c(witness[.][k])
    fp num;
    if (last) {
      num = cval;  // Z(r) = s / r is not zero: D(r) = C(r) r / s
    } else {
      const fp cz = fp_canon(cval);
      uint32_t nz = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) nz |= cz.v[w];
      if (nz) atomicOr(a.bad + b, 1u);
      // x C'(x) = Q_c(g1 x) - sum_v (d step_c / d X_v)(P(x)) Q_v(x)
      fp dsum = fp_zero();
#pragma unroll 1
      for (int v = 0; v < W; ++v) {
        const uint32_t t0 = a.dterm_begin[c * W + v], t1 = a.dterm_begin[c * W + v + 1];
        if (t0 != t1)
          dsum = fp_add(dsum, fp_mul(eval_terms<W>(a.dterm_coef, a.dterm_exps, t0, t1, P), fp_load(qe + (uint64_t)v * s + k)));
      }
      num = fp_sub(fp_load(qe + (uint64_t)c * s + knext), dsum);
    }
    fp_store(a.d_work + col + i, fp_mul(num, scale));
    if (k == 0 || last) {
      // B(1) = (P'(1) - b) / (1 - r);  B(r) = (P'(r) - b) / (r - 1),  P'(x) = Q(x) / x,  1 / r = g1
      const fp slope = fp_load(a.iab + 3 * (b * W + c) + 1);
      const fp qc = fp_load(qe + (uint64_t)c * s + k);
      const fp dp = last ? fp_mul(qc, a.g1) : qc;
      const fp v = fp_mul(fp_sub(dp, slope), a.inv_1_m_last);
      fp_store(a.b_work + col + i, last ? fp_neg(v) : v);
    }
  }
}

// ---- packed Merkle leaves (merkle_tree.py:94-119) over limb-form evaluations ---------------------------------
// Leaf x of proof b = P_1(x) .. P_W(x) || D_1(x) .. D_W(x) || B_1(x) .. B_W(x), each 32 bytes big-endian
// (stark.py:251-257).  Element e of the leaf:
__device__ __forceinline__ void leaf_elem(const StarkArgs& a, uint64_t b, uint32_t e, uint64_t x, uint32_t w[8]) {
  const uint32_t W = a.width;
  const fp* base = e < W ? a.p_evals : e < 2 * W ? a.d_work : a.b_work;
  const uint32_t c = e < W ? e : e < 2 * W ? e - W : e - 2 * W;
  fp_to_wire_words(fp_canon(fp_load(base + (b * W + c) * a.n + x)), w);
}
// WIDE: the launch fills the chip several times over and the hashes use the asm rounds (blake2s.cuh); a narrow launch (one small proof),
// where a wave per SIMD walks its compressions alone, keeps the C++ rounds (2.3 against ~4.5 us per compression there).
constexpr uint64_t STARK_WIDE_THREADS = 1ull << 19;
#ifndef SHK_STARK_SPLIT_LOG
#define SHK_STARK_SPLIT_LOG 17
#endif
// rows (times batch) up to which a row of the quotient + leaf kernel is shared by two lanes, and one of the linear combination by four
// (measured: two lanes pay up to 2^17 rows -- one 2^16-step proof --, four up to 2^16; profiles/r05_stark_narrow_lanes_ab.txt)
constexpr uint64_t STARK_SPLIT_ROWS = SHK_STARK_SPLIT_LOG ? 1ull << SHK_STARK_SPLIT_LOG : 0;
constexpr uint64_t STARK_SPLIT4_ROWS = SHK_STARK_SPLIT_LOG ? 1ull << (SHK_STARK_SPLIT_LOG - 1) : 0;

// ---- D, B off the trace points (B everywhere but x = 1, x = x_last) and the packed leaves, in ONE pass -------------------------------
// One thread per permute4 row: it computes the quotients of its four points, stores them (the linear combination reads them later) and
// hashes the two leaf pairs (k = 3 W BLAKE2s blocks each) and their parent on the spot.  (Rounds 2-4 did this in two launches -- a
// quotient kernel moving ~224 B per point, HBM-bound, then a leaf kernel re-reading P, D, B only to hash them, BLAKE2s-issue-bound,
// strictly in sequence; fused, the quotient traffic runs under the other waves' hash instructions: 10.4 -> 9.3 ms per 128 proofs of
// 2^16 steps, config 5 + 5 %, profiles/r05_c5_fused_quotients_leaves_ab.txt.)
// nodes: [batch][2n] x 32 B (only [0, n) is written: the leaves stay in the evaluation arrays).
// Runs AFTER stark_trace_points_kernel: D on the trace points (and B at x = 1, x = x_last) are its outputs and are loaded here.
// Element e of the message leafA || leafB (k = 3 W elements per leaf): leaf = e < k ? la : lb, kind = (e mod k) / W (P, D, B),
// column c = (e mod k) mod W; one BLAKE2s block = two elements.
template <int W>
struct QuotientPoint {  // what the quotients of one domain point keep in registers: the point's P values (the step polynomials read all of
  fp P[W];              // them); everything else -- x, 1 / Z2(x), 1 / Z(x) -- is one table load where it is used
  uint64_t i, inext;
  bool on_trace, special;
  __device__ __forceinline__ void load(const StarkArgs& a, const fp* pe, uint64_t pt) {
    const uint64_t N = a.n;
    i = pt;
    on_trace = ((uint32_t)pt & (a.ext - 1)) == 0;
    inext = (pt + a.ext) & (N - 1);
    special = on_trace && (pt == 0 || inext == 0);  // x = 1 and x = x_last
#pragma unroll
    for (int v = 0; v < W; ++v) P[v] = fp_load(pe + (uint64_t)v * N + pt);
  }
};
// the hash of the leaf pair (leaf A = point i + 2 s q, leaf B = point i + (2 s + 1) q) of permute4 row i, its quotients computed and stored
// on the way
template <int W, bool WIDE>
__device__ __forceinline__ void quotient_leaf_pair(const StarkArgs& a, const fp* pe, uint64_t b, uint64_t i, int s, uint32_t h[8]) {
  const uint64_t N = a.n, q = N >> 2;
  constexpr uint32_t k = 3 * W;
  b2_init(h);
  const uint64_t la = i + (uint64_t)(2 * s) * q, lb = i + (uint64_t)(2 * s + 1) * q;
  QuotientPoint<W> pt;
  pt.load(a, pe, la);
#pragma unroll 1
  for (uint32_t blk = 0; blk < k; ++blk) {
    uint32_t m[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const uint32_t e = 2 * blk + half;
      if (e == k) pt.load(a, pe, lb);  // the message passes from leaf A to leaf B (uniform over the workgroup)
      const uint32_t ee = e < k ? e : e - k, kind = ee / W, c = ee - kind * W;
      const uint64_t col = (b * W + c) * N;
      fp val;
      if (kind == 0) {
        val = fp_load(pe + (uint64_t)c * N + pt.i);  // (P[c] with a dynamic c: an L1 hit instead of a register-array index)
      } else if (kind == 1) {
        if (!pt.on_trace) {  // off the trace points: D = (P_c(g1 x) - step_c(P(x))) (x - r) / (x^s - 1)
          const fp nxt = fp_load(pe + (uint64_t)c * N + pt.inext), fz = fp_load(a.fz + pt.i);  // requested before the products
          const fp acc = eval_terms<W>(a.term_coef, a.term_exps, a.term_begin[c], a.term_begin[c + 1], pt.P);
          val = fp_mul(fp_sub(nxt, acc), fz);
          fp_store(a.d_work + col + pt.i, val);
        } else {
          val = fp_load(a.d_work + col + pt.i);  // stark_trace_points_kernel's
        }
      } else {
        if (!pt.special) {
          const fp pc = fp_load(pe + (uint64_t)c * N + pt.i), x = fp_load(a.xpow + pt.i), wz = fp_load(a.inv_z2 + pt.i);
          const fp* ab = a.iab + 3 * (b * W + c);  // a, (b, b 2^128)
          const fp interp = fp_add(fp_load(ab), fp_mul2(x, fp2_load(reinterpret_cast<const fp2*>(ab + 1))));
          val = fp_mul(fp_sub(pc, interp), wz);
          fp_store(a.b_work + col + pt.i, val);
        } else {
          val = fp_load(a.b_work + col + pt.i);
        }
      }
      fp_to_wire_words(fp_canon(val), m + 8 * half);
    }
    b2_compress<WIDE>(h, m, 64 * (blk + 1), blk + 1 == k);
  }
}
template <int W, bool WIDE>
__global__ void __launch_bounds__(TPB, (W <= 3 ? 4 : 1)) stark_quotients_leaves_kernel(StarkArgs a, uint32_t* nodes) {
  const uint64_t N = a.n, q = N >> 2;
  const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  const uint64_t b = blockIdx.y;
  if (i >= q) return;
  const fp* pe = a.p_evals + b * W * N;
  uint32_t* tree = nodes + b * 2 * N * 8;
  b2digest d[2];
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    quotient_leaf_pair<W, WIDE>(a, pe, b, i, s, d[s].h);
    store8(tree + (N / 2 + 2 * i + s) * 8, d[s].h);
  }
  b2digest top = b2_hash_pair<WIDE>(d[0].h, d[1].h);
  store8(tree + (N / 4 + i) * 8, top.h);
  if (i == 0) {
    uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    store8(tree, z);
  }
}
// A narrow launch (one small proof: fewer rows than the chip has lanes) is a latency chain of 4 W + 1 compressions and 16 W products per
// row: here TWO adjacent lanes share a row, one leaf pair each (2 W compressions, 8 W products), and the even lane hashes the parent
// from its neighbour's digest (a lane swap inside the wave).  Same stores, same bytes.
// (four waves per SIMD as the one-lane form: a 2^16-step proof's 4096 waves are resident at once; the throughput form of the hash
// rounds measured level there, profiles/r05_stark_narrow_lanes_ab.txt)
template <int W>
__global__ void __launch_bounds__(TPB, (W <= 2 ? 4 : 1)) stark_quotients_leaves_narrow_kernel(StarkArgs a, uint32_t* nodes) {
  const uint64_t N = a.n, q = N >> 2;
  const uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  const uint64_t i = g >> 1, b = blockIdx.y;
  const int s = (int)(g & 1);
  if (i >= q) return;  // both lanes of a row leave together (TPB is even)
  const fp* pe = a.p_evals + b * W * N;
  uint32_t* tree = nodes + b * 2 * N * 8;
  b2digest own, other;
  quotient_leaf_pair<W, false>(a, pe, b, i, s, own.h);
  store8(tree + (N / 2 + 2 * i + s) * 8, own.h);
#pragma unroll
  for (int w = 0; w < 8; ++w) other.h[w] = (uint32_t)__shfl_xor((int)own.h[w], 1);
  if (s == 0) {
    const b2digest top = b2_hash_pair<false>(own.h, other.h);
    store8(tree + (N / 4 + i) * 8, top.h);
    if (i == 0) {
      uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      store8(tree, z);
    }
  }
}

// ---- Fiat-Shamir scalars of the linear combination (stark.py:106-177) ---------------------------------------
// k_t = int(blake(m_root + b"0x0<t+1>")) (4-byte ASCII suffix); with c = (G2^steps)^(precision-1) -- `powers[i]` in the
// reference reads the loop variable left over from building the list, i.e. the last power --
//   l = sum_j (1 + lk_j c) (D_j + (k1 + k2 c) P_j + (k3 + k4 c) B_j),   lk = get_pseudorandom_ks(m_root, width).
// One 64-thread block per proof: thread t hashes suffix t, threads j < W then write alpha_j, alpha_j beta, alpha_j gamma.
__global__ void __launch_bounds__(64) stark_scalars_kernel(const uint32_t* mnodes, uint64_t tree_words, uint32_t width, fp cpow,
                                                           fp* scal) {
  __shared__ fp ks[16];
  const uint32_t t = threadIdx.x, b = blockIdx.x;
  const uint32_t nk = 4 + (width > 4 ? width : 0);  // k1..k4, then (width > 4 only) the "0x00".. series of l_ks
  if (t < nk) {
    uint32_t m[16];
    load8(mnodes + (uint64_t)b * tree_words + 8, m);
#pragma unroll
    for (int j = 9; j < 16; ++j) m[j] = 0;
    const uint32_t digit = t < 4 ? t + 1 : t - 4;  // stark.py:118-125
    m[8] = 0x30u | (0x78u << 8) | (0x30u << 16) | ((0x30u + digit) << 24);  // "0x0<digit>"
    b2digest d = b2_hash_short<false>(m, 36);  // a handful of lanes: latency, not throughput
    ks[t] = fp_from_wire_words(d.h);
  }
  __syncthreads();
  if (t < width) {
    const fp lk = width > 4 ? ks[4 + t] : ks[t];
    const fp beta = fp_add(ks[0], fp_mul(ks[1], cpow));
    const fp gamma = fp_add(ks[2], fp_mul(ks[3], cpow));
    const fp alpha = fp_add(fp_one(), fp_mul(lk, cpow));
    // each scalar with its image times 2^128: the combination multiplies by them through fp_mul2
    fp two128 = fp_zero();
    two128.v[4] = 1;
    const fp sc[3] = {alpha, fp_mul(alpha, beta), fp_mul(alpha, gamma)};
    fp* o = scal + ((uint64_t)b * width + t) * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      fp_store(o + 2 * k, sc[k]);
      fp_store(o + 2 * k + 1, fp_mul(sc[k], two128));
    }
  }
}
// l[b][i] = sum_j alpha_j D_j[i] + (alpha_j beta) P_j[i] + (alpha_j gamma) B_j[i] (stark.py:128-177),
// fused with the first three levels of its Merkle tree (merkelize(l_evaluations),
// stark.py:263 -> merkle_tree.py:36-56 with permute4): one thread per permute4 row i computes l at i, i + q, i + 2q, i + 3q,
// stores the four values and hashes them on the spot -- the tree's leaf pass no longer reads l back, and its hashing runs in
// the shadow of this kernel's memory traffic (the combination alone is HBM-bound).  nodes: [batch][2n] x 32 B as in
// kernels.hip (the leaf level itself is not materialised: the branch gather re-derives leaves from l).
// l at point x of proof b: sum_j alpha_j D_j + (alpha_j beta) P_j + (alpha_j gamma) B_j, the scalars as fp_mul2 pairs from LDS
template <int W, class ScPair>
__device__ __forceinline__ fp lincomb_point(const StarkArgs& a, uint64_t b, uint64_t x, uint32_t width, ScPair&& sc_pair) {
  const uint64_t N = a.n;
  fp acc = fp_zero();
  if constexpr (W != 0) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      // the three values of a column are requested together, before their products (whose inline asm would otherwise pin
      // every load right in front of its use)
      const uint64_t col = (b * W + j) * N + x;
      const fp v0 = fp_load(a.d_work + col), v1 = fp_load(a.p_evals + col), v2 = fp_load(a.b_work + col);
      asm volatile("" ::: "memory");  // one scalar pair in registers at a time
      acc = fp_add(acc, fp_mul2(v0, sc_pair(3u * j)));
      asm volatile("" ::: "memory");
      acc = fp_add(acc, fp_mul2(v1, sc_pair(3u * j + 1)));
      asm volatile("" ::: "memory");
      acc = fp_add(acc, fp_mul2(v2, sc_pair(3u * j + 2)));
    }
  } else {
    for (uint32_t j = 0; j < width; ++j) {
      const uint64_t col = (b * width + j) * N + x;
      acc = fp_add(acc, fp_mul2(fp_load(a.d_work + col), sc_pair(3 * j)));
      acc = fp_add(acc, fp_mul2(fp_load(a.p_evals + col), sc_pair(3 * j + 1)));
      acc = fp_add(acc, fp_mul2(fp_load(a.b_work + col), sc_pair(3 * j + 2)));
    }
  }
  return acc;
}
template <int W, bool WIDE>  // W = the width when it is 1 or 2 (all values of a point are requested before the first product), else 0
__global__ void __launch_bounds__(TPB) stark_lincomb_leaves_kernel(StarkArgs a, const fp* scal, fp* l_evals, uint32_t* nodes) {
  // the proof's 3 * width scalar pairs, staged in LDS once per workgroup (kept in registers they would cost 16 VGPRs each)
  // the 3 * width scalars as fp_mul2 pairs: 12 uint4 per trace column, one per thread of the first width * 12
  static_assert(TPB >= SHK_STARK_MAX_WIDTH * 12, "the scalar staging needs one thread per uint4");
  __shared__ uint4 sc_lds[SHK_STARK_MAX_WIDTH * 12];
  const uint64_t N = a.n, q = N >> 2;
  const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  const uint64_t b = blockIdx.y;
  const uint32_t width = W ? (uint32_t)W : a.width;
  if (threadIdx.x < width * 12) sc_lds[threadIdx.x] = reinterpret_cast<const uint4*>(scal + b * width * 6)[threadIdx.x];
  __syncthreads();
  if (i >= q) return;
  auto sc_pair = [&](uint32_t idx) {  // pair idx = 3 j + k
    const uint4 q0 = sc_lds[4 * idx], q1 = sc_lds[4 * idx + 1], q2 = sc_lds[4 * idx + 2], q3 = sc_lds[4 * idx + 3];
    fp2 r;
    r.w.v[0] = q0.x; r.w.v[1] = q0.y; r.w.v[2] = q0.z; r.w.v[3] = q0.w;
    r.w.v[4] = q1.x; r.w.v[5] = q1.y; r.w.v[6] = q1.z; r.w.v[7] = q1.w;
    r.w128.v[0] = q2.x; r.w128.v[1] = q2.y; r.w128.v[2] = q2.z; r.w128.v[3] = q2.w;
    r.w128.v[4] = q3.x; r.w128.v[5] = q3.y; r.w128.v[6] = q3.z; r.w128.v[7] = q3.w;
    return r;
  };
  uint32_t* tree = nodes + b * 2 * N * 8;
  uint32_t w[4][8];
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    asm volatile("" ::: "memory");  // the scalar pairs are re-read from LDS every round, not kept live across the loop
    const uint64_t x = i + (uint64_t)r * q;
    const fp acc = lincomb_point<W>(a, b, x, width, sc_pair);
    fp_store(l_evals + b * N + x, acc);
    fp_to_wire_words(fp_canon(acc), w[r]);  // x.to_bytes(): 32 bytes big-endian (modp.py:94-95)
  }
  const b2digest d0 = b2_hash_pair<WIDE>(w[0], w[1]);
  const b2digest d1 = b2_hash_pair<WIDE>(w[2], w[3]);
  store8(tree + (N / 2 + 2 * i) * 8, d0.h);
  store8(tree + (N / 2 + 2 * i + 1) * 8, d1.h);
  const b2digest d2 = b2_hash_pair<WIDE>(d0.h, d1.h);
  store8(tree + (N / 4 + i) * 8, d2.h);
  if (i == 0) {
    uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    store8(tree, z);  // nodes[0]: the reference keeps b'' there
  }
}
// A narrow launch (one small proof): FOUR adjacent lanes share a permute4 row, one point each -- 3 W products and up to three
// compressions in sequence instead of 12 W and three; lanes 0 and 2 hash the two leaf pairs from their neighbours' values, lane 0 the
// parent (lane swaps inside the wave).  Same stores, same bytes.
template <int W>
__global__ void __launch_bounds__(TPB) stark_lincomb_leaves_narrow_kernel(StarkArgs a, const fp* scal, fp* l_evals, uint32_t* nodes) {
  static_assert(TPB >= SHK_STARK_MAX_WIDTH * 12, "the scalar staging needs one thread per uint4");
  __shared__ uint4 sc_lds[SHK_STARK_MAX_WIDTH * 12];
  const uint64_t N = a.n, q = N >> 2;
  const uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  const uint64_t i = g >> 2, b = blockIdx.y;
  const uint32_t r = (uint32_t)g & 3u;
  const uint32_t width = W ? (uint32_t)W : a.width;
  if (threadIdx.x < width * 12) sc_lds[threadIdx.x] = reinterpret_cast<const uint4*>(scal + b * width * 6)[threadIdx.x];
  __syncthreads();
  if (i >= q) return;  // the four lanes of a row leave together (TPB is a multiple of 4)
  auto sc_pair = [&](uint32_t idx) {
    const uint4 q0 = sc_lds[4 * idx], q1 = sc_lds[4 * idx + 1], q2 = sc_lds[4 * idx + 2], q3 = sc_lds[4 * idx + 3];
    fp2 p;
    p.w.v[0] = q0.x; p.w.v[1] = q0.y; p.w.v[2] = q0.z; p.w.v[3] = q0.w;
    p.w.v[4] = q1.x; p.w.v[5] = q1.y; p.w.v[6] = q1.z; p.w.v[7] = q1.w;
    p.w128.v[0] = q2.x; p.w128.v[1] = q2.y; p.w128.v[2] = q2.z; p.w128.v[3] = q2.w;
    p.w128.v[4] = q3.x; p.w128.v[5] = q3.y; p.w128.v[6] = q3.z; p.w128.v[7] = q3.w;
    return p;
  };
  uint32_t* tree = nodes + b * 2 * N * 8;
  const uint64_t x = i + (uint64_t)r * q;
  const fp acc = lincomb_point<W>(a, b, x, width, sc_pair);
  fp_store(l_evals + b * N + x, acc);
  uint32_t own[8], other[8];
  fp_to_wire_words(fp_canon(acc), own);
#pragma unroll
  for (int k = 0; k < 8; ++k) other[k] = (uint32_t)__shfl_xor((int)own[k], 1);
  b2digest d = b2_hash_pair<false>(own, other);  // meaningful on the even lanes: hash(value r, value r + 1)
  if ((r & 1u) == 0) store8(tree + (N / 2 + 2 * i + (r >> 1)) * 8, d.h);
#pragma unroll
  for (int k = 0; k < 8; ++k) other[k] = (uint32_t)__shfl_xor((int)d.h[k], 2);
  if (r == 0) {
    const b2digest top = b2_hash_pair<false>(d.h, other);
    store8(tree + (N / 4 + i) * 8, top.h);
    if (i == 0) {
      uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      store8(tree, z);
    }
  }
}

// ---- spot checks (stark.py:390-402) -------------------------------------------------------------------------
// proof[b] = m_root | l_root | for each sampled pos: mk_branch(mtree, pos) | mk_branch(mtree, pos + ext) | mk_branch(ltree, pos).
// A packed branch = leaf (k values) | sibling leaf (k values) | log2(n) - 1 nodes; an l branch = log2(n) + 1 nodes.
// One thread per 32-byte slot.
__global__ void __launch_bounds__(TPB) stark_gather_kernel(StarkArgs a, const uint32_t* mnodes, const uint32_t* lnodes,
                                                           const fp* lvals, const uint32_t* ys, uint32_t samples, uint32_t lg,
                                                           uint8_t* proof, uint64_t stride) {
  const uint32_t k = 3 * a.width;
  const uint32_t pb = 2 * k + (lg - 1), lb = lg + 1, per_sample = 2 * pb + lb;
  const uint64_t per_proof = 2 + (uint64_t)samples * per_sample;
  const uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (g >= per_proof * a.batch) return;
  const uint64_t b = g / per_proof;
  uint64_t r = g - b * per_proof;
  uint32_t* out = reinterpret_cast<uint32_t*>(proof + b * stride) + 8 * r;
  const uint64_t n = a.n, q = n >> 2;
  const uint32_t* mt = mnodes + b * 2 * n * 8;
  const uint32_t* lt = lnodes + b * 2 * n * 8;
  uint32_t w[8];
  if (r < 2) {
    load8((r == 0 ? mt : lt) + 8, w);
    store8(out, w);
    return;
  }
  r -= 2;
  const uint32_t s = (uint32_t)(r / per_sample);
  uint32_t slot = (uint32_t)(r - (uint64_t)s * per_sample);
  const uint64_t pos = ys[b * samples + s];
  if (slot < 2 * pb) {
    const uint32_t which = slot / pb;
    slot -= which * pb;
    const uint64_t x = which ? ((pos + a.ext) & (n - 1)) : pos;
    const uint64_t idx = x / q + 4 * (x % q);  // get_index_in_permuted (merkle_tree.py:26-33)
    if (slot < 2 * k) {
      const uint64_t pi = slot < k ? idx : (idx ^ 1);      // the leaf, then its sibling leaf
      const uint64_t leaf = (pi & 3) * q + (pi >> 2);      // back to the natural position
      leaf_elem(a, b, slot < k ? slot : slot - k, leaf, w);
    } else {
      const uint32_t lev = slot - 2 * k + 1;               // branch entry lev + 1: tree[((n + idx) >> lev) ^ 1]
      load8(mt + (((n + idx) >> lev) ^ 1) * 8, w);
    }
  } else {
    slot -= 2 * pb;
    const uint64_t pi = pos / q + 4 * (pos % q);
    if (slot <= 1) {  // the leaf and its sibling leaf: the l tree's leaf level is not materialised either
      const uint64_t ps = pi ^ slot;
      fp_to_wire_words(fp_canon(fp_load(lvals + b * n + (ps & 3) * q + (ps >> 2))), w);
    } else {
      load8(lt + (((n + pi) >> (slot - 1)) ^ 1) * 8, w);
    }
  }
  store8(out, w);
}

}  // namespace

hipError_t shk_stark_interp(const fp* trace, const fp* inputs, uint64_t steps, uint32_t cols, const fp& inv_last_m1, fp* iab,
                            hipStream_t st) {
  hipLaunchKernelGGL(stark_interp_kernel, dim3(grid_for(cols)), dim3(TPB), 0, st, trace, inputs, steps, cols, inv_last_m1, iab);
  return hipGetLastError();
}

// D, B (all points) and mtree = merkelize_polynomial_evaluations(width, P + D + B evaluations) (stark.py:38-104, 253-257)
hipError_t shk_stark_quotients_and_merkelize(const StarkArgs& a, uint32_t* d_nodes, hipStream_t st) {
  if (a.n < 4) return hipErrorInvalidValue;
  const dim3 tgrid(grid_for(a.steps * a.batch)), grid(grid_for(a.n >> 2), a.batch), block(TPB);
  const bool wide = (a.n >> 2) * a.batch >= STARK_WIDE_THREADS;
  // up to two waves per SIMD of rows: two lanes per row (stark_quotients_leaves_narrow_kernel)
  const bool split = (a.n >> 2) * a.batch <= STARK_SPLIT_ROWS;
  const dim3 sgrid(grid_for(a.n >> 1), a.batch);
  switch (a.width) {
#define SHK_CASE(W)                                                                                   \
  case W:                                                                                             \
    hipLaunchKernelGGL(stark_trace_points_kernel<W>, tgrid, block, 0, st, a);                         \
    if (wide)                                                                                         \
      hipLaunchKernelGGL((stark_quotients_leaves_kernel<W, true>), grid, block, 0, st, a, d_nodes);  \
    else if (split)                                                                                   \
      hipLaunchKernelGGL(stark_quotients_leaves_narrow_kernel<W>, sgrid, block, 0, st, a, d_nodes);   \
    else                                                                                              \
      hipLaunchKernelGGL((stark_quotients_leaves_kernel<W, false>), grid, block, 0, st, a, d_nodes); \
    break;
    SHK_CASE(1) SHK_CASE(2) SHK_CASE(3) SHK_CASE(4) SHK_CASE(5) SHK_CASE(6) SHK_CASE(7) SHK_CASE(8) SHK_CASE(9)
#undef SHK_CASE
    default: return hipErrorInvalidValue;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return shk_merkle_upper_levels(a.n, a.batch, d_nodes, st);
}

hipError_t shk_stark_qprep(const fp* pcoef, fp* q, uint64_t steps, uint64_t cols, hipStream_t st) {
  hipLaunchKernelGGL(stark_qprep_kernel, dim3(grid_for(cols * steps)), dim3(TPB), 0, st, pcoef, q, steps, cols);
  return hipGetLastError();
}

hipError_t shk_stark_domain_tables(fp* out, uint64_t n, uint32_t ext, const fp* tw_lo, const fp* tw_hi, uint32_t tw_lb, const fp& x_last,
                                   const fp* inv_omega, hipStream_t st) {
  hipLaunchKernelGGL(stark_domain_tables_kernel, dim3(grid_for(n)), dim3(TPB), 0, st, out, n, ext, tw_lo, tw_hi, tw_lb, x_last, inv_omega);
  return hipGetLastError();
}

hipError_t shk_stark_scalars(const uint32_t* d_mnodes, uint64_t tree_words, uint32_t width, uint32_t batch, const fp& cpow,
                             fp* d_scal, hipStream_t st) {
  hipLaunchKernelGGL(stark_scalars_kernel, dim3(batch), dim3(64), 0, st, d_mnodes, tree_words, width, cpow, d_scal);
  return hipGetLastError();
}

hipError_t shk_stark_lincomb_tree(const StarkArgs& a, const fp* d_scal, fp* d_l, uint32_t* d_lnodes, hipStream_t st) {
  if (a.n < 4) return hipErrorInvalidValue;
  const dim3 grid(grid_for(a.n >> 2), a.batch), ngrid(grid_for(a.n), a.batch);
  const bool wide = (a.n >> 2) * a.batch >= STARK_WIDE_THREADS;
  const bool split = (a.n >> 2) * a.batch <= STARK_SPLIT4_ROWS;  // four lanes per row (stark_lincomb_leaves_narrow_kernel)
#define SHK_LINCOMB(W)                                                                                              \
  do {                                                                                                              \
    if (wide)                                                                                                       \
      hipLaunchKernelGGL((stark_lincomb_leaves_kernel<W, true>), grid, dim3(TPB), 0, st, a, d_scal, d_l, d_lnodes); \
    else if (split)                                                                                                 \
      hipLaunchKernelGGL(stark_lincomb_leaves_narrow_kernel<W>, ngrid, dim3(TPB), 0, st, a, d_scal, d_l, d_lnodes);  \
    else                                                                                                            \
      hipLaunchKernelGGL((stark_lincomb_leaves_kernel<W, false>), grid, dim3(TPB), 0, st, a, d_scal, d_l, d_lnodes); \
  } while (0)
  if (a.width == 1)
    SHK_LINCOMB(1);
  else if (a.width == 2)
    SHK_LINCOMB(2);
  else
    SHK_LINCOMB(0);
#undef SHK_LINCOMB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return shk_merkle_upper_levels(a.n, a.batch, d_lnodes, st);
}

hipError_t shk_stark_gather(const StarkArgs& a, const uint32_t* d_mnodes, const uint32_t* d_lnodes, const fp* d_lvals,
                            const uint32_t* d_ys, uint32_t samples, uint8_t* d_proof, uint64_t stride, hipStream_t st) {
  uint32_t lg = 0;
  while ((1ull << lg) < a.n) ++lg;
  const uint32_t k = 3 * a.width;
  const uint64_t per_proof = 2 + (uint64_t)samples * (2 * (2 * k + (lg - 1)) + (lg + 1));
  hipLaunchKernelGGL(stark_gather_kernel, dim3(grid_for(per_proof * a.batch)), dim3(TPB), 0, st, a, d_mnodes, d_lnodes, d_lvals,
                     d_ys, samples, lg, d_proof, stride);
  return hipGetLastError();
}
