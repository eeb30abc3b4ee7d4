// kernels.hip -- the non-NTT kernels of the FRI commit path: wire<->limb conversion, synthetic input,
// BLAKE2s Merkle tree (merkle_tree.py:36-56), FRI fold (fri.py:235-242), query sampling (utils.py:60-90)
// and Merkle branch gather (merkle_tree.py:59-68).
#include <stdlib.h>

#include "blake2s.cuh"
#include "internal.hpp"
#include "wave_chunks.cuh"

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(uint64_t work, int tpb = TPB) { return (unsigned)((work + tpb - 1) / tpb); }
// One thread per element of a vector of up to 2^32 elements (or a batch of that many): a launch holds fewer than 2^32 threads per grid
// dimension, so from 2^30 threads on the blocks spread over blockIdx.y; flat_thread() is the element index either way.
constexpr uint64_t FLAT_GRID_X = 1ull << 22;
inline dim3 flat_grid_for(uint64_t work) {
  const uint64_t blocks = (work + TPB - 1) / TPB;
  return blocks <= FLAT_GRID_X ? dim3((unsigned)blocks) : dim3((unsigned)FLAT_GRID_X, (unsigned)((blocks + FLAT_GRID_X - 1) / FLAT_GRID_X));
}
__device__ __forceinline__ uint64_t flat_thread() {
  return ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * TPB + threadIdx.x;
}

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

// ---- conversions ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) wire_to_limb_kernel(const uint32_t* wire, fp* limbs, uint64_t n) {
  uint64_t i = flat_thread();
  if (i >= n) return;
  uint32_t w[8];
  load8(wire + 8 * i, w);
  fp_store(limbs + i, fp_from_wire_words(w));  // may be >= p: limb form is lazily reduced
}
__global__ void __launch_bounds__(TPB) limb_to_wire_kernel(const fp* limbs, uint32_t* wire, uint64_t n) {
  uint64_t i = flat_thread();
  if (i >= n) return;
  uint32_t w[8];
  fp_to_wire_words(fp_canon(fp_load(limbs + i)), w);
  store8(wire + 8 * i, w);
}
// x_i = BLAKE2s(seed_le64 || i_le64) mod p   (same generator as tests/golden/generate.py)
__global__ void __launch_bounds__(TPB) fill_seeded_kernel(fp* out, uint64_t n, uint64_t seed) {
  uint64_t i = flat_thread();
  if (i >= n) return;
  uint32_t m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = 0;
  m[0] = (uint32_t)seed;
  m[1] = (uint32_t)(seed >> 32);
  m[2] = (uint32_t)i;
  m[3] = (uint32_t)(i >> 32);
  b2digest d = b2_hash_short(m, 16);
  fp_store(out + i, fp_canon(fp_from_wire_words(d.h)));
}
// Synthetic many-proof workload (BASELINE configs[4]): unit j of the reference's MiMC formulation (test_stark.py:265-293,
// utils.py:20-27): width 2, dimension 0 carries the round constant k, dimension 1 the state x <- x^3 + k started from
// 3 + j.  One thread per unit walks its trace (the recurrence is sequential); witness [batch][2][steps], inputs [batch][2].
__global__ void __launch_bounds__(64) fill_mimc_units_kernel(fp* wit, fp* inputs, uint64_t steps, uint32_t first_unit,
                                                            uint32_t batch, uint32_t constant) {
  const uint32_t b = blockIdx.x * 64 + threadIdx.x;
  if (b >= batch) return;
  const fp k = fp_from_u32(constant);
  fp x = fp_canon(fp_add(fp_from_u32(3u), fp_from_u32(first_unit + b)));
  fp_store(inputs + 2ull * b, k);
  fp_store(inputs + 2ull * b + 1, x);
  fp* c0 = wit + (uint64_t)b * 2 * steps;
  fp* c1 = c0 + steps;
#pragma unroll 1
  for (uint64_t i = 0; i < steps; ++i) {
    fp_store(c0 + i, k);
    fp_store(c1 + i, x);
    x = fp_canon(fp_add(fp_mul(fp_sqr(x), x), k));
  }
}
__global__ void __launch_bounds__(TPB) pointwise_mul_kernel(const fp* a, const fp* b, fp* out, uint64_t n) {
  uint64_t i = flat_thread();
  if (i >= n) return;
  fp_store(out + i, fp_mul(fp_load(a + i), fp_load(b + i)));
}
__global__ void __launch_bounds__(TPB) powers_kernel(const fp* lo, const fp* hi, uint32_t lb, fp* out, uint64_t n) {
  uint64_t i = flat_thread();
  if (i >= n) return;
  fp v = fp_load(lo + (i & ((1ull << lb) - 1)));
  if (hi) v = fp_mul(v, fp_load(hi + (i >> lb)));
  fp_store(out + i, v);
}
// inter-pass twiddles of one column pass laid out the way the MFMA tile pass reads them: out[k * S + j2] = g^(j2 * k)
// (g of order R * S given by its power table), so that the 32 adjacent columns of a tile read 1 KiB contiguous per row
__global__ void __launch_bounds__(TPB) tw2_kernel(const fp* lo, const fp* hi, uint32_t lb, fp* out, uint32_t log_R, uint32_t log_S) {
  const uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (g >> (log_R + log_S)) return;
  const uint64_t k = g >> log_S, j2 = g & ((1ull << log_S) - 1);
  const uint64_t e = (j2 * k) & ((1ull << (log_R + log_S)) - 1);
  fp v = fp_load(lo + (hi ? (e & ((1ull << lb) - 1)) : e));
  if (hi) v = fp_mul(v, fp_load(hi + (e >> lb)));
  fp_store(out + g, v);
}
__global__ void __launch_bounds__(TPB) pad_copy_kernel(const fp* src, fp* dst, uint64_t n_in, uint64_t n, uint64_t total) {
  uint64_t g = flat_thread();
  if (g >= total) return;
  uint64_t b = g / n, i = g - b * n;
  fp_store(dst + g, i < n_in ? fp_load(src + b * n_in + i) : fp_zero());
}

// ---- Merkle tree ------------------------------------------------------------------------------------
// LDS-transposed block I/O (wave_chunks.cuh explains why, and holds the index arithmetic and every kernel's hand-over plan).
template <int CH>
__device__ __forceinline__ void block_store_chunks(uint4* lds, uint4* gdst, const uint4 (&v)[CH], uint32_t t, uint32_t limit) {
#pragma unroll
  for (int c = 0; c < CH; ++c) lds[chunk_swz(CH * t + c)] = v[c];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const uint32_t c = k * TPB + t;
    if (c < limit) gdst[c] = lds[chunk_swz(c)];
  }
  __syncthreads();
}
// The same hand-over per WAVE, without a workgroup barrier (measured on the leaf kernel without the leaf-level store: 409 -> 339 us
// for 2^24 values, profiles/r04_merkle_lab.txt).  H = one entry of the kernel's handover_plan; `limit` = valid chunks of the block.
#define SHK_WAVE_SYNC()                                  \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
template <class H>
__device__ __forceinline__ void wave_store_chunks(uint4* lds, uint4* gdst, const uint4 (&v)[H::CH], uint32_t t, uint32_t limit) {
#pragma unroll
  for (int c = 0; c < H::CH; ++c) lds[H::own(t, c)] = v[c];
  SHK_WAVE_SYNC();
#pragma unroll
  for (int k = 0; k < H::CH; ++k) {
    const uint32_t g = H::gidx(t, k);
    if (g < limit) gdst[g] = lds[H::moved(t, k)];
  }
  SHK_WAVE_SYNC();
}
template <class H>
__device__ __forceinline__ void wave_load_chunks(uint4* lds, const uint4* gsrc, uint4 (&v)[H::CH], uint32_t t, uint32_t limit) {
#pragma unroll
  for (int k = 0; k < H::CH; ++k) {
    const uint32_t g = H::gidx(t, k);
    if (g < limit) lds[H::moved(t, k)] = gsrc[g];
  }
  SHK_WAVE_SYNC();
#pragma unroll
  for (int c = 0; c < H::CH; ++c) v[c] = lds[H::own(t, c)];
  SHK_WAVE_SYNC();
}
__device__ __forceinline__ uint4 pack4(const uint32_t* w) { return make_uint4(w[0], w[1], w[2], w[3]); }
__device__ __forceinline__ void unpack4(const uint4& v, uint32_t* w) { w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }

// One thread per row i of permute4 (merkle_tree.py:11-23): the 4 leaves 4i..4i+3 (wire form) and the three
// nodes above them; a workgroup handles MERKLE_ROWS consecutive tiles of 256 rows.  grid = (ceil(n/4 / (256 * MERKLE_ROWS)), batch).
// STORE = false skips writing the leaf level (a third of the kernel's traffic): the callers that keep the value array
// alive next to the tree (FRI rounds, the STARK l tree) re-derive a sampled leaf from its value when they gather branches.
// Round 4 (lab: lab/r04/merkle_lab.hip): without the leaf level the pair level goes out per wave (wave_store_chunks) from 16 KiB
// of LDS: 409 -> 339 us for 2^24 values.  (Two tiles per workgroup, MERKLE_ROWS = 2: 363 -> 334 us in the lab with the leaf
// level stored, nothing in the library -- 407 vs 411 us -- and slower on small trees: not taken.)
// WIDE = the launch fills the chip several times over: the hashes use the asm rounds (blake2s.cuh); small launches, where a wave
// per SIMD walks its three hashes alone, keep the C++ rounds (2^14-step FRI commit: 0.256 ms against 0.276 with asm rounds).
constexpr int MERKLE_ROWS = 1;
constexpr uint64_t MERKLE_WIDE_THREADS = 1ull << 19;  // threads per launch from which the asm rounds win (measured between 2^18 and 2^20)
template <bool RAW, bool STORE, bool WIDE>
__global__ void __launch_bounds__(TPB) merkle_leaves_kernel(const void* leaves, uint64_t n, uint32_t* nodes) {
  __shared__ uint4 lds[TPB * (STORE ? 8 : merkle_leaves_nostore_plan::LDS_CH)];
  const uint64_t q = n >> 2;
  const uint32_t t = threadIdx.x;
  const uint64_t b = blockIdx.y;
  uint32_t* tree = nodes + b * (2 * n) * 8;
#pragma unroll 1
  for (int rr = 0; rr < MERKLE_ROWS; ++rr) {
    const uint64_t row0 = ((uint64_t)blockIdx.x * MERKLE_ROWS + rr) * TPB, i = row0 + t;
    if (row0 >= q) break;  // whole workgroup
    const bool valid = i < q;
    const uint32_t rows_here = (uint32_t)(q - row0 < TPB ? q - row0 : TPB);
    uint32_t w[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!valid) {
#pragma unroll
        for (int k = 0; k < 8; ++k) w[j][k] = 0;
      } else if (RAW) {
        load8(reinterpret_cast<const uint32_t*>(leaves) + (b * n + i + j * q) * 8, w[j]);
      } else {
        fp v = fp_canon(fp_load(reinterpret_cast<const fp*>(leaves) + b * n + i + j * q));
        fp_to_wire_words(v, w[j]);  // x.to_bytes(): 32 bytes big-endian (modp.py:94-95)
      }
    }
    if (STORE) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] = pack4(w[j]);
        v[2 * j + 1] = pack4(w[j] + 4);
      }
      block_store_chunks<8>(lds, reinterpret_cast<uint4*>(tree + (n + 4 * row0) * 8), v, t, rows_here * 8);
    }
    b2digest d0 = b2_hash_pair<WIDE>(w[0], w[1]);
    b2digest d1 = b2_hash_pair<WIDE>(w[2], w[3]);
    {
      uint4 v[4] = {pack4(d0.h), pack4(d0.h + 4), pack4(d1.h), pack4(d1.h + 4)};
      if (STORE)
        block_store_chunks<4>(lds, reinterpret_cast<uint4*>(tree + (n / 2 + 2 * row0) * 8), v, t, rows_here * 4);
      else
        wave_store_chunks<handover_at<0, merkle_leaves_nostore_plan>::type>(lds, reinterpret_cast<uint4*>(tree + (n / 2 + 2 * row0) * 8), v, t,
                                                                             rows_here * 4);
    }
    if (valid) {
      b2digest d2 = b2_hash_pair<WIDE>(d0.h, d1.h);
      store8(tree + (n / 4 + i) * 8, d2.h);
      if (i == 0) {
        uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        store8(tree, z);  // nodes[0]: the reference keeps b'' there
      }
    }
  }
}

// Wide levels: one thread reduces 4 adjacent nodes of level L to their parent pair (level L-1) and
// grandparent (level L-2): every lane busy, three hashes per thread (two independent, one dependent).
// grid = (ceil(2^(L-2) / 256), batch).
// (the LDS hand-overs are per wave, no workgroup barrier, as in the leaf kernel)
template <bool WIDE>
__global__ void __launch_bounds__(TPB) merkle_mid_kernel(uint32_t* nodes, uint64_t n, uint32_t L) {
  __shared__ uint4 lds[TPB * merkle_mid_plan::LDS_CH];
  const uint64_t cnt = 1ull << (L - 2);  // nodes produced at level L-2, per tree
  const uint32_t t = threadIdx.x;
  const uint64_t i0 = (uint64_t)blockIdx.x * TPB, i = i0 + t;
  const uint32_t here = (uint32_t)(cnt - i0 < TPB ? cnt - i0 : TPB);
  uint32_t* tree = nodes + (uint64_t)blockIdx.y * (2 * n) * 8;
  uint4 v[8];
  wave_load_chunks<handover_at<0, merkle_mid_plan>::type>(lds, reinterpret_cast<const uint4*>(tree + ((1ull << L) + 4 * i0) * 8), v, t, here * 8);
  uint32_t w[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unpack4(v[2 * j], w[j]);
    unpack4(v[2 * j + 1], w[j] + 4);
  }
  b2digest d0 = b2_hash_pair<WIDE>(w[0], w[1]);
  b2digest d1 = b2_hash_pair<WIDE>(w[2], w[3]);
  {
    uint4 u[4] = {pack4(d0.h), pack4(d0.h + 4), pack4(d1.h), pack4(d1.h + 4)};
    wave_store_chunks<handover_at<1, merkle_mid_plan>::type>(lds, reinterpret_cast<uint4*>(tree + ((1ull << (L - 1)) + 2 * i0) * 8), u, t, here * 4);
  }
  if (i >= cnt) return;
  b2digest d2 = b2_hash_pair<WIDE>(d0.h, d1.h);
  store8(tree + (cnt + i) * 8, d2.h);
}


// Packed leaves (merkelize_polynomial_evaluations, merkle_tree.py:94-119): leaf x = evals[0][x] || ... || evals[k-1][x]
// (32 k bytes), so a first-level node hashes a 64 k-byte message = k BLAKE2s blocks.  One thread per permute4 row.
// evals: [k][n] wire-form values; leaves_out: [n][k] in permuted order (slot 4i+j = leaf i + j n/4); nodes: [n] x 32 B.
__global__ void __launch_bounds__(TPB) merkle_packed_leaves_kernel(const uint32_t* evals, uint64_t n, uint32_t k,
                                                                   uint32_t* leaves_out, uint32_t* nodes) {
  const uint64_t q = n >> 2;
  const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= q) return;
  b2digest d[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    b2_init(d[s].h);
    const uint64_t la = i + (uint64_t)(2 * s) * q, lb = i + (uint64_t)(2 * s + 1) * q;  // the pair's two leaves
    for (uint32_t blk = 0; blk < k; ++blk) {
      uint32_t m[16];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint32_t e = 2 * blk + half;  // element index inside leafA || leafB
        const uint64_t leaf = e < k ? la : lb;
        const uint32_t c = e < k ? e : e - k;
        load8(evals + ((uint64_t)c * n + leaf) * 8, m + 8 * half);
        // the leaf itself (each element is loaded exactly once per pair)
        store8(leaves_out + ((4 * i + 2 * s + (e < k ? 0 : 1)) * (uint64_t)k + c) * 8, m + 8 * half);
      }
      b2_compress(d[s].h, m, 64 * (blk + 1), blk + 1 == k);
    }
    store8(nodes + (n / 2 + 2 * i + s) * 8, d[s].h);
  }
  b2digest top = b2_hash_pair(d[0].h, d[1].h);
  store8(nodes + (n / 4 + i) * 8, top.h);
  if (i == 0) {
    uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    store8(nodes, z);
  }
}

// ---- FRI fold ---------------------------------------------------------------------------------------
// column[i] = P_i(x*), P_i the cubic through (x_i I^j, v_j), v_j = values[i + j q], x_i = w^i, I = w^q.
// With g = 1/4 * iDFT_4(v) (w.r.t. I) the cubic is sum_k g_k (x / x_i)^k, so
//   column[i] = 1/4 * (G0 + G1 t + G2 t^2 + G3 t^3),   t = x* / x_i = x* * w^(-i),
//   G0 = u0 + u2, G2 = u0 - u2, G1 = u1 + u3, G3 = u1 - u3,
//   u0 = v0 + v2, u1 = v0 - v2, u2 = v1 + v3, u3 = (v1 - v3) / I.
// Residues are unique, so this equals the reference's Lagrange route (poly_utils.py:412-440) bit for bit.
__device__ __forceinline__ fp fri_fold_row(const FoldArgs& a, uint64_t b, uint64_t i) {
  const uint64_t q = a.n >> 2;
  uint32_t sxw[8];
  if (a.nodes) {
    load8(a.nodes + (b * 2 * a.n + 1) * 8, sxw);  // special_x = field(m[1]) (fri.py:229), unreduced bytes
  } else {
    load8(a.special_x, sxw);
  }
  const fp sx = fp_from_wire_words(sxw);
  const fp* v = a.values + b * a.n + i;
  const fp v0 = fp_load(v), v1 = fp_load(v + q), v2 = fp_load(v + 2 * q), v3 = fp_load(v + 3 * q);
  // w_r^(-i) = w0^(n0 - i * 2^shift)
  const uint64_t n0m = (1ull << a.log_n0) - 1;
  const uint64_t e = ((1ull << a.log_n0) - ((i << a.round_shift) & n0m)) & n0m;
  fp winv = fp_load(a.tw_lo + (e & ((1ull << a.tw_lb) - 1)));
  if (a.tw_hi) winv = fp_mul(winv, fp_load(a.tw_hi + (e >> a.tw_lb)));
  const fp t = fp_mul(sx, winv);
  const fp u0 = fp_add(v0, v2), u1 = fp_sub(v0, v2), u2 = fp_add(v1, v3);
  const fp u3 = fp_mul(fp_sub(v1, v3), a.inv_i);
  const fp G0 = fp_add(u0, u2), G2 = fp_sub(u0, u2), G1 = fp_add(u1, u3), G3 = fp_sub(u1, u3);
  fp acc = fp_add(fp_mul(G3, t), G2);
  acc = fp_add(fp_mul(acc, t), G1);
  acc = fp_add(fp_mul(acc, t), G0);
  return fp_div4(acc);  // the 1/4 by shifting (fp256.cuh)
}
__global__ void __launch_bounds__(TPB) fri_fold_kernel(FoldArgs a) {
  const uint64_t q = a.n >> 2;
  uint64_t g = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (g >= q * a.batch) return;
  const uint64_t b = g / q, i = g - b * q;
  fp_store(a.column + b * q + i, fri_fold_row(a, b, i));
}

// The serial part of a tree, where each level waits for the one below: a workgroup of QUADS quads reduces 2^levels (<= 2 QUADS)
// adjacent nodes of level L to one node of level L - levels, one quad-lane BLAKE2s (blake2s.cuh) per parent, children handed up
// through the quads' LDS message slots: ~0.8 us per level (the compression's own chain is 0.64 us) against 2.3 us for a
// lane-per-hash compression by a wave that is alone on its SIMD.  QUADS = 64 (7 levels per launch) or 128 (8 levels: a 15-level
// serial part takes two launches instead of three).
// LEAVES = TOP_FROM_VALUES: the 2^levels nodes are LEAVES (L = log2 n), derived from the value array [batch][n] (limb form ->
// canonical, big-endian, permute4 order; the internal trees do not store their leaf level): the whole bottom of a small tree in the
// serial form, one launch instead of the leaf kernel (three levels of lane-per-hash compressions) and a first top launch.
// LEAVES = TOP_FROM_FOLD: the values are not there yet -- they are THIS round's FRI column, folded here from the previous round's
// values (one row per leaf, FoldArgs; written to fa.column on the way): the fold launch of a small round disappears into the first
// launch of its column's tree.  (profiles/r04_serial_small_trees_ab.txt)
enum { TOP_FROM_NODES = 0, TOP_FROM_VALUES = 1, TOP_FROM_FOLD = 2 };
template <int QUADS, int LEAVES = TOP_FROM_NODES>
__global__ void __launch_bounds__(4 * QUADS) merkle_top_kernel(uint32_t* nodes, uint64_t n, uint32_t L, uint32_t levels,
                                                               const fp* values, FoldArgs fa) {
  constexpr bool FROM_VALUES = LEAVES != TOP_FROM_NODES;
  __shared__ __attribute__((aligned(16))) uint32_t slots[QUADS * 16];
  uint32_t* tree = nodes + (uint64_t)blockIdx.y * (2 * n) * 8;
  const uint32_t tid = threadIdx.x, quad = tid >> 2, q = tid & 3;
  b2q_addr ad;
  b2q_addr_init(ad, quad * 64, q);
  uint32_t active = 1u << (levels - 1);  // quads hashing in this step
  if (quad < active) {
    const uint64_t node = (1ull << L) + ((uint64_t)blockIdx.x << levels) + 2 * quad;
    if (FROM_VALUES) {
      if ((q & 1u) == 0) {  // lanes 0 and 2 of the quad fetch the pair's two leaves: slot s holds the value at (s & 3) n/4 + (s >> 2)
        const uint64_t sl = node - n + (q >> 1), n4 = n >> 2;
        uint32_t w[8];
        const uint64_t idx = (sl & 3) * n4 + (sl >> 2);
        fp leaf;
        if (LEAVES == TOP_FROM_FOLD) {
          leaf = fri_fold_row(fa, blockIdx.y, idx);
          fp_store(fa.column + (uint64_t)blockIdx.y * n + idx, leaf);
        } else {
          leaf = fp_load(values + (uint64_t)blockIdx.y * n + idx);
        }
        fp_to_wire_words(fp_canon(leaf), w);
        uint4* dst = reinterpret_cast<uint4*>(slots + quad * 16 + 8 * (q >> 1));
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
      }
      if (blockIdx.x == 0 && tid == 0) {
        uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        store8(tree, z);  // nodes[0]: the reference keeps b'' there
      }
    } else {
      const uint4 v = *reinterpret_cast<const uint4*>(tree + node * 8 + 4 * q);  // 16 of the pair's 64 bytes
      *reinterpret_cast<uint4*>(slots + quad * 16 + 4 * q) = v;
    }
  }
  uint32_t lvl = L;
  for (uint32_t s = 0; s < levels; ++s) {
    __syncthreads();  // message slots written
    uint32_t h_lo = 0, h_hi = 0;
    lvl -= 1;
    if (quad < active) {
      b2q_compress(ad, slots, q, 64, h_lo, h_hi);
      uint32_t* out = tree + ((1ull << lvl) + ((uint64_t)blockIdx.x << (levels - 1 - s)) + quad) * 8;
      out[q] = h_lo;
      out[4 + q] = h_hi;
    }
    __syncthreads();  // everyone has read its slot: parents' slots may be overwritten
    if (quad < active) {
      uint32_t* dst = slots + (quad >> 1) * 16 + 8 * (quad & 1);
      dst[q] = h_lo;
      dst[4 + q] = h_hi;
    }
    active >>= 1;
  }
}

// ---- query sampling + branch gather --------------------------------------------------------------------
// get_pseudorandom_indices(root, modulus, samples, exclude_multiples_of) (utils.py:60-90): one QUAD of lanes per proof.
// data = root, then data += blake(data[-32:]) (utils.py:74-75): a serial chain, so each 32-byte block is hashed
// with the low-latency quad-lane BLAKE2s; after block k lane q holds words q and 4+q = samples 8k+q and 8k+4+q.
__device__ __forceinline__ void sample_indices_quad(uint32_t* slots, const uint32_t* root, bool live, uint32_t modulus, uint32_t samples,
                                                    uint32_t exclude, uint32_t* ys) {
  const uint32_t tid = threadIdx.x, quad = tid >> 2, q = tid & 3;
  b2q_addr ad;
  b2q_addr_init(ad, quad * 64, q);
  uint32_t w_lo = 0, w_hi = 0;
  if (live) {
    w_lo = root[q];
    w_hi = root[4 + q];
  }
  uint32_t* slot = slots + quad * 16;
  slot[8 + q] = 0;   // a 32-byte message: words 8..15 are zero padding
  slot[12 + q] = 0;
  const uint32_t real = exclude ? (uint32_t)(((uint64_t)modulus * (exclude - 1)) / exclude) : modulus;
  const uint32_t blocks = (samples + 7) / 8;
  for (uint32_t k = 0; k < blocks; ++k) {
    if (live) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint32_t j = 8 * k + 4 * half + q;
        if (j < samples) {
          const uint32_t x = __builtin_bswap32(half ? w_hi : w_lo) % real;  // int.from_bytes(data[4j:4j+4], 'big') % modulus
          ys[j] = exclude ? x + 1 + x / (exclude - 1) : x;
        }
      }
    }
    if (k + 1 == blocks) break;
    slot[q] = w_lo;
    slot[4 + q] = w_hi;
    __syncthreads();
    b2q_compress(ad, slots, q, 32, w_lo, w_hi);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(64) sample_indices_kernel(const uint32_t* nodes, uint64_t tree_words, uint32_t modulus,
                                                            uint32_t batch, uint32_t samples, uint32_t exclude,
                                                            uint32_t* ys_out) {
  __shared__ __attribute__((aligned(16))) uint32_t slots[16 * 16];
  const uint32_t b = blockIdx.x * 16 + (threadIdx.x >> 2);
  const bool live = b < batch;
  sample_indices_quad(slots, nodes + (uint64_t)b * tree_words + 8, live, modulus, samples, exclude,  // entropy = node 1 of proof b's tree
                      ys_out + (uint64_t)b * samples);
}
// all rounds of a commit: blockIdx.y = round; entropy = the root of the round's column tree (fri.py:246)
__global__ void __launch_bounds__(64) fri_sample_all_kernel(FriSampleArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t slots[16 * 16];
  const FriRound& r = a.r[blockIdx.y];
  const uint32_t b = blockIdx.x * 16 + (threadIdx.x >> 2);
  const bool live = b < a.batch;
  const uint64_t q = r.n >> 2;
  sample_indices_quad(slots, r.nodes_m2 + (uint64_t)b * (2 * q * 8) + 8, live, (uint32_t)q, r.samples, a.exclude,
                      a.ys + r.ys_off + (uint64_t)b * r.samples);
}
// mk_branch (merkle_tree.py:59-68) for the 5 branches of every sample of every round, written into the flat proofs.
__global__ void __launch_bounds__(TPB) fri_gather_all_kernel(FriSampleArgs a) {
  const uint64_t g0 = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  if (g0 >= a.work_total) {  // the final layer: [x.to_bytes() for x in values] (fri.py:214)
    const uint64_t g = g0 - a.work_total;
    if (g >= a.final_n * a.batch) return;
    const uint64_t b = g / a.final_n, i = g - b * a.final_n;
    uint32_t w[8];
    fp_to_wire_words(fp_canon(fp_load(a.final_values + g)), w);
    store8(reinterpret_cast<uint32_t*>(a.proof + b * a.proof_stride + a.final_off) + 8 * i, w);
    return;
  }
  uint32_t ri = 0;
#pragma unroll 1
  while (ri + 1 < a.rounds && g0 >= a.r[ri + 1].work_begin) ++ri;
  const FriRound& rd = a.r[ri];
  const uint64_t g = g0 - rd.work_begin;
  const uint64_t q = rd.n >> 2;
  uint32_t l1 = 1;
  while ((1ull << (l1 - 1)) < rd.n) ++l1;  // log2(n) + 1
  const uint32_t l2 = l1 - 2;              // log2(n/4) + 1
  const uint32_t per_sample = l2 + 4 * l1;
  const uint64_t per_proof = (uint64_t)rd.samples * per_sample + 1;  // +1: the root2 slot
  const uint64_t b = g / per_proof;
  uint64_t r = g - b * per_proof;
  uint32_t* out = reinterpret_cast<uint32_t*>(a.proof + b * a.proof_stride + rd.round_off);
  const uint32_t* m = rd.nodes_m + b * 2 * rd.n * 8;
  const uint32_t* m2 = rd.nodes_m2 + b * 2 * q * 8;
  uint32_t w[8];
  if (r == 0) {
    load8(m2 + 8, w);
    store8(out, w);
    return;
  }
  r -= 1;
  const uint32_t s = (uint32_t)(r / per_sample), slot = (uint32_t)(r - (uint64_t)s * per_sample);
  const uint32_t y = a.ys[rd.ys_off + b * rd.samples + s];
  const uint32_t* tree;
  const fp* vals;
  uint64_t leaves, index;
  uint32_t lev;
  if (slot < l2) {
    tree = m2; vals = rd.column + b * q; leaves = q; index = y; lev = slot;
  } else {
    const uint32_t br = (slot - l2) / l1;
    tree = m; vals = rd.values + b * rd.n; leaves = rd.n; index = y + q * br; lev = (slot - l2) - br * l1;
  }
  const uint64_t ld4 = leaves >> 2;  // get_index_in_permuted (merkle_tree.py:26-33)
  const uint64_t pi = index / ld4 + 4 * (index % ld4);
  if (lev <= 1) {
    // branch entries 0 and 1 are the leaf and its sibling leaf = x.to_bytes() of the values (merkle_tree.py:47-53), which
    // the internal trees do not materialise: permuted slot pi (or pi ^ 1) holds the value at (pi & 3) * n/4 + (pi >> 2)
    const uint64_t ps = pi ^ lev;
    fp_to_wire_words(fp_canon(fp_load(vals + (ps & 3) * ld4 + (ps >> 2))), w);
  } else {
    load8(tree + (((pi + leaves) >> (lev - 1)) ^ 1) * 8, w);
  }
  store8(out + 8 + ((uint64_t)s * per_sample + slot) * 8, w);
}
}  // namespace

hipError_t shk_wire_to_limb(const uint8_t* d_wire, fp* d_limbs, uint64_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(wire_to_limb_kernel, flat_grid_for(n), dim3(TPB), 0, st, reinterpret_cast<const uint32_t*>(d_wire),
                     d_limbs, n);
  return hipGetLastError();
}
hipError_t shk_limb_to_wire(const fp* d_limbs, uint8_t* d_wire, uint64_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(limb_to_wire_kernel, flat_grid_for(n), dim3(TPB), 0, st, d_limbs, reinterpret_cast<uint32_t*>(d_wire), n);
  return hipGetLastError();
}
hipError_t shk_fill_seeded(fp* d, uint64_t n, uint64_t seed, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(fill_seeded_kernel, flat_grid_for(n), dim3(TPB), 0, st, d, n, seed);
  return hipGetLastError();
}
hipError_t shk_fill_mimc_units(fp* wit, fp* inputs, uint64_t steps, uint32_t first_unit, uint32_t batch, uint32_t constant,
                               hipStream_t st) {
  if (!batch || !steps) return hipSuccess;
  hipLaunchKernelGGL(fill_mimc_units_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, wit, inputs, steps, first_unit, batch,
                     constant);
  return hipGetLastError();
}
hipError_t shk_pointwise_mul(const fp* a, const fp* b, fp* out, uint64_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(pointwise_mul_kernel, flat_grid_for(n), dim3(TPB), 0, st, a, b, out, n);
  return hipGetLastError();
}
hipError_t shk_powers(const fp* lo, const fp* hi, uint32_t lb, fp* out, uint64_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(powers_kernel, flat_grid_for(n), dim3(TPB), 0, st, lo, hi, lb, out, n);
  return hipGetLastError();
}
hipError_t shk_tw2(const fp* lo, const fp* hi, uint32_t lb, fp* out, uint32_t log_R, uint32_t log_S, hipStream_t st) {
  const uint64_t n = 1ull << (log_R + log_S);
  hipLaunchKernelGGL(tw2_kernel, dim3(grid_for(n)), dim3(TPB), 0, st, lo, hi, lb, out, log_R, log_S);
  return hipGetLastError();
}
hipError_t shk_pad_copy(const fp* src, fp* dst, uint64_t n_in, uint64_t n, uint32_t batch, hipStream_t st) {
  const uint64_t total = n * batch;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(pad_copy_kernel, flat_grid_for(total), dim3(TPB), 0, st, src, dst, n_in, n, total);
  return hipGetLastError();
}
// the serial levels L .. 1 -> 0 of a tree: as few launches as 8 levels per launch allow, the levels dealt evenly
// (15 = 8 + 7, 13 = 7 + 6, 11 = 6 + 5)
constexpr uint64_t MERKLE_SERIAL_MAX_LEAVES = 1ull << 15;  // trees (times batch) up to this many leaves start in serial form
static hipError_t merkle_serial_levels(int L, uint64_t n, uint32_t batch, uint32_t* d_nodes, hipStream_t st) {
  while (L > 0) {
    const int launches = (L + 7) / 8, levels = (L + launches - 1) / launches;
    if (levels == 8)
      hipLaunchKernelGGL((merkle_top_kernel<128, TOP_FROM_NODES>), dim3(1u << (L - levels), batch), dim3(512), 0, st, d_nodes, n, (uint32_t)L,
                         (uint32_t)levels, (const fp*)nullptr, FoldArgs{});
    else
      hipLaunchKernelGGL((merkle_top_kernel<64, TOP_FROM_NODES>), dim3(1u << (L - levels), batch), dim3(256), 0, st, d_nodes, n, (uint32_t)L,
                         (uint32_t)levels, (const fp*)nullptr, FoldArgs{});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    L -= levels;
  }
  return hipSuccess;
}
hipError_t shk_merkelize(const void* d_leaves, bool raw_leaves, uint64_t n, uint32_t batch, uint32_t* d_nodes,
                         hipStream_t st, bool store_leaves) {
  if (n < 4 || (n & (n - 1)) || batch == 0) return hipErrorInvalidValue;
  // small trees without a stored leaf level (the later FRI rounds): serial from the leaves up, 8 levels per launch
  if (!raw_leaves && !store_leaves && n * batch <= MERKLE_SERIAL_MAX_LEAVES) {
    uint32_t logn = 0;
    while ((1ull << logn) < n) ++logn;
    const int launches = ((int)logn + 7) / 8, levels = ((int)logn + launches - 1) / launches;
    if (levels == 8)
      hipLaunchKernelGGL((merkle_top_kernel<128, TOP_FROM_VALUES>), dim3(1u << (logn - levels), batch), dim3(512), 0, st, d_nodes, n, logn,
                         (uint32_t)levels, reinterpret_cast<const fp*>(d_leaves), FoldArgs{});
    else
      hipLaunchKernelGGL((merkle_top_kernel<64, TOP_FROM_VALUES>), dim3(1u << (logn - levels), batch), dim3(256), 0, st, d_nodes, n, logn,
                         (uint32_t)levels, reinterpret_cast<const fp*>(d_leaves), FoldArgs{});
    const hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return e0;
    return merkle_serial_levels((int)logn - levels, n, batch, d_nodes, st);
  }
  const dim3 lgrid(grid_for(n >> 2, TPB * MERKLE_ROWS), batch);
  const bool wide = (n >> 2) * batch >= MERKLE_WIDE_THREADS;
#define SHK_LEAVES(RAW, STORE)                                                                                           \
  do {                                                                                                                   \
    if (wide)                                                                                                            \
      hipLaunchKernelGGL((merkle_leaves_kernel<RAW, STORE, true>), lgrid, dim3(TPB), 0, st, d_leaves, n, d_nodes);      \
    else                                                                                                                 \
      hipLaunchKernelGGL((merkle_leaves_kernel<RAW, STORE, false>), lgrid, dim3(TPB), 0, st, d_leaves, n, d_nodes);     \
  } while (0)
  if (raw_leaves)
    SHK_LEAVES(true, true);
  else if (store_leaves)
    SHK_LEAVES(false, true);
  else
    SHK_LEAVES(false, false);
#undef SHK_LEAVES
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return shk_merkle_upper_levels(n, batch, d_nodes, st);
}
// levels log2(n)-2 .. 0 from the nodes the leaf kernels wrote (node layout of merkle_tree.py:36-56)
constexpr int MID_MIN_LOG = 15;  // a level pair goes to the (throughput-form) mid kernel while it has at least 2^this threads
hipError_t shk_merkle_upper_levels(uint64_t n, uint32_t batch, uint32_t* d_nodes, hipStream_t st) {
  hipError_t e = hipSuccess;
  uint32_t logn = 0;
  while ((1ull << logn) < n) ++logn;
  int L = (int)logn - 2;
  // wide levels: two levels per launch at full lane efficiency, while a level still fills the chip
  // (the threshold is flat: 2^15 .. 2^19 threads measure within +-1.5 % of each other on 2^20 .. 2^24 leaves and on the FRI
  // commits, profiles/r04_merkle_lab.txt)
  while (L >= 2 && ((1ull << (L - 2)) * batch) >= (1ull << MID_MIN_LOG)) {
    if (((1ull << (L - 2)) * batch) >= MERKLE_WIDE_THREADS / 2)
      hipLaunchKernelGGL(merkle_mid_kernel<true>, dim3(grid_for(1ull << (L - 2)), batch), dim3(TPB), 0, st, d_nodes, n, (uint32_t)L);
    else
      hipLaunchKernelGGL(merkle_mid_kernel<false>, dim3(grid_for(1ull << (L - 2)), batch), dim3(TPB), 0, st, d_nodes, n, (uint32_t)L);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    L -= 2;
  }
  return merkle_serial_levels(L, n, batch, d_nodes, st);
}
// column = fold(values) and the whole tree of the column (fri.py:235-243): small rounds fold inside the first launch of the tree
hipError_t shk_fri_fold_and_tree(const FoldArgs& a, uint32_t* d_nodes2, hipStream_t st) {
  const uint64_t q = a.n >> 2;
  if (q < 4 || q * a.batch > MERKLE_SERIAL_MAX_LEAVES) {
    hipError_t e = shk_fri_fold(a, st);
    if (e != hipSuccess) return e;
    return shk_merkelize(a.column, false, q, a.batch, d_nodes2, st, false);
  }
  uint32_t logq = 0;
  while ((1ull << logq) < q) ++logq;
  const int launches = ((int)logq + 7) / 8, levels = ((int)logq + launches - 1) / launches;
  if (levels == 8)
    hipLaunchKernelGGL((merkle_top_kernel<128, TOP_FROM_FOLD>), dim3(1u << (logq - levels), a.batch), dim3(512), 0, st, d_nodes2, q, logq,
                       (uint32_t)levels, (const fp*)nullptr, a);
  else
    hipLaunchKernelGGL((merkle_top_kernel<64, TOP_FROM_FOLD>), dim3(1u << (logq - levels), a.batch), dim3(256), 0, st, d_nodes2, q, logq,
                       (uint32_t)levels, (const fp*)nullptr, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return merkle_serial_levels((int)logq - levels, q, a.batch, d_nodes2, st);
}
hipError_t shk_fri_fold(const FoldArgs& a, hipStream_t st) {
  const uint64_t work = (a.n >> 2) * a.batch;
  if (!work) return hipSuccess;
  hipLaunchKernelGGL(fri_fold_kernel, dim3(grid_for(work)), dim3(TPB), 0, st, a);
  return hipGetLastError();
}
hipError_t shk_sample_indices(const uint32_t* d_nodes, uint64_t tree_words, uint32_t modulus, uint32_t batch,
                              uint32_t samples, uint32_t exclude, uint32_t* d_ys, hipStream_t st) {
  hipLaunchKernelGGL(sample_indices_kernel, dim3((batch + 15) / 16), dim3(64), 0, st, d_nodes, tree_words, modulus, batch,
                     samples, exclude, d_ys);
  return hipGetLastError();
}
hipError_t shk_fri_sample_and_gather_all(const FriSampleArgs& a, hipStream_t st) {
  if (!a.batch) return hipSuccess;
  if (!a.rounds) {  // a direct proof (maxdeg_plus_1 <= 16): the final layer is all there is
    hipLaunchKernelGGL(fri_gather_all_kernel, dim3(grid_for(a.final_n * a.batch)), dim3(TPB), 0, st, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(fri_sample_all_kernel, dim3((a.batch + 15) / 16, a.rounds), dim3(64), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fri_gather_all_kernel, dim3(grid_for(a.work_total + a.final_n * a.batch)), dim3(TPB), 0, st, a);
  return hipGetLastError();
}
hipError_t shk_merkelize_packed(const uint8_t* d_evals, uint64_t n, uint32_t k, uint8_t* d_leaves, uint32_t* d_nodes,
                                hipStream_t st) {
  if (n < 4 || (n & (n - 1)) || k == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(merkle_packed_leaves_kernel, dim3(grid_for(n >> 2)), dim3(TPB), 0, st,
                     reinterpret_cast<const uint32_t*>(d_evals), n, k, reinterpret_cast<uint32_t*>(d_leaves), d_nodes);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // interior levels: the ordinary tree over n "virtual" 32-byte leaves; only the node half of the buffer is touched
  return shk_merkle_upper_levels(n, 1, d_nodes, st);
}
