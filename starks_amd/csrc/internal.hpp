// internal.hpp -- declarations shared by the translation units of libstarkhip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp256.cuh"

struct NttPassArgs {
  const fp* src;
  fp* dst;
  uint64_t total;     // number of columns (column pass: batch*P*S) or rows (row pass: batch*P)
  uint32_t log_n;     // log2 of the transform length
  uint32_t log_S;     // column pass: log2 stride between successive points of a column
  uint32_t log_P;     // row pass: log2 number of rows per transform (n / R)
  const fp2* wR;      // w^(n/R * k), k < R/2, each with its second image w * 2^128 mod p (fp256.cuh: fp_mul2)
  // column pass: table of g = w^P, order R*S:  g^e = lo[e & mask] (* hi[e >> lb] unless direct)
  const fp* tw_lo;
  const fp* tw_hi;
  uint32_t tw_lb;
  uint32_t tw_direct;
  // row pass: radix logs of the earlier passes (digits k_1 .. k_{m-1} of the row number, k_1 first)
  uint32_t ndig;
  uint32_t dig_log[3];
  const fp* scale;    // row pass: optional factor applied to every output (n^-1 of a one-pass inverse)
  uint64_t src_n;     // first pass only: the source holds src_n <= n elements per vector, the rest of each vector is zero
                      // (fft_1d's zero padding, fft.py:323-324, never materialised); 0 = the source holds n per vector
  const fp* tw2;      // column passes: the same twiddles as [k][j2] rows, tw2[k * S + j2] = g^(j2 * k) -- one coalesced load per
                      // element; null: use the power table (tw_lo / tw_hi)
  uint32_t pass_index; // 0 = the first pass of the transform (experiments: STARKHIP_TILE_LOGS picks a tile size per pass)
  uint32_t xcd_per;   // 0: tile = blockIdx.x.  Else workgroups are dealt to the 8 XCDs round-robin and tile = (blockIdx.x & 7) *
                      // xcd_per + (blockIdx.x >> 3): adjacent tiles run on the SAME XCD (they share 128-byte lines when T < 4)
};

// ---- ntt.hip ----------------------------------------------------------------------------------
// Launch one tile pass (radix 2^log_R).  Returns hipSuccess or the launch error.
hipError_t shk_launch_ntt_pass(int log_R, bool last, const NttPassArgs& a, hipStream_t st);
hipError_t shk_launch_ntt_tiny(const fp* src, fp* dst, uint32_t n, uint32_t batch, const fp* scale, hipStream_t st);

// ---- kernels.hip: conversions, powers, Merkle, FRI fold, sampling, branch gather ------------------
hipError_t shk_wire_to_limb(const uint8_t* d_wire, fp* d_limbs, uint64_t n, hipStream_t st);
hipError_t shk_limb_to_wire(const fp* d_limbs, uint8_t* d_wire, uint64_t n, hipStream_t st);
hipError_t shk_fill_seeded(fp* d, uint64_t n, uint64_t seed, hipStream_t st);
hipError_t shk_fill_mimc_units(fp* wit, fp* inputs, uint64_t steps, uint32_t first_unit, uint32_t batch, uint32_t constant,
                               hipStream_t st);
hipError_t shk_pointwise_mul(const fp* a, const fp* b, fp* out, uint64_t n, hipStream_t st);
// out[i] = g^i for i < n, g given by its two-level table (lo, hi, lb)
hipError_t shk_powers(const fp* lo, const fp* hi, uint32_t lb, fp* out, uint64_t n, hipStream_t st);
hipError_t shk_tw2(const fp* lo, const fp* hi, uint32_t lb, fp* out, uint32_t log_R, uint32_t log_S, hipStream_t st);
// zero-pad: dst[b][0..n_in) = src[b][0..n_in), dst[b][n_in..n) = 0
hipError_t shk_pad_copy(const fp* src, fp* dst, uint64_t n_in, uint64_t n, uint32_t batch, hipStream_t st);
// Merkle tree of `batch` arrays of n limb-form values (or n raw 32-byte leaves when raw_leaves).
// store_leaves = false leaves nodes[n, 2n) unwritten (limb-form input only): for callers that gather leaves from the values
hipError_t shk_merkelize(const void* d_leaves, bool raw_leaves, uint64_t n, uint32_t batch, uint32_t* d_nodes,
                         hipStream_t st, bool store_leaves = true);
hipError_t shk_merkle_upper_levels(uint64_t n, uint32_t batch, uint32_t* d_nodes, hipStream_t st);
// packed leaves (merkle_tree.py:94-119): d_evals [k][n][32 B] wire; d_leaves [n][k][32 B] permuted; d_nodes [n][32 B]
hipError_t shk_merkelize_packed(const uint8_t* d_evals, uint64_t n, uint32_t k, uint8_t* d_leaves, uint32_t* d_nodes,
                                hipStream_t st);
struct FoldArgs {
  const fp* values;        // [batch][n]
  const uint32_t* nodes;   // [batch][2n][8 words]; challenge = node 1 (nullptr: use special_x)
  const uint32_t* special_x;  // 8 wire words (used when nodes == nullptr)
  fp* column;              // [batch][n/4]
  uint64_t n;
  uint32_t batch;
  // powers of the ROUND-0 generator w0 (order n0): w0^e = lo[e & mask] * hi[e >> lb]
  const fp* tw_lo;
  const fp* tw_hi;
  uint32_t tw_lb;
  uint32_t log_n0;
  uint32_t round_shift;    // this round's generator is w0^(2^round_shift)
  fp inv_i;                // (w0^(n0/4))^-1: inverse of the primitive 4th root of unity
};
hipError_t shk_fri_fold(const FoldArgs& a, hipStream_t st);
// the fold and the Merkle tree of its column (nodes layout of shk_merkelize without the leaf level): [batch][2 * n/4][8 words]
hipError_t shk_fri_fold_and_tree(const FoldArgs& a, uint32_t* d_nodes2, hipStream_t st);
// Query sampling + branch gather of ALL rounds of a FRI commit in two launches (fri.py:246-254 per round): every round keeps
// its values and its tree until the end of the commit, so nothing on the serial chain tree -> challenge -> fold -> tree waits
// for the 40 x 5 branch copies of the round before.
constexpr uint32_t SHK_FRI_MAX_ROUNDS = 12;  // domain <= 2^26 (the column of round 0 must stay below 2^24 rows, utils.py:69)
struct FriRound {
  const fp* values;          // [batch][n]   the values under nodes_m  (leaves are re-derived from them)
  const fp* column;          // [batch][n/4] the values under nodes_m2
  const uint32_t* nodes_m;   // [batch][2n][8]   tree of the values
  const uint32_t* nodes_m2;  // [batch][2q][8]   tree of the column (q = n/4)
  uint64_t n;
  uint64_t round_off;        // byte offset of this round inside a proof
  uint32_t samples;
  uint32_t ys_off;           // this round's slice of the ys scratch: ys[(ys_off + b * samples) ...]
  uint64_t work_begin;       // gather: first work item of this round (prefix sum of (samples * slots + 1) * batch)
};
struct FriSampleArgs {
  FriRound r[SHK_FRI_MAX_ROUNDS];
  uint32_t rounds;
  uint32_t batch;
  uint32_t exclude;
  uint32_t* ys;              // scratch, sum over rounds of batch * samples
  uint8_t* proof;            // [batch][proof_stride] device proof buffers
  uint64_t proof_stride;
  uint64_t work_total;
  // the final layer (fri.py:212-214), written by the same gather launch: canonical wire form of final_values[b][0..final_n) at
  // proof[b] + final_off
  const fp* final_values;
  uint64_t final_n;
  uint64_t final_off;
};
hipError_t shk_fri_sample_and_gather_all(const FriSampleArgs& a, hipStream_t st);
// ys[b][0..samples) = get_pseudorandom_indices(node 1 of tree b, modulus, samples, exclude) (utils.py:60-90);
// trees are tree_words u32 apart
hipError_t shk_sample_indices(const uint32_t* d_nodes, uint64_t tree_words, uint32_t modulus, uint32_t batch,
                              uint32_t samples, uint32_t exclude, uint32_t* d_ys, hipStream_t st);

// ---- stark.hip: constraint / boundary quotients, packed tree, linear combination, spot checks (stark.py:233-279) ----
constexpr uint32_t SHK_STARK_MAX_WIDTH = 9;  // get_pseudorandom_ks returns None from 10 on (stark.py:106-126)
constexpr uint32_t SHK_STARK_MAX_TERMS = 256;
struct StarkArgs {
  const fp* p_evals;  // [batch * width][n]  trace polynomials on the G2 domain
  fp* d_work;         // [batch * width][n]  D evaluations
  fp* b_work;         // [batch * width][n]  B evaluations
  const fp* q_evals;  // [batch * width][steps]  Q_c = X P_c'(X) on G1
  const fp* wit;      // [batch * width][steps]  the witness itself = the trace polynomials on G1 (read contiguously by the
                      // trace-point kernel instead of every ext-th entry of p_evals)
  const fp* iab;      // [batch * width][3]  boundary interpolant a + b X: a, b, b 2^128 (b as an fp_mul2 pair)
  uint64_t n;         // precision = steps * ext
  uint64_t steps;
  uint32_t ext;
  uint32_t width;
  uint32_t batch;
  const fp* tw_lo;    // powers of G2: G2^e = lo[e & mask] (* hi[e >> lb] when hi)
  const fp* tw_hi;
  uint32_t tw_lb;
  // per-(steps, ext) tables over the domain (cached by the context, shared by every proof of every batch):
  const fp* inv_z2;     // [n]   1 / ((x_i - 1)(x_i - x_last)), 0 at the two roots
  const fp* xpow;       // [n]   x_i = G2^i
  const fp* fz;         // [n]   (x_i - x_last) / (x_i^steps - 1) = 1 / Z(x_i), 0 on the trace points (i % ext == 0)
                        // (as fp_mul2 pairs the two factor tables measured 1 % slower: 64 B more table traffic per point)
  const fp* inv_omega;  // [ext] 1 / (omega^j - 1), omega = G2^steps, entry 0 = 0
  fp x_last;          // G2^((steps - 1) ext)  (stark.py:212)
  fp g1;              // G2^ext = 1 / x_last
  fp inv_steps;       // 1 / steps
  fp inv_1_m_last;    // 1 / (1 - x_last)
  uint32_t* bad;      // [batch] bad[b] is set to 1 when a transition constraint of proof b fails on the trace
  // step polynomials: term t = coef[t] * prod_v X_v^exps[t][v] (exps rows are width + 1 bytes: the last one flags
  // coef == 1); terms of dimension c: [term_begin[c], term_begin[c+1])
  const fp* term_coef;
  const uint8_t* term_exps;
  uint32_t term_begin[SHK_STARK_MAX_WIDTH + 1];
  // their partial derivatives: d step_c / d X_v = terms [dterm_begin[c * width + v], dterm_begin[c * width + v + 1])
  const fp* dterm_coef;
  const uint8_t* dterm_exps;
  const uint32_t* dterm_begin;
};
hipError_t shk_stark_interp(const fp* trace, const fp* inputs, uint64_t steps, uint32_t cols, const fp& inv_last_m1, fp* iab,
                            hipStream_t st);
hipError_t shk_stark_qprep(const fp* pcoef, fp* q, uint64_t steps, uint64_t cols, hipStream_t st);
// the three domain tables, out = [3][n]: inv_z2, xpow, fz (inv_omega = the [ext] table, already on the device)
hipError_t shk_stark_domain_tables(fp* out, uint64_t n, uint32_t ext, const fp* tw_lo, const fp* tw_hi, uint32_t tw_lb, const fp& x_last,
                                   const fp* inv_omega, hipStream_t st);
// D = C / Z and B = (P - I) / Z2 on the whole domain and the packed-leaf Merkle tree over (P, D, B), in one pass over the domain (the
// quotients are hashed where they are computed)
hipError_t shk_stark_quotients_and_merkelize(const StarkArgs& a, uint32_t* d_nodes, hipStream_t st);
hipError_t shk_stark_scalars(const uint32_t* d_mnodes, uint64_t tree_words, uint32_t width, uint32_t batch, const fp& cpow,
                             fp* d_scal, hipStream_t st);
// l = the Fiat-Shamir linear combination of P, D, B (scalars as (s, s 2^128) pairs) and the Merkle tree of l
// (merkelize(l_evaluations)): the combination is fused with the tree's first three levels
hipError_t shk_stark_lincomb_tree(const StarkArgs& a, const fp* d_scal, fp* d_l, uint32_t* d_lnodes, hipStream_t st);
hipError_t shk_stark_gather(const StarkArgs& a, const uint32_t* d_mnodes, const uint32_t* d_lnodes, const fp* d_lvals,
                            const uint32_t* d_ys, uint32_t samples, uint8_t* d_proof, uint64_t stride, hipStream_t st);
