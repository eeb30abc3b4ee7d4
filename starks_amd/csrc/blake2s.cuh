// blake2s.cuh -- BLAKE2s-256, unkeyed, default parameters (RFC 7693) = hashlib.blake2s as the reference
// uses it (starks/merkle_tree.py:1-5, starks/utils.py:75).  Register-resident, fully unrolled: the message
// schedule is a compile-time permutation, so every m[sigma[r][i]] is a fixed VGPR.
//
// Merkle interior nodes hash exactly one 64-byte block (two 32-byte children): one compression,
// t = 64, final.  Digests are kept as 8 little-endian u32 words = the byte string itself when stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define B2_HD __host__ __device__ __forceinline__

struct b2digest {
  uint32_t h[8];
};

B2_HD uint32_t b2_rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, n);  // v_alignbit_b32: one instruction
#else
  return (x >> n) | (x << (32 - n));
#endif
}

#define B2_G(a, b, c, d, x, y) \
  do {                         \
    a = a + b + (x);           \
    d = b2_rotr(d ^ a, 16);    \
    c = c + d;                 \
    b = b2_rotr(b ^ c, 12);    \
    a = a + b + (y);           \
    d = b2_rotr(d ^ a, 8);     \
    c = c + d;                 \
    b = b2_rotr(b ^ c, 7);     \
  } while (0)

#define B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  do {                                                                               \
    B2_G(v0, v4, v8, v12, m[s0], m[s1]);                                             \
    B2_G(v1, v5, v9, v13, m[s2], m[s3]);                                             \
    B2_G(v2, v6, v10, v14, m[s4], m[s5]);                                            \
    B2_G(v3, v7, v11, v15, m[s6], m[s7]);                                            \
    B2_G(v0, v5, v10, v15, m[s8], m[s9]);                                            \
    B2_G(v1, v6, v11, v12, m[s10], m[s11]);                                          \
    B2_G(v2, v7, v8, v13, m[s12], m[s13]);                                           \
    B2_G(v3, v4, v9, v14, m[s14], m[s15]);                                           \
  } while (0)

// The ten rounds in plain C++ (host code; device code where a hash is a serial dependency and few waves are resident: the
// compiler interleaves the rounds of two independent hashes, which the asm blocks below do not allow).
B2_HD void b2_compress_cpp(uint32_t h[8], const uint32_t m[16], uint32_t t, bool last) {
  uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
  uint32_t v12 = 0x510E527Fu ^ t, v13 = 0x9B05688Cu, v14 = last ? ~0x1F83D9ABu : 0x1F83D9ABu, v15 = 0x5BE0CD19u;
  B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
  B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4);
  B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8);
  B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13);
  B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9);
  B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11);
  B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10);
  B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5);
  B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0);
  h[0] ^= v0 ^ v8;
  h[1] ^= v1 ^ v9;
  h[2] ^= v2 ^ v10;
  h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12;
  h[5] ^= v5 ^ v13;
  h[6] ^= v6 ^ v14;
  h[7] ^= v7 ^ v15;
}

// Throughput form (device only): the ten rounds as generated asm blocks (gen_blake2s_asm.py: four columns in lock-step, a + b + m as
// one v_add3_u32, a taken branch to the next instruction after every group of rotates -- 48 G pair-hashes/s on the bare hash loop
// against 40 for the compiler's schedule of the C++ rounds; the two-adds form reaches 51 there but loses in sustained, power-limited
// runs, where the instruction count decides; profiles/r04_blake2s_issue_rate_study.txt).  It wins where many waves hash at once
// (the wide Merkle levels, the STARK leaf kernels) and loses where one wave per SIMD walks a chain.
// -DB2_NO_ASM keeps the C++ rounds everywhere (A/B builds).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(B2_NO_ASM)
#define B2_ASM_ROUNDS 1
#ifndef B2_ASM_INC
#define B2_ASM_INC "blake2s_asm.inc"
#endif
#include B2_ASM_INC
#endif

// One compression of a 64-byte block `m` (16 LE words) into chaining value h; t = byte counter, last = final-block flag.
// THROUGHPUT selects the asm rounds on the device.
template <bool THROUGHPUT = true>
B2_HD void b2_compress(uint32_t h[8], const uint32_t m[16], uint32_t t, bool last) {
#if defined(B2_ASM_ROUNDS)
  if (THROUGHPUT) {
    uint32_t v[16] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                      0x510E527Fu ^ t, 0x9B05688Cu, last ? ~0x1F83D9ABu : 0x1F83D9ABu, 0x5BE0CD19u};
    b2_rounds_asm(v, *reinterpret_cast<const uint32_t(*)[16]>(m));
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
    return;
  }
#endif
  b2_compress_cpp(h, m, t, last);
}

B2_HD void b2_init(uint32_t h[8]) {
  h[0] = 0x6A09E667u ^ 0x01010020u;  // digest_length 32, key 0, fanout 1, depth 1
  h[1] = 0xBB67AE85u;
  h[2] = 0x3C6EF372u;
  h[3] = 0xA54FF53Au;
  h[4] = 0x510E527Fu;
  h[5] = 0x9B05688Cu;
  h[6] = 0x1F83D9ABu;
  h[7] = 0x5BE0CD19u;
}

// digest of the 64-byte message left || right (two 32-byte strings given as 8 LE words each):
// nodes[i] = blake(nodes[2i] + nodes[2i+1])   (starks/merkle_tree.py:54-55)
template <bool THROUGHPUT = true>
B2_HD b2digest b2_hash_pair(const uint32_t left[8], const uint32_t right[8]) {
  uint32_t m[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[i] = left[i];
    m[8 + i] = right[i];
  }
  b2digest d;
  b2_init(d.h);
  b2_compress<THROUGHPUT>(d.h, m, 64, true);
  return d;
}

// digest of a message of len <= 64 bytes given as zero-padded words (utils.py:75 hashes 32 bytes;
// the synthetic-input generator hashes 16 bytes)
template <bool THROUGHPUT = true>
B2_HD b2digest b2_hash_short(const uint32_t m[16], uint32_t len) {
  b2digest d;
  b2_init(d.h);
  b2_compress<THROUGHPUT>(d.h, m, len, true);
  return d;
}

// ---- quad-lane BLAKE2s (device only) ---------------------------------------------------------------------
// Four adjacent lanes (a "quad", q = lane & 3) compute ONE compression together: lane q owns column q of the
// 4 x 4 state (a = v[q], b = v[4+q], c = v[8+q], d = v[12+q]); the diagonal step is the column step after
// rotating rows b, c, d by 1, 2, 3 lanes inside the quad, which DPP quad_perm does inside the VALU (no LDS).
// The 16 message words sit in a 64-byte LDS slot owned by the quad; each lane reads the 4 words per round it
// needs through 40 byte addresses it keeps in registers (b2q_addr_init).  One compression costs ~1/3 of the
// single-lane instruction stream per lane, i.e. ~3.5x lower latency -- used where the Merkle tree / the
// Fiat-Shamir chain is a serial dependency (top of the tree, index sampling), not where it is throughput bound.
#if defined(__HIPCC__)
struct b2q_addr {
  uint32_t a[40];  // LDS byte addresses of m[sigma[r][2q]], m[sigma[r][2q+1]], m[sigma[r][8+2q]], m[sigma[r][9+2q]]
};

__device__ __forceinline__ uint32_t b2q_sigma(int r, int i) {
  // sigma rows packed 4 bits per entry (entry i in bits [4i, 4i+4))
  const uint64_t S[10] = {0xfedcba9876543210ull, 0x357b20c16df984aeull, 0x491763eadf250c8bull, 0x8f04a562ebcd1397ull,
                          0xd386cb1efa427509ull, 0x91ef57d438b0a6c2ull, 0xb8293670a4def15cull, 0xa2684f05931ce7bdull,
                          0x5a417d2c803b9ef6ull, 0x0dc3e9bf5167482aull};
  return (uint32_t)(S[r] >> (4 * i)) & 15u;
}

#if defined(B2Q_NO_ASM)
#define B2Q_DIAG_SHIFT 0u
#else
#define B2Q_DIAG_SHIFT 3u
#endif
// slot_base: LDS byte address of this quad's 64-byte message slot
__device__ __forceinline__ void b2q_addr_init(b2q_addr& t, uint32_t slot_base, uint32_t q) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // k = 0,1: column step words 2q, 2q+1 ; k = 2,3: diagonal step words 8+2g, 9+2g of the diagonal g this lane works on
      // (B2Q_DIAG_OF_LANE: g = q when the rows b, c, d come to lane q; g = q - 1 when b stays and a, c, d move, b2q_compress)
      const int basei = (k < 2 ? 0 : 8) + (k & 1);
      const uint32_t g = k < 2 ? q : ((q + B2Q_DIAG_SHIFT) & 3u);
      uint32_t idx = b2q_sigma(r, basei);  // g = 0
      idx = g == 1 ? b2q_sigma(r, basei + 2) : idx;
      idx = g == 2 ? b2q_sigma(r, basei + 4) : idx;
      idx = g == 3 ? b2q_sigma(r, basei + 6) : idx;
      t.a[4 * r + k] = slot_base + 4 * idx;
    }
  }
}

#define B2Q_DPP(x, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xf, 0xf, false))

// Compress the 64-byte block in the quad's LDS slot (byte counter tcount, final block).  `slots` = the LDS array
// the byte offsets in `t` are relative to.  Returns this lane's two digest words: h[q] and h[4+q].
__device__ __forceinline__ uint32_t b2q_word(const uint32_t* slots, uint32_t byte_off) {
  return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(slots) + byte_off);
}
// One half-round (G on this lane's column, or on a diagonal) as ONE asm block of 13 instructions.  Between the column and the diagonal
// arrangement three of the four rows have to move one, two and three lanes; here row b -- the LAST one a G writes and the first one the
// next G reads -- is the row that stays, and a, c, d move, each inside the instruction that first reads it (DPP quad_perm on src0, no
// separate v_mov_b32_dpp): a was written six instructions before its DPP read, d and c likewise, so the "VALU write, then DPP read: two
// wait states" hazard never arises and no s_nop is needed (the compiler's own form: 20 v_mov_b32_dpp and 55 s_nop per compression).
// In the diagonal step lane q therefore works on the diagonal of b_q, i.e. diagonal q - 1 (its message words: B2Q_DIAG_SHIFT above), and
// afterwards holds a of column q - 1, c of column q + 1, d of column q + 2.  0.874 -> 0.62 us per compression for a wave alone on its
// SIMD (lab/r04/quad_latency.hip), which is what a serial tree level costs.  PA / PC / PD: the lane that lane q takes a / c / d from.
// (The first half-round has nothing to move and its inputs may have been written by the instruction right before it: plain form.)
// -DB2Q_NO_ASM keeps the C++ form (A/B builds).
#if !defined(B2Q_NO_ASM)
#define B2Q_DPPMOD(P) " quad_perm:" P " row_mask:0xf bank_mask:0xf\n"
// operands: %0 a, %1 b, %2 c, %3 d (in/out), %4 scratch, %5 mx, %6 my
#define B2Q_G_TAIL_ASM                  \
  "v_alignbit_b32 %3, %3, %3, 16\n"    \
  "v_add_u32 %2, %2, %3\n"             \
  "v_xor_b32 %1, %1, %2\n"             \
  "v_alignbit_b32 %1, %1, %1, 12\n"    \
  "v_add3_u32 %0, %0, %1, %6\n"        \
  "v_xor_b32 %3, %3, %0\n"             \
  "v_alignbit_b32 %3, %3, %3, 8\n"     \
  "v_add_u32 %2, %2, %3\n"             \
  "v_xor_b32 %1, %1, %2\n"             \
  "v_alignbit_b32 %1, %1, %1, 7\n"
#define B2Q_HALF_ASM(PA, PC, PD)                        \
  "v_add_u32 %4, %1, %5\n"                              \
  "v_add_u32_dpp %0, %0, %4" B2Q_DPPMOD(PA)             \
  "v_xor_b32_dpp %3, %3, %0" B2Q_DPPMOD(PD)             \
  "v_alignbit_b32 %3, %3, %3, 16\n"                     \
  "v_add_u32_dpp %2, %2, %3" B2Q_DPPMOD(PC)             \
  "v_xor_b32 %1, %1, %2\n"                              \
  "v_alignbit_b32 %1, %1, %1, 12\n"                     \
  "v_add3_u32 %0, %0, %1, %6\n"                         \
  "v_xor_b32 %3, %3, %0\n"                              \
  "v_alignbit_b32 %3, %3, %3, 8\n"                      \
  "v_add_u32 %2, %2, %3\n"                              \
  "v_xor_b32 %1, %1, %2\n"                              \
  "v_alignbit_b32 %1, %1, %1, 7\n"
#define B2Q_HALF(PA, PC, PD, mx, my)                                                                            \
  do {                                                                                                          \
    uint32_t tmp_;                                                                                              \
    asm(B2Q_HALF_ASM(PA, PC, PD) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(tmp_) : "v"(mx), "v"(my));       \
  } while (0)
// the same with nothing to move (round 0's column step)
#define B2Q_HALF_PLAIN(mx, my)                                                                                  \
  do {                                                                                                          \
    uint32_t tmp_;                                                                                              \
    asm("v_add3_u32 %0, %0, %1, %5\n"                                                                           \
        "v_xor_b32 %3, %3, %0\n" B2Q_G_TAIL_ASM                                                                 \
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(tmp_) : "v"(mx), "v"(my));                                  \
  } while (0)
#endif
__device__ __forceinline__ void b2q_compress(const b2q_addr& t, const uint32_t* slots, uint32_t q, uint32_t tcount,
                                             uint32_t& h_lo, uint32_t& h_hi) {
  const uint32_t iv_lo = q == 0 ? 0x6A09E667u : q == 1 ? 0xBB67AE85u : q == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  const uint32_t iv_hi = q == 0 ? 0x510E527Fu : q == 1 ? 0x9B05688Cu : q == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
  const uint32_t h0_lo = iv_lo ^ (q == 0 ? 0x01010020u : 0u);
  const uint32_t h0_hi = iv_hi;
  uint32_t a = h0_lo, b = h0_hi, c = iv_lo;
  uint32_t d = iv_hi ^ (q == 0 ? tcount : q == 2 ? 0xffffffffu : 0u);
#if !defined(B2Q_NO_ASM)
  {
    const uint32_t mx = b2q_word(slots, t.a[0]), my = b2q_word(slots, t.a[1]);
    B2Q_HALF_PLAIN(mx, my);
  }
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r) {  // column q: a is in lane q + 1, c in lane q - 1, d in lane q - 2 (where the diagonal step left them)
      const uint32_t mx = b2q_word(slots, t.a[4 * r + 0]), my = b2q_word(slots, t.a[4 * r + 1]);
      B2Q_HALF("[1,2,3,0]", "[3,0,1,2]", "[2,3,0,1]", mx, my);
    }
    // the diagonal through b_q: a from lane q - 1, c from lane q + 1, d from lane q + 2
    const uint32_t mx = b2q_word(slots, t.a[4 * r + 2]), my = b2q_word(slots, t.a[4 * r + 3]);
    B2Q_HALF("[3,0,1,2]", "[1,2,3,0]", "[2,3,0,1]", mx, my);
  }
  a = B2Q_DPP(a, 0x39);  // back to columns for the feed-forward: a_q from lane q + 1, c_q from lane q - 1, d_q from lane q - 2
  c = B2Q_DPP(c, 0x93);
  d = B2Q_DPP(d, 0x4e);
#else
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t mx = b2q_word(slots, t.a[4 * r + 0]), my = b2q_word(slots, t.a[4 * r + 1]);
    B2_G(a, b, c, d, mx, my);
    b = B2Q_DPP(b, 0x39);  // lane q <- lane (q+1)&3
    c = B2Q_DPP(c, 0x4e);  // lane q <- lane (q+2)&3
    d = B2Q_DPP(d, 0x93);  // lane q <- lane (q+3)&3
    mx = b2q_word(slots, t.a[4 * r + 2]);
    my = b2q_word(slots, t.a[4 * r + 3]);
    B2_G(a, b, c, d, mx, my);
    b = B2Q_DPP(b, 0x93);
    c = B2Q_DPP(c, 0x4e);
    d = B2Q_DPP(d, 0x39);
  }
#endif
  h_lo = h0_lo ^ a ^ c;
  h_hi = h0_hi ^ b ^ d;
}
#endif  // __HIPCC__
