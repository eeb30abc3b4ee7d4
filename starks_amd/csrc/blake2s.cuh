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

// One compression of a 64-byte block `m` (16 LE words) into chaining value h; t = byte counter,
// last = final-block flag.
B2_HD void b2_compress(uint32_t h[8], const uint32_t m[16], uint32_t t, bool last) {
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
B2_HD b2digest b2_hash_pair(const uint32_t left[8], const uint32_t right[8]) {
  uint32_t m[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[i] = left[i];
    m[8 + i] = right[i];
  }
  b2digest d;
  b2_init(d.h);
  b2_compress(d.h, m, 64, true);
  return d;
}

// digest of a message of len <= 64 bytes given as zero-padded words (utils.py:75 hashes 32 bytes;
// the synthetic-input generator hashes 16 bytes)
B2_HD b2digest b2_hash_short(const uint32_t m[16], uint32_t len) {
  b2digest d;
  b2_init(d.h);
  b2_compress(d.h, m, len, true);
  return d;
}
