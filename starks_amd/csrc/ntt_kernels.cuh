// ntt_kernels.cuh -- radix-2^r tile passes of the large NTT over Z/p (replaces the recursive
// _fft/_simple_ft of starks/fft.py:287-314; same outputs, natural order in and out).
//
// Decomposition.  n = R_1 * R_2 * ... * R_m, one kernel launch ("pass") per factor:
//   pass d < m ("column pass", LAST=false): the array is viewed as [P_d][R_d][S_d] (S_d = n / (P_d R_d));
//       every column (p, j2) gets an R_d-point NTT along the middle axis, in place, followed by the
//       inter-pass twiddle  g_d^(j2*k1),  g_d = w^(P_d)  (order R_d*S_d).
//   pass m ("row pass", LAST=true): contiguous rows of R_m points; the result is scattered to natural
//       order  out[drev(p) + P*k]  (drev = mixed-radix digit reversal of the row number).
// A workgroup owns a tile of R x T elements (2048 elements = 64 KiB of LDS -> 2 workgroups per CU):
// T adjacent columns (T*32 B contiguous per row: coalesced segments) or T rows.  Each thread keeps 4
// elements in registers and does two butterfly levels (radix-4) between LDS exchanges; the R-point
// transform is decimation-in-frequency, so the tile's own twiddles come from one table of R/2 powers of
// w^(n/R) shared by every pass that uses radix R; the bit-reversed order DIF leaves is undone by the
// store addresses.  The last two levels of each tile transform have twiddles 1 / w^(R/4) only.
#pragma once
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "internal.hpp"
#include "knobs.hpp"


// LDS image of a tile: 32-byte elements, 8 to a 256-byte bank row; the slot inside the row is XOR-ed with
// the higher index bits so that any power-of-two stride between lanes spreads over all 8 slots.
__host__ __device__ constexpr uint32_t lds_slot(uint32_t e) {
  uint32_t s = (e ^ (e >> 3) ^ (e >> 6) ^ (e >> 9)) & 7u;
  return (((e >> 3) << 3) | s) * 2u;  // in uint4 units
}
// lds_slot is linear over GF(2) (shifts, XOR, disjoint masks), so lds_slot(e ^ d) == lds_slot(e) ^ lds_slot(d): a
// thread computes the slot of its first element once and reaches the other three with one XOR by a constant.
__device__ __forceinline__ void lds_put_at(uint4* lds, uint32_t o, const fp& r) {
  lds[o] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  lds[o + 1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ fp lds_get_at(const uint4* lds, uint32_t o) {
  uint4 a = lds[o], b = lds[o + 1];
  fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void lds_put(uint4* lds, uint32_t e, const fp& r) {
  uint32_t o = lds_slot(e);
  lds[o] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  lds[o + 1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ fp lds_get(const uint4* lds, uint32_t e) {
  uint32_t o = lds_slot(e);
  uint4 a = lds[o], b = lds[o + 1];
  fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}

__device__ __forceinline__ fp tw_lookup(const NttPassArgs& a, uint64_t e) {
  if (a.tw_direct) return fp_load(a.tw_lo + e);
  fp lo = fp_load(a.tw_lo + (e & ((1ull << a.tw_lb) - 1)));
  fp hi = fp_load(a.tw_hi + (e >> a.tw_lb));
  return fp_mul(lo, hi);
}

template <class F>
__device__ __forceinline__ void static_for4(F&& f) {
  f(std::integral_constant<int, 0>{});
  f(std::integral_constant<int, 1>{});
  f(std::integral_constant<int, 2>{});
  f(std::integral_constant<int, 3>{});
}

// Per-thread state carried across the register groups of one tile pass.
struct TileThread {
  fp x[4];
  uint32_t t, ibase;
  bool active;
  uint64_t gbase;  // global element offset of (this thread's column/row, i = 0)
  uint64_t j2;     // column pass: column index inside the prefix block
  uint64_t obase;  // row pass: output offset of k = 0
  uint64_t sbase;  // zero-padded source (a.src_n != 0): element offset of (this column/row, i = 0) in the short source
};

// ---- which thread holds which elements ---------------------------------------------------------------------------------
// In group g a thread holds the four elements whose row index i differs in the bits (beta, beta + 1) of the group's two
// levels; the other LOG_R - 2 bits of i (and the column t) name the thread.  WAVE-LOCAL EXCHANGES: the groups fall into
// two phases.  Phase 0 (the first GA groups, whose level bits are the top 2 GA bits of i) takes the wave number from i
// bits BELOW its level bits, phase 1 (the rest) from the TOP bits of i, which its levels never touch.  Inside a phase an
// element therefore stays in the same wave from group to group: the hand-over through LDS needs no workgroup barrier (the
// LDS unit serves a wave's requests in order), only the phase change does -- one __syncthreads() per tile instead of one
// per group, and waves that drift apart instead of marching in step.  (SHK_WAVE_LOCAL=0: the plain mapping, a barrier per
// exchange.)
#ifndef SHK_WAVE_LOCAL
#define SHK_WAVE_LOCAL 1
#endif
template <int LOG_R, int LOG_T>
__host__ __device__ constexpr bool tile_wave_local() {
  constexpr int LOG_W = LOG_R + LOG_T - 8;  // log2 of the waves per workgroup
  if (!SHK_WAVE_LOCAL || LOG_W < 0 || LOG_T > 5 || LOG_R < 4) return false;
  constexpr int GA = (LOG_W + 1) / 2;
  return LOG_W <= LOG_R - 2 * GA;
}
// phase of group g: -1 = the row pass's first group (its own lane mapping), 0 / 1 as above
template <int LOG_R, int LOG_T, bool LAST>
__host__ __device__ constexpr int tile_phase(int g) {
  if (LAST && g == 0) return -1;
  if (!tile_wave_local<LOG_R, LOG_T>()) return 2 + g;  // every exchange crosses waves
  constexpr int LOG_W = LOG_R + LOG_T - 8;
  return g < (LOG_W + 1) / 2 ? 0 : 1;
}
// row index (with the group's two level bits clear) of the elements of thread tid in group g (not the row pass's group 0)
template <int LOG_R, int LOG_T, int g>
__host__ __device__ __forceinline__ uint32_t tile_ibase(uint32_t tid) {
  constexpr int beta = (LOG_R - 2 * (g + 1)) > 0 ? (LOG_R - 2 * (g + 1)) : 0;
  if constexpr (!tile_wave_local<LOG_R, LOG_T>()) {
    const uint32_t rest = tid >> LOG_T;
    return ((rest >> beta) << (beta + 2)) | (rest & ((1u << beta) - 1u));
  } else {
    constexpr int LOG_W = LOG_R + LOG_T - 8;
    constexpr int GA = (LOG_W + 1) / 2;
    constexpr int wlo = g < GA ? (LOG_R - 2 * GA) - LOG_W : LOG_R - LOG_W;  // the wave number sits at i bits [wlo, wlo + LOG_W)
    const uint32_t lane_i = (tid & 63u) >> LOG_T, wave = tid >> 6;
    uint32_t i = 0;
    int nl = 0, nw = 0;
#pragma unroll
    for (int p = 0; p < LOG_R; ++p) {
      if (p == beta || p == beta + 1) continue;
      if (p >= wlo && p < wlo + LOG_W) {
        i |= ((wave >> nw) & 1u) << p;
        ++nw;
      } else {
        i |= ((lane_i >> nl) & 1u) << p;
        ++nl;
      }
    }
    return i;
  }
}

// Column t and row base (the group's two level bits clear) of thread tid in group g of a tile pass: its four elements are the rows
// ibase | (h << beta), h = 0..3, beta = max(LOG_R - 2 (g + 1), 0).  The row pass's first group has its own mapping (lanes run along
// the contiguous row).  (Also what tests/native/tile_map_host.cpp enumerates on the host to check the wave-local exchanges.)
template <int LOG_R, int LOG_T, bool LAST, int g>
__host__ __device__ __forceinline__ void tile_thread_coords(uint32_t tid, uint32_t* t, uint32_t* ibase) {
  constexpr int beta = (LOG_R - 2 * (g + 1)) > 0 ? (LOG_R - 2 * (g + 1)) : 0;
  if constexpr (LAST && g == 0) {
    const uint32_t rest = tid & ((1u << LOG_R) / 4 - 1);
    *t = tid >> (LOG_R - 2);
    *ibase = ((rest >> beta) << (beta + 2)) | (rest & ((1u << beta) - 1u));
  } else {
    *t = tid & ((1u << LOG_T) - 1);
    *ibase = tile_ibase<LOG_R, LOG_T, g>(tid);
  }
}

// Register group g: fetch 4 elements (global memory for g == 0, LDS otherwise), do its butterfly levels,
// and hand the elements to the next group through LDS.
template <int LOG_R, int LOG_T, bool LAST, int g>
__device__ __forceinline__ void ntt_group(const NttPassArgs& a, uint4* lds, TileThread& th, uint32_t tid, uint64_t tile0) {
  constexpr int R = 1 << LOG_R;
  constexpr int G = (LOG_R + 1) / 2;
  constexpr int beta = (LOG_R - 2 * (g + 1)) > 0 ? (LOG_R - 2 * (g + 1)) : 0;  // position of local bit 0
  tile_thread_coords<LOG_R, LOG_T, LAST, g>(tid, &th.t, &th.ibase);

  if (g == 0 || (LAST && g == 1)) {
    // (re)derive the global coordinates of this thread's column/row: t changes between the
    // rfast mapping of the row pass's first group and the t-fast mapping of the later ones.
    // Threads past the end (partial last tile) work on a copy of the last valid column / row and store nothing: every
    // load is unconditional and in bounds, so the four loads of a thread are requested back to back.
    const uint64_t col_raw = tile0 + th.t;
    th.active = col_raw < a.total;
    const uint64_t col = th.active ? col_raw : a.total - 1;
    if (LAST) {
      const uint64_t b = col >> a.log_P;
      const uint32_t pp = (uint32_t)(col & ((1ull << a.log_P) - 1));
      // tiles enumerate rows with the FIRST pass's digit fastest, so that a tile's T rows land on
      // T adjacent output addresses
      const uint32_t lr1 = a.ndig ? a.dig_log[0] : 0;
      const uint32_t k1 = pp & ((1u << lr1) - 1u);
      const uint32_t rst = pp >> lr1;
      const uint32_t p = (k1 << (a.log_P - lr1)) | rst;
      uint32_t sh = a.log_P, wl = 0, acc = 0;
#pragma unroll
      for (uint32_t d = 0; d < 3; ++d) {
        if (d < a.ndig) {
          sh -= a.dig_log[d];
          acc |= ((p >> sh) & ((1u << a.dig_log[d]) - 1u)) << wl;
          wl += a.dig_log[d];
        }
      }
      th.gbase = (b << a.log_n) + ((uint64_t)p << LOG_R);
      th.obase = (b << a.log_n) + acc;
      th.sbase = b * a.src_n;  // used by one-pass transforms only (p = 0)
    } else {
      th.j2 = col & ((1ull << a.log_S) - 1);
      th.gbase = ((col >> a.log_S) << (LOG_R + a.log_S)) + th.j2;
      th.sbase = (col >> a.log_S) * a.src_n;  // first pass: P = 1, col >> log_S is the vector index
    }
  }

  // ---- the twiddles of this group's (up to) four butterflies b = 2 lv + pr: which exist is known at compile time ---------
  constexpr int qhi = LOG_R - 1 - 2 * g;
  auto tw_needed = [](int b) constexpr {
    const int q = qhi - (b >> 1), pr = b & 1;
    if (q <= 0) return false;                        // level absent, or twiddle 1
    const int lb = q - beta;
    const int hl = lb == 1 ? pr : 2 * pr;
    if (q == 1 && beta == 0) return (hl & 1) != 0;   // 1 or w^(R/4)
    return true;
  };
  auto tw_load = [&](int b) {
    const int q = qhi - (b >> 1), pr = b & 1;
    const int lb = q - beta;
    const int hl = lb == 1 ? pr : 2 * pr;
    if (q == 1 && beta == 0) return fp2_load(a.wR + (R / 4));  // w^(R/4), the 4th root of unity
    const uint32_t il = th.ibase | ((uint32_t)hl << beta);
    const uint32_t ex = (il & ((1u << q) - 1u)) << (LOG_R - 1 - q);
    return fp2_load(a.wR + ex);
  };
  // one twiddle is always in flight: the first is requested before the elements are fetched, the next before the current
  // product (the product's inline asm keeps the compiler from moving loads across it, so source order is issue order)
  constexpr int first_tw = tw_needed(0) ? 0 : tw_needed(1) ? 1 : tw_needed(2) ? 2 : tw_needed(3) ? 3 : 4;
  fp2 tw_cur, tw_nxt;
  if constexpr (first_tw < 4) tw_cur = tw_load(first_tw);

  // ---- zero-padded source whose non-zero part ends inside the first quarter of the rows (the 8x low-degree extension:
  // rows i >= R / 8 of every column are zero): this thread's elements 1..3 are zero, so the group's two levels are
  //   x2 = x0 w_a,  x1 = x0 w_b,  x3 = x2 w_c   -- three products, no addition, nothing at all where x0 is zero too
  bool sparse_done = false;
  if constexpr (g == 0 && !LAST && LOG_R >= 4) {
    if (a.src_n && a.src_n <= ((uint64_t)(R / 4) << a.log_S)) {
      sparse_done = true;
      const uint64_t off = ((uint64_t)th.ibase << a.log_S) + th.j2;  // element h = 0
      const bool in = off < a.src_n;
      th.x[0] = th.x[1] = th.x[2] = th.x[3] = fp_zero();
      if (FP_ANY(in)) {  // wave-uniform
        fp x0 = fp_load(a.src + th.sbase + (in ? off : 0));
        const fp2 tw_b = tw_load(2), tw_c = tw_load(3);  // tw_cur holds butterfly 0's
#pragma unroll
        for (int w = 0; w < 8; ++w) x0.v[w] = in ? x0.v[w] : 0u;
        th.x[0] = x0;
        th.x[2] = fp_mul2(x0, tw_cur);
        th.x[1] = fp_mul2(x0, tw_b);
        th.x[3] = fp_mul2(th.x[2], tw_c);
      }
    }
  }

  // ---- fetch this group's four elements ---------------------------------------------------------
  if (sparse_done) {
    // nothing left to do before the exchange
  } else if (g == 0) {
    if (a.src_n) {  // zero-padded source (first pass only): points at or beyond src_n are zero and are not read
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const uint32_t i = th.ibase | ((uint32_t)h << beta);
        const uint64_t off = LAST ? (uint64_t)i : (((uint64_t)i << a.log_S) + th.j2);
        const bool in = off < a.src_n;
        fp v = fp_load(a.src + th.sbase + (in ? off : 0));
#pragma unroll
        for (int w = 0; w < 8; ++w) v.v[w] = in ? v.v[w] : 0u;
        th.x[h] = v;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const uint32_t i = th.ibase | ((uint32_t)h << beta);
        th.x[h] = LAST ? fp_load(a.src + th.gbase + i) : fp_load(a.src + th.gbase + ((uint64_t)i << a.log_S));
      }
    }
  } else {
#pragma unroll
    for (int h = 0; h < 4; ++h)
      th.x[h] = lds_get_at(lds, lds_slot((th.ibase << LOG_T) | th.t) ^ lds_slot((uint32_t)h << (beta + LOG_T)));
  }

  // ---- butterfly levels (DIF: a' = a + b, b' = (a - b) * w^((i mod half) * 2^s)) ---------------------
  if (!sparse_done) static_for4([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    constexpr int q = qhi - (b >> 1), pr = b & 1;
    if constexpr (q >= 0) {
      constexpr int lb = q - beta;  // local bit
      constexpr int hl = lb == 1 ? pr : 2 * pr;  // lower element of the pair
      constexpr int hh = hl | (1 << lb);
      constexpr int nxt = (b < 1 && tw_needed(1)) ? 1 : (b < 2 && tw_needed(2)) ? 2 : (b < 3 && tw_needed(3)) ? 3 : 4;
      if constexpr (tw_needed(b) && nxt < 4) tw_nxt = tw_load(nxt);
      const fp s = fp_add(th.x[hl], th.x[hh]);
      fp d = fp_sub(th.x[hl], th.x[hh]);
      if constexpr (tw_needed(b)) {
        d = fp_mul2(d, tw_cur);
        if constexpr (nxt < 4) tw_cur = tw_nxt;
      }
      th.x[hl] = s;
      th.x[hh] = d;
    }
  });

  // ---- hand the elements to the next group through LDS ---------------------------------------------
  if (g < G - 1) {
    // in place: for g > 0 these are exactly the slots this thread read at the top of the group (element (i, t) always lives
    // at the same slot), and no other thread touches them before the barrier below -- one barrier per exchange
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      lds_put_at(lds, lds_slot((th.ibase << LOG_T) | th.t) ^ lds_slot((uint32_t)h << (beta + LOG_T)), th.x[h]);
    }
    if constexpr (tile_phase<LOG_R, LOG_T, LAST>(g) == tile_phase<LOG_R, LOG_T, LAST>(g + 1)) {
      // the next group's elements were written by lanes of this wave: order the wave's own LDS traffic, nothing more
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
  }
}

#ifndef SHK_NTT_MIN_WAVES
#define SHK_NTT_MIN_WAVES 4
#endif
template <int LOG_R, int LOG_T, bool LAST>
__device__ __forceinline__ void ntt_pass_body(const NttPassArgs& a) {
  static_assert(LOG_R >= 2 && LOG_R <= 11 && LOG_T >= 0 && LOG_R + LOG_T >= 8 && LOG_R + LOG_T <= 12, "unsupported tile");
  constexpr int G = (LOG_R + 1) / 2;  // register groups (two levels each, the last may have one)
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const uint32_t tid = threadIdx.x;
  uint64_t tile = blockIdx.x;
  if (a.xcd_per) {
    tile = (uint64_t)(blockIdx.x & 7u) * a.xcd_per + (blockIdx.x >> 3);
    if ((tile << LOG_T) >= a.total) return;  // grid padding (whole workgroup, before any barrier)
  }
  const uint64_t tile0 = tile << LOG_T;
  TileThread th;
  th.t = 0; th.ibase = 0; th.active = false; th.gbase = 0; th.j2 = 0; th.obase = 0; th.sbase = 0;
  // Column passes: the inter-pass twiddles g^(j2 k) of this thread's outputs are requested ahead of their use -- before the
  // LAST group's butterflies (whose own twiddles are 1 and w^(R/4) only) -- so that their latency (HBM, or L2 when another
  // vector has just read the same rows) is hidden behind arithmetic instead of being paid four times in the store loop.
  // One is kept in flight (the first before the last group, the others one product ahead in the store loop): all four at
  // once, or two, cost a wave per SIMD (106 / 102 VGPRs) and measured slower on single vectors and on 2^24.
  fp itw[4];
  auto itw_load = [&](int h) {
    const uint32_t i = tile_ibase<LOG_R, LOG_T, G - 1>(tid) | (uint32_t)h;  // the last group's element index (beta = 0)
    const uint32_t k = __brev(i) >> (32 - LOG_R);
    return fp_load(a.tw2 + ((uint64_t)k << a.log_S) + th.j2);
  };
  auto itw_request = [&]() {
    if (!LAST && a.tw2) itw[0] = itw_load(0);
  };
  if constexpr (G == 1) {
    ntt_group<LOG_R, LOG_T, LAST, 0>(a, lds, th, tid, tile0);
    itw_request();
  } else {
    ntt_group<LOG_R, LOG_T, LAST, 0>(a, lds, th, tid, tile0);
    if constexpr (G == 2) itw_request();
    ntt_group<LOG_R, LOG_T, LAST, 1>(a, lds, th, tid, tile0);
    if constexpr (G > 2) {
      if constexpr (G == 3) itw_request();
      ntt_group<LOG_R, LOG_T, LAST, 2>(a, lds, th, tid, tile0);
    }
    if constexpr (G > 3) {
      if constexpr (G == 4) itw_request();
      ntt_group<LOG_R, LOG_T, LAST, 3>(a, lds, th, tid, tile0);
    }
    if constexpr (G > 4) {
      if constexpr (G == 5) itw_request();
      ntt_group<LOG_R, LOG_T, LAST, 4>(a, lds, th, tid, tile0);
    }
    if constexpr (G > 5) {
      if constexpr (G == 6) itw_request();
      ntt_group<LOG_R, LOG_T, LAST, 5>(a, lds, th, tid, tile0);
    }
  }

  // ---- store: position i of the DIF output holds frequency k = bitrev(i) ----------------------------
  if (!th.active) return;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const uint32_t i = th.ibase | (uint32_t)h;  // the last group's local bits sit at position 0
    const uint32_t k = __brev(i) >> (32 - LOG_R);
    if (LAST) {
      fp v = th.x[h];
      if (a.scale) v = fp_mul(v, fp_load(a.scale));
      fp_store(a.dst + th.obase + ((uint64_t)k << a.log_P), v);
    } else {
      if (a.tw2 && h + 1 < 4) itw[h + 1] = itw_load(h + 1);
      const fp tw = a.tw2 ? itw[h] : tw_lookup(a, th.j2 * k);
      fp v = fp_mul(th.x[h], tw);
      fp_store(a.dst + th.gbase + ((uint64_t)k << a.log_S), v);
    }
  }
}

template <int LOG_R, int LOG_T, bool LAST>
__global__ void __launch_bounds__(1 << (LOG_R + LOG_T - 2), SHK_NTT_MIN_WAVES) ntt_pass_kernel(NttPassArgs a) {
  ntt_pass_body<LOG_R, LOG_T, LAST>(a);
}

// ---- narrow launches -----------------------------------------------------------------------------------------------------
// A pass over fewer tiles than the chip has CUs is a latency chain, not a throughput problem: every workgroup is alone on its CU, a
// lone wave issues an instruction every ~2 ns whatever it does, and ntt_pass_body gives a thread 2 butterflies per level (4
// elements, 14-20 butterflies and 4 inter-pass products per pass = 11-20 us for a radix 2^7..2^9 pass).  Here a thread owns ONE
// butterfly per level (twice the threads per tile, 512 for 1024 elements): the pair (i, i + half) of level q = log2(half) is read from
// the tile's LDS image, combined, and written back in place; one workgroup barrier per level; the first level reads global memory,
// the last one writes it (same addresses, twiddles and results as ntt_pass_body: DIF, position i holds frequency bitrev(i)).
// Used below STARKHIP_NTT_NARROW_TILES tiles per launch (knobs.hpp; small transforms: the commits and proofs of 2^14..2^16-step traces).
template <int LOG_R, int LOG_T, bool LAST>
__global__ void __launch_bounds__(1 << (LOG_R + LOG_T - 1)) ntt_narrow_pass_kernel(NttPassArgs a) {
  static_assert(LOG_R >= 2 && LOG_R <= 10 && LOG_T >= 0 && (LOG_R + LOG_T == 10 || LOG_R + LOG_T == 9), "1024- or 512-element tiles");
  __shared__ __attribute__((aligned(16))) uint4 lds[2 << (LOG_R + LOG_T)];
  constexpr int R = 1 << LOG_R;
  const uint32_t tid = threadIdx.x;
  // Which butterfly (column / row t, pair index p) a thread takes may change from level to level: everything goes through the LDS
  // image behind a workgroup barrier.  Column pass: adjacent lanes take adjacent columns throughout (32 contiguous bytes each, in
  // the loads and in the stores).  Row pass: adjacent lanes run along the contiguous row for the loads of the first level and take
  // adjacent rows afterwards, so that the T outputs of one frequency land on T adjacent addresses (as in ntt_pass_body).
  const uint32_t t_in = LAST ? tid >> (LOG_R - 1) : tid & ((1u << LOG_T) - 1u);
  const uint32_t p_in = LAST ? tid & (R / 2 - 1) : tid >> LOG_T;
  const uint32_t t = tid & ((1u << LOG_T) - 1u), p = tid >> LOG_T;
  const uint64_t tile0 = (uint64_t)blockIdx.x << LOG_T;
  // threads past the end work on a copy of the last column / row and store nothing
  const uint64_t col_in = tile0 + t_in < a.total ? tile0 + t_in : a.total - 1;
  const bool active = tile0 + t < a.total;
  const uint64_t col = active ? tile0 + t : a.total - 1;
  uint64_t gbase_in, gbase = 0, obase = 0, sbase_in, j2 = 0;
  if (LAST) {
    auto row_of = [&](uint64_t c, uint64_t* bb) {  // rows enumerated with the first pass's digit fastest, as in ntt_pass_body
      *bb = c >> a.log_P;
      const uint32_t pp = (uint32_t)(c & ((1ull << a.log_P) - 1));
      const uint32_t lr1 = a.ndig ? a.dig_log[0] : 0;
      const uint32_t k1 = pp & ((1u << lr1) - 1u), rst = pp >> lr1;
      return (k1 << (a.log_P - lr1)) | rst;
    };
    uint64_t b_in, b_out;
    const uint32_t row_in = row_of(col_in, &b_in), row = row_of(col, &b_out);
    uint32_t sh = a.log_P, wl = 0, acc = 0;
#pragma unroll
    for (uint32_t d = 0; d < 3; ++d) {
      if (d < a.ndig) {
        sh -= a.dig_log[d];
        acc |= ((row >> sh) & ((1u << a.dig_log[d]) - 1u)) << wl;
        wl += a.dig_log[d];
      }
    }
    gbase_in = (b_in << a.log_n) + ((uint64_t)row_in << LOG_R);
    obase = (b_out << a.log_n) + acc;
    sbase_in = b_in * a.src_n;
  } else {
    j2 = col & ((1ull << a.log_S) - 1);
    gbase = ((col >> a.log_S) << (LOG_R + a.log_S)) + j2;
    gbase_in = gbase;
    sbase_in = (col >> a.log_S) * a.src_n;
  }
  auto fetch = [&](uint32_t i) {
    if (a.src_n) {  // zero-padded source (first pass): points at or beyond src_n are zero and are not read
      const uint64_t off = LAST ? (uint64_t)i : (((uint64_t)i << a.log_S) + j2);
      const bool in = off < a.src_n;
      fp v = fp_load(a.src + sbase_in + (in ? off : 0));
#pragma unroll
      for (int w = 0; w < 8; ++w) v.v[w] = in ? v.v[w] : 0u;
      return v;
    }
    return LAST ? fp_load(a.src + gbase_in + i) : fp_load(a.src + gbase_in + ((uint64_t)i << a.log_S));
  };
  fp x0, x1;
  uint32_t i_lo = 0;
  // the twiddle pair of a level, w_R^((i mod half) * R / (2 half)), depends on the thread alone: it is requested one level ahead (the
  // product's inline asm pins a load where the source has it, and a load right in front of its product is ~0.5 us of every level)
  auto tw_of = [&](int q) {
    const uint32_t half = 1u << q, pq = q == LOG_R - 1 ? p_in : p;
    const uint32_t il = ((pq >> q) << (q + 1)) | (pq & (half - 1u));
    return fp2_load(a.wR + ((il & (half - 1u)) << (LOG_R - 1 - q)));
  };
  fp2 tw_cur, tw_nxt;
  if constexpr (LOG_R >= 2) tw_cur = tw_of(LOG_R - 1);
#pragma unroll
  for (int q = LOG_R - 1; q >= 0; --q) {
    const uint32_t half = 1u << q;
    const uint32_t tq = q == LOG_R - 1 ? t_in : t, pq = q == LOG_R - 1 ? p_in : p;
    i_lo = ((pq >> q) << (q + 1)) | (pq & (half - 1u));
    const uint32_t i_hi = i_lo | half;
    const uint32_t s_lo = lds_slot((i_lo << LOG_T) | tq), s_hi = lds_slot((i_hi << LOG_T) | tq);
    fp u, v;
    if (q == LOG_R - 1) {
      u = fetch(i_lo);
      v = fetch(i_hi);
    } else {
      u = lds_get_at(lds, s_lo);
      v = lds_get_at(lds, s_hi);
    }
    if (q > 1) tw_nxt = tw_of(q - 1);
    x0 = fp_add(u, v);
    x1 = fp_sub(u, v);
    if (q > 0) {
      x1 = fp_mul2(x1, tw_cur);
      if (q > 1) tw_cur = tw_nxt;
      lds_put_at(lds, s_lo, x0);
      lds_put_at(lds, s_hi, x1);
      __syncthreads();
    }
  }
  if (!active) return;
  // position i of the DIF output holds frequency k = bitrev(i); this thread ends with positions i_lo (even) and i_lo + 1
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t k = __brev(i_lo | (uint32_t)h) >> (32 - LOG_R);
    fp v = h ? x1 : x0;
    if (LAST) {
      if (a.scale) v = fp_mul(v, fp_load(a.scale));
      fp_store(a.dst + obase + ((uint64_t)k << a.log_P), v);
    } else {
      const fp tw = a.tw2 ? fp_load(a.tw2 + ((uint64_t)k << a.log_S) + j2) : tw_lookup(a, j2 * k);
      fp_store(a.dst + gbase + ((uint64_t)k << a.log_S), fp_mul(v, tw));
    }
  }
}

// ---- launching a tile pass (any kernel built on ntt_pass_body) --------------------------------------------------------------
// Tiles narrower than 128 bytes (T < 4: the big-radix passes) share their cache lines with the neighbouring tile; workgroups
// are dealt to the 8 XCDs round-robin, so neighbours would sit behind different L2s and every line would be fetched (or
// written back partially) twice.  Default (1): such launches map adjacent tiles to the same XCD (measured on the T = 1
// variant: 8.6 -> 10.2 G elements/s).  STARKHIP_XCD_SWZ (knobs.hpp): 0 = never, 2 = every tile pass (measured level for
// T >= 4).

// attr_done: one bit per device ordinal, per kernel instantiation (contexts on several devices, and on several host threads,
// share the launcher)
inline hipError_t shk_launch_tile_kernel(void (*k)(NttPassArgs), std::atomic<uint64_t>& attr_done, int log_t, unsigned threads,
                                         size_t lds_bytes, const NttPassArgs& a, hipStream_t st) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const uint64_t tiles = (a.total + ((1ull << log_t) - 1)) >> log_t;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7ffffff0ull) return hipErrorInvalidValue;
  NttPassArgs b = a;
  uint64_t grid = tiles;
  const int swz = shk_knobs().xcd_swz;
  if (swz && (swz == 2 || log_t < 2) && tiles >= 64) {
    b.xcd_per = (uint32_t)((tiles + 7) / 8);
    grid = 8ull * b.xcd_per;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), lds_bytes, st, b);
  return hipGetLastError();
}

// n <= 2: the reference's naive base case (_simple_ft, fft.py:287-300) is the whole transform.
static __global__ void ntt_tiny_kernel(const fp* src, fp* dst, uint32_t n, uint32_t batch, const fp* scale) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  if (n == 1) {
    fp_store(dst + b, fp_load(src + b));
    return;
  }
  fp x0 = fp_load(src + 2ull * b), x1 = fp_load(src + 2ull * b + 1);
  fp s = fp_add(x0, x1), d = fp_sub(x0, x1);
  if (scale) {
    fp sc = fp_load(scale);
    s = fp_mul(s, sc);
    d = fp_mul(d, sc);
  }
  fp_store(dst + 2ull * b, s);
  fp_store(dst + 2ull * b + 1, d);
}
