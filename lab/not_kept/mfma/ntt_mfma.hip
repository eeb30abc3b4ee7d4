// ntt_mfma.hip -- radix-2^r passes of the large NTT with the butterflies' twiddle products on the matrix cores.
// Same decomposition, same tables of values and same results as ntt_kernels.cuh (the reference's _fft/_simple_ft,
// starks/fft.py:287-314, natural order in and out); what changes is where a pass keeps its data and how it multiplies.
//
// A workgroup owns a tile of R rows x 32 COLUMNS and keeps it in registers: 2R threads, 16 elements each.  Lane l of a
// wave is column l & 31; the row index never reaches the low five lane bits.  Every butterfly twiddle of a tile
// transform depends on the row only, so the 32 lanes of a half-wave always share it and the product (a - b) * w runs as
// one byte-matrix x byte-vector MFMA for all of them (mfma_tw.cuh).  Only the inter-pass twiddle g^(j2 * k), which
// differs per column, is a general 256-bit modmul on the VALUs (fp_mul).
//   stage 1: rows i = m * (R/16) + rho, m = 0..15 in registers (rho = the thread's row group): the four top DIF levels
//            without any exchange; the two half-waves of a wave are different row groups (two twiddle matrices per MFMA
//            pair);
//   exchange through LDS (64 KiB at a time: a column half / quarter per round, so that every thread writes and reads all
//            of its 16 elements in the same round and needs no second register set);
//   stage 2: rows i = 16 * mu + m', m' = 0..15 in registers: the remaining r - 4 levels, whose twiddles are the same for
//            every thread (one matrix, compile-time index; twiddle 1 is a plain subtraction);
//   DIF leaves position i holding frequency bitrev(i); the store addresses undo that.
// Column pass (LAST = false): the 32 columns are adjacent (1 KiB contiguous per row), loads and stores are direct.
// Row pass (LAST = true): a "column" is one contiguous row of R points; rows are staged through LDS (lanes along the row
// for the global load), and the result is scattered to natural order with 32 adjacent outputs per frequency.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>
#include <utility>

#include "internal.hpp"
#include "mfma_tw.cuh"
#include "ntt_kernels.cuh"
#include "ntt_tile_common.cuh"
#include "mfma_bfly.inc"  // GENERATED (gen_bflyasm.py): the two butterfly stages as scheduled asm blocks
#include "mfma_group.inc"  // GENERATED (gen_bflyasm.py): the register groups of the LDS-resident 32-column tile

#ifdef SHK_STAMPS
// diagnostic build only (make STAMPS=1 -> libstarkhip_stamps.so): s_memtime at the phase boundaries of wave 0 of the
// first 1024 workgroups of the most recent tile pass; never compiled into the product library
__device__ unsigned long long g_stamps[8][1024];
#define STAMP(k)                                                                                     \
  do {                                                                                               \
    if (a.debug && threadIdx.x == 0 && blockIdx.x < 1024) g_stamps[k][blockIdx.x] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int sh_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : -3;
}
#else
#define STAMP(k)
#endif

namespace {

// the generated stage blocks by radix
template <int LOG_R>
__device__ __forceinline__ void stage1_asm(shk_x8 (&x)[16], const shk_v16i& offs, uint32_t lane16, uint32_t mlo, uint32_t mhi, uint32_t rho) {
  if constexpr (LOG_R == 5) shk_stage1_asm_5(x, offs, lane16, mlo, mhi, rho);
  if constexpr (LOG_R == 6) shk_stage1_asm_6(x, offs, lane16, mlo, mhi, rho);
  if constexpr (LOG_R == 7) shk_stage1_asm_7(x, offs, lane16, mlo, mhi, rho);
  if constexpr (LOG_R == 8) shk_stage1_asm_8(x, offs, lane16, mlo, mhi, rho);
}
template <int LOG_R>
__device__ __forceinline__ void stage2_asm(shk_x8 (&x)[16], const shk_v16i& offs, uint32_t lane16, uint32_t mlo, uint32_t mhi) {
  if constexpr (LOG_R == 5) shk_stage2_asm_5(x, offs, lane16, mlo, mhi);
  if constexpr (LOG_R == 6) shk_stage2_asm_6(x, offs, lane16, mlo, mhi);
  if constexpr (LOG_R == 7) shk_stage2_asm_7(x, offs, lane16, mlo, mhi);
  if constexpr (LOG_R == 8) shk_stage2_asm_8(x, offs, lane16, mlo, mhi);
}
__device__ __forceinline__ void to_x8(const fp (&x)[16], shk_x8 (&v)[16]) {
#pragma unroll
  for (int m = 0; m < 16; ++m)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[m][i] = x[m].v[i];
}
__device__ __forceinline__ void from_x8(const shk_x8 (&v)[16], fp (&x)[16]) {
#pragma unroll
  for (int m = 0; m < 16; ++m)
#pragma unroll
    for (int i = 0; i < 8; ++i) x[m].v[i] = v[m][i];
}

// ASM: the butterfly stages are the generated asm blocks (mfma_bfly.inc); otherwise the C++ butterflies of mfma_tw.cuh
// (STARKHIP_MFMA_BFLY=cxx: kept as the readable statement of the same computation and for A/B measurements)
template <int LOG_R, bool LAST, bool ASM>
__global__ void __launch_bounds__(2 << LOG_R) __attribute__((amdgpu_waves_per_eu(2, 2))) ntt_ctile_kernel(NttPassArgs a) {
  static_assert(LOG_R >= 5 && LOG_R <= 8, "unsupported radix");
  constexpr int R = 1 << LOG_R;
  constexpr int G = R / 16;                       // row groups = 2 per wave
  constexpr int THREADS = 2 * R;
  // LDS window of an exchange round: all R rows of 16 columns (R/2 KiB), two rounds.  Sized so that two waves per SIMD
  // fit whatever the radix: 8 / 4 / 2 / 1 workgroups per CU for R = 32 / 64 / 128 / 256 use 16 / 32 / 64 / 128 KiB each.
  constexpr int NR = 2;
  constexpr int LOG_CW = 4;
  constexpr int CW = 1 << LOG_CW;
  constexpr int QMAX = LOG_R - 5 < 3 ? LOG_R - 5 : 3;   // top level of stage 2
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: twiddle-matrix addresses stay scalar
  const uint32_t c = lane & 31u, hb = lane >> 5;
  const uint32_t rho = 2u * wave + hb;  // row group of stage 1, and row block mu of stage 2
  const uint64_t col_raw = ((uint64_t)blockIdx.x << 5) + c;
  const bool active = col_raw < a.total;
  // columns past the end (partial last tile) compute on a copy of the last valid column and store nothing: every load is
  // unconditional and in bounds, no zero-fill branches
  const uint64_t col = active ? col_raw : a.total - 1;
  const shk_kinit cinit = shk_mfma_kinit(lane);
  const TwMat* mats = reinterpret_cast<const TwMat*>(a.mats);

  uint64_t gbase = 0, obase = 0, j2 = 0;
  if (LAST) {
    row_coords<LOG_R>(a, col, &gbase, &obase);
  } else {
    j2 = col & ((1ull << a.log_S) - 1);
    gbase = ((col >> a.log_S) << (LOG_R + a.log_S)) + j2;
  }

  fp x[16];
  STAMP(0);
  // ---- load ------------------------------------------------------------------------------------------------------
  if (!LAST) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const uint32_t i = (uint32_t)m * G + rho;
      x[m] = fp_load(a.src + gbase + ((uint64_t)i << a.log_S));
    }
  } else {
    // rows are contiguous: lanes run along the row for the global load (all 16 loads of a thread are requested up
    // front), then the tile is transposed through LDS, CW rows (= tile columns) per round
    constexpr int ELEMS = CW * R / THREADS;  // per thread per round (= 8)
    fp ld[NR * ELEMS];
    static_for<NR * ELEMS>([&](auto ei) {
      constexpr int e = decltype(ei)::value % ELEMS, k = decltype(ei)::value / ELEMS;
      const uint32_t flat = (uint32_t)e * THREADS + tid;  // < CW * R: (row r_l inside the round, point i)
      // a wave covers (part of) ONE row when R >= 64: its coordinates are wave-uniform (scalar arithmetic)
      const uint32_t r_l = LOG_R >= 6 ? (uint32_t)__builtin_amdgcn_readfirstlane(flat >> LOG_R) : flat >> LOG_R;
      const uint32_t i = flat & (R - 1);
      uint64_t rcol = ((uint64_t)blockIdx.x << 5) + (uint32_t)k * CW + r_l;
      if (rcol >= a.total) rcol = a.total - 1;  // clamped: loaded, never stored
      uint64_t gb, ob;
      row_coords<LOG_R>(a, rcol, &gb, &ob);
      ld[decltype(ei)::value] = fp_load(a.src + gb + i);
    });
    STAMP(6);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (k > 0) __syncthreads();
      static_for<ELEMS>([&](auto ei) {
        constexpr int e = decltype(ei)::value;
        const uint32_t flat = (uint32_t)e * THREADS + tid;
        lds_put_at(lds, win_slot<LOG_CW>(flat & (R - 1), flat >> LOG_R), k == 0 ? ld[e] : ld[ELEMS + e]);
      });
      __syncthreads();
      if ((c >> LOG_CW) == (uint32_t)k) {
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = lds_get_at(lds, win_slot<LOG_CW>((uint32_t)m * G + rho, c & (CW - 1)));
      }
    }
    __syncthreads();
  }

  // ---- stage 1: levels q = LOG_R-1 .. LOG_R-4 on the register index m (bit mu = 3 .. 0) -----------------------------
  STAMP(1);
  shk_v16i offs;
  const uint32_t mlo = (uint32_t)reinterpret_cast<uintptr_t>(a.mats), mhi = (uint32_t)(reinterpret_cast<uintptr_t>(a.mats) >> 32);
  if constexpr (ASM) {
#pragma unroll
    for (int r = 0; r < 16; ++r) offs[r] = hb ? SHK_OFFS[1][r] : SHK_OFFS[0][r];
    shk_x8 xv[16];
    to_x8(x, xv);
    stage1_asm<LOG_R>(xv, offs, lane * 16u, mlo, mhi, 2u * wave);
    from_x8(xv, x);
  } else {
    // the fragments of butterfly j + 1 are requested before butterfly j is computed (L2 latency behind ~130 VALU
    // instructions); consecutive butterflies with the same twiddle keep their fragments
    const uint32_t rho_lo = 2u * wave, rho_hi = rho_lo + 1u;
    struct Frag4 {
      shk_v4i w1, n1, w2, n2;
    };
    auto frags_of = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int mu = 3 - j / 8, b = j % 8;
      constexpr int m0 = ((b >> mu) << (mu + 1)) | (b & ((1 << mu) - 1));
      constexpr uint32_t eb = (uint32_t)(m0 & ((1 << mu) - 1)) * G;
      const TwMat* t1 = mats + ((eb + rho_lo) << (3 - mu));
      const TwMat* t2 = mats + ((eb + rho_hi) << (3 - mu));
      Frag4 f;
      f.w1 = shk_ld_frag(t1->w, lane);
      f.n1 = shk_ld_frag(t1->nw, lane);
      f.w2 = shk_ld_frag(t2->w, lane);
      f.n2 = shk_ld_frag(t2->nw, lane);
      return f;
    };
    Frag4 cur = frags_of(std::integral_constant<int, 0>{});
    static_for<32>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int mu = 3 - j / 8, b = j % 8;
      constexpr int m0 = ((b >> mu) << (mu + 1)) | (b & ((1 << mu) - 1));
      constexpr int m1 = m0 | (1 << mu);
      Frag4 nxt = cur;
      if constexpr (j + 1 < 32) {
        constexpr int mu2 = 3 - (j + 1) / 8, b2 = (j + 1) % 8;
        constexpr int m02 = ((b2 >> mu2) << (mu2 + 1)) | (b2 & ((1 << mu2) - 1));
        constexpr bool same = mu2 == mu && (m02 & ((1 << mu2) - 1)) == (m0 & ((1 << mu) - 1));
        if constexpr (!same) nxt = frags_of(std::integral_constant<int, j + 1>{});
      }
      const fp d = shk_mfma_submul2(x[m0], x[m1], cur.w1, cur.n1, cur.w2, cur.n2, cinit);
      x[m0] = fp_add(x[m0], x[m1]);
      x[m1] = d;
      cur = nxt;
    });
  }

  STAMP(2);
  // ---- exchange: (m, rho) -> (mu', m') ---------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const bool mine = (c >> LOG_CW) == (uint32_t)k;
    if (k > 0) __syncthreads();
    if (mine) {
#pragma unroll
      for (int m = 0; m < 16; ++m) lds_put_at(lds, win_slot<LOG_CW>((uint32_t)m * G + rho, c & (CW - 1)), x[m]);
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int m = 0; m < 16; ++m) x[m] = lds_get_at(lds, win_slot<LOG_CW>(16u * rho + (uint32_t)m, c & (CW - 1)));
    }
  }

  STAMP(3);
  // ---- stage 2: levels q = QMAX .. 0 on the register index m' (twiddles identical for every thread) ------------------
  if constexpr (ASM) {
    shk_x8 xv[16];
    to_x8(x, xv);
    stage2_asm<LOG_R>(xv, offs, lane * 16u, mlo, mhi);
    from_x8(xv, x);
  } else
  static_for<QMAX + 1>([&](auto lv) {
    constexpr int q = QMAX - decltype(lv)::value;
    static_for<8>([&](auto bi) {
      constexpr int b = decltype(bi)::value;
      constexpr int m0 = ((b >> q) << (q + 1)) | (b & ((1 << q) - 1));
      constexpr int m1 = m0 | (1 << q);
      constexpr int e = (m0 & ((1 << q) - 1)) << (LOG_R - 1 - q);
      if constexpr (e == 0) {
        const fp s = fp_add(x[m0], x[m1]);
        x[m1] = fp_sub(x[m0], x[m1]);
        x[m0] = s;
      } else {
        const TwMat* t = mats + e;
        const shk_v4i w = shk_ld_frag(t->w, lane), nw = shk_ld_frag(t->nw, lane);
        const fp d = shk_mfma_submul(x[m0], x[m1], w, nw, cinit);
        x[m0] = fp_add(x[m0], x[m1]);
        x[m1] = d;
      }
    });
  });

  // ---- store: position i = 16 rho + m' holds frequency k = bitrev(i) -----------------------------------------------------
  STAMP(4);
  if (!active) return;
  if (LAST) {
    static_for<16>([&](auto mi) {
      constexpr int m = decltype(mi)::value;
      const uint32_t i = 16u * rho + (uint32_t)m;
      const uint32_t k = __brev(i) >> (32 - LOG_R);
      fp v = x[m];
      if (a.scale) v = fp_mul(v, fp_load(a.scale));
      fp_store(a.dst + obase + ((uint64_t)k << a.log_P), v);
    });
  } else {
    // inter-pass twiddles g^(j2 k) from the [k][j2] table (a tile's 32 columns read 1 KiB contiguous per row); eight are
    // requested at a time so that their latency overlaps the modmuls of the previous ones
    static_for<2>([&](auto gi) {
      constexpr int g8 = decltype(gi)::value * 8;
      fp tw[8];
      static_for<8>([&](auto ti) {
        constexpr int m = g8 + decltype(ti)::value;
        const uint32_t k = __brev(16u * rho + (uint32_t)m) >> (32 - LOG_R);
        tw[m - g8] = fp_load(a.tw2 + ((uint64_t)k << a.log_S) + j2);
      });
      static_for<8>([&](auto ti) {
        constexpr int m = g8 + decltype(ti)::value;
        const uint32_t k = __brev(16u * rho + (uint32_t)m) >> (32 - LOG_R);
        fp_store(a.dst + gbase + ((uint64_t)k << a.log_S), fp_mul(x[m], tw[m - g8]));
      });
    });
  }
  STAMP(5);
}

bool use_asm_butterflies() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_MFMA_BFLY");
    v = (e && !strcmp(e, "cxx")) ? 0 : 1;
  }
  return v == 1;
}

template <int LOG_R, bool LAST, bool ASM>
hipError_t launch_ctile_impl(const NttPassArgs& a, hipStream_t st) {
  constexpr int R = 1 << LOG_R;
  constexpr size_t LDS = (size_t)R * 16 * 32;  // one exchange window: R rows x 16 columns
  auto k = ntt_ctile_kernel<LOG_R, LAST, ASM>;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const uint64_t tiles = (a.total + 31) >> 5;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(2 * R), LDS, st, a);
  return hipGetLastError();
}

template <int LOG_R, bool LAST>
hipError_t launch_ctile(const NttPassArgs& a, hipStream_t st) {
  return use_asm_butterflies() ? launch_ctile_impl<LOG_R, LAST, true>(a, st) : launch_ctile_impl<LOG_R, LAST, false>(a, st);
}

// ---- the LDS-resident tile of ntt_kernels.cuh, 32 columns wide, with its register groups' butterflies on the matrix cores ------
// Same pass, same lane mapping (four elements per thread, two levels per LDS exchange, R x 32 elements in LDS) -- but the 32
// lanes of a half-wave are the 32 columns of one row set, so a butterfly's twiddle is one matrix per half-wave and the
// group's arithmetic is one generated block (mfma_group.inc) on 128 VGPRs: four waves per SIMD, twice the register-tile
// kernel above.  The row pass's first group (a row per lane) and the sparse first group of the low-degree extension stay
// on the VALU butterflies.
struct MfmaLane {
  static constexpr bool itw_prefetch = false;  // no registers to spare across the last group
  // every group but the row pass's first, whose lanes run along a row (a twiddle per lane)
  template <int LOG_R, int LOG_T, bool LAST>
  static constexpr bool group_on_mfma(int g) { return !(LAST && g == 0); }
  template <int LOG_R, int LOG_T, bool LAST>
  static constexpr int phase(int g) { return tile_phase<LOG_R, LOG_T, LAST>(g); }
  template <int LOG_R, int LOG_T, bool LAST, int g>
  static __device__ __forceinline__ uint32_t ibase(uint32_t tid) { return tile_ibase<LOG_R, LOG_T, g>(tid); }
  shk_v16i offs;
  uint32_t lane16, mlo, mhi;
#ifdef SHK_STAMPS
  uint32_t debug;
  __device__ __forceinline__ void stamp(int k) const {
    // debug = 1 + first recorded workgroup / 1024: workgroups [1024 (debug - 1), 1024 debug) record (STARKHIP_STAMP_BASE)
    const uint32_t b = blockIdx.x - 1024u * (debug - 1u);
    if (debug && threadIdx.x == 0 && b < 1024u) g_stamps[k][b] = __builtin_amdgcn_s_memtime();
  }
#else
  __device__ __forceinline__ void stamp(int) const {}
#endif

  template <int LOG_R, int g>
  __device__ __forceinline__ void butterflies(TileThread& th) const {
    constexpr int G = (LOG_R + 1) / 2;
    constexpr int beta = (LOG_R - 2 * (g + 1)) > 0 ? (LOG_R - 2 * (g + 1)) : 0;
    shk_x8 xv[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) xv[h][i] = th.x[h].v[i];
    if constexpr (beta > 0) {
      // levels q = beta + 1 (pairs (0,2), (1,3): exponents (low | pr << beta) << (LOG_R - 2 - beta)) and q = beta (pairs
      // (0,1), (2,3): low << (LOG_R - 1 - beta) for both), low = the row bits below the group's two
      static_assert(g < G - 1, "");
      const uint32_t low = th.ibase & ((1u << beta) - 1u);
      const uint32_t e0 = (low << (LOG_R - 2 - beta)) * (uint32_t)sizeof(TwMat);
      const uint32_t e1 = ((low | (1u << beta)) << (LOG_R - 2 - beta)) * (uint32_t)sizeof(TwMat);
      const uint32_t e2 = (low << (LOG_R - 1 - beta)) * (uint32_t)sizeof(TwMat);
      const uint32_t o0l = __builtin_amdgcn_readfirstlane(e0), o0h = __builtin_amdgcn_readlane(e0, 32);
      const uint32_t o1l = __builtin_amdgcn_readfirstlane(e1), o1h = __builtin_amdgcn_readlane(e1, 32);
      const uint32_t o2l = __builtin_amdgcn_readfirstlane(e2), o2h = __builtin_amdgcn_readlane(e2, 32);
      shk_group_asm_MMMM(xv, offs, lane16, mlo, mhi, o0l, o0h, o1l, o1h, o2l, o2h, o2l, o2h);
    } else if constexpr (LOG_R % 2 == 0) {
      constexpr uint32_t o = (uint32_t)((1u << LOG_R) / 4) * (uint32_t)sizeof(TwMat);  // 1, w^(R/4); then 1, 1
      shk_group_asm_AMAA(xv, offs, lane16, mlo, mhi, o, o);
    } else {
      shk_group_asm_AA(xv, offs, lane16, mlo, mhi);
    }
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) th.x[h].v[i] = xv[h][i];
  }
};

__device__ __forceinline__ void lane_setup(MfmaLane& ln, const NttPassArgs& a) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int r = 0; r < 16; ++r) ln.offs[r] = (lane >> 5) ? SHK_OFFS[1][r] : SHK_OFFS[0][r];
  ln.lane16 = lane * 16u;
  ln.mlo = (uint32_t)reinterpret_cast<uintptr_t>(a.mats);
  ln.mhi = (uint32_t)(reinterpret_cast<uintptr_t>(a.mats) >> 32);
#ifdef SHK_STAMPS
  ln.debug = a.debug;
#endif
}

template <int LOG_R, bool LAST>
__global__ void __launch_bounds__(8 << LOG_R) __attribute__((amdgpu_waves_per_eu(4, 4))) ntt_ltile_kernel(NttPassArgs a) {
  static_assert(LOG_R >= 5 && LOG_R <= 7, "unsupported radix");
  MfmaLane ln;
  lane_setup(ln, a);
  ntt_pass_body<LOG_R, 5, LAST>(a, ln);
}

// ---- the VALU tile pass with its shared-twiddle groups on the matrix cores ---------------------------------------------------
// The tile of ntt_kernels.cuh as it is (R x T elements, any radix, the same plans and the same HBM traffic as the VALU
// passes).  The twiddle of a butterfly of level q depends on the low q bits of the row index only, so in the LATER groups of a
// tile transform -- group g, level bits (beta, beta + 1), beta = LOG_R - 2 (g + 1) -- the 32 lanes of a half-wave share
// their twiddles as soon as the low beta bits of their row indices agree: take those bits from the wave number and lane
// bit 5, and let the lanes run over the columns and the HIGH row bits.  That mapping exists when beta <= 1 + log2(waves per
// workgroup): the middle groups of the big radices (groups 2 and 3 of a 2^10-row tile of 2048 elements).  Those groups run the
// generated blocks (mfma_group.inc); the first groups (a twiddle per lane) and the last one (twiddles 1 and w^(R/4)) keep the
// VALU butterflies and their wave-local mapping; an exchange into or out of a matrix-core group crosses waves (a barrier).
struct HybridLane : MfmaLane {
  static constexpr bool itw_prefetch = true;  // the last group is a VALU group
  // groups with the shared-twiddle mapping
  template <int LOG_R, int LOG_T, bool LAST>
  static constexpr bool shared(int g) {
    const int beta = LOG_R - 2 * (g + 1), log_w = LOG_R + LOG_T - 8;
    if ((LAST && g == 0) || LOG_T > 5 || log_w < 0) return false;
    return beta > 0 && beta <= 1 + log_w;
  }
  template <int LOG_R, int LOG_T, bool LAST>
  static constexpr bool group_on_mfma(int g) { return shared<LOG_R, LOG_T, LAST>(g); }
  template <int LOG_R, int LOG_T, bool LAST>
  static constexpr int phase(int g) {
    return shared<LOG_R, LOG_T, LAST>(g) ? 100 + g : tile_phase<LOG_R, LOG_T, LAST>(g);
  }
  template <int LOG_R, int LOG_T, bool LAST, int g>
  static __device__ __forceinline__ uint32_t ibase(uint32_t tid) {
    if constexpr (!shared<LOG_R, LOG_T, LAST>(g)) {
      return tile_ibase<LOG_R, LOG_T, g>(tid);
    } else {
      constexpr int beta = LOG_R - 2 * (g + 1);
      const uint32_t lane = tid & 63u, wave = tid >> 6;
      const uint32_t lane_i = (lane & 31u) >> LOG_T;   // 5 - LOG_T row bits run over the lanes of a half-wave
      const uint32_t u = (lane >> 5) | (wave << 1);    // 1 + log2(waves) bits: the low beta row bits, the rest goes on top
      const uint32_t hi = lane_i | ((u >> beta) << (5 - LOG_T));
      return (u & ((1u << beta) - 1u)) | (hi << (beta + 2));
    }
  }
};
// experiment (STARKHIP_HYBRID_MATH=valu): the hybrid's mapping and barriers with every butterfly on the VALUs -- what the
// shared-twiddle mapping costs by itself
struct HybridMapOnlyLane : HybridLane {
  template <int LOG_R, int LOG_T, bool LAST>
  static constexpr bool group_on_mfma(int) { return false; }
};

#ifdef SHK_STAMPS  // the diagnostic library only (make stamps): these two variants give WRONG RESULTS by design
// timing experiment only (STARKHIP_HYBRID_MATH=skip, WRONG RESULTS): the shared groups exchange their elements and do no
// arithmetic at all -- the ceiling of what any faster butterfly in those groups could return
struct HybridSkipLane : HybridLane {
  template <int LOG_R, int g>
  __device__ __forceinline__ void butterflies(TileThread&) const {}
};
// timing experiment only (STARKHIP_HYBRID_MATH=frag0, WRONG RESULTS): every butterfly multiplies by the table's first entry, so the
// operand-image loads always hit the L1 -- what the latency of those loads costs
struct HybridFrag0Lane : HybridLane {
  template <int LOG_R, int g>
  __device__ __forceinline__ void butterflies(TileThread& th) const {
    shk_x8 xv[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) xv[h][i] = th.x[h].v[i];
    shk_group_asm_MMMM(xv, offs, lane16, mlo, mhi, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u);
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) th.x[h].v[i] = xv[h][i];
  }
};
#endif

template <int LOG_R, int LOG_T, bool LAST, class LANE>
__global__ void __launch_bounds__(1 << (LOG_R + LOG_T - 2)) __attribute__((amdgpu_waves_per_eu(4, 4))) ntt_htile_kernel(NttPassArgs a) {
  LANE ln;
  lane_setup(ln, a);
  ntt_pass_body<LOG_R, LOG_T, LAST>(a, ln);
}

// STARKHIP_HYBRID_MATH: 0 = the blocks (default), 1 = "valu"; the diagnostic library also knows 2 = "skip", 3 = "frag0"
int hybrid_math() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_HYBRID_MATH");
    int m = (e && !strcmp(e, "valu")) ? 1 : 0;
#ifdef SHK_STAMPS
    if (e && !strcmp(e, "skip")) m = 2;
    if (e && !strcmp(e, "frag0")) m = 3;
#endif
    v = m;
  }
  return v;
}

// STARKHIP_HYBRID_TILE_LOG = 10: 1024-element tiles (256 threads, 32 KiB: four workgroups per CU, one wave of each per SIMD) for the
// radices that fit; default 11: 2048 elements, 512 threads, 64 KiB of LDS, two workgroups per CU
int hybrid_tile_log() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_HYBRID_TILE_LOG");
    v = (e && atoi(e) == 10) ? 10 : 11;
  }
  return v;
}

template <int LOG_R, bool LAST, int TILE_LOG>
hipError_t launch_htile_sized(const NttPassArgs& a, hipStream_t st) {
  constexpr int LOG_T = TILE_LOG - LOG_R;
  constexpr unsigned THREADS = 1u << (TILE_LOG - 2);
  constexpr size_t LDS = (size_t)32 << TILE_LOG;
  static std::atomic<uint64_t> attr_done{0}, attr_done_v{0};
#ifdef SHK_STAMPS
  static std::atomic<uint64_t> attr_done_s{0}, attr_done_f{0};
  if (hybrid_math() == 3)
    return shk_launch_tile_kernel(ntt_htile_kernel<LOG_R, LOG_T, LAST, HybridFrag0Lane>, attr_done_f, LOG_T, THREADS, LDS, LAST, a, st);
  if (hybrid_math() == 2)
    return shk_launch_tile_kernel(ntt_htile_kernel<LOG_R, LOG_T, LAST, HybridSkipLane>, attr_done_s, LOG_T, THREADS, LDS, LAST, a, st);
#endif
  if (hybrid_math() == 1)
    return shk_launch_tile_kernel(ntt_htile_kernel<LOG_R, LOG_T, LAST, HybridMapOnlyLane>, attr_done_v, LOG_T, THREADS, LDS, LAST, a, st);
  return shk_launch_tile_kernel(ntt_htile_kernel<LOG_R, LOG_T, LAST, HybridLane>, attr_done, LOG_T, THREADS, LDS, LAST, a, st);
}

template <int LOG_R, bool LAST>
hipError_t launch_htile(const NttPassArgs& a, hipStream_t st) {
  if constexpr (LOG_R <= 10 && LOG_R >= 5) {
    if (hybrid_tile_log() == 10 && (LAST || a.log_S >= (uint32_t)(10 - LOG_R))) return launch_htile_sized<LOG_R, LAST, 10>(a, st);
  }
  if constexpr (LOG_R >= 6) return launch_htile_sized<LOG_R, LAST, 11>(a, st);
  return hipErrorInvalidValue;
}

template <bool LAST>
hipError_t dispatch_htile(int log_R, const NttPassArgs& a, hipStream_t st) {
  switch (log_R) {
    case 5: return launch_htile<5, LAST>(a, st);
    case 6: return launch_htile<6, LAST>(a, st);
    case 7: return launch_htile<7, LAST>(a, st);
    case 8: return launch_htile<8, LAST>(a, st);
    case 9: return launch_htile<9, LAST>(a, st);
    case 10: return launch_htile<10, LAST>(a, st);
    case 11: return launch_htile<11, LAST>(a, st);
    default: return hipErrorInvalidValue;
  }
}

template <int LOG_R, bool LAST>
hipError_t launch_ltile(const NttPassArgs& a, hipStream_t st) {
  constexpr size_t LDS = (size_t)32 << (LOG_R + 5);
  auto k = ntt_ltile_kernel<LOG_R, LAST>;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const uint64_t tiles = (a.total + 31) >> 5;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(8u << LOG_R), LDS, st, a);
  return hipGetLastError();
}

template <bool LAST>
hipError_t dispatch_ltile(int log_R, const NttPassArgs& a, hipStream_t st) {
  switch (log_R) {
    case 5: return launch_ltile<5, LAST>(a, st);
    case 6: return launch_ltile<6, LAST>(a, st);
    case 7: return launch_ltile<7, LAST>(a, st);
    default: return hipErrorInvalidValue;
  }
}

template <bool LAST>
hipError_t dispatch(int log_R, const NttPassArgs& a, hipStream_t st) {
  switch (log_R) {
    case 5: return launch_ctile<5, LAST>(a, st);
    case 6: return launch_ctile<6, LAST>(a, st);
    case 7: return launch_ctile<7, LAST>(a, st);
    case 8: return launch_ctile<8, LAST>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool shk_ntt_mfma_supports(int log_R, bool last, const NttPassArgs& a) {
  if (a.mfma_kind == 2) return a.mats && log_R >= 5 && log_R <= 7 && (last || a.log_S >= 5);
  if (a.mfma_kind == 3) {
    if (!a.mats) return false;
    if (hybrid_tile_log() == 10 && log_R >= 5 && log_R <= 10 && (last || a.log_S >= (uint32_t)(10 - log_R))) return true;
    return log_R >= 6 && log_R <= 11 && (last || a.log_S >= (uint32_t)(11 - log_R));
  }
  if (log_R < 5 || log_R > 8 || !a.mats) return false;
  if (!last && (a.log_S < 5 || !a.tw2)) return false;  // a tile's 32 columns must be adjacent
  return true;
}

hipError_t shk_launch_ntt_pass_mfma(int log_R, bool last, const NttPassArgs& a, hipStream_t st) {
  if (a.mfma_kind == 2) return last ? dispatch_ltile<true>(log_R, a, st) : dispatch_ltile<false>(log_R, a, st);
  if (a.mfma_kind == 3) return last ? dispatch_htile<true>(log_R, a, st) : dispatch_htile<false>(log_R, a, st);
  return last ? dispatch<true>(log_R, a, st) : dispatch<false>(log_R, a, st);
}
