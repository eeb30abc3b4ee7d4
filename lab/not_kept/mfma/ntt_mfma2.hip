// NOT BUILT INTO THE LIBRARY: measured 25-30 % slower than the VALU passes (`profiles/r02_ntt_paths_valu_vs_mfma.txt`);
// kept as the record of the experiment.  It passed the NTT / LDE / FRI parity tests when it was wired in as STARKHIP_NTT_PATH=mfma2.
// ntt_mfma2.hip -- the matrix-core tile pass of ntt_mfma.hip with every element SPLIT over a lane pair (mfma_split.cuh):
// lane c holds limbs 0..3 of column c, lane c + 32 limbs 4..7.  That is the MFMA's own operand layout, so a butterfly needs no
// lane swaps and one accumulator, and a lane carries 4 registers per element: a wave holds 16 rows x 32 columns in 64
// VGPRs and the kernel fits 128 VGPRs -> four waves per SIMD (ntt_mfma.hip: two), the occupancy that hides the carry wait
// states, the MFMA latency and the memory phases.  The price: every carry between the halves of an element crosses lanes
// (hf_fix), about 9 % more instructions per butterfly (tools/mfma_modmul.hip: 290 against 317 G butterflies/s).
//
// A workgroup = R/16 waves owns R rows x 32 columns.  Wave w holds rows m * (R/16) + w (stage 1: the four top DIF levels on
// the register index, one twiddle matrix per butterfly for the whole wave), then rows 16 w + m' after one exchange through LDS
// (stage 2: the remaining levels, compile-time twiddles).  For the inter-pass twiddle -- a general fp_mul per element -- and
// for the store, pairs of rows are converted to whole form (4 swaps per pair: the lower lane gets one row, the upper lane the
// other).  Same plans, tables and results as the other passes.
#include <stdlib.h>

#include <atomic>

#include "mfma_split.cuh"
#include "ntt_kernels.cuh"
#include "ntt_tile_common.cuh"

namespace {

__device__ __forceinline__ hf hf_from(const uint4& q) {
  hf r;
  r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  return r;
}
__device__ __forceinline__ uint4 hf_pack(const hf& r) { return make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]); }

template <int LOG_R, bool LAST>
__global__ void __launch_bounds__(4 << LOG_R) __attribute__((amdgpu_waves_per_eu(4, 4))) ntt_stile_kernel(NttPassArgs a) {
  static_assert(LOG_R >= 5 && LOG_R <= 8, "unsupported radix");
  constexpr int R = 1 << LOG_R;
  constexpr int G = R / 16;  // waves = row groups
  constexpr int THREADS = 4 * R;
  constexpr int NR = 2;      // exchange rounds: 16 columns (R/2 KiB of LDS) each
  constexpr int LOG_CW = 4;
  constexpr int CW = 1 << LOG_CW;
  constexpr int QMAX = LOG_R - 5 < 3 ? LOG_R - 5 : 3;  // top level of stage 2
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];

  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t c = lane & 31u;
  const bool hb = lane >= 32u;
  const uint64_t col_raw = ((uint64_t)blockIdx.x << 5) + c;
  const bool active = col_raw < a.total;
  const uint64_t col = active ? col_raw : a.total - 1;  // clamped: loaded, computed, never stored
  const shk_kinit cinit = shk_mfma_kinit(lane);
  const TwMat* mats = reinterpret_cast<const TwMat*>(a.mats);

  uint64_t gbase = 0, obase = 0, j2 = 0;
  if (LAST) {
    row_coords<LOG_R>(a, col, &gbase, &obase);
  } else {
    j2 = col & ((1ull << a.log_S) - 1);
    gbase = ((col >> a.log_S) << (LOG_R + a.log_S)) + j2;
  }

  hf x[16];
  // ---- load ------------------------------------------------------------------------------------------------------
  if (!LAST) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const uint32_t i = (uint32_t)m * G + wave;
      x[m] = hf_from(reinterpret_cast<const uint4*>(a.src + gbase + ((uint64_t)i << a.log_S))[hb ? 1 : 0]);
    }
  } else {
    // rows are contiguous: lanes run along the row (whole 32-byte elements), the tile is transposed through LDS
    constexpr int ELEMS = CW * R / THREADS;  // per thread per round (= 4)
    fp ld[NR * ELEMS];
    static_for<NR * ELEMS>([&](auto ei) {
      constexpr int e = decltype(ei)::value % ELEMS, k = decltype(ei)::value / ELEMS;
      const uint32_t flat = (uint32_t)e * THREADS + tid;
      const uint32_t r_l = LOG_R >= 6 ? (uint32_t)__builtin_amdgcn_readfirstlane(flat >> LOG_R) : flat >> LOG_R;
      const uint32_t i = flat & (R - 1);
      uint64_t rcol = ((uint64_t)blockIdx.x << 5) + (uint32_t)k * CW + r_l;
      if (rcol >= a.total) rcol = a.total - 1;
      uint64_t gb, ob;
      row_coords<LOG_R>(a, rcol, &gb, &ob);
      ld[decltype(ei)::value] = fp_load(a.src + gb + i);
    });
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
        for (int m = 0; m < 16; ++m)
          x[m] = hf_from(lds[win_slot<LOG_CW>((uint32_t)m * G + wave, c & (CW - 1)) + (hb ? 1u : 0u)]);
      }
    }
    __syncthreads();
  }

  // ---- stage 1: levels LOG_R-1 .. LOG_R-4 on the register index m (bit mu = 3 .. 0); one matrix per butterfly ------------
  {
    struct Frag2 {
      shk_v4i w, nw;
    };
    auto frags_of = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int mu = 3 - j / 8, b = j % 8;
      constexpr int m0 = ((b >> mu) << (mu + 1)) | (b & ((1 << mu) - 1));
      constexpr uint32_t eb = (uint32_t)(m0 & ((1 << mu) - 1)) * G;
      const TwMat* t = mats + ((eb + wave) << (3 - mu));
      Frag2 f;
      f.w = shk_ld_frag(t->w, lane);
      f.nw = shk_ld_frag(t->nw, lane);
      return f;
    };
    Frag2 cur = frags_of(std::integral_constant<int, 0>{});
    static_for<32>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int mu = 3 - j / 8, b = j % 8;
      constexpr int m0 = ((b >> mu) << (mu + 1)) | (b & ((1 << mu) - 1));
      constexpr int m1 = m0 | (1 << mu);
      Frag2 nxt = cur;
      if constexpr (j + 1 < 32) {
        constexpr int mu2 = 3 - (j + 1) / 8, b2 = (j + 1) % 8;
        constexpr int m02 = ((b2 >> mu2) << (mu2 + 1)) | (b2 & ((1 << mu2) - 1));
        constexpr bool same = mu2 == mu && (m02 & ((1 << mu2) - 1)) == (m0 & ((1 << mu) - 1));
        if constexpr (!same) nxt = frags_of(std::integral_constant<int, j + 1>{});
      }
      const hf d = hf_submul(x[m0], x[m1], cur.w, cur.nw, cinit, hb);
      x[m0] = hf_add(x[m0], x[m1], hb);
      x[m1] = d;
      cur = nxt;
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- exchange: (m, wave) -> (wave', m') ---------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const bool mine = (c >> LOG_CW) == (uint32_t)k;
    if (k > 0) __syncthreads();
    if (mine) {
#pragma unroll
      for (int m = 0; m < 16; ++m) lds[win_slot<LOG_CW>((uint32_t)m * G + wave, c & (CW - 1)) + (hb ? 1u : 0u)] = hf_pack(x[m]);
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int m = 0; m < 16; ++m) x[m] = hf_from(lds[win_slot<LOG_CW>(16u * wave + (uint32_t)m, c & (CW - 1)) + (hb ? 1u : 0u)]);
    }
  }

  // ---- stage 2: levels QMAX .. 0 on the register index m' (twiddles identical for every thread) ------------
  {
    struct Frag2 {
      shk_v4i w, nw;
    };
    constexpr int NB = 8 * (QMAX + 1);
    auto twid_of = [](int t) {  // twiddle index of butterfly t (0 = none)
      const int q = QMAX - t / 8, b = t % 8;
      const int m0 = ((b >> q) << (q + 1)) | (b & ((1 << q) - 1));
      return (m0 & ((1 << q) - 1)) << (LOG_R - 1 - q);
    };
    auto frags_of = [&](int e) {
      Frag2 f;
      f.w = shk_ld_frag(mats[e].w, lane);
      f.nw = shk_ld_frag(mats[e].nw, lane);
      return f;
    };
    Frag2 cur = {}, nxt = {};
    static_for<NB>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      constexpr int q = QMAX - t / 8, b = t % 8;
      constexpr int m0 = ((b >> q) << (q + 1)) | (b & ((1 << q) - 1));
      constexpr int m1 = m0 | (1 << q);
      constexpr int e = twid_of(t);
      if constexpr (t + 1 < NB) {
        constexpr int e2 = twid_of(t + 1);
        if constexpr (e2 != 0 && e2 != e) nxt = frags_of(e2);
      }
      if constexpr (e == 0) {
        const hf s = hf_add(x[m0], x[m1], hb);
        x[m1] = hf_sub(x[m0], x[m1], hb);
        x[m0] = s;
      } else {
        const hf d = hf_submul(x[m0], x[m1], cur.w, cur.nw, cinit, hb);
        x[m0] = hf_add(x[m0], x[m1], hb);
        x[m1] = d;
      }
      if constexpr (t + 1 < NB) {
        constexpr int e2 = twid_of(t + 1);
        if constexpr (e2 != 0 && e2 != e) cur = nxt;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- pairs of rows to whole form (lower lane: row 2 j, upper lane: row 2 j + 1), inter-pass twiddle, store ----------------
  // position i = 16 wave + m' holds frequency bitrev(i)
  {
    auto tw_of = [&](int j) {
      const uint32_t i = 16u * wave + 2u * (uint32_t)j + (hb ? 1u : 0u);
      const uint32_t k = __brev(i) >> (32 - LOG_R);
      return fp_load(a.tw2 + ((uint64_t)k << a.log_S) + j2);
    };
    fp tw = {}, twn = {};
    fp sc = {};
    if (LAST) {
      if (a.scale) sc = fp_load(a.scale);
    } else {
      tw = tw_of(0);
    }
    static_for<8>([&](auto ji) {
      constexpr int j = decltype(ji)::value;
      if constexpr (!LAST && j + 1 < 8) twn = tw_of(j + 1);
      fp v = hf_to_whole(x[2 * j], x[2 * j + 1]);
      const uint32_t i = 16u * wave + 2u * (uint32_t)j + (hb ? 1u : 0u);
      const uint32_t k = __brev(i) >> (32 - LOG_R);
      if (LAST) {
        if (a.scale) v = fp_mul(v, sc);
        if (active) fp_store(a.dst + obase + ((uint64_t)k << a.log_P), v);
      } else {
        v = fp_mul(v, tw);
        if (active) fp_store(a.dst + gbase + ((uint64_t)k << a.log_S), v);
        tw = twn;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
}

template <int LOG_R, bool LAST>
hipError_t launch_stile(const NttPassArgs& a, hipStream_t st) {
  constexpr int R = 1 << LOG_R;
  constexpr size_t LDS = (size_t)R * 16 * 32;  // one exchange window: R rows x 16 columns
  auto k = ntt_stile_kernel<LOG_R, LAST>;
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
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(4 * R), LDS, st, a);
  return hipGetLastError();
}

template <bool LAST>
hipError_t dispatch2(int log_R, const NttPassArgs& a, hipStream_t st) {
  switch (log_R) {
    case 5: return launch_stile<5, LAST>(a, st);
    case 6: return launch_stile<6, LAST>(a, st);
    case 7: return launch_stile<7, LAST>(a, st);
    case 8: return launch_stile<8, LAST>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t shk_launch_ntt_pass_mfma2(int log_R, bool last, const NttPassArgs& a, hipStream_t st) {
  return last ? dispatch2<true>(log_R, a, st) : dispatch2<false>(log_R, a, st);
}
