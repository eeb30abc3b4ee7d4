// ntt.hip -- instantiations and launchers of the NTT tile passes (ntt_kernels.cuh).
#include <stdlib.h>

#include <atomic>

#include "knobs.hpp"
#include "ntt_kernels.cuh"

#ifndef SHK_NARROW_HALF_TILES
#define SHK_NARROW_HALF_TILES 128
#endif

namespace {

template <int LOG_R, bool LAST, int TILE_LOG>
hipError_t launch_tile(const NttPassArgs& a, hipStream_t st) {
  constexpr int LOG_T = TILE_LOG - LOG_R;
  static std::atomic<uint64_t> attr_done{0};
  return shk_launch_tile_kernel(ntt_pass_kernel<LOG_R, LOG_T, LAST>, attr_done, LOG_T, 1u << (TILE_LOG - 2), (size_t)32 << TILE_LOG,
                                a, st);
}

// narrow launches (ntt_kernels.cuh): at most kn.narrow_tiles tiles of 1024 elements, radix <= 2^10
template <int LOG_R, bool LAST>
bool launch_narrow(const NttPassArgs& a, hipStream_t st, hipError_t* err) {
  if constexpr (LOG_R <= 10) {
    const ShkKnobs& kn = shk_knobs();
    constexpr int LOG_T = 10 - LOG_R;
    const uint64_t tiles = (a.total + ((1ull << LOG_T) - 1)) >> LOG_T;
    if (kn.narrow_tiles <= 0 || tiles == 0 || tiles > (uint64_t)kn.narrow_tiles) return false;
    if (a.pass_index < 8 && kn.tile_logs[a.pass_index]) return false;  // a forced tile size means the tile-pass kernels
    // Up to SHK_NARROW_HALF_TILES tiles: 512-element tiles, 256 threads -- the launch then has at most one wave per SIMD, and a level's
    // product chain is not shared with a second wave of the same workgroup (radix <= 2^8: at least two columns / rows per tile)
    if constexpr (LOG_R <= 8) {
      if (tiles <= SHK_NARROW_HALF_TILES) {
        constexpr int LOG_T9 = 9 - LOG_R;
        const uint64_t tiles9 = (a.total + ((1ull << LOG_T9) - 1)) >> LOG_T9;
        hipLaunchKernelGGL((ntt_narrow_pass_kernel<LOG_R, LOG_T9, LAST>), dim3((unsigned)tiles9), dim3(256), 0, st, a);
        *err = hipGetLastError();
        return true;
      }
    }
    hipLaunchKernelGGL((ntt_narrow_pass_kernel<LOG_R, LOG_T, LAST>), dim3((unsigned)tiles), dim3(1u << (LOG_R + LOG_T - 1)), 0, st, a);
    *err = hipGetLastError();
    return true;
  }
  return false;
}

// The FIRST column pass of a long transform (P = 1: the rows of a tile are n / R elements apart) gets 2048-element tiles once that
// distance reaches 2 MiB (n / R >= 2^16): twice the columns per row (256-byte segments at radix 2^8) are worth + 3.6 % on a
// 2^24-point transform and + 4 % on two of them (profiles/r03_first_pass_tile_2p24.txt); below that distance the 1024-element
// tiles of the other passes stay ahead, and so they do for the radix-2^7 first pass of the four-pass plan of 2^25 points: the
// rule is radix 2^8 only.  The knobs (knobs.hpp, read once per process) override it.
template <int LOG_R, bool LAST>
hipError_t launch(const NttPassArgs& a, hipStream_t st) {
  const ShkKnobs& kn = shk_knobs();
  hipError_t ne = hipSuccess;
  if (launch_narrow<LOG_R, LAST>(a, st, &ne)) return ne;
  if (const int f = a.pass_index < 8 ? kn.tile_logs[a.pass_index] : 0) {
    if (f == 12 && LOG_R >= 4) return launch_tile<LOG_R, LAST, 12>(a, st);
    if (f == 11) return launch_tile<LOG_R, LAST, 11>(a, st);
    if (f == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
    if (f == 9) return launch_tile<LOG_R, LAST, 9>(a, st);
  }
  if constexpr (!LAST && LOG_R == 8) {
    if (a.log_S + LOG_R == a.log_n && a.log_S >= 16 && !kn.tile_forced) return launch_tile<LOG_R, LAST, 11>(a, st);
  }
  if (kn.tile_log == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
  if (kn.tile_log == 9) return launch_tile<LOG_R, LAST, 9>(a, st);
  return launch_tile<LOG_R, LAST, 11>(a, st);
}

// Radices above 2^8 (the two-pass plans of 2^17 .. 2^20 points): 2048-element tiles (64 KiB, two workgroups per CU); the knobs
// select 1024- or 4096-element tiles (128 KiB, 1024 threads, one workgroup per CU) for experiments.
template <int LOG_R, bool LAST>
hipError_t launch_big(const NttPassArgs& a, hipStream_t st) {
  const ShkKnobs& kn = shk_knobs();
  hipError_t ne = hipSuccess;
  if (launch_narrow<LOG_R, LAST>(a, st, &ne)) return ne;
  if (const int f = a.pass_index < 8 ? kn.tile_logs[a.pass_index] : 0) {
    if (f == 12) return launch_tile<LOG_R, LAST, 12>(a, st);
    if (f == 11) return launch_tile<LOG_R, LAST, 11>(a, st);
    if constexpr (LOG_R <= 10) {
      if (f == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
    }
  }
  if constexpr (LOG_R <= 10) {
    if (kn.tile_log_big == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
  }
  if (kn.tile_log_big <= 11) return launch_tile<LOG_R, LAST, 11>(a, st);
  return launch_tile<LOG_R, LAST, 12>(a, st);
}

template <bool LAST>
hipError_t dispatch(int log_R, const NttPassArgs& a, hipStream_t st) {
  switch (log_R) {
    case 2: return launch<2, LAST>(a, st);
    case 3: return launch<3, LAST>(a, st);
    case 4: return launch<4, LAST>(a, st);
    case 5: return launch<5, LAST>(a, st);
    case 6: return launch<6, LAST>(a, st);
    case 7: return launch<7, LAST>(a, st);
    case 8: return launch<8, LAST>(a, st);
    case 9: return launch_big<9, LAST>(a, st);
    case 10: return launch_big<10, LAST>(a, st);
    case 11: return launch_big<11, LAST>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t shk_launch_ntt_pass(int log_R, bool last, const NttPassArgs& a, hipStream_t st) {
  return last ? dispatch<true>(log_R, a, st) : dispatch<false>(log_R, a, st);
}

hipError_t shk_launch_ntt_tiny(const fp* src, fp* dst, uint32_t n, uint32_t batch, const fp* scale, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  hipLaunchKernelGGL(ntt_tiny_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, src, dst, n, batch, scale);
  return hipGetLastError();
}
