// ntt.hip -- instantiations and launchers of the NTT tile passes (ntt_kernels.cuh).
#include <stdlib.h>

#include <atomic>

#include "ntt_kernels.cuh"

namespace {

template <int LOG_R, bool LAST, int TILE_LOG>
hipError_t launch_tile(const NttPassArgs& a, hipStream_t st) {
  constexpr int LOG_T = TILE_LOG - LOG_R;
  static std::atomic<uint64_t> attr_done{0};
  return shk_launch_tile_kernel(ntt_pass_kernel<LOG_R, LOG_T, LAST>, attr_done, LOG_T, 1u << (TILE_LOG - 2), (size_t)32 << TILE_LOG,
                                LAST, a, st);
}

int tile_log_choice() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_TILE_LOG");
    v = e ? atoi(e) : SHK_TILE_LOG;
    if (v != 9 && v != 10 && v != 11) v = SHK_TILE_LOG;
  }
  return v;
}

// The FIRST column pass of a long transform (P = 1: the rows of a tile are n / R elements apart) gets 2048-element tiles once that
// distance reaches 2 MiB (n / R >= 2^16): twice the columns per row (256-byte segments at radix 2^8) are worth + 3.6 % on a
// 2^24-point transform and + 4 % on two of them (measured in alternation, profiles/r03_first_pass_tile_2p24.txt); below that
// distance the 1024-element tiles of the other passes stay ahead (2^21: - 0.5 .. - 4 %, 2^22 / 2^23: +- 1 %), and so they do for
// the radix-2^7 first pass of the four-pass plan of 2^25 points (16 columns per row: - 1.7 %): the rule is radix 2^8 only.
// STARKHIP_TILE_LOG_FIRST = 10 | 11 | 12 forces a size for every first pass.
int tile_log_first() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_TILE_LOG_FIRST");
    v = e ? atoi(e) : 0;
    if (v != 10 && v != 11 && v != 12) v = 0;
  }
  return v;
}

// experiments: STARKHIP_TILE_LOGS="11,10,10": elements per tile (log2) of pass 0, 1, 2, ... of every transform (0 = default rule)
int tile_log_of_pass(uint32_t d) {
  static int v[8] = {-1, 0, 0, 0, 0, 0, 0, 0};
  if (v[0] < 0) {
    int t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (const char* e = getenv("STARKHIP_TILE_LOGS")) {
      int i = 0;
      for (const char* p = e; *p && i < 8; ++i) {
        char* end = nullptr;
        const long x = strtol(p, &end, 10);
        if (end == p) break;
        t[i] = (x >= 9 && x <= 12) ? (int)x : 0;
        p = (*end == ',') ? end + 1 : end;
      }
    }
    for (int i = 7; i >= 0; --i) v[i] = t[i];
  }
  return d < 8 ? v[d] : 0;
}

template <int LOG_R, bool LAST>
hipError_t launch(const NttPassArgs& a, hipStream_t st) {
  if (const int f = tile_log_of_pass(a.pass_index)) {
    if (f == 12 && LOG_R >= 4) return launch_tile<LOG_R, LAST, 12>(a, st);
    if (f == 11) return launch_tile<LOG_R, LAST, 11>(a, st);
    if (f == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
    if (f == 9) return launch_tile<LOG_R, LAST, 9>(a, st);
  }
  if constexpr (!LAST) {
    static const bool tile_forced = getenv("STARKHIP_TILE_LOG") != nullptr;  // an explicit tile size applies to every pass
    if (a.log_S + LOG_R == a.log_n && !tile_forced) {
      const int f = tile_log_first() ? tile_log_first() : ((LOG_R == 8 && a.log_S >= 16) ? 11 : 0);
      if (f == 12) return launch_tile<LOG_R, LAST, 12>(a, st);
      if (f == 11) return launch_tile<LOG_R, LAST, 11>(a, st);
      if (f == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
    }
  }
  if (tile_log_choice() == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
  if (tile_log_choice() == 9) return launch_tile<LOG_R, LAST, 9>(a, st);
  return launch_tile<LOG_R, LAST, 11>(a, st);
}

// Radices above 2^8 (two-pass plans, STARKHIP_NTT_RADICES): 2048-element tiles (64 KiB, two workgroups per CU) or, with
// STARKHIP_TILE_LOG_BIG=12, 4096-element tiles (128 KiB, 1024 threads, one workgroup per CU).
int big_tile_log_choice() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_TILE_LOG_BIG");
    v = e ? atoi(e) : 11;
    if (v != 10 && v != 11 && v != 12) v = 11;
  }
  return v;
}
template <int LOG_R, bool LAST>
hipError_t launch_big(const NttPassArgs& a, hipStream_t st) {
  if (const int f = tile_log_of_pass(a.pass_index)) {
    if (f == 12) return launch_tile<LOG_R, LAST, 12>(a, st);
    if (f == 11) return launch_tile<LOG_R, LAST, 11>(a, st);
    if constexpr (LOG_R <= 10) {
      if (f == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
    }
  }
  if constexpr (LOG_R <= 10) {
    if (big_tile_log_choice() == 10) return launch_tile<LOG_R, LAST, 10>(a, st);
  }
  if (big_tile_log_choice() <= 11) return launch_tile<LOG_R, LAST, 11>(a, st);
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
