// ntt.hip -- instantiations and launchers of the NTT tile passes (ntt_kernels.cuh).
#include <stdlib.h>

#include <atomic>

#include "ntt_kernels.cuh"

namespace {

// Tiles narrower than 128 bytes (T < 4: the big-radix passes) share their cache lines with the neighbouring tile; workgroups
// are dealt to the 8 XCDs round-robin, so neighbours would sit behind different L2s and every line would be fetched (or
// written back partially) twice.  Default (1): such launches map adjacent tiles to the same XCD (measured on the T = 1
// variant: 8.6 -> 10.2 G elements/s).  STARKHIP_XCD_SWZ: 0 = never, 2 = every tile pass (measured level for T >= 4),
// 3 = as 1 plus the sharer-fastest order below.
int xcd_swizzle() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STARKHIP_XCD_SWZ");
    v = e ? atoi(e) : 1;
    if (v < 0 || v > 3) v = 0;
  }
  return v;
}

template <int LOG_R, bool LAST, int TILE_LOG>
hipError_t launch_tile(const NttPassArgs& a, hipStream_t st) {
  constexpr int LOG_T = TILE_LOG - LOG_R;
  constexpr int THREADS = 1 << (TILE_LOG - 2);
  constexpr size_t LDS = (size_t)32 << TILE_LOG;
  auto k = ntt_pass_kernel<LOG_R, LOG_T, LAST>;
  // the attribute is per device: one bit per device ordinal, per instantiation (contexts on several devices, and on
  // several host threads, share this function)
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
  const uint64_t tiles = (a.total + ((1ull << LOG_T) - 1)) >> LOG_T;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7ffffff0ull) return hipErrorInvalidValue;
  NttPassArgs b = a;
  uint64_t grid = tiles;
  // column passes whose twiddle rows (tw2) are shared by several vectors / prefix blocks: sharer-fastest order, XCD-local.
  // Opt-in only (STARKHIP_XCD_SWZ=3): it removes the per-vector re-fetch of the rows (2 * FETCH_SIZE of the 2^20 x 8 column
  // pass 540 -> 325 MB) but measured 3-8 % SLOWER -- the sharers sit 2^k bytes apart, so the concurrently running tiles
  // all map to the same memory channels.
  const uint64_t sharers = LAST ? 0 : (a.total >> a.log_S);
  const bool share = !LAST && a.tw2 && sharers >= 2 && sharers <= 0xffffffffull && a.log_S >= (uint32_t)LOG_T + 2 &&
                     tiles <= 0xffffffffull && xcd_swizzle() == 3;
  if (xcd_swizzle() && (xcd_swizzle() == 2 || LOG_T < 2 || share) && tiles >= 64) {
    b.xcd_per = (uint32_t)((tiles + 7) / 8);
    b.sharers = share ? (uint32_t)sharers : 0;
    grid = 8ull * b.xcd_per;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(THREADS), LDS, st, b);
  return hipGetLastError();
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

template <int LOG_R, bool LAST>
hipError_t launch(const NttPassArgs& a, hipStream_t st) {
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
