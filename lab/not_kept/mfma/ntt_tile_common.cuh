// ntt_tile_common.cuh -- helpers of the register-tile NTT passes (ntt_mfma.hip; tools/not_kept/ntt_mfma2.hip).
#pragma once
#include <type_traits>
#include <utility>

#include "internal.hpp"

namespace {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The butterfly bodies contain
// convergent operations (MFMA operands moved with v_permlane32_swap, ballots), which `#pragma unroll` refuses to unroll;
// the 16 elements of a thread must be addressed with constant indices to stay in registers.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS window of one exchange round: RW rows x CW columns of 32-byte elements, the column slot XOR-ed with the row so
// that both access patterns (lanes along the columns of one row, lanes along the rows of one column) spread over banks.
template <int LOG_CW>
__device__ __forceinline__ uint32_t win_slot(uint32_t i, uint32_t c) {
  constexpr uint32_t CW = 1u << LOG_CW;
  return ((i << LOG_CW) | ((c ^ i) & (CW - 1u))) * 2u;  // in uint4 units
}

template <int LOG_R>
__device__ __forceinline__ void row_coords(const NttPassArgs& a, uint64_t col, uint64_t* gbase, uint64_t* obase) {
  // the row pass of ntt_kernels.cuh: tiles enumerate rows with the FIRST pass's digit fastest, so that 32 rows of a
  // tile land on 32 adjacent output addresses
  const uint64_t b = col >> a.log_P;
  const uint32_t pp = (uint32_t)(col & ((1ull << a.log_P) - 1));
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
  *gbase = (b << a.log_n) + ((uint64_t)p << LOG_R);
  *obase = (b << a.log_n) + acc;
}

}  // namespace
