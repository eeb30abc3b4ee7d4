// knobs.hpp -- the experiment knobs of the launch path and the plan choice that depends on them.  Host-only, plain C++
// (no HIP types): the library includes it, and tests/native/knobs_tsan.cpp compiles it with g++ -fsanitize=thread.
//
// Every STARKHIP_* variable that shapes a launch is read from the environment ONCE per process, inside std::call_once, into
// one immutable ShkKnobs; afterwards the launch path only reads that object.  Contexts on several host threads (include/
// starkhip.h: contexts are independent) therefore share no mutable state here.  (STARKHIP_LIB / STARKHIP_DEVICE are read by
// the Python binding, STARKHIP_PLAN_CACHE_MB gives a new context its initial plan budget.)
//
//   STARKHIP_NTT_RADICES="10,10"   log2 radices of the passes (digits 2..11, at most 4), for the sizes they sum to
//   STARKHIP_TILE_LOG=9|10|11      elements per tile (log2) of every pass of radix <= 2^8 (default 10; the first pass of a
//                                  long transform 11, ntt.hip); setting it switches that first-pass rule off
//   STARKHIP_TILE_LOG_BIG=10|11|12 the same for the radix 2^9 .. 2^11 passes (default 11)
//   STARKHIP_TILE_LOGS="11,10,10"  per pass 0, 1, 2, ... (9..12; 0 = the rules above)
//   STARKHIP_XCD_SWZ=0|1|2         workgroup -> tile mapping over the 8 XCDs (ntt_kernels.cuh:shk_launch_tile_kernel)
//   STARKHIP_TW2_MAX_LOG=k         row-major inter-pass twiddle tables up to 2^k entries (default 24), else the power-table lookup
//   STARKHIP_PLAN_CACHE_MB=m       initial plan-cache budget of a context
//   STARKHIP_NTT_NARROW_TILES=k    passes of at most k 1024-element tiles (and radix <= 2^10) run in the one-butterfly-per-thread form
//                                  (ntt_narrow_pass_kernel; default 256 = the CUs of the chip; 0 = never)
// All of them exist for the parity tests over alternate plans (tests/test_gpu_parity.py::test_alternate_ntt_plans_parity,
// tools/stress_plans.py) and for A/B measurements; the defaults are the measured best.
#pragma once
#include <stdlib.h>

#include <mutex>

struct ShkKnobs {
  int tile_log = 10;         // passes of radix <= 2^8
  bool tile_forced = false;  // STARKHIP_TILE_LOG was given
  int tile_log_big = 11;     // passes of radix >= 2^9
  int tile_logs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int xcd_swz = 1;
  int tw2_max_log = 24;
  long plan_cache_mb = -1;   // -1: not given
  int n_radices = 0;         // 0: not given (or malformed)
  int radices[4] = {0, 0, 0, 0};
  int radix_sum = 0;
  long narrow_tiles = 256;   // 0: the narrow form is never used
};

namespace shk_knobs_detail {
inline void parse_list(const char* e, int* out, int n, long lo, long hi, int dflt) {
  int i = 0;
  for (const char* p = e; p && *p && i < n; ++i) {
    char* end = nullptr;
    const long x = strtol(p, &end, 10);
    if (end == p) break;
    out[i] = (x >= lo && x <= hi) ? (int)x : dflt;
    p = (*end == ',') ? end + 1 : end;
  }
}
inline void parse(ShkKnobs* k) {
  if (const char* e = getenv("STARKHIP_TILE_LOG")) {
    const int v = atoi(e);
    k->tile_forced = true;
    if (v == 9 || v == 10 || v == 11) k->tile_log = v;
  }
  if (const char* e = getenv("STARKHIP_TILE_LOG_BIG")) {
    const int v = atoi(e);
    if (v == 10 || v == 11 || v == 12) k->tile_log_big = v;
  }
  parse_list(getenv("STARKHIP_TILE_LOGS"), k->tile_logs, 8, 9, 12, 0);
  if (const char* e = getenv("STARKHIP_XCD_SWZ")) {
    const int v = atoi(e);
    k->xcd_swz = (v < 0 || v > 2) ? 0 : v;
  }
  if (const char* e = getenv("STARKHIP_TW2_MAX_LOG")) {
    const int v = atoi(e);
    k->tw2_max_log = v < 0 ? 0 : v > 28 ? 28 : v;
  }
  if (const char* e = getenv("STARKHIP_PLAN_CACHE_MB")) {
    const long v = atol(e);
    if (v >= 0) k->plan_cache_mb = v;
  }
  if (const char* e = getenv("STARKHIP_NTT_NARROW_TILES")) {
    const long v = atol(e);
    k->narrow_tiles = v < 0 ? 0 : v;
  }
  if (const char* e = getenv("STARKHIP_NTT_RADICES")) {
    int r[4] = {0, 0, 0, 0}, cnt = 0, sum = 0;
    bool ok = true;
    for (const char* p = e; *p && ok;) {
      char* end = nullptr;
      const long v = strtol(p, &end, 10);
      if (end == p || v < 2 || v > 11 || cnt == 4) {
        ok = false;
        break;
      }
      r[cnt++] = (int)v;
      sum += (int)v;
      if (*end && *end != ',') ok = false;
      p = (*end == ',') ? end + 1 : end;
    }
    if (ok && cnt >= 1) {
      k->n_radices = cnt;
      k->radix_sum = sum;
      for (int i = 0; i < 4; ++i) k->radices[i] = r[i];
    }
  }
}
}  // namespace shk_knobs_detail

inline const ShkKnobs& shk_knobs() {
  static ShkKnobs knobs;
  static std::once_flag once;
  std::call_once(once, [] { shk_knobs_detail::parse(&knobs); });
  return knobs;
}

// The passes of a 2^log_n-point transform: log2 radices into out[0..4), returns their number.
//   n <= 2^8: one pass.  2^9 .. 2^16: two passes of 1024-element tiles.  2^17 .. 2^20: two passes of radix 2^8 .. 2^10 over
//   2048-element tiles -- (9, 8), (9, 9), (9, 10), (10, 10) -- measured ahead of three passes there (one inter-pass twiddle
//   product and one read + write of the vector fewer; 2^20: 12.3 / 15.5 / 16.5 against 10.6 / 15.2 / 15.9 G elements/s at
//   1 / 8 / 32 vectors).  From 2^21 up: ceil(log_n / 8) passes of near-equal radix <= 2^8 (two-pass plans measured behind).
inline int shk_choose_radices(int log_n, int out[4]) {
  const ShkKnobs& k = shk_knobs();
  if (k.n_radices && k.radix_sum == log_n) {
    for (int i = 0; i < k.n_radices; ++i) out[i] = k.radices[i];
    return k.n_radices;
  }
  if (log_n <= 8) {
    out[0] = log_n;
    return 1;
  }
  if (log_n >= 17 && log_n <= 20) {
    out[0] = log_n == 20 ? 10 : 9;
    out[1] = log_n - out[0];
    return 2;
  }
  const int m = (log_n + 7) / 8, base = log_n / m, rem = log_n % m;
  if (m > 4) return 0;
  for (int i = 0; i < m; ++i) out[i] = base + (i < rem ? 1 : 0);
  return m;
}
