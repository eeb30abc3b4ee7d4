// knobs_tsan.cpp -- the launch path's process-wide state under ThreadSanitizer (CPU build; g++ -fsanitize=thread).
// starks_amd/csrc/knobs.hpp holds every STARKHIP_* launch knob (parsed once, std::call_once) and the plan choice built on it;
// include/starkhip.h promises that contexts on different host threads are independent, so first use from several threads at
// once -- the way two contexts' first transforms meet -- must be race-free and must give every thread the same answer.
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "knobs.hpp"

int main() {
  constexpr int NT = 8;
  std::atomic<int> go{0};
  int passes[NT][33];
  const ShkKnobs* seen[NT];
  std::vector<std::thread> th;
  for (int t = 0; t < NT; ++t)
    th.emplace_back([&, t] {
      while (!go.load(std::memory_order_acquire)) {
      }
      seen[t] = &shk_knobs();  // first use races here if the parse is not once-only
      for (int rep = 0; rep < 200; ++rep)
        for (int lg = 0; lg <= 32; ++lg) {
          int r[4] = {0, 0, 0, 0};
          const int m = shk_choose_radices(lg, r);
          int sum = 0;
          for (int i = 0; i < m; ++i) sum += r[i];
          passes[t][lg] = (m && sum == lg) ? m : -1;
        }
    });
  go.store(1, std::memory_order_release);
  for (auto& x : th) x.join();
  for (int t = 1; t < NT; ++t) {
    if (seen[t] != seen[0]) return 2;
    if (memcmp(passes[t], passes[0], sizeof passes[0])) return 3;
  }
  for (int lg = 0; lg <= 32; ++lg)
    if (passes[0][lg] < 1) return 4;
  const ShkKnobs& k = shk_knobs();
  printf("ok tile_log=%d big=%d logs0=%d swz=%d tw2=%d cache=%ld radices=%d passes20=%d passes24=%d\n", k.tile_log, k.tile_log_big,
         k.tile_logs[0], k.xcd_swz, k.tw2_max_log, k.plan_cache_mb, k.n_radices, passes[0][20], passes[0][24]);
  return 0;
}
