// Host-side check of the NTT tile passes' thread -> element mapping (csrc/ntt_kernels.cuh: tile_thread_coords, tile_phase, lds_slot),
// compiled with hipcc and run on the CPU.  For every tile shape the library instantiates:
//   * in every register group the threads' 4 elements each partition the tile (every (row, column) exactly once);
//   * where two consecutive groups are in the same phase the hand-over through LDS has NO workgroup barrier, so the elements (and
//     therefore the LDS slots) a wave reads in group g + 1 must be exactly the ones the same wave wrote in group g;
//   * lds_slot is a bijection of the tile onto its LDS image and linear over GF(2) (the kernel reaches a thread's other three
//     elements with one XOR).
#include <cstdio>
#include <set>
#include <utility>
#include <vector>

#include "ntt_kernels.cuh"

static int failures = 0, shapes = 0, wave_local_exchanges = 0;

template <int LOG_R, int LOG_T, bool LAST, int g>
static void elements_of_group(std::vector<std::vector<uint32_t>>* per_thread) {
  constexpr int threads = 1 << (LOG_R + LOG_T - 2);
  constexpr int beta = (LOG_R - 2 * (g + 1)) > 0 ? (LOG_R - 2 * (g + 1)) : 0;
  per_thread->assign(threads, {});
  for (uint32_t tid = 0; tid < (uint32_t)threads; ++tid) {
    uint32_t t = 0, ibase = 0;
    tile_thread_coords<LOG_R, LOG_T, LAST, g>(tid, &t, &ibase);
    for (uint32_t h = 0; h < 4; ++h) (*per_thread)[tid].push_back((((ibase | (h << beta)) << LOG_T) | t));
  }
}

template <int LOG_R, int LOG_T, bool LAST, int g>
static void check_groups() {
  constexpr int G = (LOG_R + 1) / 2;
  constexpr int tile = 1 << (LOG_R + LOG_T);
  std::vector<std::vector<uint32_t>> cur;
  elements_of_group<LOG_R, LOG_T, LAST, g>(&cur);
  std::set<uint32_t> all;
  for (auto& v : cur)
    for (uint32_t e : v) {
      if (e >= (uint32_t)tile || !all.insert(e).second) {
        if (!failures++) printf("R=2^%d T=2^%d last=%d group %d: element %u out of range or held twice\n", LOG_R, LOG_T, (int)LAST, g, e);
      }
    }
  if ((int)all.size() != tile && !failures++) printf("R=2^%d T=2^%d last=%d group %d: not a partition\n", LOG_R, LOG_T, (int)LAST, g);
  if constexpr (g + 1 < G) {
    if constexpr (tile_phase<LOG_R, LOG_T, LAST>(g) == tile_phase<LOG_R, LOG_T, LAST>(g + 1)) {
      ++wave_local_exchanges;
      std::vector<std::vector<uint32_t>> nxt;
      elements_of_group<LOG_R, LOG_T, LAST, g + 1>(&nxt);
      const size_t waves = (cur.size() + 63) / 64;
      for (size_t w = 0; w < waves; ++w) {
        std::set<uint32_t> wr, rd;
        for (size_t tid = 64 * w; tid < 64 * (w + 1) && tid < cur.size(); ++tid) {
          for (uint32_t e : cur[tid]) wr.insert(lds_slot(e));
          for (uint32_t e : nxt[tid]) rd.insert(lds_slot(e));
        }
        if (wr != rd && !failures++)
          printf("R=2^%d T=2^%d last=%d: wave %zu reads in group %d what another wave wrote in group %d\n", LOG_R, LOG_T, (int)LAST, w,
                 g + 1, g);
      }
    }
    check_groups<LOG_R, LOG_T, LAST, g + 1>();
  }
}

template <int LOG_R, int TILE_LOG>
static void check_shape() {
  constexpr int LOG_T = TILE_LOG - LOG_R;
  if constexpr (LOG_T >= 0 && LOG_R + LOG_T >= 8 && LOG_R + LOG_T <= 12) {
    ++shapes;
    constexpr uint32_t tile = 1u << TILE_LOG;
    std::set<uint32_t> slots;
    for (uint32_t e = 0; e < tile; ++e) {
      const uint32_t o = lds_slot(e);
      if (o >= 2 * tile || (o & 1u) || !slots.insert(o).second) {
        if (!failures++) printf("tile 2^%d: lds_slot(%u) = %u collides or leaves the image\n", TILE_LOG, e, o);
      }
      for (uint32_t d = 1; d < tile; d <<= 1)
        if (lds_slot(e ^ d) != (lds_slot(e) ^ lds_slot(d)) && !failures++) printf("lds_slot is not linear at %u ^ %u\n", e, d);
    }
    check_groups<LOG_R, LOG_T, false, 0>();
    check_groups<LOG_R, LOG_T, true, 0>();
  }
}

template <int TILE_LOG>
static void check_tile_size() {
  check_shape<2, TILE_LOG>();
  check_shape<3, TILE_LOG>();
  check_shape<4, TILE_LOG>();
  check_shape<5, TILE_LOG>();
  check_shape<6, TILE_LOG>();
  check_shape<7, TILE_LOG>();
  check_shape<8, TILE_LOG>();
  check_shape<9, TILE_LOG>();
  check_shape<10, TILE_LOG>();
  check_shape<11, TILE_LOG>();
}

int main() {
  check_tile_size<9>();
  check_tile_size<10>();
  check_tile_size<11>();
  check_tile_size<12>();
  printf("%d tile shapes, %d barrier-free exchanges, %d failures\n", shapes, wave_local_exchanges, failures);
  return failures ? 1 : 0;
}
