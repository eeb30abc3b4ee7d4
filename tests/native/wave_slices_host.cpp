// Host-side check of every barrier-free LDS hand-over of the hashing kernels (csrc/wave_chunks.cuh: the kernels take their constants
// from the plans listed there; compiled with plain g++).  For each plan (256 threads = 4 waves), each hand-over, each wave:
//   * every LDS index the wave touches -- parking its threads' chunks (own) and moving them lane-contiguously (moved) -- lies inside
//     the wave's slice [w * 64 * plan::LDS_CH, (w + 1) * 64 * plan::LDS_CH), the SAME slice in every hand-over of the kernel, and
//     inside the array the kernel declares (256 * plan::LDS_CH chunks);
//   * what a wave parks is exactly what it moves (no chunk lost or duplicated), and its global chunks are the contiguous range
//     [w * 64 * CH, (w + 1) * 64 * CH), each once.
// The plan of commit 284afb6's Merkle mid kernel (load through slices of 8 chunks per thread, store through slices of 4) must FAIL:
// that is the race of round 4, which 114 GPU tests could not see.
#include <cstdio>
#include <set>

#include "wave_chunks.cuh"

static int failures = 0, handovers = 0;

template <class H>
static void check_handover(const char* name, int index, int plan_lds_ch, bool quiet) {
  ++handovers;
  const uint32_t threads = 256, array = threads * (uint32_t)plan_lds_ch;
  for (uint32_t w = 0; w < threads / 64; ++w) {
    const uint32_t lo = w * 64u * (uint32_t)plan_lds_ch, hi = (w + 1) * 64u * (uint32_t)plan_lds_ch;
    std::set<uint32_t> own, moved, glob;
    bool ok = true;
    for (uint32_t t = 64 * w; t < 64 * (w + 1); ++t)
      for (int c = 0; c < H::CH; ++c) {
        const uint32_t a = H::own(t, c), b = H::moved(t, c), g = H::gidx(t, c);
        ok = ok && a >= lo && a < hi && b >= lo && b < hi && a < array && b < array;
        ok = own.insert(a).second && ok;
        ok = moved.insert(b).second && ok;
        ok = glob.insert(g).second && ok;
      }
    ok = ok && own == moved && own.size() == 64u * H::CH;
    ok = ok && *glob.begin() == w * 64u * H::CH && *glob.rbegin() == (w + 1) * 64u * H::CH - 1;
    if (!ok) {
      if (!failures++ && !quiet) printf("%s: hand-over %d (CH %d, slices of %d): wave %u leaves its slice [%u, %u) or loses chunks\n", name, index, H::CH, H::LDS_CH, w, lo, hi);
    }
  }
}
template <class Plan, int I = 0>
static void check_plan(const char* name, bool quiet = false) {
  if constexpr (I < Plan::count) {
    check_handover<typename handover_at<I, Plan>::type>(name, I, Plan::LDS_CH, quiet);
    check_plan<Plan, I + 1>(name, quiet);
  }
}

int main() {
  check_plan<merkle_leaves_nostore_plan>("merkle_leaves_kernel<., false, .>");
  check_plan<merkle_mid_plan>("merkle_mid_kernel");
#if defined(SHK_HAVE_STARK_FUSED_PLAN)
  check_plan<stark_fused_plan>("stark_quotients_leaves_kernel");
#endif
  const int good_failures = failures, good_handovers = handovers;
  // the round-4 bug, as it was committed (284afb6): must be caught
  using r04_bug = handover_plan<8, wave_handover<8, 8>, wave_handover<4, 4>>;
  failures = 0;
  check_plan<r04_bug>("r04 mid kernel", true);
  const bool bug_caught = failures > 0;
  printf("%d hand-overs, %d failures, r04 bug %s\n", good_handovers, good_failures, bug_caught ? "caught" : "MISSED");
  return good_failures == 0 && bug_caught ? 0 : 1;
}
