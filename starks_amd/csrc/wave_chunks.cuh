// wave_chunks.cuh -- index arithmetic of the LDS-transposed I/O of the hashing kernels (kernels.hip, stark.hip), as plain
// __host__ __device__ functions, plus the list of every kernel's barrier-free hand-overs: the kernels take their constants from
// here and tests/native/wave_slices_host.cpp enumerates the same constants on the host.
//
// A thread that owns CH adjacent 16-byte chunks (a 128-byte row of four leaves, a 64-byte node pair) would move them at a lane
// stride of CH * 16 bytes: measured 2.5 TB/s against 5.7 TB/s lane-contiguous (lab/r01_r03/membench.hip).  The chunks therefore go
// through an LDS image (XOR-swizzled by bank row, conflict-free both ways).  PER-WAVE form: a wave parks the CH chunks of each of its
// 64 threads in its OWN slice of the workgroup's LDS array and moves them lane-contiguously; the LDS unit serves one wave's requests
// in order, so no workgroup barrier is involved.  That is only safe if the slice a wave uses is the same in EVERY hand-over of the
// kernel: wave w owns [w * 64 * LDS_CH, (w + 1) * 64 * LDS_CH) whatever the hand-over's own CH <= LDS_CH.  (Round 4: the Merkle mid
// kernel loaded through slices of 512 chunks and stored through slices of 256 -- a wave that was already storing wrote into the slice
// its neighbour was still loading through; invisible while the waves of a workgroup run in step, 5-30 % wrong proofs beside a second
// stream; profiles/r04_two_context_race.txt.)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SHK_WC_HD __host__ __device__ __forceinline__
#else
#define SHK_WC_HD inline
#endif

SHK_WC_HD constexpr uint32_t chunk_swz(uint32_t c) { return (c & ~15u) | ((c & 15u) ^ ((c >> 4) & 15u)); }

// one per-wave hand-over of CH chunks per thread inside slices sized for LDS_CH chunks per thread
template <int CH_, int LDS_CH_>
struct wave_handover {
  static constexpr int CH = CH_, LDS_CH = LDS_CH_;
  static_assert(CH_ >= 1 && CH_ <= LDS_CH_, "a hand-over must fit the wave's own LDS slice");
  static SHK_WC_HD uint32_t lbase(uint32_t t) { return (t >> 6) * (64u * LDS_CH_); }  // first chunk of the wave's LDS slice
  static SHK_WC_HD uint32_t gbase(uint32_t t) { return (t >> 6) * (64u * CH_); }      // first chunk of the wave in global memory
  static SHK_WC_HD uint32_t own(uint32_t t, int c) { return lbase(t) + chunk_swz(CH_ * (t & 63u) + c); }   // chunk c of thread t
  static SHK_WC_HD uint32_t moved(uint32_t t, int k) { return lbase(t) + chunk_swz(k * 64u + (t & 63u)); }  // k-th lane-contiguous piece
  static SHK_WC_HD uint32_t gidx(uint32_t t, int k) { return gbase(t) + k * 64u + (t & 63u); }
};

// a kernel's hand-overs, in program order; LDS_CH = chunks per thread its LDS array is sized for
template <int LDS_CH_, class... H>
struct handover_plan {
  static constexpr int LDS_CH = LDS_CH_;
  static constexpr int count = sizeof...(H);
};
template <int I, class Plan>
struct handover_at;
template <int LDS_CH_, class H0, class... H>
struct handover_at<0, handover_plan<LDS_CH_, H0, H...>> {
  using type = H0;
};
template <int I, int LDS_CH_, class H0, class... H>
struct handover_at<I, handover_plan<LDS_CH_, H0, H...>> {
  using type = typename handover_at<I - 1, handover_plan<LDS_CH_, H...>>::type;
};

// ---- every kernel with barrier-free hand-overs (256 threads per workgroup) -------------------------------------------------------
// merkle_leaves_kernel<RAW, STORE = false>: the pair level (2 digests = 4 chunks per thread) goes out per wave
using merkle_leaves_nostore_plan = handover_plan<4, wave_handover<4, 4>>;
// merkle_mid_kernel: 4 nodes (8 chunks) per thread come in, the 2 parents (4 chunks) go out
using merkle_mid_plan = handover_plan<8, wave_handover<8, 8>, wave_handover<4, 8>>;
