// How fast does the chip move "column tiles" -- ROWS rows of SEG contiguous bytes each, the rows STRIDE bytes apart -- as the
// NTT column passes do?  Pass 0 of the 2^24-point transform (rows 2 MiB apart) runs ~35 % slower than pass 1 (rows 8 KiB
// apart) with the same arithmetic; this separates reads from writes and the stride from the segment length.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// one workgroup per tile: 256 rows; thread t handles row (t / LPR + k * (256 / RPI)) ...: lanes run along the segment first
template <int MODE>  // 0 read, 1 write, 2 read + write (in place)
__global__ void __launch_bounds__(256) k(uint4* buf, uint64_t stride16, uint32_t seg16, uint64_t tiles_per_block_row, uint4* sink) {
  const uint32_t lpr = seg16;                 // lanes per row segment (uint4 = 16 B each)
  const uint32_t rows_per_iter = 256 / lpr;
  const uint32_t lr = threadIdx.x / lpr, lc = threadIdx.x % lpr;
  // tile index -> (block of 256 rows, segment index inside the row)
  const uint64_t blk = blockIdx.x / tiles_per_block_row, segi = blockIdx.x % tiles_per_block_row;
  uint4* base = buf + blk * 256 * stride16 + segi * seg16 + lc;
  uint4 acc = {0, 0, 0, 0};
#pragma unroll 4
  for (uint32_t r = lr; r < 256; r += rows_per_iter) {
    uint4* p = base + (uint64_t)r * stride16;
    if (MODE != 1) {
      uint4 v = *p;
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      if (MODE == 2) { v.x += 1; *p = v; }
    } else {
      *p = uint4{r, lc, 1, 2};
    }
  }
  if (MODE != 1 && acc.x == 0x12345678u) sink[0] = acc;
}
int main() {
  const size_t bytes = (size_t)512 << 20;
  uint4 *buf, *sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 64); hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"read", "write", "read+write"};
  for (uint64_t stride : {(uint64_t)8 << 10, (uint64_t)64 << 10, (uint64_t)512 << 10, (uint64_t)2 << 20})
    for (uint32_t seg : {128u, 1024u}) {
      // the buffer as blocks of 256 rows x stride bytes; a tile is one seg-byte column segment of a block
      const uint64_t blocks = bytes / (256 * stride), per_row = stride / seg, tiles = blocks * per_row;
      for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(tiles), dim3(256), 0, 0, buf, stride / 16, seg / 16, per_row, sink);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(tiles), dim3(256), 0, 0, buf, stride / 16, seg / 16, per_row, sink);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(tiles), dim3(256), 0, 0, buf, stride / 16, seg / 16, per_row, sink);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("row stride %7llu KiB, segment %4u B, %-10s: %7.3f ms for 512 MiB -> %6.2f TB/s of traffic\n", (unsigned long long)(stride >> 10), seg,
               names[mode], best, (mode == 2 ? 2.0 : 1.0) * bytes / best / 1e9);
      }
    }
  return 0;
}
