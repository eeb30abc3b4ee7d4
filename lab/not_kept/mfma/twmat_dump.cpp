// Host-only helper for tests/test_host_cpu.py::test_twiddle_matrix_images: prints the MFMA operand image (TwMat,
// starks_amd/csrc/mfma_tw.cuh) of the twiddles given as 64-hex-digit big-endian arguments.  No GPU call is made.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfma_tw.cuh"

int main(int argc, char** argv) {
  for (int a = 1; a < argc; ++a) {
    if (strlen(argv[a]) != 64) return 2;
    fp w;
    for (int i = 0; i < 8; ++i) {
      char buf[9];
      memcpy(buf, argv[a] + 8 * (7 - i), 8);
      buf[8] = 0;
      w.v[i] = (uint32_t)strtoul(buf, nullptr, 16);
    }
    TwMat m;
    if (!shk_build_twmat(w, &m)) return 3;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(&m);
    for (size_t k = 0; k < sizeof m; ++k) printf("%02x", p[k]);
    printf("\n");
  }
  return 0;
}
