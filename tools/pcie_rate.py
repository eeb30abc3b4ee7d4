"""sh_ntt (host-buffer entry point) on 2^20 elements: pageable Python buffers vs page-locked ones (sh_host_alloc)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from starks_amd import fft, _lib
P = 2**256 - 2**32*351 + 1
n = 1 << 20
data = os.urandom(32 * n)
w = pow(7, (P - 1) // n, P)
ref = fft.ntt_bytes(data, n, w)
t = time.perf_counter()
for _ in range(5): out = fft.ntt_bytes(data, n, w)
dt = (time.perf_counter() - t) / 5
print("sh_ntt 2^20, pageable buffers: %.2f ms per call = %.2f G elements/s (PCIe + staging + conversion inclusive)" % (dt * 1e3, n / dt / 1e9))
src, dst = _lib.PinnedBuffer(32 * n), _lib.PinnedBuffer(32 * n)
src.view[:] = data
fft.ntt_bytes(src, n, w, out=dst)
assert bytes(dst.view) == ref
t = time.perf_counter()
for _ in range(10): fft.ntt_bytes(src, n, w, out=dst)
dt = (time.perf_counter() - t) / 10
print("sh_ntt 2^20, pinned buffers:   %.2f ms per call = %.2f G elements/s (PCIe + conversion inclusive)" % (dt * 1e3, n / dt / 1e9))
