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
# several vectors per call from pinned buffers: chunks pipelined over two copy streams (sh_ntt_batch) vs one call per vector
B = 8
srcb, dstb = _lib.PinnedBuffer(32 * n * B), _lib.PinnedBuffer(32 * n * B)
for b in range(B):
    srcb.view[32 * n * b:32 * n * (b + 1)] = data
fft.ntt_bytes(srcb, n, w, batch=B, out=dstb)
assert all(bytes(dstb.view[32 * n * b:32 * n * (b + 1)]) == ref for b in range(B))
t = time.perf_counter()
for _ in range(5): fft.ntt_bytes(srcb, n, w, batch=B, out=dstb)
dt = (time.perf_counter() - t) / 5
print("sh_ntt_batch 8 x 2^20, pinned buffers, pipelined chunks: %.2f ms per call = %.2f G elements/s" % (dt * 1e3, B * n / dt / 1e9))
t = time.perf_counter()
for _ in range(5):
    for b in range(B): fft.ntt_bytes(src, n, w, out=dst)
dt = (time.perf_counter() - t) / 5
print("8 x sh_ntt 2^20, pinned buffers, one call per vector:   %.2f ms = %.2f G elements/s" % (dt * 1e3, B * n / dt / 1e9))
# the drop-in call site end to end: Python ints in, field elements out (fft.py:316-331)
from starks_amd import IntegersModP
F = IntegersModP(P)
vals = [int.from_bytes(data[32 * i:32 * i + 32], "big") % P for i in range(n)]
t = time.perf_counter()
res = fft.fft_1d(F, vals, P, w)
dt = time.perf_counter() - t
t = time.perf_counter(); wire = _lib.to_wire(vals); t1 = time.perf_counter() - t
t = time.perf_counter(); ints = _lib.from_wire(ref); t2 = time.perf_counter() - t
t = time.perf_counter(); fe = [F(x) for x in ints]; t3 = time.perf_counter() - t
print("fft_1d(2^20) end to end: %.2f s (ints -> wire %.3f s, wire -> ints %.3f s, ints -> field elements %.3f s; the transform itself 0.085 ms)" % (dt, t1, t2, t3))
