import os, sys, time, hashlib, struct
sys.path.insert(0, os.getcwd())
from starks_amd import fft
P = 2**256 - 2**32*351 + 1
n = 1 << 20
data = os.urandom(32 * n)
w = pow(7, (P - 1) // n, P)
fft.ntt_bytes(data, n, w)
t = time.perf_counter()
for _ in range(5): out = fft.ntt_bytes(data, n, w)
dt = (time.perf_counter() - t) / 5
print("sh_ntt host-buffer API, 2^20: %.2f ms per call = %.2f G elements/s (PCIe + conversion inclusive)" % (dt * 1e3, n / dt / 1e9))
