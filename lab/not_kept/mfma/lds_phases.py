#!/usr/bin/env python3
"""Phase timeline of the LDS-resident matrix-core tile pass (diagnostic library: make -C starks_amd/csrc stamps).
   STARKHIP_NTT_PATH=mfma_lds STARKHIP_LIB=starks_amd/libstarkhip_stamps.so [STARKHIP_STAMP_PASS=d] python3 tools/lds_phases.py LOGN [BATCH]
Median s_memtime ticks of wave 0 of the first workgroups between the phase boundaries: group 0 (with the global loads), the
later register groups, then inter-pass twiddle + store."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev, root_of
dev = Dev(); L, ctx = dev.L, dev.ctx
logn = int(sys.argv[1]); batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 1 << logn
w = root_of(n).to_bytes(32, "big")
dx, dy = dev.alloc(32 * n * batch), dev.alloc(32 * n * batch)
dev.ck(L.sh_dev_fill_seeded(ctx, dx, n * batch, 0x5eed), "fill")
for _ in range(3):
    dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, batch, w, 0), "ntt")
dev.sync()
print("passes:", L.sh_ntt_passes(n, 0), " recorded pass:", os.environ.get("STARKHIP_STAMP_PASS", "last"))
buf = (ctypes.c_ulonglong * (8 * 1024))()
L.sh_debug_stamps.argtypes = [ctypes.c_void_p]
assert L.sh_debug_stamps(buf) == 0
st = [[buf[k * 1024 + b] for b in range(1024)] for k in range(8)]
last = max(k for k in range(8) if st[k][0])
nb = sum(1 for b in range(1024) if st[last][b] > st[0][b] > 0)
for k in range(last):
    d = [st[k + 1][b] - st[k][b] for b in range(nb) if st[k + 1][b] > st[k][b]]
    name = "group %d" % k if k + 1 < last else "twiddle+store"
    print("%-14s median %8.0f  min %8.0f  max %8.0f  (ticks of s_memtime, %d workgroups)" % (name, statistics.median(d), min(d), max(d), len(d)))
print("%-14s median %8.0f" % ("whole tile", statistics.median([st[last][b] - st[0][b] for b in range(nb)])))
