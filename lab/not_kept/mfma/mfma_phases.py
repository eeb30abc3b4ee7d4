#!/usr/bin/env python3
"""Phase timeline of the MFMA tile pass (diagnostic library: make -C starks_amd/csrc stamps).
   STARKHIP_LIB=starks_amd/libstarkhip_stamps.so [STARKHIP_STAMP_PASS=d] python3 tools/mfma_phases.py LOGN [BATCH]
Runs forward NTTs; after each, reads the s_memtime stamps of wave 0 of the first workgroups of the LAST pass launched and
prints the median cycles between phase boundaries (load | stage 1 | exchange | stage 2 | twiddle + store)."""
import ctypes, os, sys
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
st = [[buf[k * 1024 + b] for b in range(1024)] for k in range(7)]
nb = sum(1 for b in range(1024) if st[5][b] > st[0][b] > 0)
names = ["load", "stage1", "exchange", "stage2", "twiddle+store"]
import statistics
for k in range(5):
    d = [st[k + 1][b] - st[k][b] for b in range(nb) if st[k + 1][b] > st[k][b]]
    print("%-14s median %8.0f  min %8.0f  max %8.0f  (ticks of s_memtime, %d workgroups)" % (names[k], statistics.median(d), min(d), max(d), len(d)))
d = [st[6][b] - st[0][b] for b in range(nb) if st[6][b] > st[0][b]]
if d:
    print("%-14s median %8.0f  (row pass: address arithmetic + issue of the 16 global loads, part of `load`)" % ("  loads issued", statistics.median(d)))
tot = [st[5][b] - st[0][b] for b in range(nb)]
print("%-14s median %8.0f" % ("whole tile", statistics.median(tot)))
starts = sorted(st[0][b] for b in range(nb))
print("start of workgroup #%d after the first: %d ticks" % (min(255, nb - 1), starts[min(255, nb - 1)] - starts[0]))
