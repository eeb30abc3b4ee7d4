#!/usr/bin/env python3
"""fri fold timing (sh_dev_fri_fold) per domain size; algorithmic bytes 40 N, 7 modmuls per output."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev, root_of
dev = Dev(); L, ctx = dev.L, dev.ctx
for logn in [int(a) for a in sys.argv[1:]] or [19, 21, 23]:
    n = 1 << logn
    w = root_of(n).to_bytes(32, "big")
    dv, dt, dc = dev.alloc(32 * n), dev.alloc(64 * n), dev.alloc(8 * n)
    dev.ck(L.sh_dev_fill_seeded(ctx, dv, n, 3), "fill")
    dev.ck(L.sh_dev_merkelize(ctx, dv, n, 1, dt), "tree")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_fold(ctx, dv, dt, n, 1, w, dc), "fold"), 20)
    print("fold 2^%d: %.4f ms  %.1f GB/s algorithmic, %.1f G modmul/s" % (logn, ms, 40.0 * n / ms / 1e6, 7 * (n / 4) / ms / 1e6), flush=True)
    for p in (dv, dt, dc):
        dev.free(p)
