#!/usr/bin/env python3
"""forward+inverse 2^logn NTT throughput vs batch (sh_dev_ntt): args logn [batches...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev, root_of
dev = Dev(); L, ctx = dev.L, dev.ctx
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
w = root_of(n).to_bytes(32, "big")
for batch in [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16, 32]:
    dx, dy = dev.alloc(32 * n * batch), dev.alloc(32 * n * batch)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n * batch, 0x5eed), "fill")
    def step():
        dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, batch, w, 0), "ntt")
        dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, batch, w, 1), "intt")
    ms = dev.timed(step, 30)
    print("2^%d batch %2d: %.4f ms per fwd+inv step, %.2f G elements/s" % (logn, batch, ms, 2 * n * batch / ms / 1e6), flush=True)
    dev.free(dx); dev.free(dy)
