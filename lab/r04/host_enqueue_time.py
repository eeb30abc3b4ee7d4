#!/usr/bin/env python3
"""Host time to ENQUEUE one FRI commit / one STARK batch (the sh_dev_* entry points only enqueue): if it exceeds the GPU time of the
work, the stream runs dry and the commit is launch-bound on the host."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import Dev, root_of
dev = Dev(); L, ctx = dev.L, dev.ctx
for logsteps, batch in ((14, 1), (16, 1), (20, 1), (16, 32)):
    steps, ext = 1 << logsteps, 8
    n = steps * ext
    w = root_of(n).to_bytes(32, "big")
    plen = int(L.sh_fri_proof_len(n, steps, 40))
    dc, dp = dev.alloc(32 * n * batch), dev.alloc(plen * batch)
    dev.ck(L.sh_dev_fill_seeded(ctx, dc, n * batch, 0xF51), "fill")
    call = lambda: dev.ck(L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, batch, dp), "fri")
    call(); dev.sync()
    host = []
    for _ in range(5):
        dev.sync()
        t0 = time.perf_counter(); call(); t1 = time.perf_counter()
        dev.sync(); t2 = time.perf_counter()
        host.append((t1 - t0, t2 - t0))
    print("steps 2^%d batch %d: host enqueue %.3f ms (min of 5), enqueue + wait %.3f ms" % (logsteps, batch, min(h[0] for h in host) * 1e3, min(h[1] for h in host) * 1e3), flush=True)
    dev.free(dc); dev.free(dp)
