#!/usr/bin/env python3
"""forward+inverse 2^logn NTT of `batch` vectors, issued through K contexts (= K streams), batch/K vectors each, from one host
thread (sh_dev_ntt only enqueues): do the launches of different streams fill each other's ramps and tails?
args: logn batch [streams...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev, root_of
dev = Dev(); L = dev.L
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << logn
w = root_of(n).to_bytes(32, "big")
dx, dy = dev.alloc(32 * n * batch), dev.alloc(32 * n * batch)
dev.ck(L.sh_dev_fill_seeded(dev.ctx, dx, n * batch, 0x5eed), "fill")
ctxs = [dev.ctx]
for K in [int(a) for a in sys.argv[3:]] or [1, 2, 4]:
    while len(ctxs) < K:
        c = ctypes.c_void_p()
        dev.ck(L.sh_ctx_create(0, ctypes.byref(c)), "ctx")
        ctxs.append(c)
    per = batch // K
    off = lambda p, k: ctypes.c_void_p(p.value + 32 * n * per * k)
    def step():
        for k in range(K):
            dev.ck(L.sh_dev_ntt(ctxs[k], off(dx, k), off(dy, k), n, per, w, 0), "ntt")
        for k in range(K):
            dev.ck(L.sh_dev_ntt(ctxs[k], off(dy, k), off(dy, k), n, per, w, 1), "intt")
    def sync():
        for k in range(K):
            dev.ck(L.sh_sync(ctxs[k]), "sync")
    for _ in range(3):
        step()
    sync()
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    sync()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    print("2^%d batch %d over %d stream(s): %.4f ms per fwd+inv step (wall), %.2f G elements/s" % (logn, batch, K, ms, 2 * n * batch / ms / 1e6), flush=True)
