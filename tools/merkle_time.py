import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev
dev = Dev(); L, ctx = dev.L, dev.ctx
for logn in (20, 22, 24):
    n = 1 << logn
    dx, dt = dev.alloc(32 * n), dev.alloc(64 * n)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 7), "fill")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkle"), 10)
    print("variant", os.environ.get("STARKHIP_MERKLE_VARIANT", "0"), "merkelize 2^%d: %.4f ms  %.2f G leaves/s" % (logn, ms, n / ms / 1e6), flush=True)
    dev.free(dx); dev.free(dt)
