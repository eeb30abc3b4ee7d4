#!/usr/bin/env python3
"""FRI commit timing per size (GPU box): ms per proof for batch 1 and a batched run; used with rocprofv3."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev, root_of
dev = Dev(); L, ctx = dev.L, dev.ctx
cfgs = [(14, 1), (14, 16), (16, 1), (16, 16), (20, 1)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
for logsteps, batch in cfgs:
    steps, ext = 1 << logsteps, 8
    n = steps * ext
    w = root_of(n).to_bytes(32, "big")
    plen = int(L.sh_fri_proof_len(n, steps, 40))
    dc, dp = dev.alloc(32 * n * batch), dev.alloc(plen * batch)
    dev.ck(L.sh_dev_fill_seeded(ctx, dc, n * batch, 0xF51), "fill")
    if os.environ.get("FRI_PROFILE_DENSE") == "1":  # the explicitly zero-padded vector (sh_dev_fri_prove)
        z = bytes(32 * (n - steps))
        for b in range(batch):
            dev.ck(L.sh_dev_upload(ctx, z, ctypes.c_void_p(dc.value + 32 * (b * n + steps)), len(z)), "upload")
        ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, batch, dp), "fri"), 5)
    else:                                           # [batch][steps] coefficients, implicit padding
        ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove_coeffs(ctx, dc, steps, n, w, steps, ext, 40, batch, dp), "fri"), 5)
    print("steps 2^%d batch %d: %.4f ms per launch, %.4f ms per proof" % (logsteps, batch, ms, ms / batch), flush=True)
    dev.free(dc); dev.free(dp)
