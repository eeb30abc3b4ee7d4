#!/usr/bin/env python3
"""The non-NTT kernels of the hot path at the sizes the bench reports, for rocprofv3 (kernel trace or one --pmc pass):
merkelize of 2^24 stored leaves (merkle_leaves_kernel<false,true>, merkle_mid_kernel, merkle_top_kernel), a FRI commit of a
2^20-step trace (merkle_leaves_kernel<false,false>, fri_fold_kernel, sample / gather) and a batch of 2^16-step STARK proofs
(stark_quotients_leaves_kernel, stark_lincomb_leaves_kernel).  Args: reps (default 3)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev, root_of, ProofShard
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = Dev(); L, ctx = dev.L, dev.ctx
# 1. Merkle commit of 2^24 values, leaves stored
n = 1 << 24
dx, dt = dev.alloc(32 * n), dev.alloc(64 * n)
dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 7), "fill")
ms = dev.timed(lambda: dev.ck(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkle"), reps)
print("merkelize 2^24: %.4f ms" % ms, flush=True)
dev.free(dx); dev.free(dt)
# 2. FRI commit of a 2^20-step trace
steps, ext = 1 << 20, 8
n = steps * ext
w = root_of(n).to_bytes(32, "big")
plen = int(L.sh_fri_proof_len(n, steps, 40))
dc, dp = dev.alloc(32 * n), dev.alloc(plen)
dev.ck(L.sh_dev_fill_seeded(ctx, dc, n, 0xF51), "fill")
z = bytes(32 * (n - steps))
dev.ck(L.sh_dev_upload(ctx, z, ctypes.c_void_p(dc.value + 32 * steps), len(z)), "upload")
ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, 1, dp), "fri"), reps)
print("fri commit 2^20 steps: %.4f ms" % ms, flush=True)
dev.free(dc); dev.free(dp)
# 3. 128 STARK proofs of 2^16 steps in one batched launch (config 5's launch shape)
sh = ProofShard(dev, range(128), 1 << 16, chunk=128, streams=1)
ms = dev.timed(sh.prove_all, reps)
print("stark 128 x 2^16 steps: %.4f ms per launch, %.1f proofs/s" % (ms, 128 / ms * 1e3), flush=True)
