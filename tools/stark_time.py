#!/usr/bin/env python3
"""STARK.mk_proof timing on the device-resident API (GPU box): ms per proof for the reference's MiMC formulation
(width 2, step polynomials [X_1, X_1 + X_2^3], test_stark.py:265-293).  Args: logsteps:batch ...  Used with rocprofv3."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Dev
from starks_amd import stark
from starks_amd.multivariate_polynomial import generate_Xi_s
from starks_amd.modp import IntegersModP
from starks_amd._lib import MIMC_P as P

dev = Dev(); L, ctx = dev.L, dev.ctx
F = IntegersModP(P)
X1, X2 = generate_Xi_s(F, 2)
coefs, exps, counts, degree = stark.pack_step_polys([X1, X1 + X2**3], 2)
cfgs = [(14, 1), (16, 1), (16, 16), (20, 1)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
for logsteps, batch in cfgs:
    steps, ext, width = 1 << logsteps, 8, 2
    k, x = 42, 3
    col = [x]
    for _ in range(steps - 1):
        x = (x * x * x + k) % P
        col.append(x)
    wit = b"".join(v.to_bytes(32, "big") for v in [k] * steps + col) * batch
    inp = (k.to_bytes(32, "big") + (3).to_bytes(32, "big")) * batch
    plen = stark.proof_len(steps, ext, width, degree)
    dw, di, dp = dev.alloc(len(wit)), dev.alloc(len(inp)), dev.alloc(plen * batch)
    dev.ck(L.sh_dev_from_wire(ctx, inp, di, width * batch), "inputs")

    def run():
        dev.ck(L.sh_dev_from_wire(ctx, wit, dw, width * steps * batch), "witness")  # the prover overwrites its witness
        dev.ck(L.sh_dev_stark_prove(ctx, dw, di, steps, ext, width, coefs, exps, counts, 80, batch, dp), "stark")
    run(); dev.sync()
    best = 1e9
    for _ in range(5):
        dev.ck(L.sh_dev_from_wire(ctx, wit, dw, width * steps * batch), "witness")
        dev.sync()
        dev.ck(L.sh_timer_start(ctx), "t")
        dev.ck(L.sh_dev_stark_prove(ctx, dw, di, steps, ext, width, coefs, exps, counts, 80, batch, dp), "stark")
        ms = ctypes.c_float(); dev.ck(L.sh_timer_stop(ctx, ctypes.byref(ms)), "t")
        best = min(best, ms.value)
    dev.ck(L.sh_stark_status(ctx), "status")
    print("steps 2^%d batch %d: %.4f ms per launch, %.4f ms per proof (%d B each)" % (logsteps, batch, best, best / batch, plen), flush=True)
    dev.free(dw); dev.free(di); dev.free(dp)
