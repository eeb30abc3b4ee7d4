#!/usr/bin/env python3
"""Randomised STARK prover differential test (GPU box): random sparse step polynomials, widths 1..5, steps 8..64, extension factors
2..16, random inputs -- the flat proof of starks_amd.stark.prove_flat against the oracle's coefficient-form prover
(oracle/pyoracle.py), byte for byte.  usage: stress_stark.py [seconds]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from starks_amd import stark, IntegersModP, MIMC_P as P
from starks_amd.multivariate_polynomial import multivariates_over
F = IntegersModP(P)
wire = lambda vals: b"".join(int(v).to_bytes(32, "big") for v in vals)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(4242)
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    width = rng.choice([1, 2, 2, 3, 4, 5])
    steps = rng.choice([8, 16, 32, 64])
    ext = rng.choice([2, 4, 8, 8, 16])
    maxdeg = rng.randint(1, 3)
    sp = []
    for _ in range(width):
        terms = {}
        for _ in range(rng.randint(1, 3)):
            ex = [0] * width
            for _ in range(rng.randint(0, maxdeg)):
                ex[rng.randrange(width)] += 1
            terms[tuple(ex)] = rng.choice([1, 2, 3, rng.randrange(P), P - 1])
        sp.append(terms)
    deg = max(1, max(sum(e) for t in sp for e in t))
    if deg * (steps - 1) + 1 >= steps * ext:
        continue
    inputs = [rng.randrange(P) for _ in range(width)]
    w = po.get_computational_trace(inputs, steps, sp)
    want = po.stark_flat(po.mk_stark_proof(w, inputs, sp, steps, ext))
    mv = multivariates_over(F, width).factory
    got = stark.prove_flat(b"".join(wire(col) for col in w), wire(inputs), steps, ext, width, [mv(d) for d in sp])
    if got != want:
        print("MISMATCH", width, steps, ext, sp, inputs)
        sys.exit(1)
    n += 1
    if n % 10 == 0:
        print("%d systems ok, %.0f s" % (n, time.time() - t0), flush=True)
print("stress: %d random systems, every proof equal to the oracle's" % n)
