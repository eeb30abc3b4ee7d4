#!/usr/bin/env python3
"""Randomised NTT differential test over forced plan / tile variants (GPU box): for each variant a child process transforms random
vectors of random sizes and batches (forward, inverse, zero-padded) and compares every output with oracle/oracle.c.
usage: stress_plans.py [seconds [min_logn max_logn [transforms_per_variant]]]"""
import os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, random, sys
sys.path.insert(0, %r)
from oracle import coracle
from starks_amd import fft
P = 2**256 - 351*2**32 + 1
rng = random.Random(int(sys.argv[1]))
logn = int(sys.argv[2])
n = 1 << logn
w = pow(7, (P - 1) // n, P)
for it in range(int(sys.argv[3])):
    batch = rng.choice([1, 1, 2, 3, 5])
    n_in = rng.choice([n, n, n >> 1, n >> 3, max(1, n >> 3) + 1, 1])
    inv = rng.random() < 0.4
    vecs = [b"".join(rng.getrandbits(256).to_bytes(32, "big") for _ in range(n_in)) for _ in range(batch)]
    got = fft.ntt_bytes(b"".join(vecs), n, w, inverse=inv, batch=batch)
    for b, v in enumerate(vecs):
        want = coracle.fft_bytes(v + bytes(32 * (n - n_in)), n, w, inverse=inv)
        assert got[32 * n * b:32 * n * (b + 1)] == want, (logn, batch, n_in, inv, b)
print("ok")
''' % ROOT
def parts(total, k, lo=2, hi=11, rng=random):
    while True:
        r = [rng.randint(lo, hi) for _ in range(k)]
        if sum(r) == total:
            return r
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
lo_n = int(sys.argv[2]) if len(sys.argv) > 3 else 4
hi_n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
per = sys.argv[4] if len(sys.argv) > 4 else "6"
rng = random.Random(20261004)
t0 = time.time(); runs = 0
while time.time() - t0 < budget:
    logn = rng.randint(lo_n, hi_n)
    k = rng.choice([1, 2, 2, 3, 3, 4])
    if logn < 2 * k or logn > 11 * k:
        continue
    rad = parts(logn, k, rng=rng)
    env = dict(os.environ, STARKHIP_NTT_RADICES=",".join(map(str, rad)), STARKHIP_TILE_LOG=str(rng.choice([9, 10, 11])),
               STARKHIP_TILE_LOG_BIG=str(rng.choice([10, 11, 12])), STARKHIP_XCD_SWZ=str(rng.choice([0, 1, 2])),
               # the narrow-launch form: never / the default threshold / every launch
               STARKHIP_NTT_NARROW_TILES=rng.choice(["0", "0", "256", "100000000"]))
    out = subprocess.run([sys.executable, "-c", CHILD, str(rng.randrange(1 << 30)), str(logn), per], env=env, capture_output=True, text=True, timeout=300)
    runs += 1
    if out.returncode != 0 or "ok" not in out.stdout:
        print("FAILED", logn, rad, {k: v for k, v in env.items() if k.startswith("STARKHIP")}, out.stdout[-500:], out.stderr[-1500:])
        sys.exit(1)
    if runs % 10 == 0:
        print("%d variants ok, %.0f s" % (runs, time.time() - t0), flush=True)
print("stress: %d plan variants, all outputs equal to the oracle" % runs)
