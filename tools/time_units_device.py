#!/usr/bin/env python3
"""wall time of batch.prove_stark_units_device (the many-proof driver of the drop-in API: device-generated witnesses, proofs
downloaded and digested on the host) for 512 units of 2^16 steps, chunk 64 / 128"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starks_amd import batch
batch.prove_stark_units_device(0, 64, 1 << 16, 8, chunk=32)  # warm-up: tables, workspaces
for chunk in (64, 64, 128, 128):  # the first run of a size grows the workspaces
    t0 = time.perf_counter()
    digs, _ = batch.prove_stark_units_device(0, 512, 1 << 16, 8, chunk=chunk)
    dt = time.perf_counter() - t0
    print("chunk %3d: %.3f s for 512 proofs (%.0f proofs/s incl. download + SHA-256 on the host), %d distinct" % (chunk, dt, 512 / dt, len(set(digs))), flush=True)
