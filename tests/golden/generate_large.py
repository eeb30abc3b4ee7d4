#!/usr/bin/env python3
"""Digests of NTTs too long for the pure-Python reference (2^19 and 2^21 ... 2^24 points; the reference needs ~25 min and
several GB for 2^24), computed with the C oracle (oracle/oracle.c: the reference's recursive
radix-2 algorithm, fft.py:287-331).  The oracle itself is pinned to the live reference up to 2^20 points by
tests/golden/generate.py + tests/test_coracle.py, so these digests are reference-independent pins for the sizes the
reference cannot reach (BASELINE configs[3]).

    python3 tests/golden/generate_large.py          # writes tests/golden/ntt_large.json  (about 4 min, 4 GB)
    python3 tests/golden/generate_large.py --missing  # keeps the cases already in the file, adds the sizes it lacks

2^19 is the domain of config 5's proofs (plan (9, 10)), 2^21 the first three-pass plan, 2^23 the domain of the metric's
2^20-step FRI commit (the plan with the 256 MiB row table): each of these plan shapes gets its own digest.

Input: x_i = BLAKE2s(seed_le64 || i_le64) mod p with seed 0x5eed; w = 7^((p-1)/n).
Recorded: SHA-256 of the forward transform's wire bytes, of the inverse transform of the INPUT (inv(x), not the round
trip), and the first two output elements of each."""
import hashlib
import json
import os
import struct
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import coracle  # noqa: E402

P = 2**256 - 2**32 * 351 + 1
SEED = 0x5eed


LOGNS = (19, 21, 22, 23, 24)


def main():
    out = os.path.join(HERE, "ntt_large.json")
    cases = []
    if "--missing" in sys.argv[1:] and os.path.exists(out):
        with open(out) as fh:
            cases = json.load(fh)["cases"]
    have = {c["logn"] for c in cases}
    for logn in LOGNS:
        if logn in have:
            continue
        n = 1 << logn
        t0 = time.time()
        raw = b"".join(hashlib.blake2s(struct.pack("<QQ", SEED, i)).digest() for i in range(n))
        w = pow(7, (P - 1) // n, P)
        fwd = coracle.fft_bytes(raw, n, w)
        case = {"logn": logn, "n": n, "seed": SEED, "w": "%064x" % w,
                "sha_fwd": hashlib.sha256(fwd).hexdigest(), "fwd_head": fwd[:64].hex(),
                "fwd_tail": fwd[-32:].hex()}
        del fwd
        inv = coracle.fft_bytes(raw, n, w, inverse=True)
        case.update({"sha_inv": hashlib.sha256(inv).hexdigest(), "inv_head": inv[:64].hex()})
        del inv, raw
        case["oracle_seconds"] = round(time.time() - t0, 1)
        print(case, flush=True)
        cases.append(case)
    cases.sort(key=lambda c: c["logn"])
    with open(out, "w") as fh:
        json.dump({"generator": "tests/golden/generate_large.py (oracle/oracle.c, pinned to the reference <= 2^20)",
                   "cases": cases}, fh, indent=1)


if __name__ == "__main__":
    main()
