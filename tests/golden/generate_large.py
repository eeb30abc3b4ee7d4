#!/usr/bin/env python3
"""Digests of NTTs too long for the pure-Python reference (2^17 dense, 2^19 and 2^21 ... 2^24 points; the reference needs ~25 min and
several GB for 2^24), computed with the C oracle (oracle/oracle.c: the reference's recursive
radix-2 algorithm, fft.py:287-331).  The oracle itself is pinned to the live reference up to 2^20 points by
tests/golden/generate.py + tests/test_coracle.py, so these digests are reference-independent pins for the sizes the
reference cannot reach (BASELINE configs[3]).

    python3 tests/golden/generate_large.py          # writes tests/golden/ntt_large.json  (about 4 min, 4 GB)
    python3 tests/golden/generate_large.py --missing  # keeps the cases already in the file, adds the sizes it lacks
    python3 tests/golden/generate_large.py --fri      # writes tests/golden/fri_large.json (about 3 min, 3 GB): below
    python3 tests/golden/generate_large.py --merkle   # writes tests/golden/merkle_large.json (about 1 min, 2 GB)
    python3 tests/golden/generate_large.py --stark [12 14 16]  # writes / extends tests/golden/stark_large.json: see stark_main

2^17 is the domain of config 3's commit, whose default plan (9, 8) the reference-generated fixtures reach only through the
sparse first pass (a dense vector of that length: here).  2^19 is the domain of config 5's proofs (plan (9, 10)), 2^21 the first three-pass plan, 2^23 the domain of the metric's
2^20-step FRI commit (the plan with the 256 MiB row table): each of these plan shapes gets its own digest.

Input: x_i = BLAKE2s(seed_le64 || i_le64) mod p with seed 0x5eed; w = 7^((p-1)/n).
Recorded: SHA-256 of the forward transform's wire bytes, of the inverse transform of the INPUT (inv(x), not the round
trip), and the first two output elements of each.

--fri: the FRI commits bench.py times (the metric's second half, "FRI-commit ms for 2^20 trace", and the 2^14 / 2^16-step ones
beside it): coefficients c_i = BLAKE2s(seed_le64 || i_le64) mod p, seed 0xF51, i < steps (a degree < steps polynomial, as
bench.py:extras fills it), 8x extension, w = 7^((p-1)/(8 steps)), maxdeg_plus_1 = steps, exclude_multiples_of = 8, 40 samples:
SHA-256 of the flat proof oracle/oracle.c:fri_rec writes (the reference's commit loop, fri.py:189-266, with its iNTT -> NTT per
round and its Lagrange fold), its length, the first round's root2 and the final layer's first value.  The reference itself cannot
reach these sizes (2^14 steps take it 20 s, 2^20 would take hours); the C oracle is pinned to it on the 2^14-step MiMC commit and
seven smaller ones (tests/golden/fri.json, tests/test_coracle.py)."""
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


LOGNS = (17, 19, 21, 22, 23, 24, 25, 26)  # 25, 26: four-pass plans (1 and 2 GiB vectors)
FRI_LOGSTEPS = (14, 16, 18, 20, 22)  # 22: the largest commit the index sampling admits (domain 2^25, utils.py:69)
FRI_SEED = 0xF51


def fri_main():
    out = os.path.join(HERE, "fri_large.json")
    cases = []
    if "--missing" in sys.argv[1:] and os.path.exists(out):
        with open(out) as fh:
            cases = json.load(fh)["cases"]
    have = {c["logsteps"] for c in cases}
    for logsteps in FRI_LOGSTEPS:
        if logsteps in have:
            continue
        steps, ext = 1 << logsteps, 8
        n = steps * ext
        t0 = time.time()
        coeffs = b"".join((int.from_bytes(hashlib.blake2s(struct.pack("<QQ", FRI_SEED, i)).digest(), "big") % P).to_bytes(32, "big")
                          for i in range(steps))
        g2 = pow(7, (P - 1) // n, P)
        flat = coracle.fri_prove_flat(coeffs, g2, steps, ext, 40, n=n)
        case = {"logsteps": logsteps, "steps": steps, "ext": ext, "domain": n, "seed": FRI_SEED, "samples": 40,
                "exclude_multiples_of": ext, "maxdeg_plus_1": steps, "w": "%064x" % g2,
                "coeffs_sha256": hashlib.sha256(coeffs).hexdigest(), "proof_bytes": len(flat),
                "proof_sha256": hashlib.sha256(flat).hexdigest(), "first_root2": flat[:32].hex(),
                "final_layer_first": flat[len(flat) - 32 * 128:len(flat) - 32 * 127].hex(),
                "oracle_seconds": round(time.time() - t0, 1)}
        print(case, flush=True)
        cases.append(case)
    cases.sort(key=lambda c: c["logsteps"])
    with open(out, "w") as fh:
        json.dump({"generator": "tests/golden/generate_large.py --fri (oracle/oracle.c:fri_rec, pinned to the reference by fri.json)",
                   "cases": cases}, fh, indent=1)


STARK_LOGSTEPS = (12, 14, 16)


def stark_main():
    """--stark [LOGSTEPS ...]: whole STARK proofs of config 5's unit 0 (the reference's MiMC formulation, width 2, step polynomials
    [X1, X1 + X2^3], inputs [42, 3], 8x extension, 80 spot checks) at sizes far beyond the reference's own reach, written by the
    COEFFICIENT-FORM prover of oracle/pyoracle.py (the reference's construction, stark.py:27-279: schoolbook polynomial products and
    divisions, quadratic: 40 s for 2^12 steps, 9 min for 2^14, about 2.5 h for 2^16 = config 5's own size), which test_oracle_golden.py
    pins to the live reference on nine smaller cases.  The device prover evaluates the same polynomials on the domain instead; the GPU
    suite compares the flat proof bytes (tests/golden/stark_large.json)."""
    from oracle import pyoracle as po
    out = os.path.join(HERE, "stark_large.json")
    cases = []
    if os.path.exists(out):
        with open(out) as fh:
            cases = json.load(fh)["cases"]
    have = {c["logsteps"] for c in cases}
    want = [int(a) for a in sys.argv[1:] if a.isdigit()] or list(STARK_LOGSTEPS)
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    for logsteps in want:
        if logsteps in have:
            continue
        steps, ext, inputs = 1 << logsteps, 8, [42, 3]
        t0 = time.time()
        w = po.get_computational_trace(inputs, steps, sp)
        proof = po.mk_stark_proof(w, inputs, sp, steps, ext)
        flat = po.stark_flat(proof)
        case = {"logsteps": logsteps, "steps": steps, "ext": ext, "width": 2, "inputs": inputs, "unit": 0,
                "step_polys": [[[list(k), v] for k, v in sorted(d.items())] for d in sp], "samples": 80,
                "outputs": ["%064x" % col[-1] for col in w], "m_root": proof[0].hex(), "l_root": proof[1].hex(),
                "proof_bytes": len(flat), "proof_sha256": hashlib.sha256(flat).hexdigest(), "oracle_seconds": round(time.time() - t0, 1)}
        print(case, flush=True)
        # re-read: several sizes may be generated by concurrent processes
        if os.path.exists(out):
            with open(out) as fh:
                cases = json.load(fh)["cases"]
        cases = [c for c in cases if c["logsteps"] != logsteps] + [case]
        cases.sort(key=lambda c: c["logsteps"])
        with open(out, "w") as fh:
            json.dump({"generator": "tests/golden/generate_large.py --stark (oracle/pyoracle.py:mk_stark_proof, the coefficient-form prover "
                                    "pinned to the reference by stark.json)", "cases": cases}, fh, indent=1)


def merkle_main():
    """--merkle: the Merkle commitment bench.py times (2^24 leaves x_i = BLAKE2s(seed_le64 || i_le64) mod p, seed 7, and the 2^20-leaf
    one of its --quick mode) hashed by oracle/oracle.c:or_merkelize (merkle_tree.py:36-56 with permute4): the root, three interior
    nodes and the SHA-256 of the whole 2n x 32-byte node array (slot 0 = zeros, as sh_merkelize writes it)."""
    out = os.path.join(HERE, "merkle_large.json")
    cases = []
    if "--missing" in sys.argv[1:] and os.path.exists(out):
        with open(out) as fh:
            cases = json.load(fh)["cases"]
    have = {c["logn"] for c in cases}
    for logn in (20, 24, 26):  # 26: 2 GiB of leaves, 4 GiB of nodes (about 2 min)
        if logn in have:
            continue
        n = 1 << logn
        t0 = time.time()
        leaves = b"".join((int.from_bytes(hashlib.blake2s(struct.pack("<QQ", 7, i)).digest(), "big") % P).to_bytes(32, "big")
                          for i in range(n))
        nodes = coracle.merkelize_bytes(leaves)
        case = {"logn": logn, "n": n, "seed": 7, "root": nodes[32:64].hex(), "node_2": nodes[64:96].hex(), "node_n_minus_1": nodes[32 * (n - 1):32 * n].hex(),
                "node_n": nodes[32 * n:32 * n + 32].hex(), "nodes_sha256": hashlib.sha256(bytes(32) + nodes[32:]).hexdigest(),
                "oracle_seconds": round(time.time() - t0, 1)}
        print(case, flush=True)
        cases.append(case)
        del leaves, nodes
    cases.sort(key=lambda c: c["logn"])
    with open(out, "w") as fh:
        json.dump({"generator": "tests/golden/generate_large.py --merkle (oracle/oracle.c:or_merkelize, pinned to the reference by merkle.json)",
                   "cases": cases}, fh, indent=1)


def main():
    if "--merkle" in sys.argv[1:]:
        return merkle_main()
    if "--fri" in sys.argv[1:]:
        return fri_main()
    if "--stark" in sys.argv[1:]:
        return stark_main()
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
