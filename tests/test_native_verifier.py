"""The C-ABI verifiers (csrc/verify.hip: sh_fri_verify, sh_stark_verify -- host code, no GPU) on proofs whose bytes are pinned to the
live reference (tests/golden/fri.json, stark.json: regenerated here with the oracles, digests checked by test_coracle.py /
test_oracle_golden.py): accepted, byte flips and wrong public values rejected, and the decision equals the Python mirror of the
reference's verifiers (starks_amd/fri.py:verify_low_degree_proof, starks_amd/stark.py:STARK.verify_proof).  CPU only."""
import hashlib
import random

import pytest

from oracle import coracle as co
from oracle import pyoracle as po
from conftest import load_golden
from test_coracle import _fri_coeffs, wire, h2i

P = po.MIMC_P


def _decide(fn):
    try:
        return bool(fn())
    except AssertionError:
        return False


@pytest.mark.parametrize("rec", load_golden("fri.json"), ids=lambda r: r["name"])
def test_fri_verifier_native_vs_python(rec):
    from starks_amd import _lib, fri
    w = h2i(rec["w"])
    n = _lib.order_of_root(w)
    flat = co.fri_prove_flat(wire(_fri_coeffs(rec)), w, rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"])
    assert hashlib.sha256(flat).hexdigest() == rec["flat_sha"]
    root = bytes.fromhex(rec["eval_root"])
    args = (n, w, rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"])
    assert fri.verify_flat(flat, root, *args) is True
    nested = fri.unpack_proof(flat, n, rec["maxdeg_plus_1"], rec["samples"])
    assert fri.pack_proof(nested) == flat
    from starks_amd import IntegersModP
    assert fri.SmoothSubgroupFRI(IntegersModP(P)).verify_proximity_proof_native(
        nested, root, w, rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"]) is True

    def python_says(f, r):
        return _decide(lambda: fri.verify_low_degree_proof(fri.unpack_proof(f, n, rec["maxdeg_plus_1"], rec["samples"]), r, w,
                                                            rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"]))

    small = len(flat) <= 300000  # the Python verifier takes seconds on the large proofs: compared on the small ones
    if small:
        assert python_says(flat, root) is True
    rng = random.Random(rec["flat_len"])
    for _ in range(12 if small else 4):
        bad = bytearray(flat)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        got = _decide(lambda: fri.verify_flat(bytes(bad), root, *args))
        assert got is False
        if small and rec["samples"] == 40:  # (with samples != 40 the Python mirror reads only the first `samples` rows of later rounds)
            assert python_says(bytes(bad), root) is False
    # a different commitment, a wrong degree bound, a truncated proof
    assert _decide(lambda: fri.verify_flat(flat, hashlib.sha256(root).digest(), *args)) is False
    if rec["maxdeg_plus_1"] > 16:
        with pytest.raises(Exception):
            fri.verify_flat(flat, root, n, w, rec["maxdeg_plus_1"] // 4, rec["exclude_multiples_of"], rec["samples"])
    with pytest.raises(Exception):
        fri.verify_flat(flat[:-32], root, *args)
    # a polynomial of too high a degree: an honest prover run with a bound the polynomial does not meet is rejected (the folds stay
    # consistent, the final layer's degree check fails)
    if rec["name"] == "fri_mimc_2_7":
        coeffs = [pow(3, i, P) for i in range(200)]  # degree 199 against maxdeg_plus_1 = 128 on the 1024-point domain
        cheat = co.fri_prove_flat(wire(coeffs), w, 128, 8, 40)
        croot = co.merkelize_bytes(co.fft_bytes(wire(coeffs), n, w))[32:64]
        assert _decide(lambda: fri.verify_flat(cheat, croot, n, w, 128, 8, 40)) is False
        assert python_says(cheat, croot) is False
        honest = co.fri_prove_flat(wire(coeffs[:128]), w, 128, 8, 40)
        hroot = co.merkelize_bytes(co.fft_bytes(wire(coeffs[:128]), n, w))[32:64]
        assert fri.verify_flat(honest, hroot, n, w, 128, 8, 40) is True


class _Poly(object):
    def __init__(self, d):
        self.coefficients = d


@pytest.mark.parametrize("c", [c for c in load_golden("stark.json") if c["steps"] <= 32], ids=lambda c: c["name"])
def test_stark_verifier_native_vs_python(c):
    from starks_amd import stark
    sp = [{tuple(k): v for k, v in d} for d in c["step_polys"]]
    w = po.get_computational_trace(c["inputs"], c["steps"], sp)
    proof = po.mk_stark_proof(w, c["inputs"], sp, c["steps"], c["ext"])
    flat = po.stark_flat(proof)
    assert hashlib.sha256(flat).hexdigest() == c["flat_sha"]
    polys = [_Poly(d) for d in sp]
    inb = wire(c["inputs"])
    outs = [col[-1] for col in w]
    args = (c["steps"], c["ext"], c["width"], polys)
    assert stark.verify_flat(flat, inb, wire(outs), *args) is True
    assert po.verify_stark_proof(proof, outs, c["inputs"], sp, c["steps"], c["ext"])
    # the nested proof packs back into the flat bytes; the class method that verifies through the C entry takes the nested form
    assert stark.pack_proof(proof) == flat
    from starks_amd import IntegersModP
    S = stark.STARK(IntegersModP(P), c["steps"], c["ext"], c["width"], polys)
    boundary = [(0, j, v) for j, v in enumerate(c["inputs"])]
    assert S.verify_proof_native(proof, w, boundary) is True
    tampered = [proof[0], proof[1], [list(b) for b in proof[2]], proof[3]]
    tampered[2][4][1] = bytes(len(tampered[2][4][1]))
    assert _decide(lambda: S.verify_proof_native(tampered, w, boundary)) is False
    rng = random.Random(c["flat_len"])
    for _ in range(10):
        bad = bytearray(flat)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        assert _decide(lambda: stark.verify_flat(bytes(bad), inb, wire(outs), *args)) is False
    # wrong public values: the boundary check fails (stark.py:373) -- in both verifiers
    wrong = list(outs)
    wrong[-1] = (wrong[-1] + 1) % P
    assert _decide(lambda: stark.verify_flat(flat, inb, wire(wrong), *args)) is False
    assert _decide(lambda: po.verify_stark_proof(proof, wrong, c["inputs"], sp, c["steps"], c["ext"])) is False
    wrong_in = [(c["inputs"][0] + 1) % P] + list(c["inputs"][1:])
    assert _decide(lambda: stark.verify_flat(flat, wire(wrong_in), wire(outs), *args)) is False
    # another step polynomial: the transition check fails
    other = [dict(d) for d in sp]
    k0 = sorted(other[-1])[0]
    other[-1][k0] = (other[-1][k0] + 1) % P
    assert _decide(lambda: stark.verify_flat(flat, inb, wire(outs), c["steps"], c["ext"], c["width"], [_Poly(d) for d in other])) is False


def test_verifiers_survive_hostile_arguments(tmp_path):
    """The C entries take lengths and counts from the caller: whatever they are (zero, 2^32 - 1 samples, a domain of 2^63 points,
    products that wrap around 64 bits, term counts of 2^31, buffers shorter than the counts imply) the answer is an error code,
    at once -- no read past the buffer, no allocation sized by an unchecked count, no exception across the C ABI.  (Round 5's
    first fuzz run hung in the index sampling of a 12 kB proof declared to hold 2^32 - 1 samples.)  Run in a child process so
    that a crash or a hang fails the test instead of the suite."""
    import subprocess
    import sys
    import textwrap
    from conftest import ROOT
    code = textwrap.dedent("""
        import sys, random, ctypes
        sys.path.insert(0, %r)
        from starks_amd import _lib
        L = _lib.lib()
        rng = random.Random(1)
        P = _lib.MIMC_P
        seen = set()
        for it in range(3000):
            n = rng.choice([0, 1, 2, 3, 4, 8, 16, 64, 256, 1 << 10, 1 << 20, 1 << 25, 1 << 32, 1 << 40, 1 << 63, (1 << 64) - 1, rng.randrange(1 << 20)])
            md = rng.choice([0, 1, 4, 16, 17, 64, 256, 1 << 20, 1 << 62, (1 << 64) - 1, rng.randrange(1, 1 << 12)])
            ex = rng.choice([0, 1, 2, 3, 8, 16, 255, (1 << 32) - 1])
            sm = rng.choice([0, 1, 5, 40, 80, 1000, (1 << 32) - 1])
            plen = rng.choice([0, 1, 31, 32, 33, 64, 1000, 4096, rng.randrange(1 << 16)])
            buf = bytes(rng.randrange(256) for _ in range(min(plen, 512))) + bytes(max(0, plen - 512))
            ok_n = n and n & (n - 1) == 0 and n <= 1 << 32
            w = (pow(7, (P - 1) // n, P) if ok_n and rng.random() < 0.7 else rng.randrange(P)).to_bytes(32, "big")
            seen.add(L.sh_fri_verify(buf, len(buf), bytes(32), n, w, md, ex, sm))
        for it in range(3000):
            steps = rng.choice([0, 1, 2, 3, 4, 8, 64, 1 << 16, 1 << 20, 1 << 24, 1 << 40, 1 << 63, (1 << 64) - 1])
            ext = rng.choice([0, 1, 2, 3, 4, 8, 16, 1 << 20, 1 << 24, 1 << 31, (1 << 32) - 1])
            width = rng.choice([0, 1, 2, 3, 9, 10, 100, (1 << 32) - 1])
            sm = rng.choice([0, 1, 80, 1000, (1 << 32) - 1])
            plen = rng.choice([0, 1, 63, 64, 65, 1000, rng.randrange(1 << 16)])
            buf = bytes(rng.randrange(256) for _ in range(min(plen, 512))) + bytes(max(0, plen - 512))
            wd = min(width, 12)
            nterms = [rng.choice([0, 1, 2, 5, 1 << 31, (1 << 32) - 1]) for _ in range(max(1, wd))]
            tot = min(sum(t for t in nterms if t < 100), 64)
            coefs = bytes(32 * max(1, tot))
            exps = bytes(rng.randrange(8) for _ in range(max(1, tot) * (wd + 1)))
            counts = (ctypes.c_uint32 * len(nterms))(*nterms)
            seen.add(L.sh_stark_verify(buf, len(buf), bytes(32 * max(1, wd)), bytes(32 * max(1, wd)), steps, ext, width, coefs, exps, counts, sm))
        assert 0 not in seen, seen
        print("codes", sorted(seen))
    """ % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "codes" in out.stdout
