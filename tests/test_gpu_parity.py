"""GPU parity tests (run on the MI355X box with -m gpu): every call goes through the C ABI of
libstarkhip.so (via the ctypes mirror in starks_amd/) and is compared bit-for-bit with the CPU oracle on
the same inputs and with the fixtures generated from the live reference (tests/golden)."""
import hashlib
import os
import random
import struct

import pytest

from conftest import load_golden, GOLDEN, ROOT

pytestmark = pytest.mark.gpu

P = 2**256 - 2**32 * 351 + 1


def h2i(s):
    return int(s, 16)


def wire(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


def unwire(b):
    return [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]


def seeded(seed, i):
    return int.from_bytes(hashlib.blake2s(struct.pack("<QQ", seed, i)).digest(), "big") % P


def root_of(n):
    return pow(7, (P - 1) // n, P)


@pytest.fixture(scope="module")
def sa():
    import starks_amd
    from starks_amd import _lib, fft, merkle_tree, utils, fri, compression

    class NS:
        pass

    ns = NS()
    ns.pkg, ns.lib, ns.fft, ns.mt, ns.utils, ns.fri, ns.comp = starks_amd, _lib, fft, merkle_tree, utils, fri, compression
    ns.F = starks_amd.IntegersModP(P)
    _lib.ctx()  # fails loudly when the extension or the GPU is missing
    return ns


@pytest.fixture(scope="module")
def oracle():
    from oracle import coracle, pyoracle

    class NS:
        pass

    ns = NS()
    ns.c, ns.py = coracle, pyoracle
    return ns


# ---- NTT ---------------------------------------------------------------------------------------------
def test_ntt_golden_vectors(sa):
    g = load_golden("ntt.json")
    for c in g["cases"]:
        n, n_in, w = c["n"], c["n_in"], h2i(c["w"])
        data = wire(seeded(g["seed"], i) for i in range(n_in))
        fwd = sa.fft.ntt_bytes(data, n, w)
        inv = sa.fft.ntt_bytes(data, n, w, inverse=True)
        assert hashlib.sha256(fwd).hexdigest() == c["sha_fwd"], "fwd n=%d" % n
        assert hashlib.sha256(inv).hexdigest() == c["sha_inv"], "inv n=%d" % n
        if "fwd" in c:
            assert unwire(fwd) == [h2i(v) for v in c["fwd"]]
        # x == invNTT(NTT(x)), zero-padded (test_fft.py:132-149 round trip)
        assert sa.fft.ntt_bytes(fwd, n, w, inverse=True) == data + bytes(32 * (n - n_in))


@pytest.mark.parametrize("logn", list(range(0, 21)))
def test_ntt_every_size_vs_oracle(sa, oracle, logn):
    """All plan shapes (tiny, one, two and three passes; odd and even radices) against the C oracle."""
    n = 1 << logn
    w = root_of(n)
    rng = random.Random(logn)
    if True:  # every size dense (2^17 .. 2^20 as well: the scalar oracle takes 0.2 .. 2 s per transform there)
        vals = [rng.randrange(2**256) for _ in range(n)]  # unreduced inputs allowed (modp.py:33-34)
        vals[0] = 2**256 - 1
        if n > 1:
            vals[-1] = P
        data = wire(vals)
        exp_f = oracle.c.fft_bytes(data, n, w)
        exp_i = oracle.c.fft_bytes(data, n, w, inverse=True)
        assert sa.fft.ntt_bytes(data, n, w) == exp_f
        assert sa.fft.ntt_bytes(data, n, w, inverse=True) == exp_i
    if logn > 16:
        # on top of the dense comparison: round trip + point checks on a structured input (8 values repeated: a sparse spectrum)
        data = bytes(rng.getrandbits(8) for _ in range(32 * 8)) * (n // 8)
        fwd = sa.fft.ntt_bytes(data, n, w)
        assert sa.fft.ntt_bytes(fwd, n, w, inverse=True) == wire(v % P for v in unwire(data))
        xs = unwire(data)
        for k in (0, 1, n // 2, n - 1):  # out[k] = sum_j x_j w^(jk): check k=0 and k=n/2 exactly
            if k == 0:
                assert int.from_bytes(fwd[:32], "big") == sum(xs) % P
            if k == n // 2:
                assert int.from_bytes(fwd[32 * k:32 * k + 32], "big") == (sum(xs[0::2]) - sum(xs[1::2])) % P


def test_ntt_padding_and_batch(sa, oracle):
    n, w = 4096, root_of(4096)
    rng = random.Random(5)
    rows = [[rng.randrange(P) for _ in range(1000)] for _ in range(3)]
    got = sa.fft.ntt_bytes(b"".join(wire(r) for r in rows), n, w, batch=3)
    for b, r in enumerate(rows):
        assert got[b * 32 * n:(b + 1) * 32 * n] == oracle.c.fft_bytes(wire(r), n, w)
    # empty input -> all zeros; n_in > n -> error
    assert sa.fft.ntt_bytes(b"", 16, root_of(16)) == bytes(32 * 16)
    with pytest.raises(ValueError):
        sa.fft.ntt_bytes(bytes(32 * 32), 16, root_of(16))
    # root of the wrong order is refused (fft.py:319-321 would walk a different cycle)
    rc = sa.lib.lib().sh_ntt(sa.lib.ctx(), bytes(64), 2, __import__("ctypes").create_string_buffer(32 * 8), 8,
                            root_of(16).to_bytes(32, "big"), 0)
    assert rc == -2


def test_reference_call_sites_fft(sa):
    """The reference's own tests, through the drop-in API (test_fft.py:115-130, 185-194; Appendix A)."""
    F = sa.F
    from starks_amd.polynomial import polynomials_over
    polys = polynomials_over(F).factory
    g = load_golden("ntt.json")
    w8 = F(7) ** ((P - 1) // 8)
    ev = sa.fft.NonBinaryFFT(F, w8).fft(polys([0, 1, 2, 3]))
    assert len(ev) == 8 and all(isinstance(v, F) for v in ev)
    assert [int(v) for v in ev] == [h2i(v) for v in g["mimc_n8_0123"]["fwd"]]
    assert ev[4] == P - 2  # P(-1) = -2
    back = sa.fft.NonBinaryFFT(F, w8).inv_fft(ev)
    assert back == polys([0, 1, 2, 3]) and len(back.coefficients) == 4  # trailing zeros stripped
    m = g["inv_fft_strip"]
    assert [int(c) for c in sa.fft.NonBinaryFFT(F, F(7) ** ((P - 1) // 16)).inv_fft([h2i(v) for v in m["values"]]).coefficients] == [5, 6, 7]
    prod = sa.fft.mul_polys([F(v) for v in range(4)], [F(v) for v in range(4)], F(7) ** ((P - 1) // 512))
    assert [int(v) for v in prod[:16]] == g["mul_polys_0123"]["first16"]
    assert hashlib.sha256(wire(prod)).hexdigest() == g["mul_polys_0123"]["sha"]
    # mod-31 / order-6 inputs (test_fft.py:98-113) are outside the accelerated field: host recursion, reference KAT
    F31 = sa.pkg.IntegersModP(31)
    assert [int(v) for v in sa.fft.fft_1d(F31, [0, 1, 2, 3], 31, F31(26))] == g["mod31_n6"]["fwd"]
    # mul_polys over another field: host transforms (round 3); n * (a * b), the reference omits the 1/n (fft.py:345)
    assert [int(v) for v in sa.fft.mul_polys([F31(1), F31(2)], [F31(3)], F31(26))] == [(6 * 3) % 31, (6 * 6) % 31, 0, 0, 0, 0]


def test_chained_call_sites_move_bytes_not_objects(sa, oracle):
    """The reference's chain fft -> merkelize -> mk_branch (stark.py:253-263) and inv_fft -> prove_low_degree through the drop-in
    functions: every stage hands the next one a wire-backed sequence (starks_amd/wireseq.py), results equal the oracle's and the
    plain-list route's, and the whole 2^20-point chain stays far below the cost of creating 2^20 Python objects per stage."""
    import time
    from starks_amd.wireseq import WireList, NodeList
    from starks_amd.polynomial import polynomials_over
    F = sa.F
    n = 1 << 12
    w = root_of(n)
    coeffs = [seeded(31, i) for i in range(n // 8)]
    solver = sa.fft.NonBinaryFFT(F, F(w))
    ev = solver.fft(polynomials_over(F).factory(coeffs))
    assert isinstance(ev, WireList) and len(ev) == n and isinstance(ev[5], F)
    want = oracle.c.fft(coeffs, n, w)
    assert ev == want and [int(v) for v in ev[:4]] == want[:4] and ev[-1] == want[-1]
    tree = sa.mt.merkelize(ev)
    assert isinstance(tree, NodeList) and len(tree) == 2 * n and tree[0] == b""
    assert tree == oracle.c.merkelize(want) and tree == sa.mt.merkelize(list(ev)) and tree == sa.mt.merkelize(want)
    br = sa.mt.mk_branch(tree, 77)
    assert br == oracle.c.mk_branch_bytes(oracle.c.merkelize_bytes(wire(want)), 77)
    assert sa.mt.verify_branch(tree[1], 77, br, output_as_int=True) == want[77]
    back = solver.inv_fft(ev)  # Poly whose coefficients are still bytes; the padding zeros are stripped on the bytes
    assert isinstance(back.coefficients, WireList) and back.coefficients == coeffs and len(back) == n // 8
    proof = sa.fri.prove_low_degree(back, F(w), n // 8, exclude_multiples_of=8)
    assert proof == sa.fri.prove_low_degree(coeffs, F(w), n // 8, exclude_multiples_of=8)
    assert sa.utils.get_power_cycle(F(w), F)[:3] == [1, w, w * w % P] and len(sa.utils.get_power_cycle(F(w), F)) == n
    assert sa.fft.mul_polys(ev[:4], [1, 2], F(root_of(8))) == sa.fft.mul_polys(list(ev[:4]), [1, 2], F(root_of(8)))
    packed = sa.mt.merkelize_polynomial_evaluations(1, [ev, ev[::-1]])
    assert packed == sa.mt.merkelize_polynomial_evaluations(1, [want, want[::-1]]) and len(packed[n]) == 64
    # at size: 2^20 points, input already wire-backed (the output of an earlier stage)
    n = 1 << 20
    g = sa.utils.get_power_cycle(F(root_of(n)), F)
    t0 = time.perf_counter()
    ev = sa.fft.fft_1d(F, g, P, root_of(n))
    t1 = time.perf_counter()
    tree = sa.mt.merkelize(ev)
    t2 = time.perf_counter()
    br = sa.mt.mk_branch(tree, 12345)
    assert len(br) == 21 and sa.mt.verify_branch(tree[1], 12345, br) == bytes(ev.wire()[32 * 12345:32 * 12346])
    # NTT of (1, w, w^2, ...) = n at index n - 1, zero elsewhere
    assert ev[n - 1] == n and ev[0] == 0 and ev[123] == 0
    assert t1 - t0 < 0.25 and t2 - t1 < 0.25, (t1 - t0, t2 - t1)  # (measured 20-30 ms each; 2^20 objects alone cost 0.5 s)


def test_power_cycle(sa):
    g = load_golden("utils.json")
    F = sa.F
    assert [v.to_bytes().hex() for v in sa.utils.get_power_cycle(F(root_of(8)), F)] == g["power_cycle_w8"]
    assert hashlib.sha256(wire(sa.utils.get_power_cycle(F(root_of(64)), F))).hexdigest() == g["power_cycle_w64_sha"]
    n = 1 << 18
    cyc = sa.utils.get_power_cycle(F(root_of(n)), F)
    assert len(cyc) == n and cyc[1] == root_of(n) and int(cyc[-1]) * root_of(n) % P == 1
    assert int(cyc[12345]) == pow(root_of(n), 12345, P)


# ---- Merkle -------------------------------------------------------------------------------------------
def test_merkle_golden(sa):
    g = load_golden("merkle.json")
    t = sa.mt.merkelize([x.to_bytes(32, "big") for x in range(128)])
    assert t[1].hex() == g["range128"]["root"] and len(t) == 256 and t[0] == b""
    assert hashlib.sha256(b"".join(t)).hexdigest() == g["range128"]["tree_sha"]
    b = sa.mt.mk_branch(t, 59)
    assert [x.hex() for x in b] == g["range128"]["branch59"] and len(b) == 8
    assert sa.mt.verify_branch(t[1], 59, b, output_as_int=True) == 59  # test_merkle_tree.py:16-22
    t = sa.mt.merkelize([x.to_bytes(32, "big") for x in range(256)])
    assert t[1].hex() == g["range256"]["root"] and len(sa.mt.mk_branch(t, 59)) == 9
    t = sa.mt.merkelize([sa.F(1), sa.F(2), sa.F(3), sa.F(4)])
    assert [x.hex() for x in t] == g["f1234"]["tree"]
    assert sa.mt.merkelize([5, 6, 7, 8])[1].hex() == g["mixed5678_root"]
    for c in g["seeded"]:
        vals = [seeded(c["seed"], i) for i in range(c["n"])]
        t = sa.mt.merkelize(vals)
        assert t[1].hex() == c["root"]
        assert hashlib.sha256(b"".join(t)).hexdigest() == c["tree_sha"]
        for i, br in c["branches"].items():
            assert [x.hex() for x in sa.mt.mk_branch(t, int(i))] == br


@pytest.mark.parametrize("logn", [2, 3, 5, 9, 10, 11, 12, 13, 17, 20, 21, 22])
def test_merkle_sizes_vs_oracle(sa, oracle, logn):
    """Every node of the tree against the C oracle.  From 2^21 leaves up the leaf and mid kernels are launched in their WIDE form
    (asm BLAKE2s rounds, csrc/blake2s.cuh; below that the C++ rounds): both forms are compared with the oracle's hashes."""
    n = 1 << logn
    rng = random.Random(logn)
    leaves = bytes(rng.getrandbits(8) for _ in range(32 * min(n, 4096))) * max(1, n // 4096)
    assert sa.mt.merkelize_bytes(leaves) == oracle.c.merkelize_bytes(leaves)


# ---- FRI ----------------------------------------------------------------------------------------------
def test_fold_golden(sa, oracle):
    import ctypes
    for c in load_golden("fold.json"):
        n = c["n"]
        values = wire(seeded(c["seed"], i) for i in range(n))
        out = ctypes.create_string_buffer(8 * n)
        rc = sa.lib.lib().sh_fri_fold(sa.lib.ctx(), values, n, h2i(c["w"]).to_bytes(32, "big"),
                                     bytes.fromhex(c["special_x_bytes"]), out)
        assert rc == 0
        assert hashlib.sha256(out.raw).hexdigest() == c["column_sha"]
    # bigger, against the C oracle's Lagrange-interpolation fold
    n = 1 << 14
    vals = [seeded(3, i) for i in range(n)]
    sx = hashlib.blake2s(b"q").digest()
    out = ctypes.create_string_buffer(8 * n)
    assert sa.lib.lib().sh_fri_fold(sa.lib.ctx(), wire(vals), n, root_of(n).to_bytes(32, "big"), sx, out) == 0
    assert unwire(out.raw) == oracle.c.fold(vals, root_of(n), sx)


def _fri_coeffs(rec, oracle):
    d = rec["coeffs"]
    if d.startswith("(i**7)^42"):
        return [(i**7) ^ 42 for i in range(rec["n_coeffs"])]
    if d == "i, i<256":
        return list(range(256))
    if d == "i+1, i<16":
        return list(range(1, 17))
    k = int(d.split("2^")[1].rstrip("))"))
    coeffs = oracle.c.fft(oracle.py.mimc_trace(3, 2**k), 2**k, pow(h2i(rec["w"]), 8, P), inverse=True)
    return oracle.py.strip_trailing_zeros(coeffs)


@pytest.mark.parametrize("rec", load_golden("fri.json"), ids=lambda r: r["name"])
def test_fri_proofs_golden(sa, oracle, rec):
    """Proof bytes == the reference's, for every fixture incl. config C3 (2^14-step MiMC trace, N = 2^17)."""
    coeffs = _fri_coeffs(rec, oracle)
    w = h2i(rec["w"])
    n = sa.lib.order_of_root(w)
    flat = sa.fri.prove_flat(wire(coeffs), n, w, rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"])
    assert len(flat) == rec["flat_len"]
    assert hashlib.sha256(flat).hexdigest() == rec["flat_sha"]
    proof = sa.fri.prove_low_degree(coeffs, sa.F(w), rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"])
    assert len(proof) == rec["len_proof"]
    assert [r[0].hex() for r in proof[:-1]] == [r["root_m2"] for r in rec["rounds"]]
    assert [x.hex() for x in proof[-1]] == rec["final_values"]
    assert [[len(b) for b in proof[r][1][0]] for r in range(len(proof) - 1)] == rec["branch_lens"]
    pb = sa.comp.proof_bytes(sa.comp.compress_fri(proof))  # the reference's "proof bytes" (compression.py)
    assert (len(pb), hashlib.sha256(pb).hexdigest()) == (rec["proof_bytes_len"], rec["proof_bytes_sha"])
    path = os.path.join(GOLDEN, rec["name"] + ".proof.bin")
    if os.path.exists(path):
        assert open(path, "rb").read() == pb
    if rec["samples"] == 40:  # prove -> verify round trip (test_fri.py:136-182)
        assert sa.fri.SmoothSubgroupFRI(sa.F).verify_proximity_proof(
            proof, bytes.fromhex(rec["eval_root"]), sa.F(w), rec["maxdeg_plus_1"], rec["exclude_multiples_of"])


def test_fri_reference_test_shapes(sa):
    """test_fri.py:105-134 (commented in the reference): 4 proof items, 40 branch sets, 8 final values."""
    F = sa.F
    from starks_amd.polynomial import polynomials_over
    poly = polynomials_over(F).factory([F((i**7) ^ 42) for i in range(512)])
    proof = sa.fri.SmoothSubgroupFRI(F).generate_proximity_proof(poly, F(7) ** ((P - 1) // 512), 512)
    assert len(proof) == 4
    for i, rp in enumerate(proof):
        if i < 3:
            assert len(rp) == 2 and len(rp[1]) == 40
        else:
            assert len(rp) == 8


def test_fri_batch_and_oracle(sa, oracle):
    """Batched proofs of independent traces (config C5 in miniature) == one-at-a-time C oracle proofs."""
    steps, ext = 256, 8
    n = steps * ext
    g2 = root_of(n)
    coeffs = []
    for j in range(5):
        c = oracle.c.fft(oracle.py.mimc_trace(3 + j, steps), steps, pow(g2, ext, P), inverse=True)
        coeffs.append(wire(c))
    flat = sa.fri.prove_flat(b"".join(coeffs), n, g2, steps, ext, 40, batch=5)
    plen = sa.fri.proof_len(n, steps, 40)
    for j in range(5):
        assert flat[j * plen:(j + 1) * plen] == oracle.c.fri_prove_flat(coeffs[j], g2, steps, ext, 40)


def test_fri_invalid_arguments(sa):
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    out = ctypes.create_string_buffer(1 << 20)
    w = root_of(64).to_bytes(32, "big")
    assert L.sh_fri_prove(ctx, bytes(32 * 65), 65, 64, w, 64, 0, 40, 1, out, 1 << 20) == -1  # n_coeffs > n
    assert L.sh_fri_prove(ctx, bytes(32 * 64), 64, 64, w, 64, 0, 40, 1, out, 16) == -5        # buffer too small
    assert L.sh_fri_prove(ctx, bytes(32 * 64), 64, 64, root_of(128).to_bytes(32, "big"), 64, 0, 40, 1, out, 1 << 20) == -2


# ---- LDE ----------------------------------------------------------------------------------------------
def test_lde_golden(sa, oracle):
    for c in load_golden("lde.json"):
        tr = oracle.py.mimc_trace(c["trace_t0"], c["steps"])
        ext = sa.fft.low_degree_extension(sa.F, [tr], c["ext"], sa.F(h2i(c["g2"])))[0]
        assert hashlib.sha256(wire(ext)).hexdigest() == c["lde_sha"]
        assert [int(v) for v in ext[::8]] == tr
    # several columns at once
    steps = 512
    g2 = root_of(steps * 8)
    cols = [oracle.py.mimc_trace(3 + j, steps) for j in range(3)]
    got = sa.fft.low_degree_extension(sa.F, cols, 8, sa.F(g2))
    for j in range(3):
        assert wire(got[j]) == oracle.c.lde_bytes(wire(cols[j]), 8, g2)


# ---- device-resident API (what bench.py times) --------------------------------------------------------------
def test_device_resident_pipeline(sa, oracle):
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << 12
    w = root_of(n)
    d_x, d_y, d_t = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(d_x)) == 0
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(d_y)) == 0
    assert L.sh_dev_alloc(ctx, 64 * n, ctypes.byref(d_t)) == 0
    assert L.sh_dev_fill_seeded(ctx, d_x, n, 0x5eed) == 0
    host = ctypes.create_string_buffer(32 * n)
    assert L.sh_dev_to_wire(ctx, d_x, host, n) == 0
    assert host.raw == wire(seeded(0x5eed, i) for i in range(n))  # same generator as the fixtures
    assert L.sh_dev_ntt(ctx, d_x, d_y, n, 1, w.to_bytes(32, "big"), 0) == 0
    assert L.sh_dev_to_wire(ctx, d_y, host, n) == 0
    assert hashlib.sha256(host.raw).hexdigest() == [c for c in load_golden("ntt.json")["cases"] if c["n"] == n][0]["sha_fwd"]
    assert L.sh_dev_merkelize(ctx, d_y, n, 1, d_t) == 0
    tree = ctypes.create_string_buffer(64 * n)
    assert L.sh_dev_download(ctx, d_t, tree, 64 * n) == 0
    assert tree.raw == oracle.c.merkelize_bytes(host.raw)
    assert L.sh_dev_ntt(ctx, d_y, d_y, n, 1, w.to_bytes(32, "big"), 1) == 0  # in place inverse
    assert L.sh_dev_to_wire(ctx, d_y, host, n) == 0
    assert host.raw == wire(seeded(0x5eed, i) for i in range(n))
    for d in (d_x, d_y, d_t):
        assert L.sh_dev_free(ctx, d) == 0


# ---- BASELINE.json full sizes through size-independent properties ------------------------------------------
def test_full_size_ntt_2_24_roundtrip(sa):
    """Config 4: 2^24-point NTT (512 MiB vector, three radix-256 passes): invNTT(NTT(x)) == x bit for bit."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << 24
    w = root_of(n).to_bytes(32, "big")
    dx, dy = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dx)) == 0 and L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dy)) == 0
    assert L.sh_dev_fill_seeded(ctx, dx, n, 0x5eed) == 0
    assert L.sh_dev_ntt(ctx, dx, dy, n, 1, w, 0) == 0
    a = ctypes.create_string_buffer(32 * 4096)
    assert L.sh_dev_to_wire(ctx, dy, a, 4096) == 0
    fwd_head = a.raw
    assert L.sh_dev_ntt(ctx, dy, dy, n, 1, w, 1) == 0
    x, y = ctypes.create_string_buffer(32 * n), ctypes.create_string_buffer(32 * n)
    assert L.sh_dev_to_wire(ctx, dx, x, n) == 0 and L.sh_dev_to_wire(ctx, dy, y, n) == 0
    assert hashlib.sha256(x.raw).digest() == hashlib.sha256(y.raw).digest()
    assert x.raw[:64] == wire([seeded(0x5eed, 0), seeded(0x5eed, 1)])
    # out[0] = sum of the inputs: check against a host sum over the downloaded vector
    xs = x.raw
    total = 0
    for off in range(0, len(xs), 32 * 65536):
        blk = xs[off:off + 32 * 65536]
        total += sum(int.from_bytes(blk[i:i + 32], "big") for i in range(0, len(blk), 32))
    assert int.from_bytes(fwd_head[:32], "big") == total % P
    assert L.sh_dev_free(ctx, dx) == 0 and L.sh_dev_free(ctx, dy) == 0


def _max_sizes_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("max_sizes", os.path.join(ROOT, "tools", "max_sizes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("logn", [25, 27])
def test_transforms_above_the_bench_size(sa, logn):
    """Maximum sizes (the library takes orders up to 2^32, include/starkhip.h): four-pass plans, 1 and 4 GiB vectors, checked without the
    oracle through the definition -- 1024 outputs exactly from the residue-class sums of the dense input, the inverse on every element,
    and sampled outputs moving by sum_k d_k w^(j p_k) when K coefficients change (tools/max_sizes.py; 2^28 .. 2^32 points and the
    2^28 / 2^30-leaf trees run there: profiles/r05_max_sizes.txt)."""
    ms = _max_sizes_tool()
    assert ms.check_transform(ms.Dev(), logn, random.Random(logn))


def test_merkle_tree_above_the_bench_size(sa):
    """2^26 leaves (4 GiB of nodes): leaf, siblings and root of 64 branches recomputed with hashlib."""
    ms = _max_sizes_tool()
    assert ms.check_merkle(ms.Dev(), 26, random.Random(26), full=False)


def test_per_element_launches_beyond_2_30_threads(sa):
    """A launch holds fewer than 2^32 threads per grid dimension: the per-element kernels (conversions, seeded fill, pointwise product,
    padding copy) spread their blocks over blockIdx.y from 2^30 threads on (kernels.hip:flat_grid_for) -- a 2^32-element fill failed
    with `invalid configuration argument` before.  Here: 2^30 + 2^20 elements (32 GiB), the values either side of the seam."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = (1 << 30) + (1 << 20)
    d = ctypes.c_void_p()
    sa.lib.check(L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(d)), "alloc")
    try:
        sa.lib.check(L.sh_dev_fill_seeded(ctx, d, n, 31), "fill")
        b = ctypes.create_string_buffer(32 * 4)
        for i in (0, (1 << 30) - 2, (1 << 30) + 777, n - 4):
            sa.lib.check(L.sh_dev_to_wire(ctx, ctypes.c_void_p(d.value + 32 * i), b, 4), "to_wire")
            assert b.raw == wire([seeded(31, i + k) for k in range(4)])
    finally:
        L.sh_dev_free(ctx, d)


@pytest.mark.parametrize("logsteps", [16, 20, 22])
def test_full_size_fri_prove_then_verify(sa, logsteps):
    """Config 5's proof size (2^16 steps, N = 2^19), the metric's 2^20-step trace (N = 2^23) and the LARGEST commit the reference's
    index sampling allows (2^22 steps, N = 2^25: utils.py:69 asserts N/4 < 2^24): the GPU proof is
    accepted by the host verifier (fri.py:268-366 semantics), and a corrupted one is rejected."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    steps, ext = 1 << logsteps, 8
    n = steps * ext
    g2 = root_of(n)
    w = g2.to_bytes(32, "big")
    dc, dv, dt = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dc)) == 0
    assert L.sh_dev_fill_seeded(ctx, dc, steps, 0xabc) == 0                    # degree < steps polynomial
    assert L.sh_dev_upload(ctx, bytes(32 * (n - steps)), ctypes.c_void_p(dc.value + 32 * steps), 32 * (n - steps)) == 0
    plen = sa.fri.proof_len(n, steps, 40)
    dp = ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, plen, ctypes.byref(dp)) == 0
    assert L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, 1, dp) == 0
    flat = ctypes.create_string_buffer(plen)
    assert L.sh_dev_download(ctx, dp, flat, plen) == 0
    # the same commit from the `steps` coefficients alone (implicit padding: the sparse first pass of the evaluation)
    short = ctypes.create_string_buffer(plen)
    assert L.sh_dev_fri_prove_coeffs(ctx, dc, steps, n, w, steps, ext, 40, 1, dp) == 0
    assert L.sh_dev_download(ctx, dp, short, plen) == 0
    assert short.raw == flat.raw
    proof = sa.fri.unpack_proof(flat.raw, n, steps, 40)
    # the commitment the verifier starts from: root of the tree over the evaluations
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dv)) == 0 and L.sh_dev_alloc(ctx, 64 * n, ctypes.byref(dt)) == 0
    assert L.sh_dev_ntt(ctx, dc, dv, n, 1, w, 0) == 0 and L.sh_dev_merkelize(ctx, dv, n, 1, dt) == 0
    root = ctypes.create_string_buffer(64)
    assert L.sh_dev_download(ctx, dt, root, 64) == 0
    mroot = root.raw[32:64]
    assert sa.fri.verify_low_degree_proof(proof, mroot, g2, steps, ext)
    assert sa.fri.verify_flat(flat.raw, mroot, n, g2, steps, ext, 40)  # sh_fri_verify: the same decision from the flat bytes
    tampered = bytearray(flat.raw)
    tampered[40000] ^= 1
    with pytest.raises(AssertionError):
        sa.fri.verify_flat(bytes(tampered), mroot, n, g2, steps, ext, 40)
    bad = [[proof[0][0], [[list(b) for b in bs] for bs in proof[0][1]]]] + proof[1:]
    bad[0][1][7][2][0] = bytes(32)
    with pytest.raises(AssertionError):
        sa.fri.verify_low_degree_proof(bad, mroot, g2, steps, ext)
    for d in (dc, dv, dt, dp):
        assert L.sh_dev_free(ctx, d) == 0


@pytest.mark.parametrize("logsteps", [14, 16, 18, 20, 22])
def test_fri_commits_of_the_metric_vs_oracle_fixture(sa, logsteps):
    """The commits bench.py times -- 2^14 (config 3's size), 2^16 (config 5's) and 2^20 steps ("FRI-commit ms for 2^20 trace") -- and
    the largest one the reference's index sampling admits (2^22 steps, domain 2^25: utils.py:69; 5 min of oracle time), on the
    seeded degree < steps polynomial -- byte for byte against the flat proofs oracle/oracle.c:fri_rec wrote (the reference's loop,
    fri.py:189-266, with its per-round iNTT -> NTT and Lagrange fold; tests/golden/fri_large.json, generate_large.py --fri), from the
    coefficients a prover holds and from the zero-padded vector."""
    import ctypes
    c = [c for c in load_golden("fri_large.json")["cases"] if c["logsteps"] == logsteps][0]
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    steps, ext = c["steps"], c["ext"]
    n = steps * ext
    w = root_of(n).to_bytes(32, "big")
    assert w.hex() == c["w"] and n == c["domain"]
    plen = int(L.sh_fri_proof_len(n, c["maxdeg_plus_1"], c["samples"]))
    assert plen == c["proof_bytes"]
    dc, dp = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dc)) == 0 and L.sh_dev_alloc(ctx, plen, ctypes.byref(dp)) == 0
    assert L.sh_dev_fill_seeded(ctx, dc, steps, c["seed"]) == 0
    assert L.sh_dev_upload(ctx, bytes(32 * (n - steps)), ctypes.c_void_p(dc.value + 32 * steps), 32 * (n - steps)) == 0
    head = ctypes.create_string_buffer(32 * steps)
    assert L.sh_dev_to_wire(ctx, dc, head, steps) == 0
    assert hashlib.sha256(head.raw).hexdigest() == c["coeffs_sha256"]  # the input is the fixture's input
    flat = ctypes.create_string_buffer(plen)
    for dense in (False, True):
        if dense:
            assert L.sh_dev_fri_prove(ctx, dc, n, w, c["maxdeg_plus_1"], c["exclude_multiples_of"], c["samples"], 1, dp) == 0
        else:
            assert L.sh_dev_fri_prove_coeffs(ctx, dc, steps, n, w, c["maxdeg_plus_1"], c["exclude_multiples_of"], c["samples"], 1, dp) == 0
        assert L.sh_dev_download(ctx, dp, flat, plen) == 0
        assert flat.raw[:32].hex() == c["first_root2"], (logsteps, dense)
        assert flat.raw[plen - 32 * 128:plen - 32 * 127].hex() == c["final_layer_first"], (logsteps, dense)
        assert hashlib.sha256(flat.raw).hexdigest() == c["proof_sha256"], (logsteps, dense)
    assert L.sh_dev_free(ctx, dc) == 0 and L.sh_dev_free(ctx, dp) == 0


def test_batched_mimc_proofs_config5_shape(sa, oracle):
    """starks_amd.batch: independent MiMC traces sharded by unit id; digests equal the one-at-a-time oracle proofs."""
    from starks_amd import batch
    steps = 256
    units = list(batch.shard(6, 1, 2))  # rank 1 of 2 -> units 3, 4, 5
    got = batch.prove_mimc_batch(units, steps, chunk=2)
    assert [j for j, _ in got] == units
    g2 = root_of(steps * 8)
    for j, flat in got:
        c = oracle.c.fft(oracle.py.mimc_trace(3 + j, steps), steps, pow(g2, 8, P), inverse=True)
        assert flat == oracle.c.fri_prove_flat(wire(c), g2, steps, 8, 40)


def test_packed_leaf_merkle_golden(sa, oracle):
    """merkelize_polynomial_evaluations (merkle_tree.py:94-119): multi-block BLAKE2s first level."""
    for c in load_golden("packed.json"):
        evals = [[sa.F(seeded(c["seed_base"] + k, i)) for i in range(c["n"])] for k in range(c["k"])]
        t = sa.mt.merkelize_polynomial_evaluations(1, evals)
        assert len(t) == 2 * c["n"] and t[0] == b"" and len(t[c["n"]]) == 32 * c["k"]
        assert t[1].hex() == c["root"]
        assert hashlib.sha256(b"".join(t)).hexdigest() == c["tree_sha"]
        assert [b.hex() for b in sa.mt.mk_branch(t, c["branch_index"])] == c["branch"]
        assert [x.hex() for x in sa.mt.unpack_merkle_leaf(t[c["n"]], 1, c["k"])] == c["unpacked_leaf0"]
    n, k = 1 << 13, 7  # bigger, odd k (a block straddles the two leaves of a pair), against the oracle
    evals = [[seeded(300 + c, i) for i in range(n)] for c in range(k)]
    assert sa.mt.merkelize_polynomial_evaluations(1, evals) == oracle.py.merkelize_polynomial_evaluations(evals)


def test_device_lde_matches_host_api(sa, oracle):
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    steps, ext, cols = 1024, 8, 3
    n = steps * ext
    g2 = root_of(n)
    traces = [oracle.py.mimc_trace(3 + j, steps) for j in range(cols)]
    dtr, dout = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * steps * cols, ctypes.byref(dtr)) == 0 and L.sh_dev_alloc(ctx, 32 * n * cols, ctypes.byref(dout)) == 0
    assert L.sh_dev_from_wire(ctx, b"".join(wire(t) for t in traces), dtr, steps * cols) == 0
    assert L.sh_dev_lde(ctx, dtr, dout, steps, ext, cols, g2.to_bytes(32, "big")) == 0
    host = ctypes.create_string_buffer(32 * n * cols)
    assert L.sh_dev_to_wire(ctx, dout, host, n * cols) == 0
    for j in range(cols):
        assert host.raw[32 * n * j:32 * n * (j + 1)] == oracle.c.lde_bytes(wire(traces[j]), ext, g2)
    assert L.sh_dev_free(ctx, dtr) == 0 and L.sh_dev_free(ctx, dout) == 0


# ---- randomized differential testing against the C oracle ---------------------------------------------------
def test_randomized_ntt_differential(sa, oracle):
    rng = random.Random(20261003)
    for _ in range(40):
        logn = rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
        n = 1 << logn
        n_in = rng.choice([0, 1, n // 3, n - 1, n]) if n > 2 else n
        batch = rng.choice([1, 1, 2, 3, 5])
        inverse = rng.random() < 0.5
        w = pow(root_of(n), rng.choice([1, 3, 5, n - 1]), P)  # any generator of the order-n subgroup
        rows = [[rng.choice([0, 1, P - 1, P, 2**256 - 1, rng.randrange(2**256)]) for _ in range(n_in)] for _ in range(batch)]
        got = sa.fft.ntt_bytes(b"".join(wire(r) for r in rows), n, w, inverse=inverse, batch=batch) if n_in else \
            sa.fft.ntt_bytes(b"", n, w, inverse=inverse, batch=1)
        for b, r in enumerate(rows if n_in else [[]]):
            assert got[b * 32 * n:(b + 1) * 32 * n] == oracle.c.fft_bytes(wire(r), n, w, inverse=inverse), (logn, n_in, batch, inverse)


def test_randomized_fri_differential(sa, oracle):
    rng = random.Random(77)
    for _ in range(12):
        logn = rng.choice([6, 7, 8, 9, 10, 11, 12])
        n = 1 << logn
        maxdeg = n >> rng.choice([0, 1, 2, 3])
        if maxdeg < 1:
            maxdeg = 1
        exclude = rng.choice([0, 0, 2, 4, 8])
        samples = rng.choice([1, 7, 40, 41, 64])
        batch = rng.choice([1, 2, 3])
        w = root_of(n)
        # every round needs a column of >= 4 values
        nn, md, ok = n, maxdeg, True
        while md > 16:
            ok &= nn >= 16 and (not exclude or (nn // 4) * (exclude - 1) // exclude > 0)
            nn //= 4
            md //= 4
        if not ok:
            continue
        n_co = rng.choice([1, maxdeg // 2 + 1, min(maxdeg, n)])
        coeffs = [wire(rng.randrange(P) for _ in range(n_co)) for _ in range(batch)]
        flat = sa.fri.prove_flat(b"".join(coeffs), n, w, maxdeg, exclude, samples, batch=batch)
        plen = sa.fri.proof_len(n, maxdeg, samples)
        for b in range(batch):
            want = oracle.c.fri_prove_flat(coeffs[b], w, maxdeg, exclude, samples, n=n)
            assert flat[b * plen:(b + 1) * plen] == want, (logn, maxdeg, exclude, samples, batch, n_co)


def test_two_process_sharded_proving(sa, oracle, tmp_path):
    """The N > 1 path with real GPU work: two processes (gloo rendezvous, both on this box's GPU) each prove their
    shard of 6 MiMC traces and all_gather the digests; every rank must end with the oracle's digests in unit order."""
    import subprocess, sys, textwrap
    from conftest import ROOT
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import hashlib, os, sys
        sys.path.insert(0, %r)
        os.environ["STARKHIP_DEVICE"] = "0"
        import torch.distributed as dist
        from starks_amd import batch
        dist.init_process_group(backend="gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        mine = batch.shard(6, rank, world)
        proofs = batch.prove_mimc_batch(mine, 128, chunk=2)
        digs = batch.gather_digests([batch.digest(p) for _, p in proofs], 6, rank, world, dist, "cpu")
        open(os.path.join(%r, "rank%%d.txt" %% rank), "w").write(",".join(d.hex() for d in digs))
        # the same units as full STARK proofs (config 5's unit of work)
        sproofs = batch.prove_stark_batch(mine, 64, chunk=2)
        sdigs = batch.gather_digests([batch.digest(p) for _, p in sproofs], 6, rank, world, dist, "cpu")
        open(os.path.join(%r, "stark_rank%%d.txt" %% rank), "w").write(",".join(d.hex() for d in sdigs))
        dist.destroy_process_group()
    """ % (ROOT, str(tmp_path), str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    steps = 128
    g2 = root_of(steps * 8)
    want = []
    for j in range(6):
        c = oracle.c.fft(oracle.py.mimc_trace(3 + j, steps), steps, pow(g2, 8, P), inverse=True)
        want.append(hashlib.sha256(oracle.c.fri_prove_flat(wire(c), g2, steps, 8, 40)).hexdigest())
    from starks_amd import batch
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    swant = []
    for j in range(6):
        w, inp = batch.mimc_stark_unit(j, 64)
        swant.append(hashlib.sha256(oracle.py.stark_flat(oracle.py.mk_stark_proof(w, inp, sp, 64, 8))).hexdigest())
    for rank in (0, 1):
        assert (tmp_path / ("rank%d.txt" % rank)).read_text().split(",") == want
        assert (tmp_path / ("stark_rank%d.txt" % rank)).read_text().split(",") == swant


def test_rare_carry_branches(sa):
    """fp_add / fp_sub / the reduction keep their rare carry propagation behind wave-uniform branches
    (fp256.cuh); these inputs force every one of them, alone in a wave and mixed with ordinary lanes."""
    import ctypes
    M = 1 << 256
    c = M - P
    H = (2 * M - 1) // c
    pairs = [(M - 1, M - (1 << 32) + 6),          # add: carry, fold carry leaves limb 1, second wrap
             (0, M - 5),                          # sub: borrow, fold borrow leaves limb 1, second borrow
             ((7 << 64) + 2, M - 1),              # sub: borrow leaves limb 1, no second borrow
             (M - 1, M - 1), (P, P - 1), (5, 7)]
    rng = random.Random(9)
    filler = [(rng.randrange(M), rng.randrange(M)) for _ in range(70)]
    for batch_pairs in ([p] for p in pairs):
        a, b = batch_pairs[0]
        out = unwire(sa.fft.ntt_bytes(wire([a, b]), 2, P - 1))   # n = 2: (a + b, a - b)
        assert out == [(a + b) % P, (a - b) % P]
    mixed = filler[:30] + pairs + filler[30:]
    got = unwire(sa.fft.ntt_bytes(b"".join(wire([a, b]) for a, b in mixed), 2, P - 1, batch=len(mixed)))
    assert got == [v for a, b in mixed for v in ((a + b) % P, (a - b) % P)]
    # fp_mul2 (the butterflies' product by a table pair, fp256.cuh): its fold carries out of limb 5 with probability 2^-22.
    # A 4-point transform of [0, x, 0, 0] multiplies x by the 4th root of unity I = root_of(4) through it; these x were
    # found by search (exact integers: the low 192 bits of x_lo I + x_hi (I 2^128 mod p) plus the folded top overflow).
    rare = [0xff6a2ad1abf9806db549b1ee5fa97b878013931d44877dfc3495ef6a863f1f72,
            0x73733a361e4774862b504d594c9ae08db9c050b30568b5b6e336370c0edcc14a]
    I4 = root_of(4)
    C = M - P
    for x in rare:
        t = (x & ((1 << 128) - 1)) * I4 + (x >> 128) * ((I4 << 128) % P)
        assert ((t & ((1 << 192) - 1)) + (((t >> 256) * C) & ((1 << 192) - 1))) >> 192 == 1   # the vector does what it claims
        out = unwire(sa.fft.ntt_bytes(wire([0, x, 0, 0]), 4, I4))
        assert out == [x % P, x * I4 % P, (P - x) % P, (P - x) * I4 % P]
    vecs = [[rng.randrange(M) for _ in range(4)] for _ in range(40)]
    vecs[17] = [0, rare[0], 0, 0]
    vecs[18] = [5, rare[1], 9, 0]
    got = unwire(sa.fft.ntt_bytes(b"".join(wire(v) for v in vecs), 4, I4, batch=len(vecs)))
    want = []
    for v in vecs:
        want += [sum(v[j] * pow(I4, j * k, P) for j in range(4)) % P for k in range(4)]
    assert got == want
    # products through mul_polys with n = 1 (NTT of length 1 is the identity): x * y mod p
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    for x, y in [(2 * H, 1 << 255), (M - 1, M - 1), (P - 1, P - 1), (H, 1 << 255), (2 * H + 1, (1 << 255) + 12345)]:
        out = ctypes.create_string_buffer(32)
        assert L.sh_mul_polys(ctx, x.to_bytes(32, "big"), 1, y.to_bytes(32, "big"), 1, out, 1, (1).to_bytes(32, "big")) == 0
        assert int.from_bytes(out.raw, "big") == x * y % P


def test_degenerate_inputs(sa, oracle):
    """Zero / constant / maximal inputs: identical leaves, zero columns, all-(p-1) vectors."""
    n = 1 << 10
    w = root_of(n)
    for vec in ([0] * n, [P - 1] * n, [1] + [0] * (n - 1), [2**256 - 1] * n):
        data = wire(vec)
        assert sa.fft.ntt_bytes(data, n, w) == oracle.c.fft_bytes(data, n, w)
        assert sa.fft.ntt_bytes(data, n, w, inverse=True) == oracle.c.fft_bytes(data, n, w, inverse=True)
    for leaves in (bytes(32 * 64), b"\xff" * (32 * 64), wire([7] * 64)):
        assert sa.mt.merkelize_bytes(leaves) == oracle.c.merkelize_bytes(leaves)
    # FRI of the zero polynomial, a constant, and x^(maxdeg-1): every column is highly structured
    for coeffs in ([], [5], [0] * 255 + [1]):
        flat = sa.fri.prove_flat(wire(coeffs), n, w, 256, 0, 40)
        assert flat == oracle.c.fri_prove_flat(wire(coeffs), w, 256, 0, 40, n=n)
    # reference-level API with a Poly that strips to nothing (polynomial.py:58)
    from starks_amd.polynomial import polynomials_over
    zero = polynomials_over(sa.F).factory([0, 0, 0])
    assert zero.coefficients == [] and [int(v) for v in sa.fft.NonBinaryFFT(sa.F, sa.F(root_of(16))).fft(zero)] == [0] * 16


# ---- STARK.mk_proof (stark.py:233-279): SURVEY 8(f) rank 4 ---------------------------------------------------
def _stark_setup(sa, c):
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import multivariates_over
    mv = multivariates_over(sa.F, c["width"]).factory
    polys = [mv({tuple(k): v for k, v in d}) for d in c["step_polys"]]
    witness = [[h2i(x) for x in col] for col in c["witness"]]
    return stark, polys, witness


@pytest.mark.parametrize("c", load_golden("stark.json"), ids=lambda c: c["name"])
def test_stark_proofs_golden(sa, oracle, c):
    """The reference's own mk_proof output (run live by tests/golden/generate.py), byte for byte."""
    stark, polys, witness = _stark_setup(sa, c)
    flat = stark.prove_flat(b"".join(wire(col) for col in witness), wire(c["inputs"]), c["steps"], c["ext"], c["width"], polys)
    assert len(flat) == c["flat_len"]
    proof = stark.unpack_proof(flat, c["steps"], c["ext"], c["width"], c["degree"])
    assert proof[0].hex() == c["m_root"], "m_root"
    assert proof[1].hex() == c["l_root"], "l_root"
    assert [b.hex() for b in proof[2][0]] == c["branch0"]
    assert hashlib.sha256(flat).hexdigest() == c["flat_sha"]
    path = os.path.join(GOLDEN, "stark_%s.flat.bin" % c["name"])
    if os.path.exists(path):
        assert flat == open(path, "rb").read()
    # the drop-in call site, and both verifiers (the host mirror and the oracle's restatement)
    S = stark.STARK(sa.F, c["steps"], c["ext"], c["width"], polys)
    boundary = [(0, j, sa.F(v)) for j, v in enumerate(c["inputs"])]
    pr = S.mk_proof([[sa.F(v) for v in col] for col in witness], boundary)
    assert pr[0] == proof[0] and pr[1] == proof[1] and pr[2] == proof[2] and pr[3] == proof[3]
    assert S.verify_proof(pr, witness, boundary)
    sp = [{tuple(k): v for k, v in d} for d in c["step_polys"]]
    assert oracle.py.verify_stark_proof(pr, [col[-1] for col in witness], c["inputs"], sp, c["steps"], c["ext"])


def test_stark_random_vs_oracle(sa, oracle):
    """Random sparse step polynomials, widths 1..4: flat proof == the oracle's coefficient-form prover."""
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import multivariates_over
    po = oracle.py
    rng = random.Random(77)
    for width, steps, maxdeg in [(1, 8, 3), (2, 16, 2), (3, 16, 3), (4, 32, 2), (2, 64, 5), (5, 8, 2), (9, 8, 1)]:
        sp = []
        for _ in range(width):
            terms = {}
            for _ in range(rng.randint(1, 3)):
                ex = [0] * width
                for _ in range(rng.randint(0, maxdeg)):
                    ex[rng.randrange(width)] += 1
                terms[tuple(ex)] = rng.choice([1, 2, 3, rng.randrange(P), P - 1])
            sp.append(terms)
        inputs = [rng.randrange(P) for _ in range(width)]
        w = po.get_computational_trace(inputs, steps, sp)
        want = po.stark_flat(po.mk_stark_proof(w, inputs, sp, steps, 8))
        mv = multivariates_over(sa.F, width).factory
        got = stark.prove_flat(b"".join(wire(col) for col in w), wire(inputs), steps, 8, width, [mv(d) for d in sp])
        assert got == want, (width, steps, sp)


def test_stark_batch_and_invalid_witness(sa, oracle):
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    po = oracle.py
    X1, X2 = generate_Xi_s(sa.F, 2)
    polys = [X1, X1 + X2**3]
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    steps, ext = 32, 8
    ws, ins = [], []
    for j in range(3):
        inp = [2 + j, 5 + 7 * j]
        ws.append(po.get_computational_trace(inp, steps, sp))
        ins.append(inp)
    one = [stark.prove_flat(b"".join(wire(c) for c in w), wire(i), steps, ext, 2, polys) for w, i in zip(ws, ins)]
    many = stark.prove_flat(b"".join(b"".join(wire(c) for c in w) for w in ws), b"".join(wire(i) for i in ins), steps, ext, 2,
                            polys, batch=3)
    assert many == b"".join(one)
    assert one[1] == po.stark_flat(po.mk_stark_proof(ws[1], ins[1], sp, steps, ext))
    # a witness that breaks one transition: the reference asserts `cp % z == 0` (stark.py:76)
    bad = [list(c) for c in ws[0]]
    bad[1][5] = (bad[1][5] + 1) % P
    S = stark.STARK(sa.F, steps, ext, 2, polys)
    boundary = [(0, j, v) for j, v in enumerate(ins[0])]
    with pytest.raises(AssertionError):
        S.mk_proof(bad, boundary)
    # ... and the context is usable afterwards
    pr = S.mk_proof(ws[0], boundary)
    assert S.verify_proof(pr, ws[0], boundary)
    # a tampered proof is rejected by the host verifier
    leaf = bytearray(pr[2][0][0])
    leaf[5] ^= 1
    pr[2][0][0] = bytes(leaf)
    with pytest.raises(AssertionError):
        S.verify_proof(pr, ws[0], boundary)


def test_stark_shape_errors(sa):
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X = generate_Xi_s(sa.F, 1)[0]
    assert stark.proof_len(8, 8, 10, 1) == 0           # width >= 10: get_pseudorandom_ks returns None (stark.py:106-126)
    assert stark.proof_len(8, 8, 1, 9) == 0            # degree * (steps - 1) + 1 must stay below the domain
    assert stark.proof_len(12, 8, 1, 1) == 0           # steps must be a power of two
    assert stark.proof_len(8, 1, 1, 1) == 0            # exclude_multiples_of = 1 divides by zero (utils.py:90)
    with pytest.raises(NotImplementedError):
        stark.prove_flat(wire([1] * 8), wire([1]), 8, 8, 1, [X**9])


@pytest.mark.parametrize("logsteps,ext", [(10, 8), (14, 8), (16, 8), (20, 8), (14, 4), (14, 16), (12, 32), (18, 4), (16, 16)])
def test_stark_large_prove_then_verify(sa, oracle, logsteps, ext):
    """Sizes the coefficient-form oracle cannot reach: prove on the device, verify with the host verifier and with the
    oracle's restated verifier (transition + boundary identities at 80 positions, FRI on the linear combination).  Extension
    factors 4 (the smallest a cubic step admits: deg C (X - x_last) = 3 steps - 2 must stay below steps * ext), 8, 16, 32."""
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    steps = 1 << logsteps
    X1, X2 = generate_Xi_s(sa.F, 2)
    polys = [X1, X1 + X2**3]
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    inputs = [42, 3]
    w = oracle.py.get_computational_trace(inputs, steps, sp)
    S = stark.STARK(sa.F, steps, ext, 2, polys)
    boundary = [(0, j, v) for j, v in enumerate(inputs)]
    pr = S.mk_proof(w, boundary)
    assert S.verify_proof(pr, w, boundary)
    assert oracle.py.verify_stark_proof(pr, [c[-1] for c in w], inputs, sp, steps, ext)
    # the library's own verifier (sh_stark_verify, host C++ behind the C ABI) on the flat bytes, and on a flipped byte
    flat = stark.prove_flat(b"".join(wire(col) for col in w), wire(inputs), steps, ext, 2, polys)
    assert stark.unpack_proof(flat, steps, ext, 2, 3) == pr
    outs = wire(c[-1] for c in w)
    assert stark.verify_flat(flat, wire(inputs), outs, steps, ext, 2, polys)
    bad = bytearray(flat)
    bad[len(bad) // 3] ^= 4
    with pytest.raises(AssertionError):
        stark.verify_flat(bytes(bad), wire(inputs), outs, steps, ext, 2, polys)
    # the P evaluations inside the leaves are the low-degree extension of the witness
    pos = sa.utils.get_pseudorandom_indices(pr[1], steps * ext, 80, exclude_multiples_of=ext)[0]
    leaf = sa.mt.unpack_merkle_leaf(pr[2][0][0], 2, 3)
    if logsteps <= 16:
        lde = oracle.c.lde_bytes(wire(w[1]), ext, root_of(steps * ext))
        assert leaf[1] == lde[32 * pos:32 * pos + 32]
    # the committed trace values are the witness itself: P(g1^k) = witness[.][k] at an untouched trace point is implied by
    # the boundary + transition identities the verifiers checked at 80 random points of a degree < steps polynomial


def _stark_large_cases():
    try:
        return load_golden("stark_large.json")["cases"]
    except Exception:
        return []


@pytest.mark.parametrize("c", _stark_large_cases(), ids=lambda c: "steps_2^%d" % c["logsteps"])
def test_stark_proofs_at_size_vs_coefficient_form_oracle_fixture(sa, c):
    """Whole STARK proofs of config 5's unit 0 at 2^12, 2^14 and 2^16 steps (config 5's own size) byte for byte against the proofs the
    COEFFICIENT-FORM prover of oracle/pyoracle.py wrote (the reference's construction, quadratic: 9 min for 2^14 steps, hours for 2^16;
    tests/golden/stark_large.json, generate_large.py --stark) -- through the host-buffer entry point, and through the device-resident
    batched one that bench.py's config 5 times (unit 0 inside a batch of 4)."""
    import ctypes
    from starks_amd import batch, stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    steps, ext = c["steps"], c["ext"]
    X1, X2 = generate_Xi_s(sa.F, 2)
    polys = [X1, X1 + X2**3]
    w, inp = batch.mimc_stark_unit(0, steps)
    assert inp == c["inputs"] and ["%064x" % col[-1] for col in w] == c["outputs"]
    flat = stark.prove_flat(b"".join(wire(col) for col in w), wire(inp), steps, ext, 2, polys)
    assert len(flat) == c["proof_bytes"] and flat[:32].hex() == c["m_root"] and flat[32:64].hex() == c["l_root"]
    assert hashlib.sha256(flat).hexdigest() == c["proof_sha256"]
    assert stark.verify_flat(flat, wire(inp), b"".join(bytes.fromhex(v) for v in c["outputs"]), steps, ext, 2, polys)
    # the device-resident path with witnesses generated on the device (what config 5 runs): unit 0 of a batch
    pr = batch.StarkUnitProver(steps, ext, chunk=4)
    try:
        pr.generate(0, 4)
        pr.prove(4)
        assert pr.download(4)[0] == flat
    finally:
        pr.close()


def test_stark_batch_units_and_device_api(sa, oracle):
    """batch.prove_stark_batch (config 5's unit as a full STARK) == the oracle per unit; and the device-resident entry
    point with its deferred constraint status."""
    import ctypes
    from starks_amd import batch, stark
    po = oracle.py
    steps, ext = 64, 8
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    got = batch.prove_stark_batch([0, 5, 9], steps, ext, chunk=2)
    for j, flat in got:
        w, inp = batch.mimc_stark_unit(j, steps)
        assert w == po.get_computational_trace(inp, steps, sp)
        assert flat == po.stark_flat(po.mk_stark_proof(w, inp, sp, steps, ext)), j
    # device-resident: same bytes; a broken witness is reported by sh_stark_status, once
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X1, X2 = generate_Xi_s(sa.F, 2)
    coefs, exps, counts, degree = stark.pack_step_polys([X1, X1 + X2**3], 2)
    plen = stark.proof_len(steps, ext, 2, degree)
    w, inp = batch.mimc_stark_unit(5, steps)

    def dev_prove(cols):
        dw, di, dp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        for ptr, nbytes in ((dw, 64 * steps), (di, 64), (dp, plen)):
            sa.lib.check(L.sh_dev_alloc(ctx, nbytes, ctypes.byref(ptr)), "alloc")
        sa.lib.check(L.sh_dev_from_wire(ctx, b"".join(wire(c) for c in cols), dw, 2 * steps), "w")
        sa.lib.check(L.sh_dev_from_wire(ctx, wire(inp), di, 2), "i")
        sa.lib.check(L.sh_dev_stark_prove(ctx, dw, di, steps, ext, 2, coefs, exps, counts, 80, 1, dp), "prove")
        out = ctypes.create_string_buffer(plen)
        sa.lib.check(L.sh_dev_download(ctx, dp, out, plen), "dl")
        for ptr in (dw, di, dp):
            sa.lib.check(L.sh_dev_free(ctx, ptr), "free")
        return out.raw
    assert dev_prove(w) == dict(got)[5]
    assert L.sh_stark_status(ctx) == 0
    bad = [list(c) for c in w]
    bad[1][-1] = (bad[1][-1] + 1) % P  # breaks the last transition only
    dev_prove(bad)
    assert L.sh_stark_status(ctx) == -8
    assert L.sh_stark_status(ctx) == 0


@pytest.mark.parametrize("steps,ext,width", [(2, 8, 1), (4, 8, 2), (2, 2, 1), (8, 2, 2), (16, 4, 2), (8, 16, 3), (4, 32, 1),
                                              (128, 8, 1), (256, 4, 2)])
def test_stark_extension_factors_and_tiny_traces(sa, oracle, steps, ext, width):
    """Extension factors other than 8 and the smallest traces (x = 1 and x = x_last are then most of G1)."""
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import multivariates_over
    po = oracle.py
    rng = random.Random(steps * 1000 + ext * 10 + width)
    maxdeg = max(1, min(3, (steps * ext - 2) // max(steps - 1, 1) - 1))
    sp = []
    for _ in range(width):
        terms = {}
        for t in range(rng.randint(1, 3)):
            ex = [0] * width
            for _ in range(rng.randint(0 if t else 1, maxdeg)):  # degree >= 1: with degree 0 the reference's own verifier
                ex[rng.randrange(width)] += 1                    # rejects (FRI bound steps * 0, fri.py:351-366)
            terms[tuple(ex)] = rng.choice([1, 2, rng.randrange(P)])
        sp.append(terms)
    degree = max(po.mv_degree(q) for q in sp)
    if degree * (steps - 1) + 1 >= steps * ext:
        pytest.skip("degree too high for this domain")
    inputs = [rng.randrange(P) for _ in range(width)]
    w = po.get_computational_trace(inputs, steps, sp)
    want = po.mk_stark_proof(w, inputs, sp, steps, ext)
    mv = multivariates_over(sa.F, width).factory
    polys = [mv(d) for d in sp]
    got = stark.prove_flat(b"".join(wire(col) for col in w), wire(inputs), steps, ext, width, polys)
    assert got == po.stark_flat(want), (steps, ext, width, sp)
    S = stark.STARK(sa.F, steps, ext, width, polys)
    assert S.verify_proof(stark.unpack_proof(got, steps, ext, width, degree), w, [(0, j, v) for j, v in enumerate(inputs)])


def test_stark_wide_state_medium_trace(sa, oracle):
    """The reference's width-6 sextic system (test_stark.py:324-350) at 128 steps against the oracle, and a width-9
    system at 1024 steps through both verifiers."""
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import multivariates_over
    po = oracle.py
    sp = [{tuple(1 if j == i else 0 for j in range(6)): 1} for i in range(5)] + [{(1, 1, 1, 1, 1, 1): 1}]
    inputs, steps, ext = [1, 2, 3, 4, 5, 6], 128, 8
    w = po.get_computational_trace(inputs, steps, sp)
    mv = multivariates_over(sa.F, 6).factory
    got = stark.prove_flat(b"".join(wire(c) for c in w), wire(inputs), steps, ext, 6, [mv(d) for d in sp])
    assert got == po.stark_flat(po.mk_stark_proof(w, inputs, sp, steps, ext))
    # width 9: dimension j takes x_{j+1}^2 + j (cyclically) -- every dimension moves, degree 2
    sp9 = [{tuple(2 if v == (j + 1) % 9 else 0 for v in range(9)): 1, (0,) * 9: j + 1} for j in range(9)]
    inputs9, steps9 = list(range(11, 20)), 1024
    w9 = po.get_computational_trace(inputs9, steps9, sp9)
    mv9 = multivariates_over(sa.F, 9).factory
    polys9 = [mv9(d) for d in sp9]
    S = stark.STARK(sa.F, steps9, ext, 9, polys9)
    boundary = [(0, j, v) for j, v in enumerate(inputs9)]
    pr = S.mk_proof(w9, boundary)
    assert S.verify_proof(pr, w9, boundary)
    assert po.verify_stark_proof(pr, [c[-1] for c in w9], inputs9, sp9, steps9, ext)


def test_full_size_merkle_2_24_branches_verify(sa):
    """A 2^24-leaf tree (1 GiB of nodes) built on the device: random branches, fetched node by node, verify against the
    root with the host verifier, and their leaves are the seeded values (merkle_tree.py:59-86)."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << 24
    dx, dt = ctypes.c_void_p(), ctypes.c_void_p()
    sa.lib.check(L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dx)), "alloc")
    sa.lib.check(L.sh_dev_alloc(ctx, 64 * n, ctypes.byref(dt)), "alloc")
    try:
        sa.lib.check(L.sh_dev_fill_seeded(ctx, dx, n, 99), "fill")
        sa.lib.check(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkelize")

        def node(i):
            buf = ctypes.create_string_buffer(32)
            sa.lib.check(L.sh_dev_download(ctx, ctypes.c_void_p(dt.value + 32 * i), buf, 32), "download")
            return buf.raw
        root = node(1)
        assert node(0) == bytes(32)
        rng = random.Random(24)
        for index in [0, 1, n // 4, n // 4 + 1, n - 1] + [rng.randrange(n) for _ in range(6)]:
            idx = sa.mt.get_index_in_permuted(index, n) + n
            branch = [node(idx)]
            while idx > 1:
                branch.append(node(idx ^ 1))
                idx //= 2
            assert len(branch) == 25
            assert sa.mt.verify_branch(root, index, branch, output_as_int=True) == seeded(99, index)
    finally:
        L.sh_dev_free(ctx, dx)
        L.sh_dev_free(ctx, dt)


@pytest.mark.parametrize("logn", [20, 24, 26])
def test_merkle_commit_at_size_vs_oracle_fixture(sa, logn):
    """The Merkle commitment bench.py times (2^24 seeded leaves; 2^20 in its --quick mode) and a 2^26-leaf one (4 GiB of nodes): EVERY node of the tree against the tree
    oracle/oracle.c hashed (tests/golden/merkle_large.json: root, three interior nodes, SHA-256 of all 2n nodes)."""
    import ctypes
    c = [c for c in load_golden("merkle_large.json")["cases"] if c["logn"] == logn][0]
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << logn
    dx, dt = ctypes.c_void_p(), ctypes.c_void_p()
    sa.lib.check(L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dx)), "alloc")
    sa.lib.check(L.sh_dev_alloc(ctx, 64 * n, ctypes.byref(dt)), "alloc")
    try:
        sa.lib.check(L.sh_dev_fill_seeded(ctx, dx, n, c["seed"]), "fill")
        sa.lib.check(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkelize")
        h = hashlib.sha256()
        step = 1 << 20  # nodes per download
        buf = ctypes.create_string_buffer(32 * step)
        first = None
        for off in range(0, 2 * n, step):
            sa.lib.check(L.sh_dev_download(ctx, ctypes.c_void_p(dt.value + 32 * off), buf, 32 * step), "download")
            if first is None:
                first = buf.raw[:96]
            if off <= n - 1 < off + step:
                assert buf.raw[32 * (n - 1 - off):32 * (n - off)].hex() == c["node_n_minus_1"]
            if off <= n < off + step:
                assert buf.raw[32 * (n - off):32 * (n - off) + 32].hex() == c["node_n"]
            h.update(buf.raw)
        assert first[:32] == bytes(32) and first[32:64].hex() == c["root"] and first[64:96].hex() == c["node_2"]
        assert h.hexdigest() == c["nodes_sha256"]
    finally:
        L.sh_dev_free(ctx, dx)
        L.sh_dev_free(ctx, dt)


# ---- round 2: reference-independent pins for config 4, config 5 at its size, the new ABI entries -------------------------
@pytest.mark.parametrize("logn", [17, 19, 21, 22, 23, 24, 25, 26])
def test_ntt_large_digests_vs_oracle_fixture(sa, logn):
    """Config 4 (2^24), 2^25 and 2^26 (four-pass plans, 1 and 2 GiB vectors), 2^22 and the plan shapes no reference digest reaches densely -- 2^17 (config 3's domain, plan (9, 8)), 2^19 (config 5's domain, plan (9, 10)),
    2^21 (the first three-pass plan) and 2^23 (the domain of the metric's 2^20-step FRI commit, 256 MiB row table): forward and
    inverse transforms of the seeded vector against the digests the C oracle produced (tests/golden/ntt_large.json; the
    oracle is pinned to the live reference up to 2^20)."""
    import ctypes
    c = [c for c in load_golden("ntt_large.json")["cases"] if c["logn"] == logn][0]
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << logn
    w = root_of(n).to_bytes(32, "big")
    assert w.hex() == c["w"]
    dx, dy = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dx)) == 0 and L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dy)) == 0
    assert L.sh_dev_fill_seeded(ctx, dx, n, c["seed"]) == 0
    host = ctypes.create_string_buffer(32 * n)
    for inverse, key, head in ((0, "sha_fwd", "fwd_head"), (1, "sha_inv", "inv_head")):
        assert L.sh_dev_ntt(ctx, dx, dy, n, 1, w, inverse) == 0
        assert L.sh_dev_to_wire(ctx, dy, host, n) == 0
        assert bytes(host[:64]).hex() == c[head]
        assert hashlib.sha256(host).hexdigest() == c[key], (logn, key)  # (the buffer itself: no second copy of up to 2 GiB)
    assert L.sh_dev_free(ctx, dx) == 0 and L.sh_dev_free(ctx, dy) == 0


def test_config5_all_512_units_at_size(sa, oracle):
    """BASELINE configs[4] at its size on one GPU: 512 independent 2^16-step MiMC STARK proofs in batched launches of 32
    (device-generated witnesses).  A sample of units equals the same unit proved alone from a host-built witness, byte for
    byte; one of them passes both verifiers; all 512 proofs are distinct."""
    from starks_amd import batch, stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    steps, ext, total = 1 << 16, 8, 512
    sample = {0, 31, 32, 257, 511}
    digs, kept = batch.prove_stark_units_device(0, total, steps, ext, chunk=32, keep=sample)
    assert len(digs) == total and len(set(digs)) == total and set(kept) == sample
    for j in sorted(sample):
        (jj, alone), = batch.prove_stark_batch([j], steps, ext, chunk=1)
        assert jj == j and alone == kept[j], j
        assert batch.digest(alone) == digs[j]
    X1, X2 = generate_Xi_s(sa.F, 2)
    S = stark.STARK(sa.F, steps, ext, 2, [X1, X1 + X2**3])
    w, inp = batch.mimc_stark_unit(511, steps)
    pr = stark.unpack_proof(kept[511], steps, ext, 2, 3)
    assert S.verify_proof(pr, w, [(0, j, v) for j, v in enumerate(inp)])
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    assert oracle.py.verify_stark_proof(pr, [c[-1] for c in w], inp, sp, steps, ext)


def test_mimc_unit_generator_status_batch_and_trim(sa, oracle):
    """sh_dev_fill_mimc_units == batch.mimc_stark_unit; sh_stark_status_batch names the invalid proof of a batch;
    sh_ctx_trim drops the caches and the next proof is rebuilt to the same bytes."""
    import ctypes
    from starks_amd import batch, stark
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    steps, ext, k = 64, 8, 3
    pr = batch.StarkUnitProver(steps, ext, chunk=k)
    pr.generate(7, k)
    host = ctypes.create_string_buffer(64 * steps * k)
    sa.lib.check(L.sh_dev_to_wire(ctx, pr.dw, host, 2 * steps * k), "dl")
    want = b"".join(wire(col) for j in range(7, 7 + k) for col in batch.mimc_stark_unit(j, steps)[0])
    assert host.raw == want
    hin = ctypes.create_string_buffer(64 * k)
    sa.lib.check(L.sh_dev_to_wire(ctx, pr.di, hin, 2 * k), "dl")
    assert hin.raw == b"".join(wire(batch.mimc_stark_unit(j, steps)[1]) for j in range(7, 7 + k))
    pr.prove(k)
    good = pr.download(k)
    sp = [{(1, 0): 1}, {(1, 0): 1, (0, 3): 1}]
    for i, j in enumerate(range(7, 7 + k)):
        w, inp = batch.mimc_stark_unit(j, steps)
        assert good[i] == oracle.py.stark_flat(oracle.py.mk_stark_proof(w, inp, sp, steps, ext))
    # break unit 1 of the batch only
    pr.generate(7, k)
    bad_elem = wire([5])
    sa.lib.check(L.sh_dev_from_wire(ctx, bad_elem, ctypes.c_void_p(pr.dw.value + 32 * (1 * 2 * steps + steps + 9)), 1), "poke")
    pr.prove(k)
    flags = ctypes.create_string_buffer(k)
    assert L.sh_stark_status_batch(ctx, flags, k) == -8
    assert flags.raw == b"\x00\x01\x00"
    assert L.sh_stark_status_batch(ctx, flags, k) == 0 and flags.raw == b"\x00" * k
    # an unchecked failure of the device API does not leak into the next host-API call
    pr.generate(7, k)
    sa.lib.check(L.sh_dev_from_wire(ctx, bad_elem, ctypes.c_void_p(pr.dw.value + 32 * (steps + 9)), 1), "poke")
    pr.prove(k)
    w, inp = batch.mimc_stark_unit(8, steps)
    X = pr.coefs, pr.exps, pr.counts
    out = ctypes.create_string_buffer(pr.plen)
    assert L.sh_stark_prove(ctx, b"".join(wire(c) for c in w), wire(inp), steps, ext, 2, X[0], X[1], X[2], 80, 1, out, pr.plen) == 0
    assert out.raw == good[1]
    # trim: everything cached is dropped and rebuilt
    assert L.sh_ctx_trim(ctx) == 0
    pr.generate(7, k)
    pr.prove(k)
    assert pr.download(k) == good
    pr.close()


def test_pinned_batch_transform_is_pipelined_and_identical(sa, oracle):
    """sh_ntt_batch from page-locked buffers moves its vectors through in chunks (upload / transform / download overlapped on
    three streams, csrc/capi.hip:ntt_batch_pipelined): every vector must equal the one-at-a-time result and the oracle's."""
    import random
    rng = random.Random(3)
    n, B = 1 << 15, 5   # 5 MiB: above the pipelining threshold, a chunk count that does not divide evenly into 8
    w = root_of(n)
    vecs = [[rng.randrange(P) for _ in range(n)] for _ in range(B)]
    src, dst = sa.lib.PinnedBuffer(32 * n * B), sa.lib.PinnedBuffer(32 * n * B)
    try:
        src.view[:] = b"".join(wire(v) for v in vecs)
        for inverse in (False, True):
            sa.fft.ntt_bytes(src, n, w, inverse=inverse, batch=B, out=dst)
            got = bytes(dst.view)
            for b in range(B):
                one = sa.fft.ntt_bytes(wire(vecs[b]), n, w, inverse=inverse)
                assert got[32 * n * b:32 * n * (b + 1)] == one, (inverse, b)
            assert got[:32 * n] == wire(oracle.py.fft_1d(vecs[0], P, w, inv=inverse))
        # 11 vectors: more vectors than chunks
        B2 = 11
        src2, dst2 = sa.lib.PinnedBuffer(32 * n * B2), sa.lib.PinnedBuffer(32 * n * B2)
        try:
            src2.view[:] = b"".join(wire(vecs[b % B]) for b in range(B2))
            sa.fft.ntt_bytes(src2, n, w, batch=B2, out=dst2)
            got2 = bytes(dst2.view)
            one = [sa.fft.ntt_bytes(wire(v), n, w) for v in vecs]
            assert all(got2[32 * n * b:32 * n * (b + 1)] == one[b % B] for b in range(B2))
        finally:
            src2.close(); dst2.close()
    finally:
        src.close(); dst.close()


def test_plan_cache_is_lru_with_a_byte_budget(sa, oracle):
    """200 distinct (n, root) shapes interleaved with a repeated hot shape under a budget that holds only a few plans: the
    hot shape is built once and never rebuilt (least-recently-used eviction, csrc/capi.hip:evict_plans), the held bytes stay
    inside the budget + one call's tables, and every transform still equals the oracle's."""
    import ctypes, random
    L = sa.lib.lib()
    ctx = ctypes.c_void_p()
    assert L.sh_ctx_create(0, ctypes.byref(ctx)) == 0
    stats = (ctypes.c_uint64 * 4)()
    try:
        budget = 3 << 20
        assert L.sh_ctx_set_plan_budget(ctx, budget) == 0
        rng = random.Random(5)

        def run(logn, k):
            n = 1 << logn
            w = pow(7, (P - 1) // n, P)
            w = pow(w, k, P)  # another primitive root of the same order for odd k
            vals = [rng.randrange(P) for _ in range(n)]
            out = ctypes.create_string_buffer(32 * n)
            assert L.sh_ntt(ctx, wire(vals), n, out, n, w.to_bytes(32, "big"), 0) == 0
            return vals, w, out.raw

        hot_vals, hot_w, hot_out = run(12, 1)
        assert hot_out == wire(oracle.py.fft_1d(hot_vals, P, hot_w))
        assert L.sh_ctx_stats(ctx, stats) == 0 and stats[2] == 1
        shapes = [(logn, k) for k in range(1, 41, 2) for logn in range(5, 15)]
        assert len(set(shapes)) == 200
        peak = 0
        for i, (logn, k) in enumerate(shapes):
            if (logn, k) == (12, 1):
                k = 41
            vals, w, got = run(logn, k)
            if i % 23 == 0:
                assert got == wire(oracle.py.fft_1d(vals, P, w)), (logn, k)
            if i % 4 == 3:  # the hot shape again: a cache hit every time
                out = ctypes.create_string_buffer(32 << 12)
                assert L.sh_ntt(ctx, wire(hot_vals), 1 << 12, out, 1 << 12, hot_w.to_bytes(32, "big"), 0) == 0
                assert out.raw == hot_out
            assert L.sh_ctx_stats(ctx, stats) == 0
            peak = max(peak, stats[1])
        assert stats[2] == 201, list(stats)          # 200 cold shapes + the hot one, built once
        assert stats[3] >= 150 and stats[0] < 60, list(stats)
        assert peak <= budget + (2 << 20), peak
    finally:
        L.sh_ctx_destroy(ctx)


@pytest.mark.parametrize("args", [["--workload", "c5", "--quick"],
                                  ["--quick", "--steps", "2", "--warmup", "1", "--no-extras", "--logn", "14"]])
def test_bench_two_ranks_share_this_gpu(sa, args):
    """bench.py --gpus 2 starts its own two ranks (gloo control plane, both on this box's GPU): the N > 1 path of both
    workloads with real GPU work -- sharding by proof index, header all_gather, max-over-ranks timing, one JSON line."""
    import json, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--no-cpu-baseline"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    c5 = line["c5"]
    assert c5["n_gpus"] == 2 and c5["units_per_rank"] == [8, 8] and c5["check"]["ok"] is True
    assert c5["check"]["rank0"]["batch_equals_single"] is True and c5["check"]["rank0"]["verifies"] is True
    if line["metric"] == "ntt_field_elements_per_sec":
        assert line["check"]["roundtrip_ok"] is True and line["scaling"] == "weak"
        assert len(line["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in line["per_rank_ms_per_step"])
        assert line["per_rank_elements_per_step"] == [line["config"]["elements_per_step"]] * 2
    else:
        assert line["scaling"] == "strong" and line["unit"] == "proofs/s"
    # the line explains who ran where: backend, world size as torch.distributed reports it, each rank's device
    ri = line["ranks"]
    assert ri["backend"] == "gloo" and ri["world_size"] == 2 and ri["world_size_env"] == 2
    assert [r["rank"] for r in ri["ranks"]] == [0, 1] and [r["local_rank"] for r in ri["ranks"]] == [0, 1]
    assert all(r["device_index"] == 0 and r["pci_bus_id"] and r["device_name"] for r in ri["ranks"])
    assert ri["devices_distinct"] is False  # both ranks of this rehearsal sit on the box's one GPU (allowed with gloo only)
    assert len(c5["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in c5["per_rank_ms_per_step"])
    # every proof delivered to the host, and what arrived is what the device holds
    assert c5["proofs_per_s_delivered"] > 0 and c5["check"]["rank0"]["delivered_equals_device"] is True


@pytest.mark.parametrize("workload", ["ntt", "c5"])
def test_bench_rccl_collectives_with_one_rank(sa, workload):
    """The RCCL path (backend nccl) before its first N > 1 run on the driver's 8-GPU node: bench.py launched by
    torch.distributed.run with ONE rank and BENCH_FORCE_DIST=1, so that init_process_group("nccl", device_id=...), the barrier,
    the MAX / MIN all_reduce of device tensors and the all_gather of the proof headers on device buffers all run over RCCL.
    One JSON line, check.ok, and the gathered headers equal the ones the gloo control plane delivers for the same units."""
    import json, subprocess, sys
    from conftest import ROOT

    def run(backend):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000 + (7 if backend == "gloo" else 0)), os.path.join(ROOT, "bench.py"),
               "--gpus", "1", "--backend", backend, "--quick", "--no-cpu-baseline", "--no-extras", "--workload", workload]
        if workload == "ntt":
            cmd += ["--steps", "2", "--warmup", "1", "--logn", "14"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])

    line = run("nccl")
    assert line["n_gpus"] == 1 and line["value"] > 0
    c5 = line["c5"]
    assert c5["check"]["ok"] is True and c5["units_per_rank"] == [16]
    assert c5["check"]["rank0"]["batch_equals_single"] is True and c5["check"]["rank0"]["verifies"] is True
    if workload == "ntt":
        assert line["check"]["roundtrip_ok"] is True
    ri = line["ranks"]
    assert ri["backend"] == "nccl" and ri["world_size"] == 1 and ri["devices_distinct"] is True
    assert ri["ranks"][0]["device_index"] == 0 and ri["ranks"][0]["pci_bus_id"] and ri["ranks"][0]["uuid"]
    assert len(c5["per_rank_ms_per_step"]) == 1 and c5["check"]["rank0"]["delivered_equals_device"] is True
    other = run("gloo")
    assert other["c5"]["headers_sha256"] == c5["headers_sha256"] and other["ranks"]["backend"] == "gloo"


@pytest.mark.parametrize("env", [
    {"STARKHIP_NTT_RADICES": "7,7,6", "STARKHIP_XCD_SWZ": "2"},      # the three-pass plan for 2^20, every tile pass XCD-mapped
    {"STARKHIP_NTT_RADICES": "10,10", "STARKHIP_TILE_LOG_BIG": "12", "STARKHIP_XCD_SWZ": "0"},  # 4096-element tiles
    {"STARKHIP_NTT_RADICES": "11,9", "STARKHIP_TILE_LOG_BIG": "11"},  # radix 2^11: one column / one row per tile
    {"STARKHIP_NTT_RADICES": "6,6,4", "STARKHIP_TILE_LOG": "11"},     # 2^16 in three passes, 2048-element tiles
    {"STARKHIP_TILE_LOGS": "11,9,10", "STARKHIP_TW2_MAX_LOG": "20"},  # a tile size per pass; small row tables (lookup fallback)
    {"STARKHIP_NTT_NARROW_TILES": "100000000"},                       # every pass of radix <= 2^10 in the one-butterfly-per-thread form
    {"STARKHIP_NTT_NARROW_TILES": "100000000", "STARKHIP_NTT_RADICES": "5,5,4,3", "STARKHIP_TW2_MAX_LOG": "12"},  # ... four passes of 2^17, lookups
], ids=["7-7-6_swz2", "10-10_tile4096", "11-9", "6-6-4_tile2048", "per_pass_tiles", "narrow_everywhere", "narrow_5-5-4-3"])
def test_alternate_ntt_plans_parity(sa, env):
    """Every decomposition the plan / tile knobs can select gives the same bytes: the NTT golden vectors (reference digests to
    2^20), every size against the oracle and the 2^22 / 2^24 digests, in a child process with the knobs set.  (The plan / tile
    variants run with the narrow-launch form off, so that the tile-pass kernels also serve the small sizes; the last two force
    that form for every launch, the 2^24-point digests included.)"""
    env = dict(env)
    env.setdefault("STARKHIP_NTT_NARROW_TILES", "0")
    import subprocess, sys
    from conftest import ROOT
    sel = ("test_ntt_golden_vectors or test_ntt_every_size_vs_oracle or test_ntt_padding_and_batch or "
           "test_ntt_large_digests_vs_oracle_fixture or test_lde_golden or test_stark_batch_units_and_device_api")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m",
                          "gpu", "-k", sel, "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_random_plan_variants_vs_oracle(sa):
    """tools/stress_plans.py for a few seconds: random decompositions (digits 2..11, one to four passes), tile sizes and tile orders,
    each in a child process, random sizes / batches / zero padding / direction, every output against oracle/oracle.c.  (The
    full run -- 1005 variants in 7 minutes, all equal -- is recorded in DESIGN.md section 3.)"""
    import subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_plans.py"), "10"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "all outputs equal to the oracle" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_pinned_host_buffers_skip_staging(sa):
    """sh_host_alloc buffers go through the host-buffer entry points without the staging copy and give the same bytes."""
    n = 1 << 14
    w = root_of(n)
    data = wire(seeded(77, i) for i in range(n))
    want = sa.fft.ntt_bytes(data, n, w)
    src, dst = sa.lib.PinnedBuffer(32 * n), sa.lib.PinnedBuffer(32 * n)
    src.view[:] = data
    assert sa.fft.ntt_bytes(src, n, w, out=dst) is dst
    assert bytes(dst.view) == want
    back = sa.fft.ntt_bytes(dst, n, w, inverse=True)
    assert back == data
    src.close()
    dst.close()


def test_two_contexts_in_one_process(sa, oracle):
    """Contexts are independent (include/starkhip.h): a second sh_ctx on the same device, interleaved with the process-wide
    one, gives the same transforms and proofs; destroying it leaves the first one working."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    other = ctypes.c_void_p()
    assert L.sh_ctx_create(0, ctypes.byref(other)) == 0
    n = 1 << 12
    w = root_of(n).to_bytes(32, "big")
    data = wire(seeded(5, i) for i in range(n))
    want = oracle.c.fft_bytes(data, n, root_of(n))
    outs = []
    for c in (ctx, other, ctx, other):
        out = ctypes.create_string_buffer(32 * n)
        assert L.sh_ntt(c, data, n, out, n, w, 0) == 0
        outs.append(out.raw)
    assert all(o == want for o in outs)
    # a FRI commit on the second context while the first holds device-resident data
    dx = ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dx)) == 0
    assert L.sh_dev_fill_seeded(ctx, dx, n, 9) == 0
    steps = n // 8
    coeffs = wire(seeded(6, i) for i in range(steps))
    plen = int(L.sh_fri_proof_len(n, steps, 40))
    proof = ctypes.create_string_buffer(plen)
    assert L.sh_fri_prove(other, coeffs, steps, n, w, steps, 8, 40, 1, proof, plen) == 0
    assert proof.raw == oracle.c.fri_prove_flat(coeffs, root_of(n), steps, 8, 40, n=n)
    host = ctypes.create_string_buffer(32 * n)
    assert L.sh_dev_to_wire(ctx, dx, host, n) == 0
    assert host.raw == wire(seeded(9, i) for i in range(n))
    assert L.sh_ctx_trim(other) == 0
    L.sh_ctx_destroy(other)
    assert L.sh_dev_ntt(ctx, dx, dx, n, 1, w, 0) == 0
    assert L.sh_dev_to_wire(ctx, dx, host, n) == 0
    assert host.raw == oracle.c.fft_bytes(wire(seeded(9, i) for i in range(n)), n, root_of(n))
    assert L.sh_dev_free(ctx, dx) == 0


def test_async_download_delivers_what_the_device_holds(sa):
    """sh_dev_download_async: device -> page-locked host memory on the context's copy stream, ordered behind the work queued on the ctx
    stream; complete after sh_io_sync; a pageable destination is refused (it would turn the copy into a staged, synchronous one)."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << 16
    w = root_of(n).to_bytes(32, "big")
    dx = ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n, ctypes.byref(dx)) == 0
    host = sa.lib.PinnedBuffer(32 * n)
    want = ctypes.create_string_buffer(32 * n)
    for seed in (1, 2, 3):  # the copy waits for the fill + transform queued before it, and does not see the next round's
        assert L.sh_dev_fill_seeded(ctx, dx, n, seed) == 0
        assert L.sh_dev_ntt(ctx, dx, dx, n, 1, w, 0) == 0
        assert L.sh_dev_download_async(ctx, dx, host.ptr, 32 * n) == 0
        assert L.sh_io_sync(ctx) == 0
        assert L.sh_dev_download(ctx, dx, want, 32 * n) == 0
        assert bytes(host.view) == want.raw
    assert L.sh_dev_download_async(ctx, dx, want, 32 * n) == -1  # pageable
    assert L.sh_dev_download_async(ctx, dx, host.ptr, 0) == 0 and L.sh_io_sync(ctx) == 0
    host.close()
    assert L.sh_dev_free(ctx, dx) == 0


def test_pinned_buffer_outlives_its_context():
    """A page-locked buffer belongs to the process: sh_host_free(NULL, ptr) releases it after the allocating context is gone
    (what PinnedBuffer.close does after _lib.close()), and a double close is harmless."""
    import ctypes, subprocess, sys
    from conftest import ROOT
    code = (
        "import ctypes\n"
        "from starks_amd import _lib\n"
        "L = _lib.lib()\n"
        "b = _lib.PinnedBuffer(1 << 20)\n"
        "b.view[:4] = b'abcd'\n"
        "p = ctypes.c_void_p()\n"
        "assert L.sh_host_alloc(_lib.ctx(), 4096, ctypes.byref(p)) == 0\n"
        "_lib.close()\n"
        "assert L.sh_host_free(None, p) == 0\n"
        "assert L.sh_host_free(None, None) == 0\n"
        "b.close(); b.close()\n"
        "assert b.ptr is None and b.view is None\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_two_contexts_running_at_once_give_the_single_context_bytes(sa):
    """Two library contexts (two streams) proving / hashing at the same time give, byte for byte, what one context gives alone.
    Regression test of round 4: the Merkle mid kernel handed its nodes over per wave through LDS slices of two different
    geometries, so a wave that was ahead wrote into the slice its neighbour was still reading -- invisible while the waves of a
    workgroup run in step (one stream), 5-30 % wrong proofs as soon as a second stream perturbed them (bench.py deals the batched
    launches of config 5 to two contexts).  Every proof and every tree is compared, not a sample."""
    import ctypes
    from starks_amd import batch
    L = sa.lib.lib()
    steps, ext, k = 1 << 13, 8, 64
    a = batch.StarkUnitProver(steps, ext, chunk=k)
    b = batch.StarkUnitProver(steps, ext, chunk=k, second_context=True)
    try:
        a.generate(0, k)
        a.prove(k)
        ref_a = a.download(k)
        a.generate(k, k)
        a.prove(k)
        ref_b = a.download(k)      # units k .. 2k-1 proved on the FIRST context, alone
        a.generate(0, k)
        b.generate(k, k)
        a.status()
        b.status()
        bad = 0
        for _ in range(6):
            for _ in range(3):     # both streams stay busy: three launch sequences each, interleaved
                a.prove(k)
                b.prove(k)
            got_a, got_b = a.download(k), b.download(k)
            bad += sum(1 for x, y in zip(ref_a + ref_b, got_a + got_b) if x != y)
        assert bad == 0, "%d of %d proofs differ between concurrent and single-context runs" % (bad, 6 * 2 * k)
    finally:
        a.close()
        b.close()
    # the same for a bare Merkle commit wide enough for the mid kernels (2^22 leaves), both contexts hashing different data
    ctx, other = sa.lib.ctx(), sa.lib.second_ctx()
    n = 1 << 22
    bufs = []
    for c, seed in ((ctx, 11), (other, 12)):
        dx, dt = ctypes.c_void_p(), ctypes.c_void_p()
        assert L.sh_dev_alloc(c, 32 * n, ctypes.byref(dx)) == 0 and L.sh_dev_alloc(c, 64 * n, ctypes.byref(dt)) == 0
        assert L.sh_dev_fill_seeded(c, dx, n, seed) == 0
        bufs.append((c, dx, dt))

    def tree_digest(c, dt):
        host = ctypes.create_string_buffer(64 * n)
        assert L.sh_dev_download(c, dt, host, 64 * n) == 0
        return hashlib.sha256(host.raw).digest()

    try:
        alone = []
        for c, dx, dt in bufs:
            assert L.sh_dev_merkelize(c, dx, n, 1, dt) == 0
            alone.append(tree_digest(c, dt))
        for _ in range(4):
            for _ in range(8):
                for c, dx, dt in bufs:
                    assert L.sh_dev_merkelize(c, dx, n, 1, dt) == 0
            assert [tree_digest(c, dt) for c, dx, dt in bufs] == alone
    finally:
        for c, dx, dt in bufs:
            L.sh_dev_free(c, dx)
            L.sh_dev_free(c, dt)


def test_parity_holds_beside_a_busy_second_stream(sa, oracle):
    """The parity checks again while a host thread keeps a SECOND context busy with transforms and Merkle commits: kernels of two
    streams share the CUs, so the waves of a workgroup no longer run in step -- whatever depends on that timing (round 4: the mid
    kernel's overlapping per-wave LDS slices) shows here.  Compared: NTT digests of every plan shape against the committed
    fixtures, Merkle trees of the wide (asm) and narrow kernels against the oracle, the reference's FRI proof of config 3 and its
    STARK proofs, each several times."""
    import ctypes, threading
    L, ctx, other = sa.lib.lib(), sa.lib.ctx(), sa.lib.second_ctx()
    stop, errors = threading.Event(), []

    def noise():
        try:
            n, m = 1 << 20, 1 << 22
            w = root_of(n).to_bytes(32, "big")
            dx, dt = ctypes.c_void_p(), ctypes.c_void_p()
            assert L.sh_dev_alloc(other, 32 * m, ctypes.byref(dx)) == 0 and L.sh_dev_alloc(other, 64 * m, ctypes.byref(dt)) == 0
            assert L.sh_dev_fill_seeded(other, dx, m, 99) == 0
            k = 0
            while not stop.is_set():
                assert L.sh_dev_ntt(other, dx, dx, n, 4, w, k & 1) == 0
                assert L.sh_dev_merkelize(other, dx, m >> (k % 3), 1, dt) == 0
                k += 1
                if k % 8 == 0:
                    assert L.sh_sync(other) == 0  # bounded queue depth
            assert L.sh_sync(other) == 0
            L.sh_dev_free(other, dx)
            L.sh_dev_free(other, dt)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    th = threading.Thread(target=noise)
    th.start()
    try:
        for logn in (19, 21, 22, 23):
            test_ntt_large_digests_vs_oracle_fixture(sa, logn)
        for rep in range(3):
            for logn in (13, 20, 22):
                test_merkle_sizes_vs_oracle(sa, oracle, logn)
        fri_recs = [r for r in load_golden("fri.json") if r["name"] in ("fri_mimc_2_14",)] or load_golden("fri.json")[-1:]
        for rep in range(3):
            for rec in fri_recs:
                test_fri_proofs_golden(sa, oracle, rec)
        for c in load_golden("stark.json"):
            test_stark_proofs_golden(sa, oracle, c)
        test_fri_batch_and_oracle(sa, oracle)
    finally:
        stop.set()
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("n_coeffs,logn,batch", [(256, 11, 3), (100, 11, 2), (1, 6, 1), (64, 6, 2), (4096, 15, 2), (5000, 16, 1)])
def test_device_fri_commit_from_short_coefficient_vectors(sa, oracle, n_coeffs, logn, batch):
    """sh_dev_fri_prove_coeffs: [batch][n_coeffs] device-resident coefficients (any count up to n, powers of two or not) stand for
    their zero-padded extension; the proofs equal the C oracle's on the padded polynomials and sh_dev_fri_prove's on the padded
    vectors."""
    import ctypes
    L, ctx = sa.lib.lib(), sa.lib.ctx()
    n = 1 << logn
    g2 = root_of(n)
    w = g2.to_bytes(32, "big")
    md = max(32, 1 << (n_coeffs - 1).bit_length()) if n_coeffs * 4 <= n else n // 4
    md = min(md, n // 2)
    polys = [[seeded(77 + b, i) for i in range(n_coeffs)] for b in range(batch)]
    plen = sa.fri.proof_len(n, md, 40)
    ds, dd, dp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.sh_dev_alloc(ctx, 32 * n_coeffs * batch, ctypes.byref(ds)) == 0
    assert L.sh_dev_alloc(ctx, 32 * n * batch, ctypes.byref(dd)) == 0 and L.sh_dev_alloc(ctx, plen * batch, ctypes.byref(dp)) == 0
    assert L.sh_dev_from_wire(ctx, b"".join(wire(p) for p in polys), ds, n_coeffs * batch) == 0
    assert L.sh_dev_from_wire(ctx, b"".join(wire(p + [0] * (n - n_coeffs)) for p in polys), dd, n * batch) == 0
    out = []
    for call in (lambda: L.sh_dev_fri_prove_coeffs(ctx, ds, n_coeffs, n, w, md, 8, 40, batch, dp),
                 lambda: L.sh_dev_fri_prove(ctx, dd, n, w, md, 8, 40, batch, dp)):
        assert call() == 0
        host = ctypes.create_string_buffer(plen * batch)
        assert L.sh_dev_download(ctx, dp, host, plen * batch) == 0
        out.append(host.raw)
    assert out[0] == out[1]
    for b in range(batch):
        assert out[0][b * plen:(b + 1) * plen] == oracle.c.fri_prove_flat(wire(polys[b]), g2, md, 8, 40, n=n), b
    assert L.sh_dev_fri_prove_coeffs(ctx, ds, 0, n, w, md, 8, 40, batch, dp) == -1       # SH_ERR_INVALID
    assert L.sh_dev_fri_prove_coeffs(ctx, ds, n + 1, n, w, md, 8, 40, batch, dp) == -1
    for d in (ds, dd, dp):
        assert L.sh_dev_free(ctx, d) == 0
