"""Pins oracle/oracle.c (the C restatement, also the timed CPU baseline) to the reference-generated
fixtures and cross-checks it against oracle/pyoracle.py on random inputs.  CPU only."""
import hashlib
import os
import random
import struct
import subprocess

import pytest

from oracle import coracle as co
from oracle import pyoracle as po
from conftest import load_golden, GOLDEN, ROOT

P = po.MIMC_P


def h2i(s):
    return int(s, 16)


def wire(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


def seeded(seed, i):
    return int.from_bytes(hashlib.blake2s(struct.pack("<QQ", seed, i)).digest(), "big") % P


def root_of(n):
    return pow(7, (P - 1) // n, P)


def test_blake2s_rfc7693():
    for c in load_golden("merkle.json")["blake2s"]:
        assert co.blake2s(bytes.fromhex(c["msg"])).hex() == c["digest"]
    # RFC 7693 appendix B test vector
    assert co.blake2s(b"abc").hex() == "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"
    rng = random.Random(1)
    for n in (1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000):
        m = bytes(rng.randrange(256) for _ in range(n))
        assert co.blake2s(m) == hashlib.blake2s(m).digest()


def test_field_golden():
    g = load_golden("field.json")
    a = [h2i(c["a"]) for c in g["cases"]]
    b = [h2i(c["b"]) for c in g["cases"]]
    assert co.field_op("add", a, b) == [h2i(c["add"]) for c in g["cases"]]
    assert co.field_op("sub", a, b) == [h2i(c["sub"]) for c in g["cases"]]
    assert co.field_op("mul", a, b) == [h2i(c["mul"]) for c in g["cases"]]
    ia = [h2i(c["a"]) for c in g["inverse"]]
    assert co.field_op("inv", ia, ia) == [h2i(c["inv"]) for c in g["inverse"]]
    # unreduced wire inputs (modp.py:33-34): 2^256-1 and p itself
    assert co.field_op("mul", [2**256 - 1, P], [1, 1]) == [(2**256 - 1) % P, 0]


def test_field_random_vs_python():
    rng = random.Random(7)
    a = [rng.randrange(2**256) for _ in range(2000)] + [P - 1, P, P + 1, 2**256 - 1, 0, 1]
    b = [rng.randrange(2**256) for _ in range(2000)] + [P - 1, 2**256 - 1, 2**256 - 1, 2**256 - 1, 0, P - 1]
    assert co.field_op("add", a, b) == [(x + y) % P for x, y in zip(a, b)]
    assert co.field_op("sub", a, b) == [(x - y) % P for x, y in zip(a, b)]
    assert co.field_op("mul", a, b) == [(x * y) % P for x, y in zip(a, b)]


def test_ntt_golden_all_sizes():
    g = load_golden("ntt.json")
    for c in g["cases"]:
        n, n_in, w = c["n"], c["n_in"], h2i(c["w"])
        if n > 2**16:
            continue
        data = wire(seeded(g["seed"], i) for i in range(n_in))
        fwd = co.fft_bytes(data, n, w)
        inv = co.fft_bytes(data, n, w, inverse=True)
        assert hashlib.sha256(fwd).hexdigest() == c["sha_fwd"]
        assert hashlib.sha256(inv).hexdigest() == c["sha_inv"]
        assert co.fft_bytes(fwd, n, w, inverse=True) == data + bytes(32 * (n - n_in))
    m = g["mimc_n8_0123"]
    assert co.fft([0, 1, 2, 3], 8, h2i(m["w"])) == [h2i(v) for v in m["fwd"]]
    with pytest.raises(ValueError):
        co.fft([1, 2], 8, root_of(16))  # order of w is not n


def test_merkle_golden():
    g = load_golden("merkle.json")
    t = co.merkelize(list(range(128)))
    assert t[1].hex() == g["range128"]["root"]
    assert hashlib.sha256(b"".join(t)).hexdigest() == g["range128"]["tree_sha"]
    raw = co.merkelize_bytes(wire(range(128)))
    assert [x.hex() for x in co.mk_branch_bytes(raw, 59)] == g["range128"]["branch59"]
    for c in g["seeded"]:
        vals = [seeded(c["seed"], i) for i in range(c["n"])]
        raw = co.merkelize_bytes(wire(vals))
        assert raw[32:64].hex() == c["root"]
        assert hashlib.sha256(raw[32:]).hexdigest() == c["tree_sha"]
        for i, br in c["branches"].items():
            assert [x.hex() for x in co.mk_branch_bytes(raw, int(i))] == br


def test_utils_golden():
    g = load_golden("utils.json")
    for c in g["pseudorandom_indices"]:
        assert co.pseudorandom_indices(bytes.fromhex(c["entropy"]), c["modulus"], c["count"], c["exclude"]) == c["out"]
    assert [v.to_bytes(32, "big").hex() for v in co.power_cycle(root_of(8), 8)] == g["power_cycle_w8"]
    with pytest.raises(ValueError):
        co.power_cycle(root_of(8), 4)


def test_fold_golden():
    for c in load_golden("fold.json"):
        values = [seeded(c["seed"], i) for i in range(c["n"])]
        col = co.fold(values, h2i(c["w"]), bytes.fromhex(c["special_x_bytes"]))
        assert hashlib.sha256(wire(col)).hexdigest() == c["column_sha"]


def test_lde_golden():
    for c in load_golden("lde.json"):
        tr = po.mimc_trace(c["trace_t0"], c["steps"])
        ext = co.lde_bytes(wire(tr), c["ext"], h2i(c["g2"]))
        assert hashlib.sha256(ext).hexdigest() == c["lde_sha"]


def _fri_coeffs(rec):
    d = rec["coeffs"]
    if d.startswith("(i**7)^42"):
        return [(i**7) ^ 42 for i in range(rec["n_coeffs"])]
    if d == "i, i<256":
        return list(range(256))
    if d == "i+1, i<16":
        return list(range(1, 17))
    k = int(d.split("2^")[1].rstrip("))"))
    coeffs = co.fft(po.mimc_trace(3, 2**k), 2**k, pow(h2i(rec["w"]), 8, P), inverse=True)
    return po.strip_trailing_zeros(coeffs)


@pytest.mark.parametrize("rec", load_golden("fri.json"), ids=lambda r: r["name"])
def test_fri_flat_golden(rec):
    """Every reference-generated FRI proof, including the 2^14-step MiMC case (config C3)."""
    coeffs = _fri_coeffs(rec)
    assert len(coeffs) == rec["n_coeffs"]
    flat = co.fri_prove_flat(wire(coeffs), h2i(rec["w"]), rec["maxdeg_plus_1"], rec["exclude_multiples_of"], rec["samples"])
    assert len(flat) == rec["flat_len"]
    assert hashlib.sha256(flat).hexdigest() == rec["flat_sha"]
    path = os.path.join(GOLDEN, rec["name"] + ".flat.bin")
    if os.path.exists(path):
        assert open(path, "rb").read() == flat
    # round roots appear at the predicted offsets
    off, n, first = 0, rec["rounds"][0]["n"] if rec["rounds"] else 0, True
    for r in rec["rounds"]:
        assert flat[off:off + 32].hex() == r["root_m2"]
        lg = r["n"].bit_length() - 1
        off += 32 + (rec["samples"] if first else 40) * 32 * ((lg - 1) + 4 * (lg + 1))
        first = False
    assert flat[off:] == b"".join(bytes.fromhex(v) for v in rec["final_values"])


@pytest.mark.parametrize("logsteps", [14, 16])
def test_fri_large_fixture_is_what_the_oracle_writes(logsteps):
    """tests/golden/fri_large.json (the commits bench.py times; the GPU suite compares the library with it) regenerates from the
    oracle: the 2^14- and 2^16-step cases here (0.5 s and 2 s), the 2^20-step one (74 s) only in generate_large.py --fri."""
    import struct
    c = [c for c in load_golden("fri_large.json")["cases"] if c["logsteps"] == logsteps][0]
    coeffs = b"".join((int.from_bytes(hashlib.blake2s(struct.pack("<QQ", c["seed"], i)).digest(), "big") % P).to_bytes(32, "big")
                      for i in range(c["steps"]))
    assert hashlib.sha256(coeffs).hexdigest() == c["coeffs_sha256"]
    flat = co.fri_prove_flat(coeffs, int(c["w"], 16), c["maxdeg_plus_1"], c["exclude_multiples_of"], c["samples"], n=c["domain"])
    assert len(flat) == c["proof_bytes"] and hashlib.sha256(flat).hexdigest() == c["proof_sha256"]
    assert flat[:32].hex() == c["first_root2"]


def test_sanitized_build():
    """-fsanitize=address,undefined build of the oracle runs the FRI path clean."""
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    code = (
        "import ctypes,sys\n"
        "L=ctypes.CDLL(%r)\n"
        "P=2**256-2**32*351+1\n"
        "w=pow(7,(P-1)//512,P).to_bytes(32,'big')\n"
        "c=b''.join(((i**7)^42).to_bytes(32,'big') for i in range(512))\n"
        "out=ctypes.create_string_buffer(1<<20)\n"
        "L.or_fri_prove.restype=ctypes.c_int64\n"
        "L.or_fri_prove.argtypes=[ctypes.c_char_p,ctypes.c_uint64,ctypes.c_char_p,ctypes.c_uint64,ctypes.c_uint32,ctypes.c_uint32,ctypes.c_char_p,ctypes.c_uint64]\n"
        "n=L.or_fri_prove(c,512,w,512,0,40,out,1<<20)\n"
        "import hashlib;print(n,hashlib.sha256(out.raw[:n]).hexdigest())\n" % so)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.check_output(["python3", "-c", code], env=env).decode().split()
    rec = [r for r in load_golden("fri.json") if r["name"] == "fri_deg512"][0]
    assert int(out[0]) == rec["flat_len"] and out[1] == rec["flat_sha"]
