"""ctypes loader for oracle/liboracle.so (the C restatement, oracle/oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by starks_amd/."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        u8p, u64, u32, i32 = ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        L.or_blake2s.argtypes = [u8p, u64, u8p]
        L.or_blake2s.restype = None
        L.or_fft.argtypes = [u8p, u64, u8p, i32, u8p, u64]
        L.or_fft.restype = i32
        L.or_power_cycle.argtypes = [u8p, u64, u8p]
        L.or_power_cycle.restype = i32
        L.or_merkelize.argtypes = [u8p, u64, u8p]
        L.or_merkelize.restype = None
        L.or_mk_branch.argtypes = [u8p, u64, u64, u8p]
        L.or_mk_branch.restype = u64
        L.or_pseudorandom_indices.argtypes = [u8p, u32, u32, u32, ctypes.POINTER(u32)]
        L.or_pseudorandom_indices.restype = i32
        L.or_fold.argtypes = [u8p, u64, u8p, u8p, u8p]
        L.or_fold.restype = i32
        L.or_fri_prove.argtypes = [u8p, u64, u8p, u64, u32, u32, u8p, u64]
        L.or_fri_prove.restype = ctypes.c_int64
        L.or_lde.argtypes = [u8p, u64, u32, u8p, u8p]
        L.or_lde.restype = i32
        L.or_field_op.argtypes = [i32, u8p, u8p, u64, u8p]
        L.or_field_op.restype = None
        _LIB = L
    return _LIB


def _wire(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


def _unwire(buf):
    return [int.from_bytes(buf[i:i + 32], "big") for i in range(0, len(buf), 32)]


def blake2s(msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().or_blake2s(msg, len(msg), out)
    return out.raw


def fft_bytes(data: bytes, n: int, w: int, inverse=False) -> bytes:
    out = ctypes.create_string_buffer(32 * n)
    rc = lib().or_fft(data, len(data) // 32, int(w).to_bytes(32, "big"), int(inverse), out, n)
    if rc:
        raise ValueError("or_fft failed: %d" % rc)
    return out.raw


def fft(vals, n, w, inverse=False):
    return _unwire(fft_bytes(_wire(vals), n, w, inverse))


def power_cycle(w, n):
    out = ctypes.create_string_buffer(32 * n)
    rc = lib().or_power_cycle(int(w).to_bytes(32, "big"), n, out)
    if rc:
        raise ValueError("w^n != 1")
    return _unwire(out.raw)


def merkelize_bytes(leaves: bytes) -> bytes:
    n = len(leaves) // 32
    out = ctypes.create_string_buffer(64 * n)
    lib().or_merkelize(leaves, n, out)
    return out.raw


def merkelize(vals):
    """-> list of 2n nodes like the reference (nodes[0] = b'')."""
    raw = merkelize_bytes(_wire(vals))
    nodes = [raw[i:i + 32] for i in range(0, len(raw), 32)]
    nodes[0] = b""
    return nodes


def mk_branch_bytes(nodes: bytes, index: int):
    n = len(nodes) // 64
    out = ctypes.create_string_buffer(32 * 64)
    k = lib().or_mk_branch(nodes, n, index, out)
    return [out.raw[32 * i:32 * i + 32] for i in range(k)]


def pseudorandom_indices(entropy: bytes, modulus, count, exclude=0):
    out = (ctypes.c_uint32 * count)()
    rc = lib().or_pseudorandom_indices(entropy, modulus, count, exclude, out)
    assert rc == 0, "modulus must be < 2**24"
    return list(out)


def fold(vals, w, special_x_bytes: bytes):
    n = len(vals)
    out = ctypes.create_string_buffer(8 * n)
    lib().or_fold(_wire(vals), n, int(w).to_bytes(32, "big"), special_x_bytes, out)
    return _unwire(out.raw)


def fri_flat_len(n, maxdeg_plus_1, samples=40):
    """Length in bytes of the flat proof for domain size n (layout in oracle.c:fri_rec)."""
    total, first = 0, True
    while maxdeg_plus_1 > 16:
        lg = n.bit_length() - 1
        s = samples if first else 40
        total += 32 + s * 32 * ((lg - 1) + 4 * (lg + 1))
        n //= 4
        maxdeg_plus_1 //= 4
        first = False
    return total + 32 * n


def fri_prove_flat(coeff_bytes: bytes, w, maxdeg_plus_1, exclude=0, samples=40, n=None) -> bytes:
    if n is None:
        n, t = 1, int(w)
        P = 2**256 - 2**32 * 351 + 1
        while t != 1:
            t = t * t % P
            n *= 2
    cap = fri_flat_len(n, maxdeg_plus_1, samples)
    out = ctypes.create_string_buffer(cap)
    got = lib().or_fri_prove(coeff_bytes, len(coeff_bytes) // 32, int(w).to_bytes(32, "big"), maxdeg_plus_1,
                             exclude, samples, out, cap)
    if got < 0:
        raise ValueError("or_fri_prove failed: %d" % got)
    assert got == cap, (got, cap)
    return out.raw


def lde_bytes(trace: bytes, ext, g2) -> bytes:
    steps = len(trace) // 32
    out = ctypes.create_string_buffer(32 * steps * ext)
    rc = lib().or_lde(trace, steps, ext, int(g2).to_bytes(32, "big"), out)
    if rc:
        raise ValueError("or_lde failed: %d" % rc)
    return out.raw


def field_op(op, a_vals, b_vals):
    n = len(a_vals)
    out = ctypes.create_string_buffer(32 * n)
    lib().or_field_op({"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], _wire(a_vals), _wire(b_vals), n, out)
    return _unwire(out.raw)
