"""starks/utils.py call sites on the hot path: get_power_cycle (GPU), get_pseudorandom_indices (host;
the FRI prover samples on the device with the same algorithm), mimc."""
import ctypes

from . import _lib
from ._lib import MIMC_P
from .merkle_tree import blake
from .wireseq import WireList


def get_power_cycle(r, field):
    """utils.py:30-38: [1, r, r^2, ...] until the powers return to 1.  The MiMC field with a root of power-of-two order -- every
    domain of the proving path (stark.py:223, fri.py:211) -- on the GPU; another field or order (the reference's unit test walks a
    6th root of unity mod 31, test_utils.py:20-30) is outside the hot path and walked here as the reference walks it (orders up to
    2^12, like fft._host_dft)."""
    n = _lib.order_of_root(r) if int(field.p) == MIMC_P else None
    if n is None:
        r = field(r)
        out, one = [field(1)], field(1)
        while True:
            nxt = out[-1] * r
            if nxt == one:
                return out
            if len(out) >= 1 << 12:
                raise NotImplementedError("host power cycle: order above 2^12 (use a power-of-two order in the MiMC field on the GPU)")
            out.append(nxt)
    out = ctypes.create_string_buffer(32 * n)
    _lib.check(_lib.lib().sh_power_cycle(_lib.ctx(), int(r).to_bytes(32, "big"), n, out), "sh_power_cycle")
    return WireList(out.raw, field)  # lazy: elements are created on index / iteration (wireseq.py)


def get_pseudorandom_indices(entropy, modulus, count, exclude_multiples_of=0):
    """utils.py:60-90 (Fiat-Shamir query positions).  Host restatement for verifiers / callers; the
    prover's device kernel (csrc/kernels.hip:sample_indices_kernel) produces the same numbers."""
    assert modulus < 2**24
    data = entropy
    while len(data) < 4 * count:
        data += blake(data[-32:])
    words = [int.from_bytes(data[i:i + 4], "big") for i in range(0, count * 4, 4)]
    if exclude_multiples_of == 0:
        return [w % modulus for w in words]
    real_modulus = modulus * (exclude_multiples_of - 1) // exclude_multiples_of
    return [(w % real_modulus) + 1 + (w % real_modulus) // (exclude_multiples_of - 1) for w in words]


def mimc(inp, steps, round_constants):
    """utils.py:20-27 (input generator of the benchmark configurations)."""
    for i in range(steps - 1):
        inp = (inp**3 + round_constants[i % len(round_constants)]) % MIMC_P
    return inp


def mimc_trace(t0, steps):
    """All intermediate states of mimc() with the constants of test_fri.py:112 (k_i = i^7 xor 42, 64 of them)."""
    ks = [(i**7) ^ 42 for i in range(64)]
    out = [t0 % MIMC_P]
    for i in range(steps - 1):
        out.append((out[-1]**3 + ks[i % 64]) % MIMC_P)
    return out


def generate_Xi_s(field, width):
    """utils.py:40-56: the index polynomials X_1 .. X_width"""
    from .multivariate_polynomial import generate_Xi_s as gen
    return gen(field, width)


def is_a_power_of_2(x):
    """utils.py:93-94"""
    return x >= 1 and x & (x - 1) == 0


def plus_one(num):
    """utils.py:11-12 (the reference's typed-function example, test_utils.py:32-34)"""
    return num + 1
