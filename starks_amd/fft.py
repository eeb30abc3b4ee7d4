"""NTT / inverse NTT on the MI355X behind the reference's call sites (starks/fft.py:256-272, 316-345):

    fft_1d(field, vals, modulus, root_of_unity, inv=False) -> list
    NonBinaryFFT(field, root_of_unity).fft(poly) -> list ; .inv_fft(values) -> Poly
    mul_polys(a, b, root_of_unity) -> list            (unscaled, exactly like the reference)

Supported: the MiMC prime p = 2^256 - 351*2^32 + 1 and roots of unity of power-of-two order (every STARK
call site of the reference: stark.py:31-34,217-225,253-256, fri.py:207-208,260-261).  Anything else raises
NotImplementedError -- there is deliberately no CPU fallback.
"""
import ctypes

from . import _lib
from ._lib import MIMC_P
from .modp import IntegersModP
from .polynomial import polynomials_over


def _check_field(modulus):
    if int(modulus) != MIMC_P:
        raise NotImplementedError(
            "starks_amd accelerates Z/p for the MiMC prime 2^256 - 351*2^32 + 1 only (got modulus %d)" % int(modulus))


def _order(root_of_unity):
    n = _lib.order_of_root(root_of_unity)
    if n is None:
        raise NotImplementedError("root_of_unity must have power-of-two order (<= 2^32) in the MiMC field")
    return n


def ntt_bytes(data, n, root_of_unity, inverse=False, batch=1):
    """Wire-form fast path: `data` = batch * n_in 32-byte big-endian values -> batch * n outputs."""
    n_in = len(data) // (32 * batch)
    if n_in > n:
        raise ValueError("more input values (%d) than the order of the root of unity (%d)" % (n_in, n))
    out = ctypes.create_string_buffer(32 * n * batch)
    rc = _lib.lib().sh_ntt_batch(_lib.ctx(), data, n_in, out, n, batch, int(root_of_unity).to_bytes(32, "big"),
                                 1 if inverse else 0)
    _lib.check(rc, "sh_ntt_batch")
    return out.raw


def fft_1d(field, vals, modulus, root_of_unity, inv=False):
    """starks/fft.py:316-331 -- the transform length is the order of root_of_unity; `vals` is zero-padded."""
    _check_field(modulus)
    n = _order(root_of_unity)
    vals = list(vals)
    out = ntt_bytes(_lib.to_wire(vals), n, int(root_of_unity), inverse=inv)
    return [field(x) for x in _lib.from_wire(out)]


class FFT(object):
    pass


class NonBinaryFFT(FFT):
    """starks/fft.py:256-272"""

    def __init__(self, field, root_of_unity):
        self.field = field
        self.root_of_unity = root_of_unity
        self.polysOver = polynomials_over(field).factory

    def fft(self, poly):
        coeffs = poly.coefficients if hasattr(poly, "coefficients") else list(poly)
        return fft_1d(self.field, coeffs, self.field.p, self.root_of_unity, inv=False)

    def inv_fft(self, values):
        coeffs = fft_1d(self.field, values, self.field.p, self.root_of_unity, inv=True)
        return self.polysOver(coeffs)


def mul_polys(a, b, root_of_unity):
    """starks/fft.py:334-345: returns n * (a*b) -- the reference omits the 1/n of the inverse transform."""
    field = None
    for v in list(a) + list(b) + [root_of_unity]:
        if hasattr(v, "p"):
            field = type(v)
            break
    if field is None:
        field = IntegersModP(MIMC_P)
    _check_field(field.p)
    n = _order(root_of_unity)
    a, b = list(a), list(b)
    if len(a) > n or len(b) > n:
        raise ValueError("operand longer than the order of the root of unity")
    out = ctypes.create_string_buffer(32 * n)
    rc = _lib.lib().sh_mul_polys(_lib.ctx(), _lib.to_wire(a), len(a), _lib.to_wire(b), len(b), out, n,
                                 int(root_of_unity).to_bytes(32, "big"))
    _lib.check(rc, "sh_mul_polys")
    return [field(x) for x in _lib.from_wire(out.raw)]


def low_degree_extension(field, trace_columns, extension_factor, G2):
    """The LDE step of STARK.mk_proof (stark.py:27-36 + 253-256): per column, inverse NTT over
    G1 = G2^extension_factor, then NTT over G2.  trace_columns: list of equal-length lists."""
    _check_field(field.p)
    cols = [list(c) for c in trace_columns]
    steps = len(cols[0])
    if any(len(c) != steps for c in cols):
        raise ValueError("trace columns must have equal length")
    n = steps * extension_factor
    if _order(G2) != n:
        raise ValueError("G2 must have order steps * extension_factor")
    out = ctypes.create_string_buffer(32 * n * len(cols))
    data = b"".join(_lib.to_wire(c) for c in cols)
    rc = _lib.lib().sh_lde(_lib.ctx(), data, out, steps, extension_factor, len(cols), int(G2).to_bytes(32, "big"))
    _lib.check(rc, "sh_lde")
    flat = _lib.from_wire(out.raw)
    return [[field(x) for x in flat[c * n:(c + 1) * n]] for c in range(len(cols))]
