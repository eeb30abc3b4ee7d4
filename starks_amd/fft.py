"""NTT / inverse NTT on the MI355X behind the reference's call sites (starks/fft.py:256-272, 316-345):

    fft_1d(field, vals, modulus, root_of_unity, inv=False) -> list
    NonBinaryFFT(field, root_of_unity).fft(poly) -> list ; .inv_fft(values) -> Poly
    mul_polys(a, b, root_of_unity) -> list            (unscaled, exactly like the reference)

The hot path -- the MiMC prime p = 2^256 - 351*2^32 + 1 with roots of unity of power-of-two order, i.e. every STARK
call site of the reference (stark.py:31-34,217-225,253-256, fri.py:207-208,260-261) -- ALWAYS runs on the GPU through
libstarkhip.so and raises when the library or the device is missing: there is no CPU fallback for it.

Inputs the device code cannot represent at all -- another modulus (the reference's unit tests use Z/31,
test_fft.py:98-113,132-149) or a root whose order is not a power of two (n = 6 there) -- are outside the hot path; for
those `fft_1d` evaluates the transform directly on the host (`_host_dft`: Horner evaluation at every power of the
root, O(n^2) on field elements, orders up to 2^12), so that the reference's unit tests run unchanged through this
module.  It is never used for the MiMC field with a power-of-two order.
"""
import ctypes

from . import _lib
from ._lib import MIMC_P
from .modp import IntegersModP
from .polynomial import polynomials_over
from .wireseq import WireList


def _check_field(modulus):
    if int(modulus) != MIMC_P:
        raise NotImplementedError(
            "starks_amd accelerates Z/p for the MiMC prime 2^256 - 351*2^32 + 1 only (got modulus %d)" % int(modulus))


def _order(root_of_unity):
    n = _lib.order_of_root(root_of_unity)
    if n is None:
        raise NotImplementedError("root_of_unity must have power-of-two order (<= 2^32) in the MiMC field")
    return n


def ntt_bytes(data, n, root_of_unity, inverse=False, batch=1, out=None):
    """Wire-form fast path: `data` = batch * n_in 32-byte big-endian values -> batch * n outputs.  `data` may be bytes
    or a `_lib.PinnedBuffer`; with `out` (a PinnedBuffer of 32 * n * batch bytes) the result is written there and `out`
    is returned -- both ends then skip the staging copy (include/starkhip.h, host-buffer API)."""
    pinned_in = isinstance(data, _lib.PinnedBuffer)
    nbytes = data.nbytes if pinned_in else len(data)
    n_in = nbytes // (32 * batch)
    if n_in > n:
        raise ValueError("more input values (%d) than the order of the root of unity (%d)" % (n_in, n))
    src = data.ptr if pinned_in else ctypes.cast(ctypes.c_char_p(bytes(data) if not isinstance(data, bytes) else data), ctypes.c_void_p)
    dst = out.ptr if out is not None else ctypes.create_string_buffer(32 * n * batch)
    if out is not None and out.nbytes < 32 * n * batch:
        raise ValueError("output buffer too small")
    rc = _lib.lib().sh_ntt_batch(_lib.ctx(), src, n_in, dst, n, batch, int(root_of_unity).to_bytes(32, "big"),
                                 1 if inverse else 0)
    _lib.check(rc, "sh_ntt_batch")
    return out if out is not None else dst.raw


def _elements(field, raw):
    """device output (wire form, canonical residues) -> a lazy sequence of elements of the caller's field type (wireseq.py):
    nothing is converted until somebody indexes or iterates, and the next stage takes the bytes as they are"""
    return WireList(raw, field)


def _on_device(modulus, root_of_unity):
    return int(modulus) == MIMC_P and _lib.order_of_root(root_of_unity) is not None


_HOST_MAX_ORDER = 1 << 12


def _host_dft(field, vals, modulus, root_of_unity, inv=False):
    """Transform over a field / order the device code does not cover (another modulus, or an order that is not a power
    of two -- the reference's Z/31 unit tests, test_fft.py:98-113,132-149): never the hot path, so no fast algorithm --
    value k of the result is the input polynomial evaluated at root^k by Horner's rule (root^-k for the inverse, then
    scaled by 1/n).  Every quantity is an exact residue, so the result equals fft_1d's (starks/fft.py:316-331) whatever
    recursion that uses.  Element arithmetic is the field type's own."""
    g = field(root_of_unity)
    one = field(1)
    order, t = 1, g
    while t != one:
        t = t * g
        order += 1
        if order > _HOST_MAX_ORDER:
            raise NotImplementedError("host transform: root order above %d (use the MiMC field on the GPU)" % _HOST_MAX_ORDER)
    coeffs = [field(v) for v in vals]
    if len(coeffs) > order:
        raise ValueError("more input values (%d) than the order of the root of unity (%d)" % (len(coeffs), order))
    step = g ** (order - 1) if inv else g          # g^-1 = g^(order-1)
    scale = field(order) ** (int(modulus) - 2) if inv else None
    out, point = [], one
    for _ in range(order):
        acc = field(0)
        for c in reversed(coeffs):
            acc = acc * point + c
        out.append(acc * scale if inv else acc)
        point = point * step
    return out


def fft_1d(field, vals, modulus, root_of_unity, inv=False):
    """starks/fft.py:316-331 -- the transform length is the order of root_of_unity; `vals` is zero-padded."""
    if not _on_device(modulus, root_of_unity):
        return _host_dft(field, list(vals), modulus, root_of_unity, inv)
    n = _order(root_of_unity)
    if not isinstance(vals, (list, tuple, WireList)):
        vals = list(vals)
    out = ntt_bytes(_lib.to_wire(vals), n, int(root_of_unity), inverse=inv)
    return _elements(field, out)


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
    for seq in (a, b):
        if isinstance(seq, WireList):
            field = seq.field
            break
    if field is None:
        for v in list(a) + list(b) + [root_of_unity]:
            if hasattr(v, "p"):
                field = type(v)
                break
    if field is None:
        field = IntegersModP(MIMC_P)
    if not _on_device(field.p, root_of_unity):
        # another field or an order the device code does not cover: three direct transforms on the host (never the hot path)
        fa = _host_dft(field, list(a), field.p, root_of_unity)
        fb = _host_dft(field, list(b), field.p, root_of_unity)
        unscale = field(len(fa))  # _host_dft's inverse divides by n, mul_polys (fft.py:345) does not
        return [x * unscale for x in _host_dft(field, [u * v for u, v in zip(fa, fb)], field.p, root_of_unity, inv=True)]
    n = _order(root_of_unity)
    a, b = (a if isinstance(a, WireList) else list(a)), (b if isinstance(b, WireList) else list(b))
    if len(a) > n or len(b) > n:
        raise ValueError("operand longer than the order of the root of unity")
    out = ctypes.create_string_buffer(32 * n)
    rc = _lib.lib().sh_mul_polys(_lib.ctx(), _lib.to_wire(a), len(a), _lib.to_wire(b), len(b), out, n,
                                 int(root_of_unity).to_bytes(32, "big"))
    _lib.check(rc, "sh_mul_polys")
    return _elements(field, out.raw)


def low_degree_extension(field, trace_columns, extension_factor, G2):
    """The LDE step of STARK.mk_proof (stark.py:27-36 + 253-256): per column, inverse NTT over
    G1 = G2^extension_factor, then NTT over G2.  trace_columns: list of equal-length lists."""
    _check_field(field.p)
    cols = [c if isinstance(c, WireList) else list(c) for c in trace_columns]
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
    raw = out.raw
    return [_elements(field, memoryview(raw)[32 * n * c:32 * n * (c + 1)]) for c in range(len(cols))]
