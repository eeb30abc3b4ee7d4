"""Minimal dense univariate polynomial, enough for the hot path's boundary: `NonBinaryFFT.fft` takes one
(its `.coefficients`, trailing zeros stripped) and `.inv_fft` returns one (starks/fft.py:263-272,
starks/polynomial.py:13-21,58,158-164).  Schoolbook multiplication / division are out of scope."""
from .wireseq import WireList

_POLYS = {}


class Poly(object):
    pass


def polynomials_over(ring):
    if ring in _POLYS:
        return _POLYS[ring]

    class Polynomial(Poly):
        def __init__(self, c):
            if isinstance(c, WireList) and c.field is ring:
                # a transform's output (wireseq.py): trailing zeros are stripped on the bytes, no element is created
                k = (len(c.wire_bytes().rstrip(b"\0")) + 31) // 32
                self.coefficients = c[:k]
                return
            if isinstance(c, Polynomial):
                coeffs = list(c.coefficients)
            elif isinstance(c, bytes):
                if len(c) % 32:
                    raise ValueError("Bytelength must be multiple of 32")
                coeffs = [ring(c[i:i + 32]) for i in range(0, len(c), 32)]
            elif hasattr(c, "__iter__"):
                coeffs = [x if isinstance(x, ring) else ring(x) for x in c]
            else:
                coeffs = [c if isinstance(c, ring) else ring(c)]
            k = len(coeffs)
            while k and coeffs[k - 1] == 0:  # strip trailing zeros (polynomial.py:13-21,58)
                k -= 1
            self.coefficients = coeffs[:k]

        @classmethod
        def factory(cls, L):
            return cls(L)

        def is_zero(self):
            return not self.coefficients

        def degree(self):
            return len(self.coefficients) - 1

        def __len__(self):
            return len(self.coefficients)

        def __iter__(self):
            return iter(self.coefficients)

        def __eq__(self, other):
            if not isinstance(other, Polynomial):
                try:
                    other = Polynomial(other)
                except Exception:
                    return False
            return self.coefficients == other.coefficients

        def __ne__(self, other):
            return not self == other

        def __call__(self, x):  # polynomial.py:158-164
            y = ring(0)
            pw = ring(1)
            for a in self.coefficients:
                y = y + pw * a
                pw = pw * x
            return y

        def __repr__(self):
            return "0" if self.is_zero() else " + ".join(
                ("%s *x**%d" % (a, i)) if i else "%s" % a for i, a in enumerate(self.coefficients))

    Polynomial.ring = ring
    Polynomial.__name__ = "(%s)[x]" % ring.__name__
    _POLYS[ring] = Polynomial
    return Polynomial
