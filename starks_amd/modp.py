"""Host-side element type for Z/p with the call surface of the reference's `IntegersModP(p)` objects
(starks/modp.py:25-106): `.n`, `.p`, `int()`, `+ - * / **`, `-x`, `==` against ints, `.inverse()`,
`.to_bytes()` (32 bytes big-endian) and the non-reducing bytes constructor (modp.py:33-34).

This is boundary glue, not the accelerated path: the reference's classes never travel to the GPU box,
so the drop-in API needs an element type of its own.  Bulk arithmetic happens in libstarkhip.so.
"""
_FIELDS = {}


class FieldElement(object):
    """Common base so `isinstance(x, FieldElement)` works for every modulus."""
    __slots__ = ("n",)
    p = None


def IntegersModP(p):
    """Memoised class factory, like starks/modp.py:25 (one class object per modulus)."""
    p = int(p)
    if p in _FIELDS:
        return _FIELDS[p]

    class IntegerModP(FieldElement):
        __slots__ = ()

        def __init__(self, n):
            if isinstance(n, bytes):
                self.n = int.from_bytes(n, "big")  # NOT reduced (modp.py:33-34)
            elif isinstance(n, FieldElement):
                self.n = n.n % p
            else:
                try:
                    self.n = int(n) % p
                except Exception:
                    raise TypeError("Can't cast type %s to %s" % (type(n).__name__, type(self).__name__))

        @property
        def field(self):
            return IntegerModP

        @staticmethod
        def _num(other):
            if isinstance(other, IntegerModP):
                return other.n
            if isinstance(other, (int, bytes)):
                return IntegerModP(other).n
            return None

        def __add__(self, other):
            o = self._num(other)
            return NotImplemented if o is None else IntegerModP(self.n + o)

        __radd__ = __add__

        def __sub__(self, other):
            o = self._num(other)
            return NotImplemented if o is None else IntegerModP(self.n - o)

        def __rsub__(self, other):
            o = self._num(other)
            return NotImplemented if o is None else IntegerModP(o - self.n)

        def __mul__(self, other):
            o = self._num(other)
            return NotImplemented if o is None else IntegerModP(self.n * o)

        __rmul__ = __mul__

        def __neg__(self):
            return IntegerModP(-self.n)

        def __eq__(self, other):
            o = self._num(other)
            return o is not None and self.n == o

        def __ne__(self, other):
            return not self.__eq__(other)

        def __hash__(self):
            return hash((self.n, p))

        def inverse(self):
            if self.n % p == 0:
                raise ZeroDivisionError("0 has no inverse mod p")
            return IntegerModP(pow(self.n, p - 2, p))  # same residue as the reference's extended Euclid

        def __truediv__(self, other):
            o = self._num(other)
            return NotImplemented if o is None else self * IntegerModP(o).inverse()

        def __rtruediv__(self, other):
            o = self._num(other)
            return NotImplemented if o is None else self.inverse() * o

        def __pow__(self, e):
            if type(e) is not int:
                raise TypeError
            return IntegerModP(pow(self.n, e, p))  # numbertype.py:68-84 computes the same value

        def __int__(self):
            return self.n

        __index__ = __int__

        def __abs__(self):
            return abs(self.n)

        def __str__(self):
            return str(self.n)

        def __repr__(self):
            return "%d (mod %d)" % (self.n, p)

        def to_bytes(self):
            return self.n.to_bytes(32, "big")  # modp.py:94-95

        @classmethod
        def wrap_canonical(cls, ints):
            """Elements for values ALREADY in [0, p) -- what the device returns -- without the per-value type tests and the
            `% p` of the constructor (about a tenth faster: creating 2^20 Python objects costs 0.4-0.5 s either way)."""
            new, out = object.__new__, []
            for x in ints:
                e = new(cls)
                e.n = x
                out.append(e)
            return out

    IntegerModP.p = p
    IntegerModP.m = 1
    IntegerModP.field_size = p
    IntegerModP.__name__ = "Z/%d" % p
    _FIELDS[p] = IntegerModP
    return IntegerModP
