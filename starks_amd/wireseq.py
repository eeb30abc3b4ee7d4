"""Lazy sequences over the library's wire form, so that chained reference call sites move bytes instead of Python objects.

The reference passes Python lists of field elements between its stages (fft -> merkelize -> mk_branch, stark.py:253-263;
get_power_cycle -> ..., stark.py:223).  A 2^20-point transform takes 0.09 ms on the GPU and 16 ms to cross PCIe from
pageable memory, but turning its 32 MiB of output into 2^20 element objects costs 0.5 s, and turning them back into bytes
for the next stage another 0.3 s.  `WireList` is what the drop-in functions return instead of a list: a read-only sequence
backed by the 32-byte big-endian values the device wrote (modp.py:94-95); elements are created when somebody indexes or
iterates, equality and `int()` keep the reference's semantics, and the next stage (`merkelize`, `fft_1d`, `prove_low_degree`,
`merkelize_polynomial_evaluations`) takes the bytes as they are.  `NodeList` is the same idea for `merkelize`'s 2n-entry list
of 32-byte nodes (entry 0 is b"" as in the reference, merkle_tree.py:45).

Both are `collections.abc.Sequence`s, not `list` subclasses: `len`, indexing (negative, slices, strides), iteration,
`in`, `==` against lists / tuples / each other, `+` (gives a list), `list(x)`.  Code that needs a real list calls
`list(x)` or `x.tolist()` and pays for the objects there.
"""
from collections.abc import Sequence


class WireList(Sequence):
    """n field elements stored as n * 32 big-endian bytes.  `canonical`: every value is in [0, p) (what the device returns),
    which lets two WireLists compare by their bytes."""
    __slots__ = ("_mv", "_field", "_n", "_canonical")

    def __init__(self, buf, field, canonical=True):
        mv = memoryview(buf)
        if mv.format != "B" or mv.ndim != 1:
            mv = mv.cast("B")
        if len(mv) % 32:
            raise ValueError("wire form is a multiple of 32 bytes")
        self._mv, self._field, self._n, self._canonical = mv, field, len(mv) // 32, canonical

    # ---- what the next stage takes ------------------------------------------------------------------------------------------
    def wire(self):
        """The backing bytes (a memoryview: no copy)."""
        return self._mv

    def wire_bytes(self):
        """The values as one `bytes` object (what ctypes takes): the backing object itself when the view covers all of it."""
        obj = self._mv.obj
        if isinstance(obj, bytes) and len(obj) == len(self._mv):
            return obj
        return bytes(self._mv)

    @property
    def field(self):
        return self._field

    # ---- Sequence -----------------------------------------------------------------------------------------------------------
    def __len__(self):
        return self._n

    def _elem(self, i):
        x = int.from_bytes(self._mv[32 * i:32 * i + 32], "big")
        wrap = getattr(self._field, "wrap_canonical", None)
        if wrap is not None and self._canonical:
            return wrap((x,))[0]
        return self._field(x)

    def __getitem__(self, i):
        if isinstance(i, slice):
            start, stop, step = i.indices(self._n)
            if step == 1:
                return WireList(self._mv[32 * start:32 * max(start, stop)], self._field, self._canonical)
            idx = range(start, stop, step)
            if not len(idx):
                return WireList(b"", self._field, self._canonical)
            try:  # strided: one gather over a (n, 32) byte matrix
                import numpy as np
                rows = np.frombuffer(self._mv, dtype=np.uint8).reshape(self._n, 32)[np.arange(start, stop, step)]
                return WireList(rows.tobytes(), self._field, self._canonical)
            except ImportError:  # pragma: no cover
                return WireList(b"".join(bytes(self._mv[32 * k:32 * k + 32]) for k in idx), self._field, self._canonical)
        i = i.__index__()
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError("WireList index out of range")
        return self._elem(i)

    def __iter__(self):
        conv, mv = int.from_bytes, self._mv
        wrap = getattr(self._field, "wrap_canonical", None) if self._canonical else None
        if wrap is not None:
            # elements in blocks: one wrap call per 4096 values keeps the per-element cost at the constructor-free rate
            for base in range(0, self._n, 4096):
                top = min(self._n, base + 4096)
                for e in wrap([conv(mv[32 * k:32 * k + 32], "big") for k in range(base, top)]):
                    yield e
        else:
            for k in range(self._n):
                yield self._field(conv(mv[32 * k:32 * k + 32], "big"))

    def __eq__(self, other):
        if isinstance(other, WireList):
            if self._n != other._n:
                return False
            if self._canonical and other._canonical and getattr(self._field, "p", None) == getattr(other._field, "p", None):
                return self._mv == other._mv
        elif not isinstance(other, (list, tuple, Sequence)) or isinstance(other, (bytes, str)):
            return NotImplemented
        if len(other) != self._n:
            return False
        return all(a == b for a, b in zip(self, other))

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __repr__(self):
        head = ", ".join(str(int(v)) for v in self[:3])
        return "WireList(%d values%s%s)" % (self._n, ": " if self._n else "", head + (", ..." if self._n > 3 else ""))

    def tolist(self):
        return list(self)

    def ints(self):
        """Plain ints (no element objects)."""
        conv, mv = int.from_bytes, self._mv
        return [conv(mv[32 * k:32 * k + 32], "big") for k in range(self._n)]


class NodeList(Sequence):
    """merkelize's result: 2n nodes of `width` bytes each over one buffer; entry 0 reads as b"" (merkle_tree.py:45).
    `tail` (optional): a second buffer of wider entries that follows the nodes (merkelize_polynomial_evaluations returns
    the n hash nodes and then the n packed leaves of 32 k bytes, merkle_tree.py:94-119)."""
    __slots__ = ("_mv", "_n", "_tail", "_tw", "_tn")

    def __init__(self, buf, tail=None, tail_width=0):
        mv = memoryview(buf)
        self._mv = mv if (mv.format == "B" and mv.ndim == 1) else mv.cast("B")
        self._n = len(self._mv) // 32
        self._tail = None
        self._tw = self._tn = 0
        if tail is not None:
            tv = memoryview(tail)
            self._tail = tv if (tv.format == "B" and tv.ndim == 1) else tv.cast("B")
            self._tw = tail_width
            self._tn = len(self._tail) // tail_width

    def wire(self):
        return self._mv

    def __len__(self):
        return self._n + self._tn

    def _one(self, i):
        if i == 0:
            return b""
        if i < self._n:
            return bytes(self._mv[32 * i:32 * i + 32])
        j = i - self._n
        return bytes(self._tail[self._tw * j:self._tw * (j + 1)])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._one(k) for k in range(*i.indices(len(self)))]
        i = i.__index__()
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("NodeList index out of range")
        return self._one(i)

    def __iter__(self):
        for k in range(len(self)):
            yield self._one(k)

    def __eq__(self, other):
        if isinstance(other, NodeList) and self._tail is None and other._tail is None:
            return self._n == other._n and self._mv[32:] == other._mv[32:]
        if not isinstance(other, (list, tuple, Sequence)) or isinstance(other, (bytes, str)):
            return NotImplemented
        return len(other) == len(self) and all(a == b for a, b in zip(self, other))

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __repr__(self):
        return "NodeList(%d nodes, root %s)" % (len(self), self._one(1).hex() if self._n > 1 else "-")
