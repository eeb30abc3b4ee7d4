"""Sparse multivariate polynomials over Z/p -- the host-side mirror of the reference's step-polynomial type
(starks/multivariate_polynomial.py:55-347: `multivariates_over(ring, num_vars).factory({power tuple: coeff})`) and of
`generate_Xi_s` (starks/utils.py:40-57), so `STARK(field, steps, ext, width, step_polys)` call sites read unchanged:

    X_1, X_2 = generate_Xi_s(field, 2)
    step_polys = [X_1, X_1 + X_2**3]

Only what a step polynomial needs is mirrored (+, -, *, **, scalar operands, degree(), __call__, iteration in sorted
monomial order); the sympy-backed division and string parsing of the reference (:171-327) are not on the proving path.
"""
from ._lib import MIMC_P


def multivariates_over(ring, num_vars):
    p = getattr(ring, "p", MIMC_P)

    class MultivariatePolynomial(object):
        def __init__(self, c):
            if isinstance(c, MultivariatePolynomial):
                coeffs = dict(c.coefficients)
            elif isinstance(c, dict):
                coeffs = {}
                for k, v in c.items():
                    if len(k) != num_vars:
                        raise ValueError("power tuple %r does not have %d entries" % (k, num_vars))
                    coeffs[tuple(int(e) for e in k)] = int(v) % p
            elif isinstance(c, int) or hasattr(c, "n"):
                coeffs = {(0,) * num_vars: int(c) % p}
            else:
                raise ValueError
            self.coefficients = {k: v for k, v in coeffs.items() if v}  # remove_zero_coefficients (:24-30)

        @classmethod
        def factory(cls, coefficients=None, step_fn=None):
            if coefficients is None:
                raise NotImplementedError("starks_amd: build step polynomials from a coefficient dict")
            return cls(coefficients)

        def __len__(self):
            return len(self.coefficients)

        def is_zero(self):
            return not self.coefficients

        def __iter__(self):
            for key in sorted(self.coefficients):  # :122-126
                yield key, self.coefficients[key]

        def degree(self):
            return max([sum(k) for k in self.coefficients] + [0])  # :111-117

        def __getitem__(self, power_tup):
            return self.coefficients.get(tuple(power_tup), 0)

        def _coerce(self, other):
            return other if isinstance(other, MultivariatePolynomial) else MultivariatePolynomial(other)

        def __neg__(self):
            return MultivariatePolynomial({k: -v for k, v in self.coefficients.items()})

        def __add__(self, other):
            other = self._coerce(other)
            out = dict(self.coefficients)
            for k, v in other.coefficients.items():
                out[k] = (out.get(k, 0) + v) % p
            return MultivariatePolynomial(out)

        __radd__ = __add__

        def __sub__(self, other):
            return self + (-self._coerce(other))

        def __rsub__(self, other):
            return self._coerce(other) - self

        def __mul__(self, other):
            other = self._coerce(other)
            out = {}
            for a, ca in self.coefficients.items():
                for b, cb in other.coefficients.items():
                    k = tuple(x + y for x, y in zip(a, b))
                    out[k] = (out.get(k, 0) + ca * cb) % p
            return MultivariatePolynomial(out)

        __rmul__ = __mul__

        def __pow__(self, e):
            if not isinstance(e, int) or e < 0:
                raise ValueError("exponent must be a non-negative int")
            out = MultivariatePolynomial(1)
            for _ in range(e):
                out = out * self
            return out

        def __eq__(self, other):
            try:
                return self.coefficients == self._coerce(other).coefficients
            except ValueError:
                return NotImplemented

        def __call__(self, vals):
            assert len(vals) == num_vars  # :330
            y = 0
            for a, coeff in self:
                prod = 1
                for v, power in zip(vals, a):
                    prod = prod * pow(int(v), power, p) % p
                y = (y + coeff * prod) % p
            return ring(y) if callable(ring) else y

        def __repr__(self):
            if self.is_zero():
                return "0"
            return " + ".join("%d %s" % (c, "".join("*X_%d**%d" % (i + 1, e) for i, e in enumerate(k))) for k, c in self)

    MultivariatePolynomial.ring = ring
    MultivariatePolynomial.num_vars = num_vars
    return MultivariatePolynomial


def generate_Xi_s(field, width):
    """utils.py:40-57: the index polynomials X_1 .. X_width."""
    mv = multivariates_over(field, width).factory
    return [mv({tuple(1 if j == i else 0 for j in range(width)): 1}) for i in range(width)]
