"""The witness side of the prover: `get_computational_trace` and the parts of `AIR` that `STARK.mk_proof`'s callers
use (starks/air.py:32-52, 94-131): `generate_witness()` and `generate_boundary_constraints()`.

Host-side glue on Python ints.  One deliberate difference: the reference's constructor asserts `steps == 2**9 - 1`
(air.py:94), which contradicts every STARK call site in its own tests (steps 8 / 32 / 512, test_stark.py:215-350) and
the power-of-two domain `STARK` needs (stark.py:205-208); that assert and the monotone-circuit bookkeeping
(air.py:113-118, 133-160) are not mirrored.
"""


def get_computational_trace(inp, steps, width, step_polys):
    """air.py:32-52 -> (trace, output); trace[step][dim]."""
    trace = [list(inp)]
    for _ in range(steps - 1):
        trace.append([step_polys[i](trace[-1]) for i in range(width)])
    return trace, trace[-1]


class AIR(object):
    def __init__(self, field, width, inp, steps, step_polys, extension_factor):
        self.field = field
        self.width = width
        if isinstance(inp, int):  # air.py:89-90
            inp = [inp]
        self.inp = [field(v) for v in inp]
        self.steps = steps
        self.step_polys = step_polys
        self.extension_factor = extension_factor
        self.computational_trace, self.output = get_computational_trace(self.inp, steps, width, step_polys)
        self.F, self.T, self.w = field, steps, width

    def generate_witness(self):
        """air.py:121-123: witness[dim][step]"""
        return [[self.computational_trace[i][j] for i in range(self.steps)] for j in range(self.w)]

    def generate_boundary_constraints(self):
        """air.py:125-131: (step 0, dim, input value)"""
        return [(0, ind, self.inp[ind]) for ind in range(self.w)]

    def get_degree(self):
        return max(poly.degree() for poly in self.step_polys)
