"""The module-level building blocks of `STARK.mk_proof` under the reference's names (starks/stark.py:27-177):

    construct_trace_polynomials(witness, field, root_of_unity)                                             stark.py:27-36
    construct_constraint_polynomials(step_polys, trace_polys, field, root_of_unity, width)                 stark.py:38-55
    construct_remainder_polynomials(constraint_polys, field, steps, last_step_position)                    stark.py:57-78
    construct_boundary_polynomials(trace_polys, witness, boundary, field, last_step_position, width)       stark.py:80-104
    compute_pseudorandom_linear_combination_1d(entropy, trace_polys, remainder_polys, boundary_polys,
                                               root_of_unity, steps, root_of_unity_degree)                 stark.py:130-162
    compute_pseudorandom_linear_combination(entropy, ..., field, root_of_unity, root_of_unity_degree, steps, width)   :164-177

`STARK.mk_proof` of this package does not go through them -- the device prover works on the evaluation domain (csrc/stark.hip) --
but code written against the reference's pieces (its commented tests, test_stark.py:63-213, build a proof's first half from them)
gets the SAME polynomials here, coefficient for coefficient, without the reference's quadratic polynomial arithmetic:

  * trace polynomials: one inverse NTT per state variable on the GPU (`NonBinaryFFT.inv_fft`);
  * constraint polynomials P_d(g1 X) - step_d(P(X)): a composition of degree deg * (steps - 1), obtained on a power-of-two domain that
    holds it -- the P_j and the shifted P_d(g1 X) evaluated there by NTTs on the GPU, the step polynomials applied point by point
    (Python ints; this is what bounds the practical size to ~2^18 points), one inverse NTT back;
  * remainder polynomials C / Z, Z = (X^steps - 1) / (X - x_last): C (X - x_last) divided by X^steps - 1, an O(n) recurrence
    (AssertionError when the division leaves a remainder, as the reference's `assert cp % z == 0`);
  * boundary polynomials (P - I) / ((X - 1)(X - x_last)): two synthetic divisions;
  * the linear combination: coefficient-wise, with the reference's scalars (stark.py:149-177: every `powers[i]` there reads the loop
    variable left over from building the list, i.e. the LAST power -- kept, since the proof bytes depend on it).
"""
from . import _lib
from ._lib import MIMC_P
from .fft import NonBinaryFFT, fft_1d
from .polynomial import polynomials_over
from .stark import get_pseudorandom_ks


def _ints(poly):
    c = poly.coefficients if hasattr(poly, "coefficients") else poly
    return c.ints() if hasattr(c, "ints") else [int(x) for x in c]


def construct_trace_polynomials(witness, field, root_of_unity):
    solver = NonBinaryFFT(field, root_of_unity)
    return [solver.inv_fft(col) for col in witness]


def construct_constraint_polynomials(step_polys, trace_polys, field, root_of_unity, width):
    p = int(field.p)
    if p != MIMC_P:
        raise NotImplementedError("starks_amd accelerates the MiMC prime field only")
    steps = _lib.order_of_root(root_of_unity)
    if steps is None:
        raise NotImplementedError("root_of_unity must have power-of-two order")
    g1 = int(root_of_unity)
    tps = [_ints(tp) for tp in trace_polys]
    terms = [sorted((tuple(k), int(c) % p) for k, c in sp.coefficients.items()) for sp in step_polys]
    degree = max([sum(k) for ts in terms for k, c in ts if c] + [1])
    need = degree * (max(len(t) for t in tps) - 1) + 1 if any(tps) else 1
    m = max(4, steps)
    while m < need:
        m *= 2
    w = pow(7, (p - 1) // m, p)  # any root of order m: evaluations and interpolation use the same one
    evals = [fft_1d(field, tp, p, w).ints() for tp in tps]
    out = []
    polys_over = polynomials_over(field).factory
    for d in range(len(step_polys)):
        shifted, pw = [], 1
        for c in tps[d]:  # P_d(g1 X): coefficient k times g1^k
            shifted.append(c * pw % p)
            pw = pw * g1 % p
        nxt = fft_1d(field, shifted, p, w).ints()
        vals = []
        for x in range(m):
            acc = 0
            for k, c in terms[d]:
                t = c
                for v in range(width):
                    if k[v]:
                        t = t * pow(evals[v][x], k[v], p) % p
                acc += t
            vals.append((nxt[x] - acc) % p)
        out.append(polys_over(fft_1d(field, vals, p, w, inv=True)))
    return out


def construct_remainder_polynomials(constraint_polys, field, steps, last_step_position):
    p, last = int(field.p), int(last_step_position)
    polys_over = polynomials_over(field).factory
    out = []
    for cp in constraint_polys:
        c = _ints(cp)
        num = [0] * (len(c) + 1)  # C (X - x_last)
        for k, a in enumerate(c):
            num[k + 1] = (num[k + 1] + a) % p
            num[k] = (num[k] - last * a) % p
        q = [0] * max(len(num) - steps, 0)  # num = q (X^steps - 1) + r:  q_k = num_(k + steps) + q_(k + steps)
        for k in range(len(q) - 1, -1, -1):
            q[k] = (num[k + steps] + (q[k + steps] if k + steps < len(q) else 0)) % p
        for k in range(min(steps, len(num))):
            assert (num[k] + (q[k] if k < len(q) else 0)) % p == 0, "constraint polynomial is not a multiple of Z"  # stark.py:76
        out.append(polys_over(q))
    return out


def _div_linear(c, root, p):
    """c(X) / (X - root), exact part (the remainder c(root) is dropped, as the reference's floor division does)"""
    q = [0] * max(len(c) - 1, 0)
    carry = 0
    for k in range(len(c) - 1, 0, -1):
        carry = (c[k] + carry * root) % p
        q[k - 1] = carry
    return q


def construct_boundary_polynomials(trace_polys, witness, boundary, field, last_step_position, width):
    p, last = int(field.p), int(last_step_position)
    polys_over = polynomials_over(field).factory
    inv = pow((last - 1) % p, p - 2, p)
    out = []
    for dim in range(width):
        input_value, output = int(boundary[dim][2]), int(witness[dim][-1])
        slope = (output - input_value) * inv % p  # the line through (1, input), (x_last, output)
        c = list(_ints(trace_polys[dim]))
        c += [0] * (2 - len(c))
        c[0] = (c[0] - (input_value - slope)) % p
        c[1] = (c[1] - slope) % p
        out.append(polys_over(_div_linear(_div_linear(c, 1, p), last, p)))
    return out


def _last_power(root_of_unity, steps, root_of_unity_degree, p):
    if root_of_unity_degree < 2:
        raise NameError("root_of_unity_degree must be at least 2 (the reference reads the loop variable of range(1, degree))")
    return pow(pow(int(root_of_unity), steps, p), root_of_unity_degree - 1, p)  # powers[i], i = degree - 1


def _combine(polys_and_scalars, p):
    n = max([len(c) for c, _ in polys_and_scalars] + [0])
    acc = [0] * n
    for c, s in polys_and_scalars:
        for k, a in enumerate(c):
            acc[k] = (acc[k] + a * s) % p
    return acc


def _lincomb_1d(entropy, trace_polys, remainder_polys, boundary_polys, root_of_unity, steps, root_of_unity_degree, p):
    k1, k2, k3, k4 = get_pseudorandom_ks(entropy, 4)
    c = _last_power(root_of_unity, steps, root_of_unity_degree, p)
    beta, gamma = (k1 + k2 * c) % p, (k3 + k4 * c) % p
    return [_combine([(_ints(d), 1), (_ints(t), beta), (_ints(b), gamma)], p)
            for t, d, b in zip(trace_polys, remainder_polys, boundary_polys)], c


def compute_pseudorandom_linear_combination_1d(entropy, trace_polys, remainder_polys, boundary_polys, root_of_unity, steps,
                                               root_of_unity_degree):
    field = trace_polys[0].ring
    ls, _ = _lincomb_1d(entropy, trace_polys, remainder_polys, boundary_polys, root_of_unity, steps, root_of_unity_degree, int(field.p))
    polys_over = polynomials_over(field).factory
    return [polys_over(l) for l in ls]


def compute_pseudorandom_linear_combination(entropy, trace_polys, remainder_polys, boundary_polys, field, root_of_unity,
                                            root_of_unity_degree, steps, width):
    p = int(field.p)
    ls, c = _lincomb_1d(entropy, trace_polys, remainder_polys, boundary_polys, root_of_unity, steps, root_of_unity_degree, p)
    l_ks = get_pseudorandom_ks(entropy, width)  # None from width 10 on, as in the reference (stark.py:106-126): zip() then raises
    joint = _combine([(l, (1 + lk * c) % p) for l, lk in zip(ls, l_ks)], p)
    return polynomials_over(field).factory(joint)
