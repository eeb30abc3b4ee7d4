"""The reference's OWN unit tests for the hot path, restated against `starks_amd` on the MI355X: same names, same inputs, same
assertions (each test cites the reference test it restates; the bodies are this repo's wording, the values are the reference's).
What `python -m pytest starks/test/test_{fft,merkle_tree,utils,fri,stark,compression}.py` checks for the functions of SURVEY 8(a),
a user who swaps `starks.` for `starks_amd.` gets here -- including the tests the reference keeps commented out because its own
`stark.py` / `SmoothSubgroupFRI` do not import: they run here.

Where the reference's test only checks a shape (`len(evaluations) == 8`), the restated test also checks the values against a direct
evaluation in Python ints, so that "passes the reference's tests" cannot be satisfied by the right number of wrong elements."""
import pytest

pytestmark = pytest.mark.gpu

P = 2**256 - 2**32 * 351 + 1


@pytest.fixture(scope="module")
def F():
    from starks_amd import _lib, IntegersModP
    _lib.ctx()  # fails loudly when the extension or the GPU is missing
    return IntegersModP(P)


def _eval(coeffs, x, p):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + int(c)) % p
    return acc


# ---- starks/test/test_fft.py -----------------------------------------------------------------------------------------------
def test_fft__test_basic_and_output_type():
    """test_fft.py:98-113 and :151-169 -- 0 + x + 2x^2 + 3x^3 over Z/31 at the powers of a 6th root of unity: six values of the field"""
    from starks_amd import IntegersModP
    from starks_amd.fft import NonBinaryFFT
    from starks_amd.polynomial import polynomials_over
    field = IntegersModP(31)
    poly = polynomials_over(field).factory(list(range(4)))
    root = field(3) ** ((31 - 1) // 6)
    evaluations = NonBinaryFFT(field, root).fft(poly)
    assert len(evaluations) == 6
    assert all(isinstance(v, field) for v in evaluations)
    assert [int(v) for v in evaluations] == [_eval(range(4), pow(int(root), k, 31), 31) for k in range(6)]


def test_fft__test_large_modulus(F):
    """test_fft.py:115-130 -- the same polynomial over the MiMC prime at the powers of 7^((p-1)/8): eight values (on the GPU)"""
    from starks_amd.fft import NonBinaryFFT
    from starks_amd.polynomial import polynomials_over
    poly = polynomials_over(F).factory(list(range(4)))
    root = F(7) ** ((P - 1) // 8)
    evaluations = NonBinaryFFT(F, root).fft(poly)
    assert len(evaluations) == 8
    assert [int(v) for v in evaluations] == [_eval(range(4), pow(int(root), k, P), P) for k in range(8)]


def test_fft__test_fft_inv(F):
    """test_fft.py:132-149 -- inv_fft(fft(poly)) == poly, over Z/31 as the reference has it and over the MiMC prime"""
    from starks_amd import IntegersModP
    from starks_amd.fft import NonBinaryFFT
    from starks_amd.polynomial import polynomials_over
    f31 = IntegersModP(31)
    for field, root in ((f31, f31(3) ** 5), (F, F(7) ** ((P - 1) // 8)), (F, F(7) ** ((P - 1) // 4096))):
        poly = polynomials_over(field).factory(list(range(4)))
        solver = NonBinaryFFT(field, root)
        assert solver.inv_fft(solver.fft(poly)) == poly


def test_fft__test_mul_polys(F):
    """test_fft.py:185-196 -- (x + 2x^2 + 3x^3)^2 through a 512-point transform; mul_polys leaves the factor n in (fft.py:345)"""
    from starks_amd.fft import mul_polys
    root = F(7) ** ((P - 1) // 512)
    a = [F(v) for v in range(4)]
    prod = mul_polys(a, list(a), root)
    assert len(prod) == 512
    want = [0, 0, 1, 4, 10, 12, 9] + [0] * 505
    assert [int(v) for v in prod] == [512 * c % P for c in want]


# ---- starks/test/test_merkle_tree.py ---------------------------------------------------------------------------------------
def test_merkle_tree__test_merkletree_mk_branch_verify_branch(F):
    """test_merkle_tree.py:16-22, :41-60 -- trees over the 32-byte integers 0..127 and 0..255; the branch of leaf 59"""
    from starks_amd.merkle_tree import merkelize, mk_branch, verify_branch
    t = merkelize([x.to_bytes(32, "big") for x in range(128)])
    assert len(t) == 256
    b = mk_branch(t, 59)
    assert len(b) == 8
    assert verify_branch(t[1], 59, b, output_as_int=True) == 59
    assert t[1].hex() == "3338fbbd9d7a079de681387210ccb901f68aa1bf4a51f53e7aa70ccb56de3462"  # SURVEY appendix A, from the live reference
    t = merkelize([x.to_bytes(32, "big") for x in range(256)])
    assert len(t) == 512 and len(mk_branch(t, 59)) == 9


def test_merkle_tree__test_merkletree_zmodp():
    """test_merkle_tree.py:24-30 -- 128 elements of Z/7 (their to_bytes() are 32-byte leaves: the device tree)"""
    from starks_amd import IntegersModP
    from starks_amd.merkle_tree import merkelize, mk_branch, verify_branch
    mod7 = IntegersModP(7)
    tree = merkelize([mod7(i) for i in range(128)])
    assert len(tree) == 256
    assert verify_branch(tree[1], 100, mk_branch(tree, 100), output_as_int=True) == 100 % 7


def test_merkle_tree__test_unpack_merkle_leaf():
    """test_merkle_tree.py:62-80 -- a packed leaf of 3 polynomials x 2 dimensions splits back into its six 32-byte parts"""
    from starks_amd.merkle_tree import unpack_merkle_leaf
    parts = [c.to_bytes(32, "big") for c in range(6)]
    assert unpack_merkle_leaf(b"".join(parts), 2, 3) == parts


# ---- starks/test/test_utils.py ---------------------------------------------------------------------------------------------
def test_utils__test_get_power_cycle(F):
    """test_utils.py:20-30 -- the cycle of a 6th root of unity mod 31; and (GPU) of a 2^12-th root over the MiMC prime"""
    from starks_amd import IntegersModP
    from starks_amd.utils import get_power_cycle
    mod = IntegersModP(31)
    assert get_power_cycle(mod(3) ** 5, mod) == [1, 26, 25, 30, 5, 6]
    g = F(7) ** ((P - 1) // 4096)
    cyc = get_power_cycle(g, F)
    assert len(cyc) == 4096 and int(cyc[1]) == int(g) and int(cyc[4095]) * int(g) % P == 1


def test_utils__test_plus_one_and_the_small_helpers(F):
    """test_utils.py:32-34, and the helpers the reference's STARK tests import from utils (generate_Xi_s, is_a_power_of_2)"""
    from starks_amd.utils import generate_Xi_s, is_a_power_of_2, plus_one
    assert plus_one(4) == 5
    assert [is_a_power_of_2(x) for x in (1, 2, 3, 4, 6, 1 << 20, (1 << 20) + 2)] == [True, True, False, True, False, True, False]
    X1, X2 = generate_Xi_s(F, 2)
    assert int((X1 + X2**3)([F(2), F(5)])) == 127


def test_utils__test_mimc():
    """test_utils.py:11-18 -- mimc(5, 3, [2, 7]) runs (the reference asserts nothing); here also its value"""
    from starks_amd.utils import mimc
    x = 5
    for i in range(2):
        x = (x**3 + [2, 7][i % 2]) % P
    assert int(mimc(5, 3, [2, 7])) == x


# ---- starks/test/test_fri.py (commented out in the reference: SmoothSubgroupFRI is inside a comment block, fri.py:176-366) ------
def test_fri__test_basic_prove(F):
    """test_fri.py:34-52 -- degree < 4 on an 8-point domain: the direct case, the proof is the list of the 8 evaluations"""
    from starks_amd.fri import SmoothSubgroupFRI
    from starks_amd.polynomial import polynomials_over
    poly = polynomials_over(F).factory(list(range(4)))
    root = F(7) ** ((P - 1) // 8)
    proof = SmoothSubgroupFRI(F).generate_proximity_proof(poly, root, 4)
    assert len(proof[0]) == 8
    assert [int.from_bytes(v, "big") for v in proof[0]] == [_eval(range(4), pow(int(root), k, P), P) for k in range(8)]


def _mimc_constants_poly(F, steps):
    from starks_amd.polynomial import polynomials_over
    return polynomials_over(F).factory([F((i**7) ^ 42) for i in range(steps)])


def test_fri__test_high_degree_prove(F):
    """test_fri.py:105-134 -- 512 MiMC round constants as coefficients, maxdeg_plus_1 = 512: rounds at 512, 128, 32, then 8 values"""
    from starks_amd.fri import SmoothSubgroupFRI
    proof = SmoothSubgroupFRI(F).generate_proximity_proof(_mimc_constants_poly(F, 512), F(7) ** ((P - 1) // 512), 512)
    assert len(proof) == 4
    for rec in proof[:3]:
        assert len(rec) == 2 and len(rec[1]) == 40
    assert len(proof[3]) == 8


def test_fri__test_verify_low_degree_proof(F):
    """test_fri.py:136-157 -- that proof against the root of the tree over the evaluations"""
    from starks_amd.fft import NonBinaryFFT
    from starks_amd.fri import SmoothSubgroupFRI
    from starks_amd.merkle_tree import merkelize
    poly, root = _mimc_constants_poly(F, 512), F(7) ** ((P - 1) // 512)
    fri = SmoothSubgroupFRI(F)
    proof = fri.generate_proximity_proof(poly, root, 512)
    mroot = merkelize(NonBinaryFFT(F, root).fft(poly))[1]
    assert fri.verify_proximity_proof(proof, mroot, root, 512)


def test_fri__test_fri(F):
    """test_fri.py:159-197 -- coefficients 0..255 on a 1024-point domain, proved, compressed (the length is printed there) and
    verified; and what the reference leaves as a TODO: a claim of a lower degree than the polynomial has is rejected"""
    from starks_amd.compression import bin_length, compress_fri
    from starks_amd.fft import NonBinaryFFT
    from starks_amd.fri import SmoothSubgroupFRI
    from starks_amd.merkle_tree import merkelize
    from starks_amd.polynomial import polynomials_over
    degree = 256
    poly = polynomials_over(F).factory([F(v) for v in range(degree)])
    root = F(7) ** ((P - 1) // (degree * 4))
    fri = SmoothSubgroupFRI(F)
    proof = fri.generate_proximity_proof(poly, root, degree)
    assert bin_length(compress_fri(proof)) > 0
    mroot = merkelize(NonBinaryFFT(F, root).fft(poly))[1]
    assert fri.verify_proximity_proof(proof, mroot, root, degree)
    with pytest.raises(AssertionError):
        assert fri.verify_proximity_proof(fri.generate_proximity_proof(poly, root, degree // 4), mroot, root, degree // 4)


# ---- starks/test/test_compression.py ---------------------------------------------------------------------------------------
def test_compression__test_compress_fri(F):
    """test_compression.py:18-42 -- the direct proof of a degree-3 polynomial on the 8 powers of 3^((p-1)/8) compresses to a
    non-empty stream (the reference passes its modulus in the exclude_multiples_of position; the direct case ignores it)"""
    from starks_amd.compression import bin_length, compress_fri, decompress_fri
    from starks_amd.fri import SmoothSubgroupFRI
    from starks_amd.polynomial import polynomials_over
    poly = polynomials_over(F).factory(list(range(4)))
    root = F(3) ** ((P - 1) // 8)
    proof = SmoothSubgroupFRI(F).generate_proximity_proof(poly, root, 4, P)
    compressed = compress_fri(proof)
    assert bin_length(compressed) > 0
    assert decompress_fri(compressed) == proof


# ---- starks/test/test_stark.py (every test commented out in the reference: `import starks.stark` fails) -----------------
def _prove_and_verify(F, width, steps, inp, step_polys, ext=8):
    from starks_amd.air import AIR
    from starks_amd.stark import STARK
    comp = AIR(F, width, [F(v) for v in inp], steps, step_polys, ext)
    witness, boundary = comp.generate_witness(), comp.generate_boundary_constraints()
    stark = STARK(F, steps, ext, width, step_polys)
    proof = stark.mk_proof(witness, boundary)
    assert isinstance(proof, list) and len(proof) == 4
    assert stark.verify_proof(proof, witness, boundary)
    return stark, proof, witness, boundary


def test_stark__test_higher_dim_proof_verification(F):
    """test_stark.py:215-234 -- Fibonacci as a width-2 state, 32 steps"""
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X1, X2 = generate_Xi_s(F, 2)
    _prove_and_verify(F, 2, 32, [0, 1], [X2, X1 + X2])


def test_stark__test_quadratic_stark_and_mimc_stark_verification(F):
    """test_stark.py:236-263 and :265-293 -- [X1, X1 + X2^3] from (2, 5), 8 steps; a proof about another witness is rejected"""
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X1, X2 = generate_Xi_s(F, 2)
    stark, proof, witness, boundary = _prove_and_verify(F, 2, 8, [2, 5], [X1, X1 + X2**3])
    other = [list(col) for col in witness]
    other[1][-1] = other[1][-1] + 1
    with pytest.raises(AssertionError):
        assert stark.verify_proof(proof, other, boundary)


def test_stark__test_affine_stark(F):
    """test_stark.py:295-322 -- [X1, X1 + 3 X2], 32 steps"""
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X1, X2 = generate_Xi_s(F, 2)
    _prove_and_verify(F, 2, 32, [2, 5], [X1, X1 + 3 * X2])


def test_stark__test_varying_quintic_stark(F):
    """test_stark.py:324-350 -- six state variables, the last one the product of all six (degree 6), 8 steps"""
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X = generate_Xi_s(F, 6)
    _prove_and_verify(F, 6, 8, [1, 2, 3, 4, 5, 6], X[:5] + [X[0] * X[1] * X[2] * X[3] * X[4] * X[5]])


# ---- the module-level pieces of mk_proof (stark.py:27-177), as the reference's commented tests assemble them -----------------------
def _system(F, c):
    from starks_amd.multivariate_polynomial import multivariates_over
    mv = multivariates_over(F, c["width"]).factory
    return [mv({tuple(k): F(v) for k, v in d}) for d in c["step_polys"]]


def test_stark__test_trace_polynomials(F):
    """test_stark.py:63-87 -- [X2, X1 + 2 X2^2] from (2, 5), 128 steps: every trace polynomial takes the witness's values on the
    powers of G1"""
    from starks_amd.air import AIR
    from starks_amd.multivariate_polynomial import generate_Xi_s
    from starks_amd.stark import STARK, construct_trace_polynomials
    from starks_amd.utils import get_power_cycle
    X1, X2 = generate_Xi_s(F, 2)
    step_polys = [X2, X1 + 2 * X2**2]
    comp = AIR(F, 2, [F(2), F(5)], 128, step_polys, 8)
    witness = comp.generate_witness()
    params = STARK(F, 128, 8, 2, step_polys)
    trace_polys = construct_trace_polynomials(witness, params.field, params.G1)
    assert len(trace_polys) == 2
    xs = get_power_cycle(params.G1, params.field)
    for dim in range(2):
        for ind, x in enumerate(xs):
            assert witness[dim][ind] == trace_polys[dim](x)


def test_stark__test_constraint_polynomials_and_the_pieces_of_mk_proof(F):
    """test_stark.py:89-109, :111-163, :165-213 -- the constraint, remainder and boundary polynomials, the packed tree over their
    evaluations, the pseudorandom linear combination and its tree, assembled from the module-level functions as those tests do;
    checked against what the LIVE reference's functions of the same names returned (tests/golden/stark.json: coefficients of
    every trace / remainder / boundary polynomial, m_root, l_root) and against mk_proof's own roots."""
    from conftest import load_golden
    from starks_amd import stark
    from starks_amd.fft import NonBinaryFFT
    from starks_amd.merkle_tree import merkelize, merkelize_polynomial_evaluations
    from starks_amd.utils import get_pseudorandom_indices
    for c in load_golden("stark.json"):
        if c["steps"] > 256:
            continue
        width, steps, ext = c["width"], c["steps"], c["ext"]
        step_polys = _system(F, c)
        witness = [[F(int(v, 16)) for v in col] for col in c["witness"]]
        boundary = [(0, j, F(v)) for j, v in enumerate(c["inputs"])]
        S = stark.STARK(F, steps, ext, width, step_polys)
        tps = stark.construct_trace_polynomials(witness, F, S.G1)
        cps = stark.construct_constraint_polynomials(step_polys, tps, F, S.G1, width)
        assert len(cps) == width
        dps = stark.construct_remainder_polynomials(cps, F, steps, S.last_step_position)
        bps = stark.construct_boundary_polynomials(tps, witness, boundary, F, S.last_step_position, width)
        as_hex = lambda polys: [["%064x" % int(a) for a in pl.coefficients] for pl in polys]  # noqa: E731
        assert as_hex(tps) == c["trace_polys"], c["name"]
        assert as_hex(dps) == c["remainder_polys"], c["name"]
        assert as_hex(bps) == c["boundary_polys"], c["name"]
        solver = NonBinaryFFT(F, S.G2)
        mtree = merkelize_polynomial_evaluations(width, [solver.fft(pl) for pl in tps + dps + bps])
        assert mtree[1].hex() == c["m_root"], c["name"]
        l_poly = stark.compute_pseudorandom_linear_combination(mtree[1], tps, dps, bps, F, S.G2, S.precision, steps, width)
        l_mtree = merkelize(solver.fft(l_poly))
        assert l_mtree[1].hex() == c["l_root"], c["name"]
        indices = get_pseudorandom_indices(l_mtree[1], S.precision, count=80, exclude_multiples_of=ext)
        assert len(indices) == 80 and all(i % ext for i in indices)
        proof = S.mk_proof(witness, boundary)  # the device prover commits to the same two roots
        assert proof[0] == mtree[1] and proof[1] == l_mtree[1]
        # a witness that is not a trace: the division by Z leaves a remainder (stark.py:76)
        if steps == 32:
            bad = [list(col) for col in witness]
            bad[-1][steps // 2] = bad[-1][steps // 2] + 1
            with pytest.raises(AssertionError):
                btps = stark.construct_trace_polynomials(bad, F, S.G1)
                stark.construct_remainder_polynomials(stark.construct_constraint_polynomials(step_polys, btps, F, S.G1, width), F, steps,
                                                      S.last_step_position)
