"""STARK.mk_proof / verify_proof (starks/stark.py:179-402) with the prover on the MI355X.

    stark = STARK(field, steps, extension_factor, width, step_polys)
    proof = stark.mk_proof(witness, boundary)          # [m_root, l_root, branches, fri_proof]
    assert stark.verify_proof(proof, witness, boundary)

`mk_proof` is one call into libstarkhip.so (sh_stark_prove, csrc/capi.hip:run_stark): low-degree extension, the
constraint quotient D and boundary quotient B, the packed Merkle tree, the pseudorandom linear combination, the spot
checks and the FRI commit all run on the device and one copy brings the flat proof back.  The reference builds D and B
with O(n^2) coefficient arithmetic (stark.py:38-104); the device computes the same polynomials through the evaluation
domain (see csrc/stark.hip), so every root and proof byte is identical.  The verifier is host-side Python on ints.

starks/stark.py does not import at the reference snapshot (stark.py:13 wants a class fri.py keeps commented out), but
its code runs unchanged once that name exists; tests/golden/generate.py does exactly that to produce the fixtures this
module is tested against (tests/golden/stark.json).
"""
import ctypes

from . import _lib, fri
from ._lib import MIMC_P
from .merkle_tree import blake, mk_branch, verify_branch, unpack_merkle_leaf
from .utils import get_pseudorandom_indices

SPOT_CHECKS = 80  # compute_merkle_spot_checks' default, which mk_proof never overrides (stark.py:265, 390)


def get_pseudorandom_ks(m_root, num):
    """stark.py:106-126.  NB the suffixes are the ASCII strings "0x01".. (4 bytes each), not single bytes, and the
    two branches number from 1 and from 0 respectively -- both exactly as in the reference."""
    if 0 <= num <= 4:
        suffixes = [b"0x01", b"0x02", b"0x03", b"0x04"]
    elif num < 10:
        suffixes = [("0x0%s" % i).encode("UTF-8") for i in range(num)]
    else:
        return None  # the reference falls off the end and returns None
    return [int.from_bytes(blake(m_root + suffixes[i]), "big") for i in range(num)]


def compute_merkle_spot_checks(mtree, l_mtree, precision, extension_factor, samples=80):
    """STARK.compute_merkle_spot_checks (stark.py:390-402) on host-side trees: for each sampled position (multiples of
    the extension factor excluded) the branches of mtree at pos and pos + extension_factor and of l_mtree at pos."""
    branches = []
    positions = get_pseudorandom_indices(l_mtree[1], precision, samples, exclude_multiples_of=extension_factor)
    for pos in positions:
        branches.append(mk_branch(mtree, pos))
        branches.append(mk_branch(mtree, (pos + extension_factor) % precision))
        branches.append(mk_branch(l_mtree, pos))
    return branches


def pack_step_polys(step_polys, width, p=MIMC_P):
    """-> (term_coefs bytes, term_exps bytes, term_counts array, degree): the sparse-term form sh_stark_prove takes.
    Terms go in sorted monomial order, the order MultivariatePolynomial.__call__ visits them (:329-338)."""
    coefs, exps, counts, degree = bytearray(), bytearray(), [], 0
    for poly in step_polys:
        items = sorted(poly.coefficients.items())
        if not items:  # the zero polynomial: one zero term keeps the layout rectangular
            items = [((0,) * width, 0)]
        counts.append(len(items))
        for k, c in items:
            if len(k) != width or max(k) > 255:
                raise ValueError("bad power tuple %r" % (k,))
            coefs += (int(c) % p).to_bytes(32, "big")
            exps += bytes(k)
            degree = max(degree, sum(k) if int(c) % p else 0)
    return bytes(coefs), bytes(exps), (ctypes.c_uint32 * width)(*counts), degree


def proof_len(steps, ext, width, degree, samples=SPOT_CHECKS):
    return int(_lib.lib().sh_stark_proof_len(steps, ext, width, degree, samples))


def unpack_proof(flat, steps, ext, width, degree, samples=SPOT_CHECKS):
    """Flat device proof (layout in include/starkhip.h) -> [m_root, l_root, branches, fri_proof]."""
    n = steps * ext
    lg = n.bit_length() - 1
    k = 3 * width
    m_root, l_root = flat[0:32], flat[32:64]
    off = 64
    branches = []
    for _ in range(samples):
        for _ in range(2):
            br = [flat[off:off + 32 * k], flat[off + 32 * k:off + 64 * k]]
            off += 64 * k
            br.extend(flat[off + 32 * i:off + 32 * i + 32] for i in range(lg - 1))
            off += 32 * (lg - 1)
            branches.append(br)
        branches.append([flat[off + 32 * i:off + 32 * i + 32] for i in range(lg + 1)])
        off += 32 * (lg + 1)
    return [m_root, l_root, branches, fri.unpack_proof(flat[off:], n, steps * degree, 40)]


def pack_proof(proof):
    """[m_root, l_root, branches, fri_proof] -> the flat layout (the inverse of unpack_proof)."""
    m_root, l_root, branches, fri_proof = proof
    return m_root + l_root + b"".join(b"".join(br) for br in branches) + fri.pack_proof(fri_proof)


def prove_flat(witness_bytes, input_bytes, steps, ext, width, step_polys, batch=1, samples=SPOT_CHECKS):
    """witness_bytes: [batch][width][steps] wire form, input_bytes: [batch][width] -> batch flat proofs (concatenated)."""
    if len(witness_bytes) != 32 * batch * width * steps or len(input_bytes) != 32 * batch * width:
        raise ValueError("witness must hold batch*width*steps and inputs batch*width 32-byte elements "
                         "(got %d and %d bytes for batch=%d, width=%d, steps=%d)"
                         % (len(witness_bytes), len(input_bytes), batch, width, steps))
    coefs, exps, counts, degree = pack_step_polys(step_polys, width)
    plen = proof_len(steps, ext, width, degree, samples)
    if plen == 0:
        raise NotImplementedError("starks_amd.STARK: unsupported shape (steps=%d, ext=%d, width=%d, degree=%d)"
                                  % (steps, ext, width, degree))
    out = ctypes.create_string_buffer(plen * batch)
    rc = _lib.lib().sh_stark_prove(_lib.ctx(), witness_bytes, input_bytes, steps, ext, width, coefs, exps, counts, samples,
                                   batch, out, plen * batch)
    if rc == -8:  # SH_ERR_CONSTRAINT: the reference's `assert cp % z == 0` (stark.py:76)
        raise AssertionError("constraint polynomial is not a multiple of Z: the witness is not a valid trace")
    _lib.check(rc, "sh_stark_prove")
    return out.raw


def verify_flat(flat, input_bytes, output_bytes, steps, ext, width, step_polys, samples=SPOT_CHECKS):
    """The library's own verifier (sh_stark_verify: host C++ behind the C ABI; the decisions of STARK.verify_proof below) on a FLAT
    proof as prove_flat returns it.  input_bytes / output_bytes: [width] wire-form boundary inputs and witness[dim][-1]."""
    coefs, exps, counts, _ = pack_step_polys(step_polys, width)
    rc = _lib.lib().sh_stark_verify(bytes(flat), len(flat), bytes(input_bytes), bytes(output_bytes), steps, ext, width, coefs, exps,
                                    counts, samples)
    if rc == -9:
        raise AssertionError("STARK proof rejected")
    _lib.check(rc, "sh_stark_verify")
    return True


class STARK(object):
    """Generates and verifies STARKs (stark.py:179-402); same constructor arguments."""

    def __init__(self, field, steps, extension_factor, width, step_polys, spot_check_security_factor=80):
        self.field = field
        self.width = width
        self.steps = steps
        self.step_polys = step_polys
        self.extension_factor = extension_factor
        self.precision = steps * extension_factor
        self.spot_check_security_factor = spot_check_security_factor
        modulus = getattr(field, "p", MIMC_P)
        if modulus != MIMC_P:
            raise NotImplementedError("starks_amd.STARK is compiled for the MiMC prime 2^256 - 351*2^32 + 1")
        self.modulus = modulus
        self.G2 = field(7) ** ((modulus - 1) // self.precision)   # stark.py:205
        self.G1 = self.G2 ** extension_factor                      # stark.py:208
        self.last_step_position = self.G2 ** ((steps - 1) * extension_factor)  # xs[(steps-1)*extension_factor], :212

    def get_degree(self):
        return max([poly.degree() for poly in self.step_polys])

    # ---- prover ----
    def mk_proof(self, witness, boundary):
        """stark.py:233-279.  witness[dim][step]; boundary[dim] = (step, dim, input value)."""
        if len(witness) != self.width or any(len(col) != self.steps for col in witness):
            raise ValueError("witness must be width x steps")
        if len(boundary) < self.width:
            raise IndexError("boundary must hold one (step, dim, value) constraint per dimension")  # boundary[dim], stark.py:91
        wb = b"".join(_lib.to_wire(col) for col in witness)
        ib = _lib.to_wire([constraint[2] for constraint in boundary[:self.width]])
        flat = prove_flat(wb, ib, self.steps, self.extension_factor, self.width, self.step_polys)
        return unpack_proof(flat, self.steps, self.extension_factor, self.width, self.get_degree())

    def compute_merkle_spot_checks(self, mtree, l_mtree, samples=80):
        return compute_merkle_spot_checks(mtree, l_mtree, self.precision, self.extension_factor, samples)

    # ---- verifier (host) ----
    def verify_proof(self, proof, witness, boundary):
        """stark.py:281-316"""
        m_root, l_root, branches, fri_proof = proof
        assert fri.verify_low_degree_proof(fri_proof, l_root, self.G2, self.steps * self.get_degree(),
                                           exclude_multiples_of=self.extension_factor)
        samples = self.spot_check_security_factor
        positions = get_pseudorandom_indices(l_root, self.precision, samples, exclude_multiples_of=self.extension_factor)
        ks = get_pseudorandom_ks(m_root, 4)
        for i, pos in enumerate(positions):
            self.verify_proof_at_position(witness, boundary, ks, proof, i, pos)
        return True

    def verify_proof_native(self, proof, witness, boundary):
        """The same decision from the library's C verifier (sh_stark_verify) on the packed proof -- about 50 times faster than the
        Python verifier above, which stays the line-by-line mirror of the reference (no arithmetic shared with the device code)."""
        inputs = _lib.to_wire([constraint[2] for constraint in boundary[:self.width]])
        outputs = _lib.to_wire([col[-1] for col in witness])
        return verify_flat(pack_proof(proof), inputs, outputs, self.steps, self.extension_factor, self.width, self.step_polys,
                           self.spot_check_security_factor)

    def verify_proof_at_position(self, witness, boundary, ks, proof, i, pos):
        """stark.py:318-388 (the linear-combination check is commented out there, :376-384, and is not made here)."""
        p, width, ext = self.modulus, self.width, self.extension_factor
        m_root, l_root, branches, _ = proof
        x = pow(int(self.G2), pos, p)
        last = int(self.last_step_position)
        leaf1 = unpack_merkle_leaf(verify_branch(m_root, pos, branches[i * 3]), width, 3)
        leaf2 = unpack_merkle_leaf(verify_branch(m_root, (pos + ext) % self.precision, branches[i * 3 + 1]), width, 3)
        verify_branch(l_root, pos, branches[i * 3 + 2], output_as_int=True)
        v1 = [int.from_bytes(v, "big") % p for v in leaf1]
        p_of_x, d_of_x, b_of_x = v1[:width], v1[width:2 * width], v1[2 * width:]
        p_of_g1x = [int.from_bytes(v, "big") % p for v in leaf2[:width]]
        zvalue = (pow(x, self.steps, p) - 1) * pow((x - last) % p, p - 2, p) % p
        f_of_p_of_x = [int(self.step_polys[d](p_of_x)) for d in range(width)]
        for dim in range(width):  # transition constraints C(P(x)) = Z(x) * D(x)
            assert (p_of_g1x[dim] - f_of_p_of_x[dim] - zvalue * d_of_x[dim]) % p == 0
        z2 = (x - 1) * (x - last) % p
        inv = pow((last - 1) % p, p - 2, p)
        for dim in range(width):  # boundary constraints B(x) * Z2(x) + I(x) = P(x)
            input_value = int(boundary[dim][2])
            output_dim = int(witness[dim][-1])
            slope = (output_dim - input_value) * inv % p
            interpolant = (input_value - slope + slope * x) % p
            assert (p_of_x[dim] - b_of_x[dim] * z2 - interpolant) % p == 0


# the module-level building blocks of mk_proof under the reference's names (stark.py:27-177): see stark_polys.py
from .stark_polys import (construct_trace_polynomials, construct_constraint_polynomials, construct_remainder_polynomials,  # noqa: E402,F401
                          construct_boundary_polynomials, compute_pseudorandom_linear_combination_1d,
                          compute_pseudorandom_linear_combination)
