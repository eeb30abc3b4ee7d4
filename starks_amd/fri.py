"""FRI commit loop on the MI355X behind the reference's (commented-out) prime-field FRI driver,
SmoothSubgroupFRI (starks/fri.py:176-366): NTT -> Merkle commit -> fold-by-4 at the challenge taken from
the root -> Merkle commit -> sample 40 rows -> 5 branches per row, recursing on a domain 4x smaller until
maxdeg_plus_1 <= 16.  The whole loop runs on the device (csrc/capi.hip:run_fri); one copy brings the flat
proof back and `unpack_proof` rebuilds the reference's nested lists.

    SmoothSubgroupFRI(field).generate_proximity_proof(f, root_of_unity, maxdeg_plus_1,
                                                      exclude_multiples_of=0, fri_spot_check_security_factor=40)
    prove_low_degree(...)  -- the upstream name of the same function (test_fri.py:170)
    SmoothSubgroupFRI(field).verify_proximity_proof(...) -- host-side verifier (fri.py:268-366)
"""
import ctypes

from . import _lib
from ._lib import MIMC_P
from .merkle_tree import verify_branch, blake
from .utils import get_pseudorandom_indices
from .wireseq import WireList


def proof_len(n, maxdeg_plus_1, samples=40):
    return int(_lib.lib().sh_fri_proof_len(n, maxdeg_plus_1, samples))


def unpack_proof(flat, n, maxdeg_plus_1, samples=40):
    """Flat device proof (layout in include/starkhip.h) -> [[root2, [[branch x5] x S]] x rounds, [final values]]."""
    out, off, first = [], 0, True
    while maxdeg_plus_1 > 16:
        lg = n.bit_length() - 1
        s = samples if first else 40  # the reference's recursion falls back to 40 (fri.py:262-266)
        l2, l1 = lg - 1, lg + 1
        root2 = flat[off:off + 32]
        off += 32
        branches = []
        for _ in range(s):
            bset = []
            for ln in (l2, l1, l1, l1, l1):
                bset.append([flat[off + 32 * k:off + 32 * k + 32] for k in range(ln)])
                off += 32 * ln
            branches.append(bset)
        out.append([root2, branches])
        n //= 4
        maxdeg_plus_1 //= 4
        first = False
    out.append([flat[off + 32 * k:off + 32 * k + 32] for k in range(n)])
    assert off + 32 * n == len(flat)
    return out


def pack_proof(proof):
    """The reference's nested proof -> the flat layout (the inverse of unpack_proof): every entry in order."""
    parts = []
    for root2, branches in proof[:-1]:
        parts.append(root2)
        for bset in branches:
            for br in bset:
                parts.extend(br)
    parts.extend(proof[-1])
    return b"".join(parts)


def prove_flat(coeff_bytes, n, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0, samples=40, batch=1):
    """coeff_bytes: batch * n_coeffs wire-form coefficients -> batch flat proofs (bytes, concatenated)."""
    n_coeffs = len(coeff_bytes) // (32 * batch)
    plen = proof_len(n, maxdeg_plus_1, samples)
    out = ctypes.create_string_buffer(plen * batch)
    rc = _lib.lib().sh_fri_prove(_lib.ctx(), coeff_bytes, n_coeffs, n, int(root_of_unity).to_bytes(32, "big"),
                                 maxdeg_plus_1, exclude_multiples_of, samples, batch, out, plen * batch)
    _lib.check(rc, "sh_fri_prove")
    return out.raw


def prove_low_degree(f, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0, fri_spot_check_security_factor=40):
    """fri.py:189-266.  `f`: a Poly (its .coefficients) or a list of coefficients."""
    coeffs = f.coefficients if hasattr(f, "coefficients") else f
    if isinstance(coeffs, WireList):  # e.g. an inverse transform's output: the bytes go to the prover as they are
        if int(coeffs.field.p) != MIMC_P:
            raise NotImplementedError("starks_amd accelerates the MiMC prime field only")
    else:
        coeffs = list(coeffs)
        for c in coeffs:
            if hasattr(c, "p") and int(c.p) != MIMC_P:
                raise NotImplementedError("starks_amd accelerates the MiMC prime field only")
            break
    n = _lib.order_of_root(root_of_unity)
    if n is None:
        raise NotImplementedError("root_of_unity must have power-of-two order")
    if len(coeffs) > n:
        raise ValueError("polynomial has more coefficients than the evaluation domain has points")
    flat = prove_flat(_lib.to_wire(coeffs), n, int(root_of_unity), maxdeg_plus_1, exclude_multiples_of,
                      fri_spot_check_security_factor)
    return unpack_proof(flat, n, maxdeg_plus_1, fri_spot_check_security_factor)


# ---- host-side verifier (fri.py:268-366); the prover is the accelerated path ------------------------------
def _interp4_eval(xs, ys, x, p):
    """Value at x of the cubic through (xs[k], ys[k]) -- what multi_interp_4 + Poly.__call__ compute
    (poly_utils.py:412-440, polynomial.py:158-164)."""
    total = 0
    for k in range(4):
        num, den = ys[k], 1
        for l in range(4):
            if l != k:
                num = num * (x - xs[l]) % p
                den = den * (xs[k] - xs[l]) % p
        total = (total + num * pow(den, p - 2, p)) % p
    return total


def _lagrange_eval_all(xs, ys, pts, p):
    """Values at `pts` of the interpolant through (xs, ys) (the final-layer degree check, fri.py:352-358)."""
    out = []
    for x in pts:
        total = 0
        for k in range(len(xs)):
            num, den = ys[k], 1
            for l in range(len(xs)):
                if l != k:
                    num = num * (x - xs[l]) % p
                    den = den * (xs[k] - xs[l]) % p
            total = (total + num * pow(den, p - 2, p)) % p
        out.append(total)
    return out


def verify_low_degree_proof(proof, merkle_root, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0,
                            fri_spot_check_security_factor=40, modulus=MIMC_P):
    """fri.py:268-366.  Returns True or raises AssertionError, like the reference."""
    p = modulus
    w = int(root_of_unity) % p
    roudeg = _lib.order_of_root(w, p)
    assert roudeg is not None
    for prf in proof[:-1]:
        root2, branches = prf
        special_x = int.from_bytes(merkle_root, "big")
        q = roudeg // 4
        ys = get_pseudorandom_indices(root2, q, fri_spot_check_security_factor, exclude_multiples_of=exclude_multiples_of)
        quartic = [pow(w, q * j, p) for j in range(4)]
        for i, y in enumerate(ys):
            x1 = pow(w, y, p)
            xcoords = [quartic[j] * x1 % p for j in range(4)]
            row = [verify_branch(merkle_root, y + q * j, b, output_as_int=True) % p
                   for j, b in zip(range(4), branches[i][1:])]
            colval = verify_branch(root2, y, branches[i][0], output_as_int=True) % p
            assert _interp4_eval(xcoords, row, special_x, p) == colval
        merkle_root = root2
        w = pow(w, 4, p)
        maxdeg_plus_1 //= 4
        roudeg //= 4
    data = [int.from_bytes(x, "big") for x in proof[-1]]
    assert maxdeg_plus_1 <= 16
    # the final layer's Merkle root must match the last committed root (host hashing: tiny)
    nodes = [b""] * len(data) + [data[i + j * (len(data) // 4)].to_bytes(32, "big")
                                 for i in range(len(data) // 4) for j in range(4)]
    for i in range(len(data) - 1, 0, -1):
        nodes[i] = blake(nodes[2 * i] + nodes[2 * i + 1])
    assert nodes[1] == merkle_root
    powers = [pow(w, i, p) for i in range(len(data))]
    pts = [x for x in range(len(data)) if x % exclude_multiples_of] if exclude_multiples_of else list(range(len(data)))
    head, tail = pts[:maxdeg_plus_1], pts[maxdeg_plus_1:]
    got = _lagrange_eval_all([powers[x] for x in head], [data[x] % p for x in head], [powers[x] for x in tail], p)
    assert got == [data[x] % p for x in tail]
    return True


def verify_flat(flat, merkle_root, n, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0, samples=40):
    """The library's own verifier (sh_fri_verify: host C++ behind the C ABI, same decisions as verify_low_degree_proof above) on a
    FLAT proof as prove_flat returns it.  True / AssertionError like the reference's verifier."""
    rc = _lib.lib().sh_fri_verify(bytes(flat), len(flat), bytes(merkle_root), n, int(root_of_unity).to_bytes(32, "big"),
                                  maxdeg_plus_1, exclude_multiples_of, samples)
    if rc == -9:
        raise AssertionError("FRI proof rejected")
    _lib.check(rc, "sh_fri_verify")
    return True


class SmoothSubgroupFRI(object):
    """fri.py:176-366 (class name and method signatures of the reference's commented-out driver)."""

    def __init__(self, field):
        self.field = field

    def generate_proximity_proof(self, f, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0,
                                 fri_spot_check_security_factor=40):
        return prove_low_degree(f, root_of_unity, maxdeg_plus_1, exclude_multiples_of, fri_spot_check_security_factor)

    def verify_proximity_proof(self, proof, merkle_root, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0,
                               fri_spot_check_security_factor=40):
        return verify_low_degree_proof(proof, merkle_root, root_of_unity, maxdeg_plus_1, exclude_multiples_of,
                                       fri_spot_check_security_factor, modulus=int(self.field.p))


    def verify_proximity_proof_native(self, proof, merkle_root, root_of_unity, maxdeg_plus_1, exclude_multiples_of=0,
                                      fri_spot_check_security_factor=40):
        """The same decision from the library's C verifier (sh_fri_verify) on the packed proof: milliseconds where the Python
        verifier above takes tenths of a second.  (verify_proximity_proof stays the line-by-line mirror of the reference: it
        shares no arithmetic with the device code, which is what the parity tests want from a checker.)"""
        n = _lib.order_of_root(root_of_unity)
        if n is None:
            raise NotImplementedError("root_of_unity must have power-of-two order")
        return verify_flat(pack_proof(proof), merkle_root, n, root_of_unity, maxdeg_plus_1, exclude_multiples_of,
                           fri_spot_check_security_factor)


FRI = SmoothSubgroupFRI  # the name starks/stark.py:13 tries to import
