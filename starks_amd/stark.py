"""The caller-side helpers around the hot path in STARK.mk_proof (starks/stark.py), SURVEY 8(f) rank 3: the
Fiat-Shamir constants for the linear combination and the Merkle spot checks.  starks/stark.py does not import at
the reference snapshot (stark.py:13), so these are restated from its text; they only sequence functions that ARE
pinned to the live reference (blake, get_pseudorandom_indices, mk_branch)."""
from .merkle_tree import blake, mk_branch
from .utils import get_pseudorandom_indices


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
    """STARK.compute_merkle_spot_checks (stark.py:390-402): for each sampled position (multiples of the extension
    factor excluded) the branches of mtree at pos and pos + extension_factor and of l_mtree at pos."""
    branches = []
    positions = get_pseudorandom_indices(l_mtree[1], precision, samples, exclude_multiples_of=extension_factor)
    for pos in positions:
        branches.append(mk_branch(mtree, pos))
        branches.append(mk_branch(mtree, (pos + extension_factor) % precision))
        branches.append(mk_branch(l_mtree, pos))
    return branches
