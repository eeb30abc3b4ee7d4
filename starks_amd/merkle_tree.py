"""BLAKE2s Merkle commitment on the MI355X behind the reference's call sites (starks/merkle_tree.py:5-86):
merkelize(L) -> list[bytes] of length 2n, mk_branch(tree, index), verify_branch(root, index, proof)."""
import ctypes
from hashlib import blake2s

from . import _lib
from .wireseq import NodeList, WireList


def blake(x):
    return blake2s(x).digest()  # merkle_tree.py:5 (host side: verifier and Fiat-Shamir glue only)


def permute4(values):
    """merkle_tree.py:11-23: out[4i + j] = in[i + j * n/4]"""
    q = len(values) // 4
    return [values[i + j * q] for i in range(q) for j in range(4)]


def get_index_in_permuted(x, L):
    """merkle_tree.py:26-33"""
    q = L // 4
    return x // q + 4 * (x % q)


def _leaf_list(L):
    """merkle_tree.py:47-53: ints as 32 big-endian bytes, byte strings as they are, field elements through to_bytes()"""
    return [x if isinstance(x, bytes) else x.to_bytes(32, "big") if isinstance(x, int) else x.to_bytes() for x in L]


def _leaf_bytes(L):
    out = _leaf_list(L)
    if any(len(x) != 32 for x in out):
        raise NotImplementedError("the device tree hashes 32-byte leaves (field elements) only")
    return b"".join(out)


def _host_merkelize(leaves):
    """Trees the device code does not cover -- leaves that are not 32-byte values, or a leaf count that is not a power of two
    >= 4 (the reference hashes whatever to_bytes gives and any length, merkle_tree.py:36-56) -- on the host, never the hot path:
    the same heap layout, filled a generation at a time (the parents lo..hi-1 of the nodes 2 lo..2 hi-1)."""
    leaves = permute4(leaves)
    n = len(leaves)
    nodes = [b""] * n + leaves
    hi = n
    while hi > 1:
        lo = (hi + 1) // 2
        nodes[lo:hi] = [blake(a + b) for a, b in zip(nodes[2 * lo:2 * hi:2], nodes[2 * lo + 1:2 * hi:2])]
        hi = lo
    return nodes


def merkelize_bytes(leaves):
    """leaves: n * 32 bytes in natural order -> 2n * 32 bytes (nodes[0] = zeros, nodes[1] = root)."""
    n = len(leaves) // 32
    if n < 4 or n & (n - 1):
        raise NotImplementedError("starks_amd.merkelize needs a power-of-two number of leaves >= 4 (got %d)" % n)
    out = ctypes.create_string_buffer(64 * n)
    _lib.check(_lib.lib().sh_merkelize(_lib.ctx(), leaves, n, out), "sh_merkelize")
    return out.raw


def merkelize(L):
    """merkle_tree.py:36-56.  Leaves may be ints, byte strings or field elements (:47-53).  Power-of-two counts (>= 4) of
    32-byte leaves -- every tree of the proving path -- are hashed on the GPU; anything else on the host."""
    if isinstance(L, WireList) and len(L) >= 4 and not len(L) & (len(L) - 1):
        return NodeList(merkelize_bytes(L.wire_bytes()))  # a transform's output goes to the tree as the bytes it is
    leaves = _leaf_list(list(L))
    n = len(leaves)
    if n < 4 or n & (n - 1) or any(len(x) != 32 for x in leaves):
        return _host_merkelize(leaves)
    # 2n nodes over one buffer, entry 0 reads as b"" like the reference's slot 0 (wireseq.NodeList: no 2n bytes objects)
    return NodeList(merkelize_bytes(b"".join(leaves)))


def mk_branch(tree, index):
    """merkle_tree.py:59-68: the leaf, then one sibling per level (root excluded)."""
    half = len(tree) // 2
    index = get_index_in_permuted(index, half) + half
    o = [tree[index]]
    while index > 1:
        o.append(tree[index ^ 1])
        index //= 2
    return o


def verify_branch(root, index, proof, output_as_int=False):
    """merkle_tree.py:71-86"""
    half = 2**len(proof) // 2
    index = get_index_in_permuted(index, half) + half
    v = proof[0]
    for p in proof[1:]:
        v = blake(p + v) if index % 2 else blake(v + p)
        index //= 2
    assert v == root
    return int.from_bytes(proof[0], "big") if output_as_int else proof[0]


def merkelize_polynomial_evaluations(dims, polynomial_evals):
    """merkle_tree.py:94-119: one tree over several polynomials' evaluations; leaf x is the concatenation of every
    polynomial's 32-byte value at x.  Returns the reference's list: 2n entries, [n:] are the (32 k)-byte leaves."""
    k = len(polynomial_evals)
    n = len(polynomial_evals[0])
    if any(len(e) != n for e in polynomial_evals):
        raise ValueError("all polynomials must be evaluated on the same domain")
    if n < 4 or n & (n - 1):
        raise NotImplementedError("starks_amd.merkelize needs a power-of-two number of leaves >= 4 (got %d)" % n)
    data = b"".join(e.wire_bytes() if isinstance(e, WireList) else _leaf_bytes(list(e)) for e in polynomial_evals)
    nodes = ctypes.create_string_buffer(32 * n)
    leaves = ctypes.create_string_buffer(32 * k * n)
    _lib.check(_lib.lib().sh_merkelize_packed(_lib.ctx(), data, n, k, nodes, leaves), "sh_merkelize_packed")
    return NodeList(nodes.raw, tail=leaves.raw, tail_width=32 * k)  # n hash nodes (entry 0 = b""), then the n packed leaves


def unpack_merkle_leaf(leaf, dims, num_polys):
    """merkle_tree.py:121-147: split a packed leaf back into its 32-byte values."""
    return [leaf[32 * i:32 * i + 32] for i in range(num_polys * dims)]
