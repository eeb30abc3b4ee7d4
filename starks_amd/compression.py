"""Proof wire format of the reference (starks/compression.py:1-105): a stream of objects in which a
repeated 32-byte node is replaced by a 2-byte back-reference to its first position.  Host side."""

_MARK = {"item": b"----", "branch": b"++++", "sample": b"====", "final": b"////"}


def _dedup_writer():
    out, first_pos = [], {}

    def put(x):
        if x in first_pos:
            out.append(first_pos[x].to_bytes(2, "big"))
        else:
            first_pos[x] = len(out)
            out.append(x)

    return out, put


def compress_fri(prf):
    """compression.py:1-32"""
    out, put = _dedup_writer()
    for root, yproofs in prf[:-1]:
        put(_MARK["item"])
        put(root)
        for yproof in yproofs:
            for branch in yproof:
                for node in branch:
                    put(node)
                put(_MARK["branch"])
            put(_MARK["sample"])
    put(_MARK["final"])
    for x in prf[-1]:
        put(x)
    assert decompress_fri(out) == prf
    return out


def _tokens(stream):
    """The stream with every 2-byte back-reference replaced by the object it points at."""
    for x in stream:
        yield stream[int.from_bytes(x, "big")] if len(x) == 2 else x


def decompress_fri(proof):
    """Inverse of compress_fri (format: compression.py:35-64), as one pass over the resolved tokens: a marker closes
    whatever list is open at its level, anything else is a node of the open branch (or the root right after an item
    marker); everything behind the final marker is the last layer's values."""
    layers, tok = [], _tokens(proof)
    sample, branch = [], []
    for x in tok:
        if x == _MARK["final"]:
            break
        if x == _MARK["item"]:
            layers.append([next(tok), []])
        elif x == _MARK["branch"]:
            sample.append(branch)
            branch = []
        elif x == _MARK["sample"]:
            layers[-1][1].append(sample)
            sample = []
        else:
            branch.append(x)
    layers.append(list(tok))
    return layers


def compress_branches(branches):
    """compression.py:67-82"""
    out, put = _dedup_writer()
    for branch in branches:
        for node in branch:
            put(node)
        put(_MARK["item"])
    assert decompress_branches(out) == branches
    return out


def decompress_branches(proof):
    """Inverse of compress_branches (format: compression.py:85-101): an item marker closes a branch; like the
    reference, a stream that does not end on a marker yields its open branch as the last one."""
    branches, branch, closed = [], [], True
    for x in _tokens(proof):
        closed = x == _MARK["item"]
        if closed:
            branches.append(branch)
            branch = []
        else:
            branch.append(x)
    if not closed:
        branches.append(branch)
    return branches


def proof_bytes(c):
    """The byte string whose length bin_length reports (compression.py:104-105)."""
    return b"".join((b"\xff" if len(x) == 32 else b"") + x for x in c)


def bin_length(c):
    return len(proof_bytes(c))
