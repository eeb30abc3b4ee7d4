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


def decompress_fri(proof):
    """compression.py:35-64"""

    def get(pos):
        x = proof[pos]
        return proof[int.from_bytes(x, "big")] if len(x) == 2 else x

    o, pos = [], 0
    while proof[pos] != _MARK["final"]:
        assert get(pos) == _MARK["item"]
        root = get(pos + 1)
        pos += 2
        yproofs = []
        while get(pos) not in (_MARK["item"], _MARK["final"]):
            yproof = []
            while get(pos) != _MARK["sample"]:
                branch = []
                while get(pos) != _MARK["branch"]:
                    branch.append(get(pos))
                    pos += 1
                yproof.append(branch)
                pos += 1
            yproofs.append(yproof)
            pos += 1
        o.append([root, yproofs])
    pos += 1
    o.append([get(x) for x in range(pos, len(proof))])
    return o


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
    """compression.py:85-101"""

    def get(pos):
        x = proof[pos]
        return proof[int.from_bytes(x, "big")] if len(x) == 2 else x

    o, pos = [], 0
    while pos < len(proof):
        branch = []
        while pos < len(proof) and get(pos) != _MARK["item"]:
            branch.append(get(pos))
            pos += 1
        o.append(branch)
        pos += 1
    return o


def proof_bytes(c):
    """The byte string whose length bin_length reports (compression.py:104-105)."""
    return b"".join((b"\xff" if len(x) == 32 else b"") + x for x in c)


def bin_length(c):
    return len(proof_bytes(c))
