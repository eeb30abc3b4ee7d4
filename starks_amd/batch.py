"""Many independent proofs over the GPUs of one node (BASELINE config 5).

Proofs are independent, so the path shards at proof granularity: one process per GPU (torch.distributed,
backend "nccl" = RCCL on ROCm, "gloo" in CPU tests), rank r proves the units `shard(total, r, world)`, and
the only exchange is one all_gather of the 32-byte proof digests at the end -- no collective inside a proof.
"""
import hashlib

from ._lib import MIMC_P


def shard(total, rank, world):
    """Contiguous, balanced partition of range(total): the first total % world ranks get one extra unit."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def gather_digests(local, total, rank, world, dist=None, device="cpu"):
    """all_gather of fixed-size 32-byte digests; returns the list for units 0..total-1 on every rank."""
    if world == 1 or dist is None:
        return list(local)
    import torch
    width = len(shard(total, 0, world))  # the largest shard
    buf = torch.zeros(width * 32, dtype=torch.uint8, device=device)
    flat = b"".join(local)
    if flat:
        buf[:len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r in range(world):
        raw = bytes(parts[r].cpu().numpy().tobytes())
        k = len(shard(total, r, world))
        out.extend(raw[32 * i:32 * i + 32] for i in range(k))
    return out


def prove_mimc_batch(unit_ids, steps, ext=8, samples=40, chunk=16):
    """FRI proofs of the MiMC traces t0 = 3 + j for j in unit_ids (SURVEY 8(d) synthetic input), `chunk`
    proofs per batched launch.  Returns [(j, flat_proof_bytes)].  Runs on this process's GPU."""
    from . import fft, fri
    from .utils import mimc_trace
    n = steps * ext
    g2 = pow(7, (MIMC_P - 1) // n, MIMC_P)
    g1 = pow(g2, ext, MIMC_P)
    plen = fri.proof_len(n, steps, samples)
    out = []
    ids = list(unit_ids)
    for c in range(0, len(ids), chunk):
        part = ids[c:c + chunk]
        traces = b"".join(b"".join(v.to_bytes(32, "big") for v in mimc_trace(3 + j, steps)) for j in part)
        coeffs = fft.ntt_bytes(traces, steps, g1, inverse=True, batch=len(part))  # trace polynomials (stark.py:27-36)
        flat = fri.prove_flat(coeffs, n, g2, steps, ext, samples, batch=len(part))
        out.extend((j, flat[i * plen:(i + 1) * plen]) for i, j in enumerate(part))
    return out


def digest(proof_bytes):
    return hashlib.sha256(proof_bytes).digest()
