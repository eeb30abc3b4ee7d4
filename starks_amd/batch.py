"""Many independent proofs over the GPUs of one node (BASELINE config 5).

Proofs are independent, so the path shards at proof granularity: one process per GPU (torch.distributed,
backend "nccl" = RCCL on ROCm, "gloo" in CPU tests), rank r proves the units `shard(total, r, world)`, and
the only exchange is one all_gather of the 32-byte proof digests at the end -- no collective inside a proof.
"""
import hashlib
import os

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


def mimc_stark_unit(j, steps, constant=42):
    """Unit j of the synthetic many-proof workload as a full STARK: the reference's MiMC formulation of
    test_stark.py:265-293 -- width 2, step polynomials [X_1, X_1 + X_2^3], i.e. x <- x^3 + k with the round constant k
    carried in dimension 1 -- started from (k, 3 + j).  Returns (witness[dim][step], inputs)."""
    x = (3 + j) % MIMC_P
    col = [x]
    for _ in range(steps - 1):
        x = (x * x * x + constant) % MIMC_P
        col.append(x)
    return [[constant] * steps, col], [constant, (3 + j) % MIMC_P]


def prove_stark_batch(unit_ids, steps, ext=8, chunk=16):
    """Full STARK proofs (STARK.mk_proof, stark.py:233-279) of the units `mimc_stark_unit(j)`, `chunk` proofs per
    batched launch.  Returns [(j, flat_proof_bytes)] (layout: include/starkhip.h).  Runs on this process's GPU."""
    from . import stark
    from .modp import IntegersModP
    from .multivariate_polynomial import generate_Xi_s
    X1, X2 = generate_Xi_s(IntegersModP(MIMC_P), 2)
    polys = [X1, X1 + X2**3]
    plen = stark.proof_len(steps, ext, 2, 3)
    out = []
    ids = list(unit_ids)
    for c in range(0, len(ids), chunk):
        part = ids[c:c + chunk]
        wit, inp = bytearray(), bytearray()
        for j in part:
            w, i = mimc_stark_unit(j, steps)
            for col in w:
                wit += b"".join(v.to_bytes(32, "big") for v in col)
            inp += b"".join(v.to_bytes(32, "big") for v in i)
        flat = stark.prove_flat(bytes(wit), bytes(inp), steps, ext, 2, polys, batch=len(part))
        out.extend((j, flat[i * plen:(i + 1) * plen]) for i, j in enumerate(part))
    return out


def prove_stark_units_device(first_unit, count, steps, ext=8, chunk=32, keep=None, constant=42):
    """BASELINE configs[4] on this process's GPU with device-resident data: the STARK proofs of units first_unit ..
    first_unit + count - 1 (`mimc_stark_unit`), `chunk` proofs per batched launch.  Witnesses are generated on the device
    (sh_dev_fill_mimc_units), proofs stay on the device and only their SHA-256 digests come back -- unless `keep` is a set of
    unit ids whose flat proofs are returned too.  -> (digests in unit order, {unit: flat proof}).

    This is what bench.py --workload c5 times; `StarkUnitProver` below exposes the pieces so that generation can stay untimed."""
    # Two provers, the second on a library context (stream + workspaces) of its own: batch j is launched before batch j - 1 is
    # downloaded and digested, so that the host side of one batch runs beside the device side of the next.  Downloads land in
    # page-locked buffers (no staging copy) and are hashed in place, several proofs at a time (hashlib releases the GIL).
    from concurrent.futures import ThreadPoolExecutor
    from . import _lib
    provers = [StarkUnitProver(steps, ext, chunk, constant)]
    if count > chunk:
        provers.append(StarkUnitProver(steps, ext, chunk, constant, second_context=True))
    bufs = [_lib.PinnedBuffer(pr.plen * chunk) for pr in provers]
    pool = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1))
    try:
        digs, kept = [], {}

        def collect(which, first, k):
            pr, plen = provers[which], provers[which].plen
            mv = memoryview(pr.download_into(k, bufs[which]))
            digs.extend(pool.map(lambda i: hashlib.sha256(mv[i * plen:(i + 1) * plen]).digest(), range(k)))
            if keep:
                for i in range(k):
                    if first + i in keep:
                        kept[first + i] = bytes(mv[i * plen:(i + 1) * plen])

        pending = None
        for j, c in enumerate(range(0, count, chunk)):
            k = min(chunk, count - c)
            which = j % len(provers)
            provers[which].generate(first_unit + c, k)
            provers[which].prove(k)
            if pending is not None:
                collect(*pending)
            pending = (which, first_unit + c, k)
        if pending is not None:
            collect(*pending)
        return digs, kept
    finally:
        pool.shutdown()
        for b in bufs:
            b.close()
        for pr in provers:
            pr.close()


class StarkUnitProver(object):
    """Device buffers for `chunk` MiMC STARK units of `steps` steps: generate() fills witnesses and inputs on the device,
    prove() launches sh_dev_stark_prove on them (asynchronous), status()/download() synchronise.  second_context: the
    process's second library context instead of the first (another stream: prove_stark_units_device pipelines two provers)."""

    def __init__(self, steps, ext=8, chunk=32, constant=42, second_context=False):
        import ctypes
        from . import _lib, stark
        from .modp import IntegersModP
        from .multivariate_polynomial import generate_Xi_s
        self._lib, self.L, self.ctx = _lib, _lib.lib(), _lib.ctx()
        if second_context:  # the process's second stream on the same GPU (_lib.second_ctx)
            self.ctx = _lib.second_ctx()
        self.steps, self.ext, self.chunk, self.constant = steps, ext, chunk, constant
        X1, X2 = generate_Xi_s(IntegersModP(MIMC_P), 2)
        self.coefs, self.exps, self.counts, self.degree = stark.pack_step_polys([X1, X1 + X2**3], 2)
        self.plen = stark.proof_len(steps, ext, 2, self.degree)
        if self.plen == 0:
            raise NotImplementedError("unsupported shape (steps=%d, ext=%d)" % (steps, ext))
        self.dw, self.di, self.dp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        for ptr, nbytes in ((self.dw, 64 * steps * chunk), (self.di, 64 * chunk), (self.dp, self.plen * chunk)):
            _lib.check(self.L.sh_dev_alloc(self.ctx, nbytes, ctypes.byref(ptr)), "sh_dev_alloc")

    def generate(self, first_unit, k):
        self._lib.check(self.L.sh_dev_fill_mimc_units(self.ctx, self.dw, self.di, self.steps, first_unit, k, self.constant),
                        "sh_dev_fill_mimc_units")

    def prove(self, k):
        """Proves the k units generate() left on the device (the witness is read only and can be proved again)."""
        self._lib.check(self.L.sh_dev_stark_prove(self.ctx, self.dw, self.di, self.steps, self.ext, 2, self.coefs, self.exps,
                                                  self.counts, 80, k, self.dp), "sh_dev_stark_prove")

    def status(self):
        rc = self.L.sh_stark_status(self.ctx)
        if rc == -8:
            raise AssertionError("a witness of the batch is not a valid trace")
        self._lib.check(rc, "sh_stark_status")

    def download_into(self, k, pinned):
        """The k flat proofs into a page-locked buffer (_lib.PinnedBuffer of at least k * plen bytes); returns its ctypes view."""
        self.status()
        self._lib.check(self.L.sh_dev_download(self.ctx, self.dp, pinned.view, self.plen * k), "sh_dev_download")
        return pinned.view

    def download(self, k):
        import ctypes
        self.status()
        out = ctypes.create_string_buffer(self.plen * k)
        self._lib.check(self.L.sh_dev_download(self.ctx, self.dp, out, self.plen * k), "sh_dev_download")
        raw = out.raw
        return [raw[i * self.plen:(i + 1) * self.plen] for i in range(k)]

    def close(self):
        for ptr in (self.dw, self.di, self.dp):
            if ptr:
                self.L.sh_dev_free(self.ctx, ptr)
        self.dw = self.di = self.dp = None


def digest(proof_bytes):
    return hashlib.sha256(proof_bytes).digest()
