#!/usr/bin/env python3
"""bench.py -- the hot-path benchmark (driver contract: one JSON line on rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ntt|c5]

Launch.  With --gpus N > 1 and no torch.distributed environment, bench.py starts N ranks itself
(`python -m torch.distributed.run`, one process per GPU, as a CHILD process and before this process touches the GPU);
launched by the driver through torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

Workload `ntt` (default; BASELINE.json's metric on configs[3], the largest single-GPU configuration and the one the
HBM-roofline target is quoted on): ONE 2^24-point forward NTT followed by the inverse NTT over the MiMC prime, data
resident in HBM (`--logn`, `--batch` select other shapes: `--logn 20 --batch 8` is the trace-columns shape).  Ranks
transform their own vectors: weak scaling, no data-path collective.  `value` = transformed elements of all
ranks / max-over-ranks time.  The other shapes of the metric are top-level keys measured after the timed region (rank 0,
N = 1): `single_vector_elements_per_s` (configs[1] literally: one 2^20-point pair), `ntt_2^20_x8_elements_per_s` (8 trace
columns per launch sequence), `fri_commit_ms_2^20_trace` / `fri_commit_ms_2^14_trace` (the metric's second half).

Workload `c5` (BASELINE configs[4]): `--units` (512) independent 2^16-step MiMC STARK proofs (STARK.mk_proof,
stark.py:233-279; unit j = test_stark.py:265-293 started from 3 + j), sharded over the ranks by proof index
(starks_amd/batch.py:shard), `--chunk` (256; fewer when a rank's shard would leave a context idle) proofs per batched launch, the launches dealt in turn to `--c5-streams` (2) library
contexts; a step = the whole batch once; `value` = proofs/s; strong
scaling; the only exchange is one all_gather of the 64-byte proof headers (m_root | l_root) per step (RCCL).  Every run of
the default workload also runs three such steps after its timed region and reports it under `c5` / `c5_proofs_per_s`, so
that the driver's N = 1, 2, 4, 8 runs record the proofs/s curve too.

Also on the line:
  roofline      -- the dominant kernel: algorithmic bytes / HIP-event time of the timed region on the
                   library's stream.
  cpu_baseline  -- the C oracle (oracle/oracle.c: the reference's recursive algorithm) on a bounded sample of the same
                   workload, rank 0 at N = 1 only.
  extra         -- FRI commits, Merkle commit of 2^24 leaves, LDE, whole STARK proofs.
"""
import argparse
import datetime
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = 2**256 - 2**32 * 351 + 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def root_of(n):
    return pow(7, (P - 1) // n, P)


class Dev:
    """Thin helper over the device-resident C ABI."""

    def __init__(self):
        from starks_amd import _lib
        self.L = _lib.lib()
        self.ctx = _lib.ctx()
        self._lib = _lib

    def ck(self, rc, what):
        self._lib.check(rc, what)

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        self.ck(self.L.sh_dev_alloc(self.ctx, nbytes, ctypes.byref(p)), "sh_dev_alloc")
        return p

    def free(self, p):
        self.ck(self.L.sh_dev_free(self.ctx, p), "sh_dev_free")

    def sync(self):
        self.ck(self.L.sh_sync(self.ctx), "sh_sync")

    def timed(self, fn, reps):
        """HIP-event time (ms per rep) of `reps` calls of fn on the library stream."""
        fn()
        self.sync()
        self.ck(self.L.sh_timer_start(self.ctx), "timer")
        for _ in range(reps):
            fn()
        ms = ctypes.c_float()
        self.ck(self.L.sh_timer_stop(self.ctx, ctypes.byref(ms)), "timer")
        return ms.value / reps


# ---------------------------------------------------------------------------------------------------------------------
# config 5: many independent STARK proofs, sharded by proof index
# ---------------------------------------------------------------------------------------------------------------------
class ProofShard:
    """This rank's units of the many-proof workload, resident in HBM: the witnesses (generated on the device,
    untimed; the prover leaves them intact) and the flat proofs of the whole shard."""

    def __init__(self, dev, units, steps, ext=8, chunk=128, streams=1):
        from starks_amd import stark
        from starks_amd.modp import IntegersModP
        from starks_amd.multivariate_polynomial import generate_Xi_s
        self.dev, self.units, self.steps, self.ext, self.chunk = dev, list(units), steps, ext, chunk
        X1, X2 = generate_Xi_s(IntegersModP(P), 2)
        self.polys = [X1, X1 + X2**3]
        self.coefs, self.exps, self.counts, self.degree = stark.pack_step_polys(self.polys, 2)
        self.plen = stark.proof_len(steps, ext, 2, self.degree)
        k = max(1, len(self.units))
        L, ctx = dev.L, dev.ctx
        self.wbytes = 64 * steps  # one unit's witness: 2 columns
        self.d_wit = dev.alloc(self.wbytes * k)
        self.d_inp = dev.alloc(64 * k)
        self.d_proofs = dev.alloc(self.plen * k)
        if self.units:  # shards are contiguous ranges (batch.shard)
            assert self.units == list(range(self.units[0], self.units[0] + len(self.units)))
            dev.ck(L.sh_dev_fill_mimc_units(ctx, self.d_wit, self.d_inp, steps, self.units[0], len(self.units), 42), "fill units")
        dev.sync()
        # `streams` library contexts on this GPU (each one stream and its own workspaces): successive batched launches go to them
        # in turn, so that the latency-bound tail of one launch sequence (the small FRI rounds) runs beside the wide kernels of
        # the next (include/starkhip.h: contexts are independent; device buffers belong to the device, not to a context)
        self.host = None  # page-locked host copy of the shard's proofs (deliver())
        self.ctxs = [ctx]
        for _ in range(1, max(1, streams)):
            other = ctypes.c_void_p()
            dev.ck(L.sh_ctx_create(dev._lib.default_device(), ctypes.byref(other)), "sh_ctx_create")
            self.ctxs.append(other)

    def prove_all(self, deliver=False):
        """One step: every unit of the shard, `chunk` per launch sequence; asynchronous on the library stream.  deliver: every
        launch sequence is followed by the copy of its flat proofs into page-locked host memory on the context's copy stream
        (sh_dev_download_async): batch k's proofs cross PCIe while batch k + 1 is being proved; delivered() waits for them."""
        dev, L = self.dev, self.dev.L
        if deliver and self.host is None and self.units:
            self.host = dev._lib.PinnedBuffer(self.plen * len(self.units))
        for j, (c, k) in enumerate(self.schedule(deliver)):
            ctx = self.ctxs[j % len(self.ctxs)]
            dev.ck(L.sh_dev_stark_prove(ctx, ctypes.c_void_p(self.d_wit.value + self.wbytes * c),
                                        ctypes.c_void_p(self.d_inp.value + 64 * c), self.steps, self.ext, 2,
                                        self.coefs, self.exps, self.counts, 80, k,
                                        ctypes.c_void_p(self.d_proofs.value + self.plen * c)), "stark prove")
            if deliver:
                dev.ck(L.sh_dev_download_async(ctx, ctypes.c_void_p(self.d_proofs.value + self.plen * c),
                                               ctypes.c_void_p(self.host.ptr.value + self.plen * c), self.plen * k), "deliver")

    def schedule(self, deliver):
        return launch_schedule(len(self.units), self.chunk, deliver)

    def delivered(self):
        """Waits for the copies deliver=True queued; returns the host view of the shard's proofs (None for an empty shard)."""
        for ctx in self.ctxs:
            self.dev.ck(self.dev.L.sh_io_sync(ctx), "sh_io_sync")
        return self.host.view if self.host is not None else None

    def headers(self):
        """m_root | l_root of every proof of the shard (64 B each); synchronises; raises on an invalid witness."""
        k = len(self.units)
        for ctx in self.ctxs:  # every context's stream drained, every context's witness flags read
            rc = self.dev.L.sh_stark_status(ctx)
            if rc != 0:
                self.dev.ck(rc, "stark status")
        out = ctypes.create_string_buffer(64 * max(k, 1))
        if k:
            self.dev.ck(self.dev.L.sh_dev_download_2d(self.dev.ctx, self.d_proofs, self.plen, out, 64, k), "headers")
        return [out.raw[64 * i:64 * i + 64] for i in range(k)]

    def proof(self, i):
        out = ctypes.create_string_buffer(self.plen)
        self.dev.ck(self.dev.L.sh_dev_download(self.dev.ctx, ctypes.c_void_p(self.d_proofs.value + self.plen * i), out, self.plen), "dl")
        return out.raw

    def digest_all(self):
        """SHA-256 over every flat proof of the shard, in unit order (downloaded 32 proofs at a time)."""
        h = hashlib.sha256()
        k = len(self.units)
        out = ctypes.create_string_buffer(self.plen * min(32, max(k, 1)))
        for c in range(0, k, 32):
            m = min(32, k - c)
            self.dev.ck(self.dev.L.sh_dev_download(self.dev.ctx, ctypes.c_void_p(self.d_proofs.value + self.plen * c), out, self.plen * m), "dl")
            h.update(memoryview(out)[:self.plen * m])
        return h.hexdigest()

    def close(self):
        if self.host is not None:
            self.delivered()
            self.host.close()
            self.host = None
        for p in (self.d_wit, self.d_inp, self.d_proofs):
            self.dev.free(p)
        for other in self.ctxs[1:]:
            self.dev.L.sh_ctx_destroy(other)
        self.ctxs = self.ctxs[:1]


def launch_schedule(units, chunk, deliver):
    """[(first unit, units)] of a step's launch sequences.  On the device: `chunk` units each.  Delivering: the same, but the tail
    halves from launch to launch (256, 128, 64, 32, 32 for 512 units) -- the copy of a launch's proofs can only start when the launch
    has finished, so the last launch's copy is the one nothing hides, and it should be small."""
    out, c, rem = [], 0, units
    while rem > 0:
        k = min(chunk, rem)
        if deliver and rem - k < k and k > 32:
            k = max(32, (rem // 2 + 15) // 16 * 16)
        out.append((c, k))
        c += k
        rem -= k
    return out


def gather_headers(local, total, rank, world, dist, tdev, force=False):
    """all_gather of the fixed-size proof headers (the only exchange of the many-proof path)."""
    if world == 1 and not force:
        return list(local)
    import torch
    from starks_amd.batch import shard
    width = len(shard(total, 0, world))
    buf = torch.zeros(width * 64, dtype=torch.uint8, device=tdev)
    flat = b"".join(local)
    if flat:
        buf[:len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(tdev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r in range(world):
        raw = parts[r].cpu().numpy().tobytes()
        out.extend(raw[64 * i:64 * i + 64] for i in range(len(shard(total, r, world))))
    return out


def c5_check(dev, sh, rank):
    """What was timed is right: a unit proved inside a batch == the same unit proved alone (sh_stark_prove from host
    buffers), byte for byte; on rank 0 the host verifier (stark.py:281-388) accepts it."""
    from starks_amd import batch, stark
    from starks_amd.modp import IntegersModP
    if not sh.units:
        return {"batch_equals_single": True, "verifies": None}
    i = len(sh.units) - 1
    got = sh.proof(i)
    w, inp = batch.mimc_stark_unit(sh.units[i], sh.steps)
    wit = b"".join(b"".join(v.to_bytes(32, "big") for v in col) for col in w)
    single = stark.prove_flat(wit, b"".join(v.to_bytes(32, "big") for v in inp), sh.steps, sh.ext, 2, sh.polys)
    res = {"unit": sh.units[i], "batch_equals_single": got == single, "sha256": hashlib.sha256(got).hexdigest(), "verifies": None}
    if sh.units[0] == 0:
        # unit 0 against the proof the coefficient-form oracle wrote for this size (tests/golden/stark_large.json; None when the file
        # has no case of this size)
        try:
            gold = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "stark_large.json")))["cases"]
                    if c["steps"] == sh.steps and c["ext"] == sh.ext and c["unit"] == 0]
        except Exception:
            gold = []
        res["unit0_matches_oracle_fixture"] = (hashlib.sha256(sh.proof(0)).hexdigest() == gold[0]["proof_sha256"]) if gold else None
    if rank == 0:
        S = stark.STARK(IntegersModP(P), sh.steps, sh.ext, 2, sh.polys)
        pr = stark.unpack_proof(got, sh.steps, sh.ext, 2, sh.degree)
        try:  # the verifier asserts, like the reference's (stark.py:281-388): a rejected proof is a failed check, not a crash
            res["verifies"] = bool(S.verify_proof(pr, w, [(0, j, v) for j, v in enumerate(inp)]))
            # and the library's own verifier (sh_stark_verify) on the flat bytes
            res["verifies"] = res["verifies"] and bool(stark.verify_flat(got, b"".join(v.to_bytes(32, "big") for v in inp),
                                                                         b"".join(col[-1].to_bytes(32, "big") for col in w),
                                                                         sh.steps, sh.ext, 2, sh.polys))
        except AssertionError:
            res["verifies"] = False
    return res


# ---------------------------------------------------------------------------------------------------------------------
def extras(dev, quick):
    """Secondary legs, rank 0 at N=1 only (not part of `value`)."""
    L, ctx = dev.L, dev.ctx
    out = {}
    # ---- Merkle commit -----------------------------------------------------------------------------
    logn = 20 if quick else 24
    n = 1 << logn
    dx, dt = dev.alloc(32 * n), dev.alloc(64 * n)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 7), "fill")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkle"), 10)
    # ten commits back to back is long enough for the chip to leave its boost clock (sustained hashing is power-limited: the leaf
    # kernel of launches 1-3 takes 370 us, that of launches 5-9 440-490 us, profiles/r04_merkle_launch_series.txt); `ms` is that
    # sustained figure, `ms_after_idle` one commit after 0.3 s of idling
    time.sleep(0.3)
    ms1 = dev.timed(lambda: dev.ck(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkle"), 1)
    root = ctypes.create_string_buffer(32)
    dev.ck(L.sh_dev_download(ctx, ctypes.c_void_p(dt.value + 32), root, 32), "dl")
    try:  # the root against the tree oracle/oracle.c hashed for the same leaves (tests/golden/merkle_large.json)
        gold = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "merkle_large.json")))["cases"] if c["logn"] == logn and c["seed"] == 7]
    except Exception:
        gold = []
    out["merkelize_2^%d" % logn] = {"ms": round(ms, 4), "leaves_per_s": n / ms * 1e3,
                                    "algorithmic_GBps": 64.0 * n / ms / 1e6, "ms_after_idle": round(ms1, 4), "root": root.raw.hex(),
                                    "matches_fixture": (root.raw.hex() == gold[0]["root"]) if gold else None}
    dev.free(dx)
    dev.free(dt)
    # ---- LDE: 2^16-step trace, 8x extension, 4 columns (stark.py:27-36 + 253-256) ---------------------------
    steps, ext, cols = (1 << 12 if quick else 1 << 16), 8, 4
    n = steps * ext
    dtr, dout = dev.alloc(32 * steps * cols), dev.alloc(32 * n * cols)
    dev.ck(L.sh_dev_fill_seeded(ctx, dtr, steps * cols, 11), "fill")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_lde(ctx, dtr, dout, steps, ext, cols, root_of(n).to_bytes(32, "big")), "lde"), 10)
    out["lde_%dx2^%d_x8" % (cols, steps.bit_length() - 1)] = {"ms": round(ms, 4), "out_elements_per_s": n * cols / ms * 1e3}
    dev.free(dtr)
    dev.free(dout)
    # ---- batches of independent 2^16-step FRI commits (N = 2^19 each) ---------------------------
    steps, ext, bsz = (1 << 12 if quick else 1 << 16), 8, 32
    n = steps * ext
    w = root_of(n).to_bytes(32, "big")
    plen = int(L.sh_fri_proof_len(n, steps, 40))
    # the polynomials as a prover holds them: [batch][steps] coefficients (the 8x zero padding is implicit, sh_dev_fri_prove_coeffs)
    dc, dp = dev.alloc(32 * steps * bsz), dev.alloc(plen * bsz)
    dev.ck(L.sh_dev_fill_seeded(ctx, dc, steps * bsz, 0xC5), "fill")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove_coeffs(ctx, dc, steps, n, w, steps, ext, 40, bsz, dp), "fri"), 3)
    out["fri_commit_batch%d_steps_2^%d" % (bsz, steps.bit_length() - 1)] = {
        "ms_per_batch": round(ms, 4), "proofs_per_s": bsz / ms * 1e3, "ms_per_proof": round(ms / bsz, 5)}
    dev.free(dc)
    dev.free(dp)
    # ---- FRI commit: 2^14-step (config 3), 2^16-step (config 5) and 2^20-step (the metric) traces, 8x extension --------
    # every proof is compared with the bytes oracle/oracle.c:fri_rec wrote for the same input (tests/golden/fri_large.json)
    try:
        fri_gold = {c["logsteps"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "fri_large.json")))["cases"]}
    except Exception:
        fri_gold = {}
    for logsteps in ([14] if quick else [14, 16, 20]):
        steps, ext = 1 << logsteps, 8
        n = steps * ext
        g2 = root_of(n)
        w = g2.to_bytes(32, "big")
        plen = int(L.sh_fri_proof_len(n, steps, 40))
        dc, dp = dev.alloc(32 * n), dev.alloc(plen)
        # synthetic coefficients: the degree < steps polynomial with seeded coefficients (same cost as a trace poly)
        dev.ck(L.sh_dev_fill_seeded(ctx, dc, n, 0xF51), "fill")
        # zero the upper 7/8 so that deg < steps
        z = bytes(32 * (n - steps))
        dev.ck(L.sh_dev_upload(ctx, z, ctypes.c_void_p(dc.value + 32 * steps), len(z)), "upload")
        # the commit from the `steps` coefficients a prover holds (8x low-degree extension + FRI rounds; the padding is implicit) ...
        ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove_coeffs(ctx, dc, steps, n, w, steps, ext, 40, 1, dp), "fri"), 5)
        short = ctypes.create_string_buffer(plen)
        dev.ck(L.sh_dev_download(ctx, dp, short, plen), "dl")
        # ... and from the explicitly zero-padded vector of n coefficients (the dense evaluation): same bytes
        ms_dense = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, 1, dp), "fri"), 5)
        dense = ctypes.create_string_buffer(plen)
        dev.ck(L.sh_dev_download(ctx, dp, dense, plen), "dl")
        sha = hashlib.sha256(short.raw).hexdigest()
        gold = fri_gold.get(logsteps)
        out["fri_commit_steps_2^%d" % logsteps] = {"ms": round(ms, 4), "ms_from_padded_vector": round(ms_dense, 4), "domain": n,
                                                   "proof_bytes": plen, "algorithmic_GBps": 203.0 * n / ms / 1e6,
                                                   "same_bytes_both_ways": short.raw == dense.raw, "proof_sha256": sha,
                                                   "matches_fixture": (sha == gold["proof_sha256"] and gold["seed"] == 0xF51) if gold else None,
                                                   "fixture": "tests/golden/fri_large.json (oracle/oracle.c:fri_rec)" if gold else None}
        dev.free(dc)
        dev.free(dp)
    # ---- the reference's Python call sites, end to end (host conversion + PCIe + GPU): what a user of the drop-in functions sees ----
    out["callsites"] = callsite_times(quick)
    # ---- whole prover: STARK.mk_proof (stark.py:233-279) for the reference's MiMC formulation, width 2 ------------
    # step polynomials [X_1, X_1 + X_2^3] (test_stark.py:265-293); config 5's unit of work is one such proof
    from starks_amd import stark as _stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    from starks_amd.modp import IntegersModP
    X1, X2 = generate_Xi_s(IntegersModP(P), 2)
    coefs, exps, counts, degree = _stark.pack_step_polys([X1, X1 + X2**3], 2)
    for logsteps, bsz in ([(12, 4)] if quick else [(14, 1), (16, 1), (20, 1)]):
        steps, ext, width = 1 << logsteps, 8, 2
        plen = _stark.proof_len(steps, ext, width, degree)
        dw, di, dp = dev.alloc(64 * steps * bsz), dev.alloc(64 * bsz), dev.alloc(plen * bsz)
        dev.ck(L.sh_dev_fill_mimc_units(ctx, dw, di, steps, 0, bsz, 42), "units")
        best = None
        for _ in range(4):  # the prover leaves its witness intact: best of four calls on the same buffers
            dev.sync()
            dev.ck(L.sh_timer_start(ctx), "timer")
            dev.ck(L.sh_dev_stark_prove(ctx, dw, di, steps, ext, width, coefs, exps, counts, 80, bsz, dp), "stark")
            t = ctypes.c_float()
            dev.ck(L.sh_timer_stop(ctx, ctypes.byref(t)), "timer")
            best = t.value if best is None else min(best, t.value)
        dev.ck(L.sh_stark_status(ctx), "stark status")
        first = ctypes.create_string_buffer(plen)  # unit 0's flat proof: against the coefficient-form oracle's where a fixture exists
        dev.ck(L.sh_dev_download(ctx, dp, first, plen), "dl")
        sha = hashlib.sha256(first.raw).hexdigest()
        try:
            gold = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "stark_large.json")))["cases"]
                    if c["steps"] == steps and c["ext"] == ext and c["unit"] == 0]
        except Exception:
            gold = []
        out["stark_prove_batch%d_steps_2^%d" % (bsz, logsteps)] = {
            "ms_per_batch": round(best, 4), "ms_per_proof": round(best / bsz, 5), "proofs_per_s": bsz / best * 1e3,
            "proof_bytes": plen, "m_root": first.raw[:32].hex(), "proof_sha256": sha,
            "matches_fixture": (sha == gold[0]["proof_sha256"]) if gold else None}
        for p_ in (dw, di, dp):
            dev.free(p_)
    return out


def callsite_times(quick):
    """Wall time of the reference's own call sites through starks_amd (fft.py:316-331, merkle_tree.py:36-56, fri.py:189-266), best of
    three.  `wire` = the input is the output of an earlier stage (a wire-backed sequence, starks_amd/wireseq.py): bytes in, bytes
    out.  `list` = the input is a Python list of 2^k ints, as a first call has it: int -> bytes conversion included (0.15-0.3 us per
    value, i.e. most of the time)."""
    from starks_amd import fft, fri, merkle_tree, utils
    from starks_amd.modp import IntegersModP
    F = IntegersModP(P)

    def best(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3, r

    lg = 14 if quick else 20
    n = 1 << lg
    w = root_of(n)
    res = {}
    g = utils.get_power_cycle(F(w), F)  # a wire-backed sequence of n elements
    ms, ev = best(lambda: fft.fft_1d(F, g, P, w))
    res["fft_1d_2^%d_ms" % lg] = round(ms, 3)
    ok = ev[n - 1] == n and ev[0] == 0  # NTT of (1, w, w^2, ...) is n at index n - 1
    ints = g.ints()
    ms, ev2 = best(lambda: fft.fft_1d(F, ints, P, w), 2)
    res["fft_1d_2^%d_ms_from_a_python_list" % lg] = round(ms, 3)
    ok = ok and ev2 == ev
    ms, tree = best(lambda: merkle_tree.merkelize(ev))
    res["merkelize_2^%d_ms" % lg] = round(ms, 3)
    ms, tree2 = best(lambda: merkle_tree.merkelize(ints), 2)
    res["merkelize_2^%d_ms_from_a_python_list" % lg] = round(ms, 3)
    br = merkle_tree.mk_branch(tree, 12345 % n)
    ok = ok and merkle_tree.verify_branch(tree[1], 12345 % n, br) == bytes(ev.wire()[32 * (12345 % n):32 * (12345 % n) + 32])
    ok = ok and tree2[1] == merkle_tree.merkelize(g)[1]
    steps = 1 << (10 if quick else 14)
    g2 = root_of(8 * steps)
    trace = utils.mimc_trace(3, steps)
    poly = fft.NonBinaryFFT(F, F(pow(g2, 8, P))).inv_fft(trace)  # Poly with wire-backed coefficients
    ms, proof = best(lambda: fri.prove_low_degree(poly, F(g2), steps, exclude_multiples_of=8))
    res["prove_low_degree_2^%d_ms" % (steps.bit_length() - 1)] = round(ms, 3)
    ok = ok and len(proof[-1]) == 128
    # STARK.mk_proof / verify_proof (stark.py:233-279, 281-388) as a user of the reference calls them: the witness a list of Python ints
    # per state variable (generated here by the step polynomials, untimed), the proof the reference's nested list
    from starks_amd import stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    X1, X2 = generate_Xi_s(F, 2)
    wit = [[0] * steps, [0] * steps]
    a, b = 42, 3
    for i in range(steps):
        wit[0][i], wit[1][i] = a, b
        a, b = a, (a + b * b * b) % P
    S = stark.STARK(F, steps, 8, 2, [X1, X1 + X2**3])
    boundary = [(0, 0, 42), (0, 1, 3)]
    ms, pr = best(lambda: S.mk_proof(wit, boundary))
    res["mk_proof_2^%d_from_python_lists_ms" % (steps.bit_length() - 1)] = round(ms, 3)
    ms, good = best(lambda: S.verify_proof(pr, wit, boundary))
    res["verify_proof_2^%d_ms" % (steps.bit_length() - 1)] = round(ms, 3)
    ms, good2 = best(lambda: S.verify_proof_native(pr, wit, boundary))
    res["verify_proof_native_2^%d_ms" % (steps.bit_length() - 1)] = round(ms, 3)
    ok = ok and bool(good) and bool(good2) and len(pr) == 4
    res["checks_ok"] = bool(ok)
    res["note"] = ("end to end per call: conversion + PCIe from pageable memory + GPU; *_from_a_python_list = a first call whose input is "
                   "a list of Python ints; the others take the wire-backed output of an earlier stage (starks_amd/wireseq.py)")
    return res


CPU_WHOLE_MAX_LOG = 21  # oracle/oracle.c: a 2^21-point forward + inverse pair takes ~17 s on one core, 2^24 ~4 min


def cpu_baseline(logn, vectors, budget_s=10.0):
    """The C oracle (the reference's algorithm, scalar code) on a bounded sample of the same workload, one child process
    per host core (oracle/cpu_worker.py), at most the cores this box gives us.  Transforms up to 2^21 points run whole,
    one of the step's vectors per core.  A longer transform is sampled through the reference's own recursion
    (fft.py:303-314: a 2^k-point transform = two 2^(k-1)-point transforms of the even / odd inputs + one combining level):
    each core runs one of the 2^(k-21) branches of depth k - 21, i.e. a 2^21-point transform of x[o::2^(k-21)]; the rate
    is scaled by 21/k for the combining levels the sample does not contain."""
    from oracle import coracle
    coracle.build()  # once, before the workers race for it
    n = 1 << logn
    sub = min(logn, CPU_WHOLE_MAX_LOG)
    log_stride = logn - sub
    cores = max(1, min(vectors << log_stride, os.cpu_count() or 1, 16))
    t0 = time.time()
    if log_stride == 0:
        cmds = [[str(sub), str(b), str(budget_s)] for b in range(cores)]
    else:
        cmds = [[str(sub), "0", str(budget_s), str(log_stride), str(o)] for o in range(cores)]
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker"] + c, cwd=ROOT, stdout=subprocess.PIPE) for c in cmds]
    res = []
    for pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("cpu_baseline worker failed")
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    wall = time.time() - t0
    # the workers run concurrently for their whole window
    rate = sum(2 * (1 << sub) * r["reps"] / r["seconds"] for r in res) * sub / logn
    if log_stride == 0:
        sample = ("%d of the step's %d vectors, one per core; %s forward+inverse 2^%d NTTs each with oracle/oracle.c "
                  "(%.1f s per worker, %.1f s wall)" % (cores, vectors, "/".join(str(r["reps"]) for r in res), logn,
                                                        max(r["seconds"] for r in res), wall))
    else:
        sample = ("%d of the %d depth-%d branches of the reference's recursion on the 2^%d-point vector (2^%d-point forward + "
                  "inverse transforms of x[o::%d], o = 0..%d, one per core, oracle/oracle.c; %s pairs, %.1f s per worker, "
                  "%.1f s wall); elements/s of the sample x %d/%d for the %d combining levels it leaves out" %
                  (cores, 1 << log_stride, log_stride, logn, sub, 1 << log_stride, cores - 1,
                   "/".join(str(r["reps"]) for r in res), max(r["seconds"] for r in res), wall, sub, logn, log_stride))
    return {"value": rate, "unit": "elements/s", "cores": cores, "kind": "port", "sample": sample,
            "roundtrip_ok": all(r["roundtrip_ok"] for r in res), "digests": [r["fwd_sha256"] for r in res],
            "sub_logn": sub, "log_stride": log_stride}


def gpu_digests_of_cpu_sample(dev, logn, cpu):
    """The forward transforms the CPU leg ran (cpu_baseline: vector b whole, or the branch x[o::2^log_stride] of vector 0), on the
    GPU through the host-buffer entry point, as SHA-256 of the wire bytes -- what `cpu_baseline.digest_matches_gpu` compares."""
    import numpy as np
    L, ctx = dev.L, dev.ctx
    sub, ls, k = cpu["sub_logn"], cpu["log_stride"], len(cpu["digests"])
    n, m = 1 << logn, 1 << sub
    got = []
    out = ctypes.create_string_buffer(32 * m)
    root = root_of(m).to_bytes(32, "big")
    if ls == 0 and k != 1:
        return None  # (several whole vectors: the seeded fill has no offset argument, only vector 0 is regenerated here)
    dx = dev.alloc(32 * n)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 0x5eed), "fill")
    host = ctypes.create_string_buffer(32 * n)
    dev.ck(L.sh_dev_to_wire(ctx, dx, host, n), "dl")
    dev.free(dx)
    x = np.frombuffer(host, dtype=np.uint8).reshape(m, 1 << ls, 32)  # ls == 0: the vector itself
    for o in range(k):
        branch = np.ascontiguousarray(x[:, o, :]).tobytes()  # x[o::2^ls]
        dev.ck(L.sh_ntt(ctx, branch, m, out, m, root, 0), "ntt")
        got.append(hashlib.sha256(out.raw).hexdigest())
    return got


def cpu_baseline_c5(steps, ext=8, dev=None):
    """Config 5's CPU baseline: the oracle has no whole-prover in C (the STARK oracle is coefficient-form Python, O(n^2));
    what it can time at this size is the FRI commit of one unit's trace polynomial (oracle/oracle.c:fri_rec, the
    reference's algorithm) -- a LOWER bound of one proof's CPU cost, stated as such."""
    from oracle import coracle, pyoracle
    coracle.build()
    g2 = root_of(steps * ext)
    trace = pyoracle.mimc_trace(3, steps)
    wire = b"".join(v.to_bytes(32, "big") for v in trace)
    t0 = time.time()
    coeffs = coracle.fft_bytes(wire, steps, pow(g2, ext, P), inverse=True)
    flat = coracle.fri_prove_flat(coeffs, g2, steps, ext, 40)
    dt = time.time() - t0
    same = None
    if dev is not None:  # the same commit on the GPU (host-buffer entry points): the CPU leg's bytes are the library's bytes
        n = steps * ext
        gc = ctypes.create_string_buffer(32 * steps)
        dev.ck(dev.L.sh_ntt(dev.ctx, wire, steps, gc, steps, pow(g2, ext, P).to_bytes(32, "big"), 1), "intt")
        gp = ctypes.create_string_buffer(len(flat))
        dev.ck(dev.L.sh_fri_prove(dev.ctx, gc.raw, steps, n, g2.to_bytes(32, "big"), steps, ext, 40, 1, gp, len(flat)), "fri")
        same = gc.raw == coeffs and gp.raw == flat
    return {"value": 1.0 / dt, "unit": "proofs/s", "cores": 1, "kind": "port", "proof_matches_gpu": same,
            "proof_sha256": hashlib.sha256(flat).hexdigest(),
            "sample": "one unit: inverse NTT of the trace + FRI commit (N = %d) with oracle/oracle.c, %.1f s; this is only "
                      "the FRI part of a proof (a lower bound of the CPU cost: the reference's quotient construction is "
                      "O(n^2) and not runnable at this size)" % (steps * ext, dt)}


def alu_mix_peak(achieved):
    """Wall-clock integer-VALU issue peak of the NTT pass's instruction mix (tools/alu_mix_bench: independent instructions in the
    pass's opcode proportions, HIP-event timed, 4 and 5 waves per SIMD), measured now on this GPU by the prebuilt binary
    (__graft_entry__.build); falls back to the committed measurement when the binary is missing."""
    exe = os.path.join(ROOT, "tools", "alu_mix_bench")
    res, src = None, None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        src = "tools/alu_mix_bench, run by this bench.py on this GPU"
    except Exception:
        for name in ("r05_alu_mix_peak.json", "r04_alu_mix_peak.json"):
            try:
                res = json.load(open(os.path.join(ROOT, "profiles", name)))
                src = "profiles/%s (tools/alu_mix_bench not runnable here)" % name
                break
            except Exception:
                continue
        if res is None:
            return {}
    if "waves4" not in res:
        return {"peak_error": res.get("error")}
    peak = max(res["waves4"], res["waves5"])
    return {"peak": peak, "peak_waves4": res["waves4"], "peak_waves5": res["waves5"], "frac": achieved / peak, "peak_source": src,
            "peak_mix": res.get("mix"), "peak_method": res.get("method")}


# ---------------------------------------------------------------------------------------------------------------------
_REAL_STDOUT = None


def emit(line):
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """--gpus N > 1 without a launcher: start the N ranks as a child torch.distributed.run (never exec: this process may
    already have touched the GPU, e.g. under rocprofv3) and hand back its exit code."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """Launcher / sharding / gather plumbing WITHOUT the GPU (CPU tests only; `value` is null and the line says so)."""
    import torch
    import torch.distributed as dist
    from starks_amd.batch import shard
    if world > 1:
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=args.dist_timeout))
    mine = shard(args.units, rank, world)
    t0 = time.perf_counter()
    if args.fail_rank == rank:
        # what an invalid witness does to a real rank (STARK.mk_proof / sh_stark_status raise before the header exchange): the
        # rank dies, the launcher (torch.distributed.run) ends the others and returns non-zero -- nobody waits in a collective
        raise RuntimeError("rank %d: injected failure before the header all_gather" % rank)
    local = [hashlib.sha256(b"unit-%d" % j).digest() * 2 for j in mine]  # stand-in headers
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    allh = gather_headers(local, args.units, rank, world, dist, "cpu")
    ok = allh == [hashlib.sha256(b"unit-%d" % j).digest() * 2 for j in range(args.units)]
    # the rank records of a real run (who runs where), without the device properties: the same all_gather_object over the control plane
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "host": socket.gethostname(), "units": len(mine)}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    if rank == 0:
        emit({"metric": "dry-run", "value": None, "unit": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "data": "none (plumbing only, no GPU work)",
                          "config": {"workload": args.workload, "units": args.units}, "gather_ok": ok,
                          "units_per_rank": [len(shard(args.units, r, world)) for r in range(world)],
                          "ranks": {"backend": dist.get_backend() if world > 1 else None,
                                    "world_size": dist.get_world_size() if world > 1 else 1, "ranks": ranks}})
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20 for ntt, 2 for c5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3 for ntt, 1 for c5)")
    ap.add_argument("--workload", choices=["ntt", "c5"], default="ntt")
    ap.add_argument("--logn", type=int, default=24, help="ntt: log2 transform length (configs[3] = 24, configs[1] = 20)")
    ap.add_argument("--batch", type=int, default=1, help="ntt: independent vectors transformed per step (e.g. the columns "
                    "of a trace: one launch sequence covers all of them)")
    ap.add_argument("--units", type=int, default=None, help="c5: proofs in the batch (512; 16 with --quick)")
    ap.add_argument("--logsteps", type=int, default=None, help="c5: log2 trace length (16; 10 with --quick)")
    ap.add_argument("--chunk", type=int, default=256, help="c5: proofs per batched launch (measured: 256 x 2 contexts 5.27-5.29 k proofs/s "
                    "against 128 x 2 5.19-5.24 k, 64 x 2 5.07-5.12 k)")
    ap.add_argument("--c5-streams", type=int, default=2, help="c5: library contexts (streams) the batched launches are dealt to "
                    "in turn (measured: 2 is worth 1-1.5 %% over 1; 3 and 4 no more)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="ntt: skip the many-proof leg")
    ap.add_argument("--no-single", action="store_true", help="ntt: skip the single-vector leg (profiling runs: keeps the "
                    "kernel trace to the timed workload's launches)")
    ap.add_argument("--quick", action="store_true", help="smaller secondary legs / c5 shape")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (profiling runs)")
    ap.add_argument("--no-alu-peak", action="store_true", help="do not start tools/alu_mix_bench (profiling runs: its launches would "
                    "land in the profiled session's kernel trace)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse "
                    "the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--dist-timeout", type=float, default=900.0, help="seconds a collective may wait for the other ranks")
    ap.add_argument("--fail-rank", type=int, default=-1, help="--dry-run only: this rank raises before the header exchange "
                    "(the launcher must exit non-zero, not hang)")
    ap.add_argument("--dry-run", action="store_true", help="CPU tests: launcher, sharding and gather only -- no GPU work, "
                    "value = null")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.workload == "ntt" else 2
    if args.warmup is None:
        args.warmup = 3 if args.workload == "ntt" else 1
    if args.units is None:
        args.units = 16 if args.quick else 512
    if args.logsteps is None:
        args.logsteps = 10 if args.quick else 16

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # before torch / HIP are touched in this process

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout carries exactly ONE line, the JSON of rank 0: whatever libraries print there (RCCL's version banner at
    # communicator creation, ...) goes to stderr instead
    sys.stdout.flush()
    global _REAL_STDOUT
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))

    import torch
    import torch.distributed as dist
    from starks_amd.batch import shard
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    os.environ["STARKHIP_DEVICE"] = str(dev_index)
    torch.cuda.set_device(dev_index)
    tdev = "cuda" if args.backend == "nccl" else "cpu"  # where the tiny control tensors live
    # BENCH_FORCE_DIST=1: run the collectives even with one rank (rehearses the RCCL calls on a one-GPU box)
    use_dist = world > 1 or (os.environ.get("BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index),
                                    timeout=datetime.timedelta(seconds=args.dist_timeout))
        else:
            dist.init_process_group(backend=args.backend, timeout=datetime.timedelta(seconds=args.dist_timeout))

    dev = Dev()
    L, ctx = dev.L, dev.ctx

    def per_rank(x):
        """[x of rank 0, x of rank 1, ...] on every rank (all_gather_object: a few bytes, outside every timed region)."""
        if not use_dist:
            return [x]
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, x)
        return out

    # who runs where -- so that the first multi-GPU run explains itself: backend and world size as torch.distributed reports them,
    # every rank's device index, PCI bus id and UUID.  With RCCL two ranks on one device are a launch error and fail the run
    # (with gloo that is the one-GPU rehearsal of the N > 1 path and is only reported).
    props = torch.cuda.get_device_properties(dev_index)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": props.name,
          "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "uuid": str(getattr(props, "uuid", "")), "host": socket.gethostname(), "visible_devices": ndev}
    ranks = per_rank(me)
    ids = [(r["host"], r["uuid"] or r["pci_bus_id"]) for r in ranks]
    rank_info = {"backend": dist.get_backend() if use_dist else None, "world_size": dist.get_world_size() if use_dist else 1,
                 "world_size_env": world, "devices_distinct": len(set(ids)) == len(ids), "ranks": ranks}
    if use_dist and args.backend == "nccl" and not rank_info["devices_distinct"]:
        raise SystemExit("bench.py: two ranks share a GPU under RCCL (%r): check LOCAL_RANK / HIP_VISIBLE_DEVICES" % (ids,))

    def fence():
        dev.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], device=tdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def run_c5(steps_k, warm_k):
        """K timed steps of the many-proof workload; returns the result dict (rank-independent fields agree on all ranks)."""
        steps = 1 << args.logsteps
        mine = shard(args.units, rank, world)
        # proofs per batched launch: --chunk, but never so many that one of the contexts stays idle (a rank of the 8-GPU run holds 64
        # units: two launches of 32 on two contexts measure 1 % ahead of one launch of 64; 128 units: 64 x 2 against 128 x 1 likewise)
        chunk = min(args.chunk, max(1, (len(mine) + args.c5_streams - 1) // max(1, args.c5_streams)))
        sh = ProofShard(dev, mine, steps, 8, chunk, args.c5_streams)
        heads = None
        for _ in range(warm_k):
            sh.prove_all()
            gather_headers(sh.headers(), args.units, rank, world, dist, tdev, use_dist)
        fence()
        dev.ck(L.sh_timer_start(ctx), "timer")
        t0 = time.perf_counter()
        for _ in range(steps_k):
            sh.prove_all()
            heads = gather_headers(sh.headers(), args.units, rank, world, dist, tdev, use_dist)  # the step's only exchange
        ev = ctypes.c_float()
        dev.ck(L.sh_timer_stop(ctx, ctypes.byref(ev)), "timer")
        fence()
        dt_own = time.perf_counter() - t0
        dt = max_over_ranks(dt_own)
        # the same steps with every proof DELIVERED: the flat proofs of each launch sequence are copied into page-locked host memory
        # on the contexts' copy streams while the next sequence is being proved (STARK.mk_proof returns its proof, stark.py:233-279);
        # a step ends when the last byte has arrived.  Reported beside the on-device figure, never as `value`.
        for _ in range(warm_k):
            sh.prove_all(deliver=True)
            sh.headers()
            sh.delivered()
        fence()
        t1 = time.perf_counter()
        for _ in range(steps_k):
            sh.prove_all(deliver=True)
            gather_headers(sh.headers(), args.units, rank, world, dist, tdev, use_dist)
            host_view = sh.delivered()
        fence()
        dt_deliv = max_over_ranks(time.perf_counter() - t1)
        host_sha = hashlib.sha256(memoryview(host_view)).hexdigest() if host_view is not None else hashlib.sha256(b"").hexdigest()
        chk = c5_check(dev, sh, rank)
        # EVERY proof of the timed run (launches dealt to several contexts = streams running at once) against the same shard proved
        # again through ONE context: all bytes, not a sample (round 4: a race between the waves of a Merkle kernel showed only
        # when a second stream perturbed them, and only in 5-30 % of the proofs)
        chk["all_proofs_sha256"] = sh.digest_all()
        chk["delivered_equals_device"] = host_sha == chk["all_proofs_sha256"]
        if len(sh.ctxs) > 1 and sh.units:
            dealt, sh.ctxs = sh.ctxs, sh.ctxs[:1]
            sh.prove_all()
            sh.headers()
            chk["equals_one_context_run"] = sh.digest_all() == chk["all_proofs_sha256"]
            sh.ctxs = dealt
        else:
            chk["equals_one_context_run"] = True
        ok = all_ok(chk["batch_equals_single"] and chk["verifies"] is not False and chk["equals_one_context_run"] and
                    chk["delivered_equals_device"] and chk.get("unit0_matches_oracle_fixture") is not False and
                    len(set(heads)) == len(heads))
        n = steps * 8
        res = {"units": args.units, "trace_steps": steps, "domain": n, "proofs_per_launch": chunk,
               "proofs_per_s": args.units * steps_k / dt, "ms_per_step": dt / steps_k * 1e3, "ms_per_proof": dt / steps_k / args.units * 1e3,
               "n_gpus": world, "units_per_rank": [len(shard(args.units, r, world)) for r in range(world)],
               "proof_bytes": sh.plen, "headers_sha256": hashlib.sha256(b"".join(heads)).hexdigest(),
               "rank0_event_ms_per_step": ev.value / steps_k,
               "proofs_per_s_delivered": args.units * steps_k / dt_deliv,
               "delivered": {"ms_per_step": dt_deliv / steps_k * 1e3, "bytes_per_step_this_rank": sh.plen * len(mine),
                             "proofs_per_launch": [k for _, k in sh.schedule(True)],
                             "over_on_device": dt_deliv / dt,
                             "what": "every flat proof of the step in page-locked host memory at the end of the step (copies on the contexts' "
                                     "copy streams, under the proving of the next launch sequence); witness generation is outside both "
                                     "timed regions (sh_dev_fill_mimc_units: one sequential x <- x^3 + k chain of 2^16 steps per unit)"},
               "per_rank_ms_per_step": per_rank(dt_own / steps_k * 1e3),
               "check": {"ok": ok, "rank0": chk}}
        sh.close()
        return res

    # =================================================================================================================
    if args.workload == "c5":
        res = run_c5(args.steps, args.warmup)
        steps = 1 << args.logsteps
        n = steps * 8
        # SURVEY 8(d) gives per-unit bytes for the NTT (64 B per element per transform) and the FRI commit (203 B per domain
        # point) only: per proof, two columns x (inverse NTT_s, NTT_N for P, NTT_s for Q) + one FRI commit whose round-0 NTT
        # the prover does not need (l is built on evaluations).  The quotient / packed-leaf / combination kernels have no
        # 8(d) figure and are left out: `achieved` is a lower bound of the bytes the proof moves.
        alg = 2 * 64.0 * (2 * steps + n) + (203.0 - 64.0) * n
        achieved = alg * len(shard(args.units, rank, world)) / (res["rank0_event_ms_per_step"] * 1e-3) / 1e9
        line = {
            "metric": "stark_proofs_per_sec", "value": res["proofs_per_s"], "unit": "proofs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u256 (8 x u32 limbs, integer VALU)", "data": "synthetic",
            "config": {"workload": "configs[4]: batch of %d independent 2^%d-step MiMC STARK proofs (STARK.mk_proof, width 2, "
                                   "8x extension, 80 spot checks, FRI 40 samples), sharded by proof index over %d GPU(s), %d "
                                   "proofs per batched launch; a step proves the whole batch once" %
                                   (args.units, args.logsteps, world, res["proofs_per_launch"]),
                       "units": args.units, "trace_steps": steps, "parallelism": "proof sharding x%d" % world},
            "c5": res, "c5_proofs_per_s": res["proofs_per_s"], "c5_proofs_per_s_delivered": res["proofs_per_s_delivered"],
            "ranks": rank_info,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole proof (NTT passes ~30 %, packed-leaf hashing ~18 %, quotients ~20 %: DESIGN.md section 5)",
                         "algorithmic_bytes_per_proof": alg,
                         "note": "SURVEY 8(d) bytes of the proof's transforms and FRI commit only (the quotient, packed-leaf and "
                                 "combination kernels have no 8(d) figure); integer-VALU / BLAKE2s-ALU bound, not HBM bound"},
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_c5(steps, dev=dev)
        if rank == 0:
            emit(line)
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- workload ntt ---------------------------------------------------------------------------------------------------
    n = 1 << args.logn
    w = root_of(n).to_bytes(32, "big")
    B = max(1, args.batch)
    dx, dy = dev.alloc(32 * n * B), dev.alloc(32 * n * B)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n * B, 0x5eed + rank), "fill")  # independent vectors per rank; vector 0 of
    # rank 0 is the input of the reference-generated fixture

    def step():
        dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, B, w, 0), "ntt")   # y = NTT(x), every vector
        dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, B, w, 1), "intt")  # y = invNTT(y) == x

    for _ in range(args.warmup):
        step()
    fence()
    dev.ck(L.sh_timer_start(ctx), "timer")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctypes.c_float()
    dev.ck(L.sh_timer_stop(ctx, ctypes.byref(ev_ms)), "timer")  # HIP events on the library's stream
    fence()
    dt_own = time.perf_counter() - t0
    dt_max = max_over_ranks(dt_own)
    ms_by_rank = per_rank(dt_own / args.steps * 1e3)

    # correctness of what was timed: x == invNTT(NTT(x)) on every element, and the forward digest of vector 0 against the
    # committed fixture (rank 0): reference-generated up to 2^20 (tests/golden/ntt.json), the pinned C oracle's above
    # (tests/golden/ntt_large.json)
    CH = min(n * B, 1 << 20)
    a, b = ctypes.create_string_buffer(32 * CH), ctypes.create_string_buffer(32 * CH)
    roundtrip_ok = True
    for off in range(0, n * B, CH):
        k = min(CH, n * B - off)  # the last chunk of a batch that is not a multiple of 2^20 elements is shorter
        dev.ck(L.sh_dev_to_wire(ctx, ctypes.c_void_p(dx.value + 32 * off), a, k), "dl")
        dev.ck(L.sh_dev_to_wire(ctx, ctypes.c_void_p(dy.value + 32 * off), b, k), "dl")
        roundtrip_ok = roundtrip_ok and a.raw[:32 * k] == b.raw[:32 * k]
    dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, B, w, 0), "ntt")
    h = hashlib.sha256()
    for off in range(0, n, CH):  # vector 0
        k = min(CH, n - off)
        dev.ck(L.sh_dev_to_wire(ctx, ctypes.c_void_p(dy.value + 32 * off), b, k), "dl")
        h.update(b.raw[:32 * k])
    fwd_digest = h.hexdigest()
    golden_ok, golden_src = None, None
    for gf in ("ntt.json", "ntt_large.json"):
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", gf)))
            hit = [c for c in gold["cases"] if c["n"] == n and c.get("n_in", n) == n]
            if hit and rank == 0:
                golden_ok, golden_src = hit[0]["sha_fwd"] == fwd_digest, "tests/golden/" + gf
        except Exception:
            pass
    ok = all_ok(roundtrip_ok and golden_ok is not False)  # the only exchange: a 1-word status gather over RCCL

    elems_per_step = 2 * n * B
    value = elems_per_step * args.steps * world / dt_max
    alg_bytes = 64.0 * elems_per_step * args.steps  # 64 B per element per transform
    achieved = alg_bytes / (ev_ms.value * 1e-3) / 1e9
    traffic, traffic_src, lane_instr, lane_src, prof_us = None, None, None, None, None
    for tf in ("r05_traffic_2p%d.json" % args.logn, "r04_traffic_2p%d.json" % args.logn, "r03_traffic_2p%d.json" % args.logn,
               "r02_traffic_2p%d.json" % args.logn):
        try:  # HBM bytes and VALU instructions per launch from the committed PMC runs of this same command (tools/prof_bench.sh)
            tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
            if tj.get("logn", 20) == args.logn and tj.get("vectors_per_step", 1) == B:
                traffic = tj["ntt_pass_kernel_mean_hbm_bytes_per_launch"]
                # the launch durations of the profiled session, next to this run's (drift between the two = the counters are stale)
                ks = [k for k in tj.get("kernels", {}).values() if k.get("launches")]
                if ks:
                    prof_us = sum(k["avg_duration_us_kernel_trace"] * k["launches"] for k in ks) / sum(k["launches"] for k in ks)
                traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2*FETCH+WRITE)" % tf
                if "lane_instructions_per_element_per_transform" in tj:
                    lane_instr = tj["lane_instructions_per_element_per_transform"]
                    lane_src = "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU of the same command: sum over the passes x 64 lanes / elements)" % tf
                break
        except Exception:
            pass
    passes = int(L.sh_ntt_passes(n, B))
    cfg = {20: "configs[1]", 24: "configs[3]"}.get(args.logn, "2^%d" % args.logn)
    line = {
        "metric": "ntt_field_elements_per_sec", "value": value, "unit": "elements/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u256 (8 x u32 limbs, integer VALU)",
        "data": "synthetic", "config": {
            "workload": "%s: 2^%d-point NTT + inverse NTT over the MiMC prime, %d vector%s per step per GPU, resident in "
                        "HBM; x == invNTT(NTT(x)) checked on every element and the forward digest against the committed "
                        "fixture" % (cfg, args.logn, B, "" if B == 1 else "s (independent, one launch sequence)"),
            "n": n, "vectors_per_step": B, "elements_per_step": elems_per_step,
            "parallelism": "independent vectors x%d" % world},
        "field_mul_eq_per_s": (n // 2) * args.logn * 2 * B * args.steps * world / dt_max,
        "ranks": rank_info, "per_rank_ms_per_step": ms_by_rank, "per_rank_elements_per_step": [elems_per_step] * world,
        "check": {"roundtrip_ok": ok, "fwd_sha256": fwd_digest, "matches_fixture": golden_ok, "fixture": golden_src},
        # `roofline` is the metric's own quantity -- algorithmic HBM bytes against the 8 TB/s peak (bound / achieved / peak / unit / frac
        # all speak of HBM); what binds this kernel is the integer-VALU issue rate: `binding_resource` and roofline_alu below
        "roofline": {"bound": "hbm", "binding_resource": "integer VALU issue (see roofline_alu)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": 64.0 * n * B / passes,
                     "kernel": "ntt_pass_kernel (%d launches per 2^%d transform; all passes of both directions averaged)" %
                               (passes, args.logn),
                     "avg_launch_us": ev_ms.value * 1e3 / (2 * passes * args.steps),
                     "traffic_profile_avg_launch_us": prof_us,
                     # the same launches by the bytes they really move (PMC traffic per launch / live launch duration): how far
                     # the pass is from the memory roof, as opposed to `frac`, which counts every element once per transform
                     "hbm_traffic_GBps": (traffic / (ev_ms.value * 1e-3 / (2 * passes * args.steps)) / 1e9) if traffic else None,
                     "hbm_traffic_frac": (traffic / (ev_ms.value * 1e-3 / (2 * passes * args.steps)) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "note": "achieved / peak / frac are the algorithmic HBM bytes of SURVEY 8(d) against the HBM peak, as the metric "
                             "asks; the kernel is bound by integer-VALU issue: see roofline_alu (DESIGN.md section 5)"},
    }
    if lane_instr:
        # the roof that binds: VALU lane-instructions the transform executes (committed counters of this command) x elements/s
        # of THIS run, against the wall-clock issue peak of the pass's own instruction mix (tools/alu_mix_bench, run below)
        knobs_set = sorted(k for k in os.environ if k.startswith("STARKHIP_") and k not in ("STARKHIP_DEVICE", "STARKHIP_LIB"))
        line["roofline_alu"] = {"bound": "integer VALU issue", "unit": "T lane-ops/s", "derived": True,
                                "derived_note": "achieved = a COMMITTED instruction count (lane_instructions_source) x this run's elements/s; "
                                                "the count goes stale when kernels or plans change" +
                                                ("; STARKHIP_* knobs are set (%s): the count may not describe this run" % ",".join(knobs_set)
                                                 if knobs_set else ""),
                                "lane_instructions_per_element_per_transform": lane_instr, "lane_instructions_source": lane_src,
                                "achieved": lane_instr * (n * B * 2 * args.steps) / (ev_ms.value * 1e-3) / 1e12,
                                "peak": None, "frac": None}
    if rank == 0 and world == 1 and "roofline_alu" in line and not args.no_alu_peak:
        line["roofline_alu"].update(alu_mix_peak(line["roofline_alu"]["achieved"]))
    if rank == 0 and world == 1 and not args.no_single:
        # the other shapes of the metric: configs[1] literally (ONE 2^20-point pair: a launch's load / store phases are
        # exposed) and 8 independent 2^20-point vectors per launch sequence (the columns of a trace)
        n20 = 1 << (14 if args.quick else 20)
        w20 = root_of(n20).to_bytes(32, "big")
        sx, sy = dev.alloc(32 * n20 * 8), dev.alloc(32 * n20 * 8)
        dev.ck(L.sh_dev_fill_seeded(ctx, sx, n20 * 8, 0x5eed), "fill")
        for vecs, key in ((1, "single_vector"), (8, "ntt_2^%d_x8" % (n20.bit_length() - 1))):
            ms = dev.timed(lambda: (dev.ck(L.sh_dev_ntt(ctx, sx, sy, n20, vecs, w20, 0), "ntt"),
                                    dev.ck(L.sh_dev_ntt(ctx, sy, sy, n20, vecs, w20, 1), "intt")), 50)
            line[key + "_elements_per_s"] = 2 * n20 * vecs / ms * 1e3
            line[key + "_ms_per_fwd_inv"] = round(ms, 5)
            line[key + "_hbm_frac"] = 64.0 * 2 * n20 * vecs / ms / 1e6 / HBM_PEAK_GBS
        s1, s2 = ctypes.create_string_buffer(32 * n20 * 8), ctypes.create_string_buffer(32 * n20 * 8)
        dev.ck(L.sh_dev_to_wire(ctx, sx, s1, n20 * 8), "dl")
        dev.ck(L.sh_dev_to_wire(ctx, sy, s2, n20 * 8), "dl")
        line["check"]["secondary_roundtrip_ok"] = s1.raw == s2.raw
        del s1, s2
        dev.free(sx)
        dev.free(sy)
        if B >= 2 and B % 2 == 0 and n * B <= (1 << 23):
            a = ctypes.create_string_buffer(32 * n * B)
            b = ctypes.create_string_buffer(32 * n * B)
            dev.ck(L.sh_dev_to_wire(ctx, dx, a, n * B), "dl")
            # The same step with its vectors split over TWO contexts (= two streams), free-running: the launches of one
            # stream fill the ramps and tails of the other's.  Reported beside `value`, never as `value`: with two kernels
            # resident at once a per-launch duration no longer maps to a per-launch byte count.
            other = ctypes.c_void_p()
            dev.ck(L.sh_ctx_create(dev_index, ctypes.byref(other)), "sh_ctx_create")
            half = B // 2
            dx2 = ctypes.c_void_p(dx.value + 32 * n * half)
            dy2 = ctypes.c_void_p(dy.value + 32 * n * half)

            def step2():
                dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, half, w, 0), "ntt")
                dev.ck(L.sh_dev_ntt(other, dx2, dy2, n, half, w, 0), "ntt")
                dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, half, w, 1), "intt")
                dev.ck(L.sh_dev_ntt(other, dy2, dy2, n, half, w, 1), "intt")

            def sync2():
                dev.sync()
                dev.ck(L.sh_sync(other), "sh_sync")

            for _ in range(3):
                step2()
            sync2()
            t2 = time.perf_counter()
            for _ in range(30):
                step2()
            sync2()
            ms2 = (time.perf_counter() - t2) * 1e3 / 30
            dev.ck(L.sh_dev_to_wire(ctx, dy, b, n * B), "dl")  # canonical wire form of both; a = the input (above)
            ok2 = a.raw == b.raw
            line["two_stream_elements_per_s"] = 2 * n * B / ms2 * 1e3
            line["two_stream_check"] = {"roundtrip_ok": ok2, "ms_per_step": round(ms2, 5)}
            L.sh_ctx_destroy(other)
    dev.free(dx)
    dev.free(dy)
    if not args.no_c5:
        # three steps of the many-proof workload over the same ranks (after one warm-up step): the proofs/s curve
        res = run_c5(3, 1)
        line["c5"] = res
        line["c5_proofs_per_s"] = res["proofs_per_s"]
        line["c5_proofs_per_s_delivered"] = res["proofs_per_s_delivered"]
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            cb = cpu_baseline(args.logn, B)
            # the transforms the CPU leg timed, run on the GPU: every one of them must give the same bytes
            gd = gpu_digests_of_cpu_sample(dev, args.logn, cb)
            cb["digest_matches_gpu"] = None if gd is None else gd == cb["digests"]
            cb["digests_compared"] = 0 if gd is None else len(gd)
            for k_ in ("digests", "sub_logn", "log_stride"):
                cb.pop(k_)
            line["cpu_baseline"] = cb
        if not args.no_extras:
            line["extra"] = extras(dev, args.quick)
            k20 = "fri_commit_steps_2^20"
            if k20 in line["extra"]:
                line["fri_commit_ms_2^20_trace"] = line["extra"][k20]["ms"]
            line["fri_commit_ms_2^14_trace"] = line["extra"]["fri_commit_steps_2^14"]["ms"]
            for k_, v_ in line["extra"]["callsites"].items():
                if k_.endswith("_ms") :
                    line["callsite_" + k_] = v_
    if rank == 0:
        emit(line)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
