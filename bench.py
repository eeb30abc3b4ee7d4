#!/usr/bin/env python3
"""bench.py -- the hot-path benchmark (driver contract: one JSON line on rank 0).

Metric (BASELINE.json): NTT field elements per second.  Workload at every N: BASELINE configs[1] -- a
2^20-point forward NTT followed by the inverse NTT over the MiMC prime, data resident in HBM, one
independent vector per GPU (weak scaling, no data-path collective: SURVEY 8(e)).  A "step" = one forward +
one inverse transform = 2 * 2^20 transformed elements.  `value` = elements of all ranks / max-over-ranks time.

Also on the same line:
  roofline      -- the NTT tile-pass kernel (the dominant kernel): algorithmic bytes (64 B per element per
                   transform) / HIP-event time of the timed region on the library's stream.
  cpu_baseline  -- the C oracle (oracle/oracle.c: the reference's recursive algorithm, one core) on a
                   bounded sample of the same workload, rank 0 at N=1 only.
  extra         -- 2^24-point NTT (config 4), FRI commit of a 2^14-step (config 3) and a 2^20-step MiMC trace,
                   Merkle commit of 2^24 leaves: ms, elements/s, field-mul-equivalents/s, algorithmic GB/s.
"""
import argparse
import ctypes
import hashlib
import json
import os
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


def extras(dev, quick):
    """Secondary legs, rank 0 at N=1 only (not part of `value`)."""
    L, ctx = dev.L, dev.ctx
    out = {}
    # ---- config 4: 2^24-point NTT ------------------------------------------------------------------
    for logn in ([22] if quick else [24]):
        n = 1 << logn
        w = root_of(n).to_bytes(32, "big")
        dx, dy = dev.alloc(32 * n), dev.alloc(32 * n)
        dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 0x5eed), "fill")
        ms = dev.timed(lambda: dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, 1, w, 0), "ntt"), 10)
        # size-independent check: inverse brings the input back (digest of both)
        dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, 1, w, 1), "intt")
        chk = 1 << 16
        a, b = ctypes.create_string_buffer(32 * chk), ctypes.create_string_buffer(32 * chk)
        dev.ck(L.sh_dev_to_wire(ctx, dx, a, chk), "dl")
        dev.ck(L.sh_dev_to_wire(ctx, dy, b, chk), "dl")
        out["ntt_2^%d" % logn] = {
            "ms": round(ms, 4), "elements_per_s": n / ms * 1e3,
            "field_mul_eq_per_s": (n // 2) * logn / ms * 1e3,
            "algorithmic_GBps": 64.0 * n / ms / 1e6, "hbm_frac": 64.0 * n / ms / 1e6 / HBM_PEAK_GBS,
            "roundtrip_ok": a.raw == b.raw}
        dev.free(dx)
        dev.free(dy)
    # ---- Merkle commit -----------------------------------------------------------------------------
    logn = 20 if quick else 24
    n = 1 << logn
    dx, dt = dev.alloc(32 * n), dev.alloc(64 * n)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 7), "fill")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_merkelize(ctx, dx, n, 1, dt), "merkle"), 10)
    out["merkelize_2^%d" % logn] = {"ms": round(ms, 4), "leaves_per_s": n / ms * 1e3,
                                    "algorithmic_GBps": 64.0 * n / ms / 1e6}
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
    # ---- config 5 on one GPU: batches of independent 2^16-step proofs (N = 2^19 each) ---------------------------
    steps, ext, bsz = (1 << 12 if quick else 1 << 16), 8, 32
    n = steps * ext
    w = root_of(n).to_bytes(32, "big")
    plen = int(L.sh_fri_proof_len(n, steps, 40))
    dc, dp = dev.alloc(32 * n * bsz), dev.alloc(plen * bsz)
    dev.ck(L.sh_dev_fill_seeded(ctx, dc, n * bsz, 0xC5), "fill")
    z = bytes(32 * (n - steps))
    for b in range(bsz):
        dev.ck(L.sh_dev_upload(ctx, z, ctypes.c_void_p(dc.value + 32 * (b * n + steps)), len(z)), "upload")
    ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, bsz, dp), "fri"), 3)
    out["fri_commit_batch%d_steps_2^%d" % (bsz, steps.bit_length() - 1)] = {
        "ms_per_batch": round(ms, 4), "proofs_per_s": bsz / ms * 1e3, "ms_per_proof": round(ms / bsz, 5)}
    dev.free(dc)
    dev.free(dp)
    # ---- FRI commit: 2^14-step (config 3) and 2^20-step MiMC trace, 8x extension ---------------------
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
        ms = dev.timed(lambda: dev.ck(L.sh_dev_fri_prove(ctx, dc, n, w, steps, ext, 40, 1, dp), "fri"), 5)
        out["fri_commit_steps_2^%d" % logsteps] = {"ms": round(ms, 4), "domain": n, "proof_bytes": plen,
                                                   "algorithmic_GBps": 203.0 * n / ms / 1e6}
        dev.free(dc)
        dev.free(dp)
    # ---- whole prover: STARK.mk_proof (stark.py:233-279) for the reference's MiMC formulation, width 2 ------------
    # step polynomials [X_1, X_1 + X_2^3] (test_stark.py:265-293); config 5's unit of work is one such proof
    from starks_amd import stark as _stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    from starks_amd.modp import IntegersModP
    X1, X2 = generate_Xi_s(IntegersModP(P), 2)
    coefs, exps, counts, degree = _stark.pack_step_polys([X1, X1 + X2**3], 2)
    for logsteps, bsz in ([(12, 4)] if quick else [(14, 1), (16, 1), (16, 32), (20, 1)]):
        steps, ext, width = 1 << logsteps, 8, 2
        k, x = 42, 3
        col = [x]
        for _ in range(steps - 1):
            x = (x * x * x + k) % P
            col.append(x)
        wit = b"".join(v.to_bytes(32, "big") for v in [k] * steps + col) * bsz
        inp = (k.to_bytes(32, "big") + (3).to_bytes(32, "big")) * bsz
        plen = _stark.proof_len(steps, ext, width, degree)
        dw, di, dp = dev.alloc(len(wit)), dev.alloc(len(inp)), dev.alloc(plen * bsz)
        dev.ck(L.sh_dev_from_wire(ctx, inp, di, width * bsz), "inputs")
        best = None
        for _ in range(4):  # the prover consumes its witness: re-upload (untimed) before every timed call
            dev.ck(L.sh_dev_from_wire(ctx, wit, dw, width * steps * bsz), "witness")
            dev.sync()
            dev.ck(L.sh_timer_start(ctx), "timer")
            dev.ck(L.sh_dev_stark_prove(ctx, dw, di, steps, ext, width, coefs, exps, counts, 80, bsz, dp), "stark")
            t = ctypes.c_float()
            dev.ck(L.sh_timer_stop(ctx, ctypes.byref(t)), "timer")
            best = t.value if best is None else min(best, t.value)
        dev.ck(L.sh_stark_status(ctx), "stark status")
        head = ctypes.create_string_buffer(64)
        dev.ck(L.sh_dev_download(ctx, dp, head, 64), "dl")
        out["stark_prove_batch%d_steps_2^%d" % (bsz, logsteps)] = {
            "ms_per_batch": round(best, 4), "ms_per_proof": round(best / bsz, 5), "proofs_per_s": bsz / best * 1e3,
            "proof_bytes": plen, "m_root": head.raw[:32].hex()}
        dev.free(dw)
        dev.free(di)
        dev.free(dp)
    return out


def cpu_baseline(logn, vectors, budget_s=10.0):
    """The C oracle (the reference's algorithm, scalar code) on the same workload: the step's independent vectors are
    spread over host cores, one child process per vector (oracle/cpu_worker.py), at most the cores this box gives us."""
    import subprocess
    from oracle import coracle
    coracle.build()  # once, before the workers race for it
    n = 1 << logn
    cores = max(1, min(vectors, os.cpu_count() or 1, 16))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", str(logn), str(b), str(budget_s)], cwd=ROOT,
                              stdout=subprocess.PIPE) for b in range(cores)]
    res = []
    for pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("cpu_baseline worker failed")
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    wall = time.time() - t0
    rate = sum(2 * n * r["reps"] / r["seconds"] for r in res)  # the workers run concurrently for their whole window
    return {"value": rate, "unit": "elements/s", "cores": cores, "kind": "port",
            "sample": "%d of the step's %d vectors, one per core; %s forward+inverse 2^%d NTTs each with oracle/oracle.c "
                      "(%.1f s per worker, %.1f s wall)" % (cores, vectors, "/".join(str(r["reps"]) for r in res), logn,
                                                            max(r["seconds"] for r in res), wall),
            "roundtrip_ok": all(r["roundtrip_ok"] for r in res), "digest": res[0]["fwd_sha256"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--logn", type=int, default=20, help="log2 transform length (configs[1] = 20)")
    ap.add_argument("--batch", type=int, default=8, help="independent vectors transformed per step (the columns of a "
                    "trace: one launch sequence covers all of them)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--quick", action="store_true", help="smaller secondary legs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse "
                    "the N > 1 path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    os.environ["STARKHIP_DEVICE"] = str(dev_index)
    torch.cuda.set_device(dev_index)
    tdev = "cuda" if args.backend == "nccl" else "cpu"  # where the tiny control tensors live
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)

    dev = Dev()
    L, ctx = dev.L, dev.ctx
    n = 1 << args.logn
    w = root_of(n).to_bytes(32, "big")
    B = max(1, args.batch)
    dx, dy = dev.alloc(32 * n * B), dev.alloc(32 * n * B)
    dev.ck(L.sh_dev_fill_seeded(ctx, dx, n * B, 0x5eed + rank), "fill")  # independent vectors per rank; vector 0 of
    # rank 0 is the input of the reference-generated fixture

    def step():
        dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, B, w, 0), "ntt")   # y = NTT(x), every vector
        dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, B, w, 1), "intt")  # y = invNTT(y) == x

    def fence():
        dev.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

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
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    # correctness of what was timed: x == invNTT(NTT(x)) and the forward digest against the fixture (rank 0)
    a, b = ctypes.create_string_buffer(32 * n * B), ctypes.create_string_buffer(32 * n * B)
    dev.ck(L.sh_dev_to_wire(ctx, dx, a, n * B), "dl")
    dev.ck(L.sh_dev_to_wire(ctx, dy, b, n * B), "dl")
    roundtrip_ok = a.raw == b.raw
    dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, B, w, 0), "ntt")
    dev.ck(L.sh_dev_to_wire(ctx, dy, b, n), "dl")
    fwd_digest = hashlib.sha256(b.raw[:32 * n]).hexdigest()  # vector 0
    golden_ok = None
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ntt.json")))
        hit = [c for c in gold["cases"] if c["n"] == n and c["n_in"] == n]
        if hit and rank == 0:
            golden_ok = hit[0]["sha_fwd"] == fwd_digest
    except Exception:
        pass
    ok = torch.tensor([1 if roundtrip_ok and golden_ok is not False else 0], device=tdev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # the only exchange: a 1-word status gather over RCCL

    elems_per_step = 2 * n * B
    value = elems_per_step * args.steps * world / dt_max
    alg_bytes = 64.0 * elems_per_step * args.steps  # 64 B per element per transform
    achieved = alg_bytes / (ev_ms.value * 1e-3) / 1e9
    traffic, traffic_src = None, None
    try:  # HBM bytes per launch from the committed PMC run of this same command (tools/prof_traffic.sh)
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if args.logn == 20 and tj.get("vectors_per_step", 1) == B:
            traffic = tj["ntt_pass_kernel_mean_hbm_bytes_per_launch"]
            traffic_src = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2*FETCH+WRITE)"
    except Exception:
        pass
    passes = 1 if args.logn <= 8 else (args.logn + 7) // 8
    line = {
        "metric": "ntt_field_elements_per_sec", "value": value, "unit": "elements/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u256 (8 x u32 limbs, integer VALU)",
        "data": "synthetic", "config": {
            "workload": "configs[1]: 2^%d-point NTT + inverse NTT over the MiMC prime, %d independent vectors per step "
                        "per GPU (the columns of a trace, one launch sequence), x == invNTT(NTT(x)) checked on all; "
                        "the single-vector figure is extra.ntt_2^%d_single_vector" % (args.logn, B, args.logn),
            "n": n, "vectors_per_step": B, "elements_per_step": elems_per_step,
            "parallelism": "independent vectors x%d" % world},
        "field_mul_eq_per_s": (n // 2) * args.logn * 2 * B * args.steps * world / dt_max,
        "check": {"roundtrip_ok": bool(int(ok.item())), "fwd_sha256": fwd_digest, "matches_reference_fixture": golden_ok},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": 64.0 * n * B / passes,
                     "kernel": "ntt_pass_kernel (%d launches per 2^%d transform)" % (passes, args.logn),
                     "avg_launch_us": ev_ms.value * 1e3 / (2 * passes * args.steps),
                     "note": "integer-VALU bound, not HBM bound: ~11 256-bit modmuls + 20 add/sub per element per "
                             "transform at ~95% of the half-rate VALU issue ceiling; see DESIGN.md section 5"},
    }
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.logn, B)
            line["cpu_baseline"]["digest_matches_gpu"] = line["cpu_baseline"].pop("digest") == fwd_digest
        if not args.no_extras:
            line["extra"] = extras(dev, args.quick)
            # the same transform pair on ONE vector (a launch's load / store phases are then exposed: nothing else runs)
            single = dev.timed(lambda: (dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, 1, w, 0), "ntt"),
                                        dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, 1, w, 1), "intt")), 50)
            line["extra"]["ntt_2^%d_single_vector" % args.logn] = {
                "ms_per_fwd_inv": round(single, 5), "elements_per_s": 2 * n / single * 1e3}
    dev.free(dx)
    dev.free(dy)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
