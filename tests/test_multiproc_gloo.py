"""The N > 1 path on CPU: two processes, gloo backend.  The GPU work is replaced by a deterministic stand-in
(the sharding, rendezvous, max-over-ranks timing and the digest all_gather are what is under test)."""
import hashlib
import os
import subprocess
import sys
import textwrap

from conftest import ROOT


def test_shard_partitions():
    from starks_amd.batch import shard
    for total in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            parts = [list(shard(total, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


WORKER = textwrap.dedent("""
    import hashlib, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from starks_amd.batch import shard, gather_digests
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 11
    mine = shard(total, rank, world)
    t0 = time.perf_counter()
    local = [hashlib.sha256(b"proof-%%d" %% j).digest() for j in mine]   # stand-in for the per-unit GPU proof
    dt = torch.tensor([time.perf_counter() - t0 + 0.01 * rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)                            # bench.py's max-over-ranks time
    allp = gather_digests(local, total, rank, world, dist, "cpu")
    want = [hashlib.sha256(b"proof-%%d" %% j).digest() for j in range(total)]
    assert allp == want, (rank, len(allp))
    assert dt.item() >= 0.01 * (world - 1)
    if rank == 0:
        print("GATHER_OK", world, len(allp), hashlib.sha256(b"".join(allp)).hexdigest())
    dist.destroy_process_group()
""")


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29531", str(script)],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    want = hashlib.sha256(b"".join(hashlib.sha256(b"proof-%d" % j).digest() for j in range(11))).hexdigest()
    assert "GATHER_OK 2 11 " + want in out.stdout


def _bench(args, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # ONE JSON line, printed by rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts 2 ranks itself (gloo rendezvous on 127.0.0.1), shards the
    units and gathers the fixed-size headers; --dry-run replaces the GPU work, so `value` is null and labelled."""
    line = _bench(["--gpus", "2", "--workload", "c5", "--units", "11", "--dry-run"])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["units_per_rank"] == [6, 5] and line["gather_ok"] is True
    one = _bench(["--gpus", "1", "--dry-run", "--units", "5"])
    assert one["n_gpus"] == 1 and one["units_per_rank"] == [5]


def test_bench_eight_ranks_dry_run():
    """The shape of the driver's 8-GPU run, on CPU: eight ranks over gloo, config 5's 512 units sharded 64 per rank, the header
    all_gather and the all_gather_object of the rank records (what a real run reports under `ranks`)."""
    line = _bench(["--gpus", "8", "--workload", "c5", "--units", "512", "--dry-run"], timeout=600)
    assert line["n_gpus"] == 8 and line["units_per_rank"] == [64] * 8 and line["gather_ok"] is True
    ri = line["ranks"]
    assert ri["backend"] == "gloo" and ri["world_size"] == 8
    assert [r["rank"] for r in ri["ranks"]] == list(range(8)) and [r["local_rank"] for r in ri["ranks"]] == list(range(8))
    assert [r["units"] for r in ri["ranks"]] == [64] * 8


def test_bench_refuses_to_run_without_gpu():
    """No CPU fallback in the bench either: without --dry-run and without a GPU it fails loudly."""
    import __graft_entry__ as ge
    ge.build()
    from starks_amd import _lib
    if _lib.lib().sh_device_count() > 0:
        import pytest
        pytest.skip("a GPU is present")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--no-cpu-baseline"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode != 0 and "{" not in out.stdout


def test_failing_rank_ends_the_job_instead_of_hanging_it():
    """One rank raises before the header all_gather (what an invalid witness does: mk_proof asserts, stark.py:76): the
    launcher bench.py starts for --gpus 2 must come back non-zero promptly -- the surviving rank is ended by the launcher,
    it does not sit in the barrier until a collective timeout."""
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5", "--units", "11",
                          "--dry-run", "--fail-rank", "1"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode != 0
    assert "injected failure before the header all_gather" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]   # no result line from a failed job
    assert time.time() - t0 < 200
