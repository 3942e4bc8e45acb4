"""N > 1 path on CPU: world_size 2 over gloo (the GPU path uses the same code with backend nccl = RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from densepose_torchscript_amd import parallel
    rank, local_rank, world = parallel.init_distributed(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    # weights: rank 0 owns the values, rank 1 starts from zeros; one coalesced broadcast per dtype
    g = torch.Generator().manual_seed(0)
    shapes = [(128, 64), (7,), (3, 5, 2), (1000,)]
    ref = [torch.randn(s, generator=g) for s in shapes] + [torch.arange(12, dtype=torch.int32).view(3, 4)]
    mine = [t.clone() if rank == 0 else torch.zeros_like(t) for t in ref]
    parallel.broadcast_tensors(mine, src=0, bucket_bytes=4096)
    for a, b in zip(mine, ref):
        assert torch.equal(a, b)
    # frame sharding: contiguous, disjoint, complete
    for n in (1, 2, 7, 64, 129):
        lo, hi = parallel.shard_range(n, rank, world)
        t = torch.zeros(n)
        t[lo:hi] = 1
        dist.all_reduce(t)
        assert torch.equal(t, torch.ones(n)), n
    # results gathered in frame order on rank 0 (variable R per frame)
    lo, hi = parallel.shard_range(5, rank, world)
    local = [{"idx": torch.tensor([i]), "boxes": torch.full((i, 4), float(i))} for i in range(lo, hi)]
    allr = parallel.gather_results(local, dst=0)
    if rank == 0:
        assert [int(r["idx"]) for r in allr] == [0, 1, 2, 3, 4]
        assert [r["boxes"].shape[0] for r in allr] == [0, 1, 2, 3, 4]
    else:
        assert allr is None
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_shard_range_properties():
    from densepose_torchscript_amd.parallel import shard_range
    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_bench_self_launch_spawns_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher starts its own two ranks (RANK / WORLD_SIZE / MASTER_* per child, rank 0
    owns stdout); `--spawn-selftest` makes each rank stop after the first collective, so the path runs without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()          # ONE JSON line, printed by rank 0
    rec = json.loads(lines[0])
    assert rec["world"] == 2 and rec["max_over_ranks"] == 2.0   # MAX over ranks reached rank 0
    # the multi-rank part of the benchmark record (bench.multi_gpu_record over parallel.rank_stats): schema and MIN / MAX semantics
    m = rec["multi_gpu"]
    assert m["ranks_in_process_group"] == 2 and m["world"] == 2 and m["backend"] == "gloo"
    for key in ("sustained_images_per_s", "host_frames_images_per_s", "weight_broadcast_s", "weight_broadcast_bytes"):
        assert set(m[key]) == {"min", "max", "per_rank"} and len(m[key]["per_rank"]) == 2
        assert m[key]["min"] == min(m[key]["per_rank"]) and m[key]["max"] == max(m[key]["per_rank"])
    assert m["sustained_images_per_s"]["per_rank"] == [100.0, 101.0] and m["host_frames_images_per_s"]["per_rank"] == [90.0, 89.0]


def test_bench_self_launch_propagates_rank_failure():
    """Without a GPU every rank of the real benchmark exits with 'needs a GPU'; the parent must return non-zero (and must
    not hang waiting for a rank that lost its peer)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-container check")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode != 0
    assert b"needs a GPU" in p.stderr + p.stdout


def test_launch_local_ranks_stops_peers_when_one_rank_fails(tmp_path):
    from densepose_torchscript_amd.parallel import launch_local_ranks
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    import time
    t0 = time.time()
    rc = launch_local_ranks([str(script)], 3)
    assert rc == 7 and time.time() - t0 < 30
