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
    info = m["per_rank_info"]          # who ran where: rank, device name, CPU share, gather workers - one entry per rank, in rank order
    assert [i["rank"] for i in info] == [0, 1] and all(i["device"] == "selftest" and i["host_gather_workers"] >= 1 for i in info)


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


BCAST_WORKER = textwrap.dedent("""
    import hashlib, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state, parallel
    from densepose_torchscript_amd import lib as L
    from densepose_torchscript_amd.pack import PackedModel
    from densepose_torchscript_amd.weights import param_shapes
    cpus = parallel.pin_rank_to_cpus()
    rank, local_rank, world = parallel.init_distributed(backend="gloo")
    cfg = get_config("densepose_rcnn_R_50_FPN_DL_s1x", TINY_OPTS)
    # bench.py's start-up: rank 0 owns the weights, every other rank packs a zeros / unit-variance state of the same shapes
    if rank == 0:
        state = make_synthetic_state(cfg, 3)
    else:
        state = {k: np.zeros(s, dtype=np.float32) for k, s in param_shapes(cfg).items()}
        for k in state:
            if k.endswith("running_var"):
                state[k] += 1.0
    model = PackedModel(cfg, state, L.DP_BF16, "cpu")      # the host-side packer (dp_pack_conv_weights); tensors stay on the CPU for gloo
    tensors = model.parameter_tensors()
    digest = lambda ts: hashlib.sha256(b"".join(t.contiguous().view(torch.uint8).numpy().tobytes() for t in ts)).hexdigest()
    before = digest(tensors)
    parallel.broadcast_tensors(tensors, src=0, bucket_bytes=1 << 20)      # the REAL collective, several buckets per dtype
    after = digest(tensors)
    both = [None, None]
    dist.all_gather_object(both, (rank, before, after, None if cpus is None else len(cpus)))
    assert both[0][2] == both[1][2], "rank 1 does not hold rank 0's packed blob after the broadcast"
    assert both[0][1] == both[0][2], "the source rank's tensors changed"
    assert both[1][1] != both[1][2], "rank 1 started from the same values: the test proves nothing"
    dtypes = sorted({str(t.dtype) for t in tensors})
    assert "torch.bfloat16" in dtypes and "torch.float32" in dtypes and "torch.int32" in dtypes, dtypes
    if cpus is not None:
        assert set(cpus) <= set(os.sched_getaffinity(0)) and len(os.sched_getaffinity(0)) == len(cpus)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok", len(tensors), "tensors", dtypes, "cpus", None if cpus is None else len(cpus))
""")


def test_weight_broadcast_reproduces_rank0_blob_over_gloo(tmp_path):
    """The multi-GPU start-up of bench.py with the REAL collective (gloo here, RCCL on the GPU box): a non-zero rank packs a zero state
    and receives PackedModel.parameter_tensors() - bf16 weight matrices, int32 tap tables, fp32 biases / GroupNorm parameters - through
    parallel.broadcast_tensors in several buckets per dtype; afterwards it holds rank 0's packed blob byte for byte. Each rank pins
    itself to its share of the host CPUs first (parallel.pin_rank_to_cpus: in-process sched_setaffinity)."""
    script = tmp_path / "bcast_worker.py"
    script.write_text(BCAST_WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_rank_cpu_plan():
    """parallel.plan_rank_cpus: disjoint, non-empty shares; the NUMA node of the rank's GPU when sysfs knows it; never outside the
    process's allowed set; a host with fewer CPUs than ranks keeps everything allowed."""
    from densepose_torchscript_amd.parallel import _parse_cpulist, gpu_numa_cpus, host_workers, plan_rank_cpus
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(256))
    shares = [plan_rank_cpus(r, 8, allowed) for r in range(8)]           # no NUMA information: an even contiguous split
    assert all(len(s) == 32 for s in shares) and sorted(c for s in shares for c in s) == allowed
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    shares = [plan_rank_cpus(r, 8, allowed, node0 if r < 4 else node1) for r in range(8)]      # 2 sockets x 4 GPUs
    assert all(len(s) == 32 for s in shares) and sorted(c for s in shares for c in s) == allowed
    assert all(set(shares[r]) <= set(node0 if r < 4 else node1) for r in range(8))
    assert plan_rank_cpus(1, 2, [4, 5, 6, 7], numa_cpus=[0, 1, 2, 3]) in ([4, 5, 6, 7], [6, 7])    # NUMA list outside the cgroup: ignored
    assert plan_rank_cpus(5, 8, [0, 1, 2]) == [0, 1, 2]                  # fewer CPUs than ranks: no pinning below one CPU
    assert gpu_numa_cpus(0, sysfs="/nonexistent") is None
    assert host_workers(4) >= 1
    # the NUMA node of EVERY rank's GPU known: ranks of a node need not be contiguous or evenly spread (5 GPUs on node 0, 3 on node 1, interleaved)
    nodes = [0, 1, 0, 0, 1, 0, 1, 0]
    shares = [plan_rank_cpus(r, 8, allowed, node0 if nodes[r] == 0 else node1, node_of_rank=nodes) for r in range(8)]
    assert all(shares) and len(set(c for sh in shares for c in sh)) == sum(len(sh) for sh in shares)      # disjoint
    assert all(set(sh) <= set(node0 if nodes[r] == 0 else node1) for r, sh in enumerate(shares))
    assert len(shares[0]) == len(node0) // 5 and len(shares[1]) == len(node1) // 3
    # HIP_VISIBLE_DEVICES re-numbers the devices: HIP device 1 of "2,5" is physical GPU 5; ROCR filters first
    from densepose_torchscript_amd.parallel import visible_device_index
    assert visible_device_index(1, {"HIP_VISIBLE_DEVICES": "2,5"}) == 5
    assert visible_device_index(0, {"ROCR_VISIBLE_DEVICES": "4,6", "HIP_VISIBLE_DEVICES": "1"}) == 6
    assert visible_device_index(3, {}) == 3 and visible_device_index(1, {"HIP_VISIBLE_DEVICES": "GPU-abc,GPU-def"}) == 1
