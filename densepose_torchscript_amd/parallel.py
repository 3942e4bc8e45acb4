"""Frame-level data parallelism over the GPUs of one node: one process per GPU, the packed weights are
broadcast once from rank 0 over RCCL/xGMI, then every rank runs the whole path on its own shard of frames - no
collective in the hot loop. (The reference has no distributed code at all; frames are its only independent unit.)
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def visible_device_index(local_rank, env=None):
    """The physical GPU index behind HIP device `local_rank`: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES re-number
    the devices a process sees (the launcher of a shared box hands every job its own subset); sysfs lists the physical ones. Integer lists
    only (UUID forms are not resolved: -> local_rank). Read from the environment without touching the GPU: a rank pins itself BEFORE it
    creates its context."""
    env = os.environ if env is None else env
    idx = local_rank
    # HIP device i = entry i of HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is its alias) among the devices ROCr shows, which are the entries of
    # ROCR_VISIBLE_DEVICES among the physical ones
    for var in ("HIP_VISIBLE_DEVICES" if env.get("HIP_VISIBLE_DEVICES", "").strip() else "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = env.get(var, "").strip()
        if not v:
            continue
        parts = [p.strip() for p in v.split(",") if p.strip() != ""]
        if not all(p.isdigit() for p in parts) or idx >= len(parts):
            return local_rank
        idx = int(parts[idx])
    return idx


def gpu_numa_node(local_rank, sysfs="/sys", env=None):
    """NUMA node of the GPU behind HIP device `local_rank`: the visible_device_index-th AMD render device in PCI-address order
    (/sys/class/drm/card*/device/numa_node). None when sysfs does not say (numa_node = -1 on single-node hosts, containers without the files)."""
    import glob
    cards = []
    for dev in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*/device")):
        try:
            if not os.path.exists(os.path.join(dev, "numa_node")):
                continue
            vendor = open(os.path.join(dev, "vendor")).read().strip() if os.path.exists(os.path.join(dev, "vendor")) else ""
            if vendor not in ("0x1002", ""):       # AMD GPUs only (a BMC's VGA function has a card node too)
                continue
            cards.append((os.path.realpath(dev), dev))
        except OSError:
            continue
    cards.sort()
    phys = visible_device_index(local_rank, env)
    if phys >= len(cards):
        return None
    try:
        node = int(open(os.path.join(cards[phys][1], "numa_node")).read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def gpu_numa_cpus(local_rank, sysfs="/sys", env=None):
    """CPUs of the NUMA node GPU `local_rank` hangs off (/sys/devices/system/node/node<N>/cpulist), None when unknown."""
    node = gpu_numa_node(local_rank, sysfs, env)
    if node is None:
        return None
    try:
        return _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read()) or None
    except (OSError, ValueError):
        return None


def plan_rank_cpus(local_rank, local_world, allowed, numa_cpus=None, node_of_rank=None):
    """The host CPUs rank `local_rank` of `local_world` on this node may run on: its GPU's NUMA node's CPUs (when known) restricted
    to what the process is allowed, split evenly among the ranks that share that node - or, without NUMA information, an even
    contiguous split of the allowed CPUs. node_of_rank: the NUMA node of EVERY local rank's GPU (list, None entries = unknown): the
    ranks that share this rank's node and its position among them are read off that list (ranks of a node need not be contiguous nor
    evenly spread); without it the ranks are assumed to be spread evenly over the nodes. A rank whose own node is unknown while
    others' are known takes an even split of the CPUs of no known node (so that it overlaps nobody). Never empty: a share smaller than
    one CPU falls back to the whole allowed set."""
    allowed = sorted(allowed)
    if node_of_rank is not None and len(node_of_rank) == local_world and any(n is not None for n in node_of_rank):
        mine = node_of_rank[local_rank]
        same = [r for r in range(local_world) if node_of_rank[r] == mine]
        if mine is not None and numa_cpus:
            pool = sorted(set(allowed) & set(numa_cpus)) or allowed
        else:
            pool = allowed          # caller passes the CPUs of no known node when it has them (pin_rank_to_cpus)
            if numa_cpus:
                pool = sorted(set(allowed) & set(numa_cpus)) or allowed
        idx, sharers = same.index(local_rank), len(same)
    else:
        pool = sorted(set(allowed) & set(numa_cpus)) if numa_cpus else allowed
        if not pool:
            pool = allowed
        # ranks that share this pool: without per-rank NUMA knowledge of the others, assume the ranks are spread evenly over the pools
        n_pools = max(1, len(allowed) // max(len(pool), 1))
        sharers = max(1, -(-local_world // n_pools))
        idx = local_rank % sharers if numa_cpus else local_rank
        if not numa_cpus:
            sharers = local_world
    per = len(pool) // sharers
    if per < 1:
        return allowed
    return pool[idx * per:(idx + 1) * per]


def pin_rank_to_cpus(local_rank=None, local_world=None):
    """Pin THIS process (in place: os.sched_setaffinity - no taskset / numactl hop, which on a GPU box would be an exec in front of a
    process that may already hold the GPU) to its share of the host CPUs and size torch's intra-op pool to it. 8 ranks on a
    256-thread host otherwise each start as many OpenMP / gather threads as there are CPUs and migrate freely across sockets; the
    `multi_gpu` record of bench.py would show the slow rank but nothing would prevent it. Returns the CPU list (None = not applied:
    single rank, or a platform without sched_setaffinity)."""
    _, lr, _ = env_rank()
    local_rank = lr if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else local_world
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    nodes = [gpu_numa_node(r) for r in range(local_world)]
    numa = gpu_numa_cpus(local_rank)
    if nodes[local_rank] is None and any(n is not None for n in nodes):
        # this rank's GPU says nothing while others do: stay off the CPUs of the nodes the others pinned themselves to
        taken = set()
        for n in set(x for x in nodes if x is not None):
            try:
                taken |= set(_parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % n).read()))
            except (OSError, ValueError):
                pass
        numa = sorted(set(allowed) - taken) or None
    cpus = plan_rank_cpus(local_rank, local_world, allowed, numa, node_of_rank=nodes)
    os.sched_setaffinity(0, cpus)
    torch.set_num_threads(max(1, min(len(cpus), 16)))
    return cpus


def host_workers(default=4):
    """Worker threads of a rank's host-side frame gather (predictor._HostFrameRing): `default` for a single process; for one rank of several,
    one per eight CPUs of the share the rank is pinned to (pin_rank_to_cpus ran first: sched_getaffinity IS the share) - 4 on the 32-CPU
    share of an 8-rank run on a 256-thread host, never more than `default`, at least one. (Unmeasured on an 8-GPU node: SCALE_rNN.json has
    been `skipped` every round.)"""
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    try:
        n_cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        n_cpus = os.cpu_count() or 1
    if local_world <= 1:
        return max(1, min(default, n_cpus))
    return max(1, min(default, n_cpus // 8 if n_cpus >= 8 else 1))


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items frames for this rank (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_buckets(tensors, bucket_bytes=256 << 20):
    """Fixed-order list of tensors -> list of buckets (lists of tensors of ONE dtype, at most bucket_bytes each; a tensor
    larger than that gets a bucket of its own). Every rank computes the same plan from the same shapes."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    buckets = []
    for group in by_dtype.values():
        bucket, size = [], 0
        for t in group:
            nb = t.numel() * t.element_size()
            if bucket and size + nb > bucket_bytes:
                buckets.append(bucket)
                bucket, size = [], 0
            bucket.append(t)
            size += nb
        if bucket:
            buckets.append(bucket)
    return buckets


def broadcast_tensors(tensors, src=0, bucket_bytes=256 << 20, transport=None):
    """Broadcast a fixed-order list of tensors from `src`, coalesced per dtype into few large messages
    (xGMI links are per-peer: a handful of 100+ MB messages beats hundreds of small ones).
    `transport(flat)` replaces the collective on one flat buffer (tests drive the flatten / unflatten path with it);
    the default is `dist.broadcast(flat, src)` and a no-op outside a multi-rank process group."""
    if transport is None:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        transport = lambda flat: dist.broadcast(flat, src=src)  # noqa: E731
    for bucket in plan_buckets(tensors, bucket_bytes):
        flat = torch.cat([b.reshape(-1) for b in bucket])
        transport(flat)
        off = 0
        for b in bucket:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()


def rank_stats(values, device=None):
    """Per-rank numbers (a dict with the SAME keys on every rank) -> {key: {"min", "max", "per_rank": [...]}} on every rank: one
    all_gather of a small vector. What a multi-GPU record needs beside the MAX-over-ranks step time: a rank that is slow at
    feeding its GPU (host gather, H2D copy - the scaling risk of this path, run.py:42-57) shows up as the MIN."""
    keys = sorted(values)
    vec = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, vec)
    else:
        parts = [vec]
    table = torch.stack([p.cpu() for p in parts])          # [world, keys]
    return {k: {"min": float(table[:, i].min()), "max": float(table[:, i].max()), "per_rank": [float(x) for x in table[:, i]]}
            for i, k in enumerate(keys)}


def gather_results(local_results, dst=0):
    """Optional: collect the per-frame result dicts (variable R) on rank `dst`, in frame order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_results
    world = dist.get_world_size()
    cpu = [{k: v.cpu() for k, v in r.items()} for r in local_results]
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(cpu, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [r for part in out for r in part]


def launch_local_ranks(argv, n_ranks, timeout_s=None):
    """Start `n_ranks` copies of `python argv...` on this node, one per GPU, with the torch.distributed rendezvous
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) - what `torch.distributed.run --nnodes=1
    --nproc-per-node N` would set. MUST be called before the calling process touches the GPU: the children are fresh
    processes (never an exec of an initialised one). Rank 0 inherits stdout, the other ranks' stdout goes to stderr.
    Returns the first non-zero exit code of any rank (0 if all succeeded); when one rank fails the others are stopped."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None if timeout_s is None else time.time() + timeout_s
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:      # a failed rank would leave the others waiting in a collective forever
                    other.terminate()
        if deadline is not None and time.time() > deadline:
            for other in live:
                other.kill()
            return rc or 124
        if live:
            time.sleep(0.05)
    return rc
