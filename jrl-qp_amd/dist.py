"""Multi-GPU front end of the batched path: shard the batch, one process per GPU (torch.distributed, backend
"nccl" == RCCL on ROCm, or "gloo" in the CPU tests).

The QPs of a batch are independent, so the data path has NO collective: rank r owns the contiguous index range
shard_range(B_total, r, world) and generates / receives only that slice.  Collectives exist only
in the epilogue: a tiny all-reduce of counters and the max-over-ranks timing, and an optional all-gather of the
results when a single consumer needs them (x: 8 n B bytes; ring over xGMI is per-link bound, ~0.4 ms for the
32768-QP result set -- negligible against the solve and overlappable).
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total: int, rank: int, world: int):
    """Contiguous, balanced partition: the first (total % world) ranks get one extra problem."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_process_group(backend: str | None = None, force: bool = False):
    """Join the process group described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.  A lone process (world 1) needs no
    group and gets none unless `force` is set (used to exercise RCCL itself on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def reduce_stats(n_success: int, sum_iters: int, elapsed_s: float, device="cpu"):
    """Epilogue all-reduce: (sum of successes, sum of iterations, MAX elapsed time over ranks)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return n_success, sum_iters, elapsed_s
    s = torch.tensor([float(n_success), float(sum_iters)], dtype=torch.float64, device=device)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s[0].item()), int(s[1].item()), float(t[0].item())


def all_gather_results(x_local, total: int):
    """Optional: assemble the (total, n) solution matrix on every rank from the per-rank shards (ragged shards are
    padded to the largest one for the collective)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return x_local
    world = dist.get_world_size()
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(e - s for s, e in sizes)
    pad = torch.zeros((mx,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]] = x_local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: e - s] for o, (s, e) in zip(out, sizes)], dim=0)


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(device_index: int, sysfs: str = "/sys"):
    """CPUs of the NUMA node GPU `device_index` hangs off (PCI sysfs: <bdf>/numa_node -> node<N>/cpulist), or None when the
    platform does not say (single-node boxes report -1).  Reads torch's device properties only: no kernel is launched."""
    import torch

    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read())
        if node < 0:
            return None
        return _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read()) or None
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def bind_rank_to_gpu_numa(device_index: int, local_world: int = 1, sysfs: str = "/sys"):
    """Pin this rank process to the cores next to its GPU: the launch thread and the pinned mailbox of the host-driven
    path then stay on the socket the GPU's PCIe root is on.  Intersects with the affinity the process already has (a
    container's cpuset wins); with several ranks on one NUMA node the node's cores are dealt out evenly.  Returns a short
    description for the bench line; never raises and never widens the mask."""
    try:
        have = os.sched_getaffinity(0)
    except AttributeError:
        return "no sched_getaffinity on this platform"
    near = gpu_numa_cpus(device_index, sysfs)
    if not near:
        return f"not bound: no NUMA node reported for GPU {device_index} ({len(have)} cpus allowed)"
    cpus = sorted(have & near)
    if not cpus:
        return f"not bound: none of the {len(near)} NUMA-local cpus is in this process's affinity mask"
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return f"not bound: sched_setaffinity failed ({e})"
    return f"bound to {len(cpus)} NUMA-local cpus of GPU {device_index} ({cpus[0]}..{cpus[-1]})"
