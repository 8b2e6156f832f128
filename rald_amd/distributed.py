"""Multi-GPU host logic.  The only parallelism in the reference is data parallel
(utils/misc.py:167-246, main_generation.py:66-68, :157-159): one process per GPU, samples
independent.  Sampling / evaluation therefore shards the batch across ranks with NO data-path
collective; the only exchanges are the start-up barrier (C2) and the end-of-eval metric sum (C3,
utils/misc.py:45-47: barrier + all_reduce(SUM) of a 2-element float64).

Backend: 'nccl' (= RCCL over xGMI on ROCm) when a GPU is visible, 'gloo' otherwise (CPU tests).
Rendezvous always on 127.0.0.1 unless MASTER_ADDR is set (single node).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank).  No-op for a single process (utils/misc.py:219-246)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=backend)
    dist.barrier()
    return rank, world, local


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `total` independent units: the first total % world ranks get
    one extra.  Empty shards are legal (total < world)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sample_indices(total: int, rank: int, world: int) -> List[int]:
    """DistributedSampler(shuffle=False)-style interleaved assignment used by the reference's eval
    loader (main_generation.py:66-68): rank r takes samples r, r+world, ... (padded by wrap-around
    so every rank sees the same count, as DistributedSampler does)."""
    if total == 0:
        return []
    n_per = -(-total // world)
    return [(rank + i * world) % total for i in range(n_per)]


def reduce_sum_count(total: float, count: float) -> Tuple[float, float]:
    """SmoothedValue.synchronize_between_processes (utils/misc.py:40-50): barrier + all_reduce(SUM)
    of [count, total] as float64."""
    if not is_dist():
        return total, count
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([count, total], dtype=torch.float64, device=dev)
    dist.barrier()
    dist.all_reduce(t)
    return float(t[1]), float(t[0])


def all_reduce_mean(x: float) -> float:
    """utils/misc.py:367-374."""
    if not is_dist():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(x, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return float(t) / dist.get_world_size()


def max_over_ranks(x: float) -> float:
    if not is_dist():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def gather_shards(local: torch.Tensor, counts: Sequence[int]) -> torch.Tensor | None:
    """Collects per-rank result shards (different lengths allowed) on rank 0, in rank order - used
    only to assemble outputs for saving; never on the compute path."""
    if not is_dist():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def all_gather_shards(local: torch.Tensor, counts: Sequence[int], dim: int = 0) -> torch.Tensor:
    """Every rank gets the concatenation (rank order) of the per-rank shards along `dim`; shard lengths may differ
    (padded to the longest for the collective) and may be zero."""
    if not is_dist():
        return local
    world = dist.get_world_size()
    loc = local.movedim(dim, 0).contiguous()
    mx = max(max(counts), 1)
    pad = torch.zeros((mx,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
    pad[: loc.shape[0]] = loc
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).movedim(0, dim)


def decode_queries_sharded(vae, z: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    """Second-level split for evaluation at batch 1: the Q ~ 1.2 M decode queries of ONE sample are
    independent given the latents, so each rank decodes a contiguous slice of them against its own copy of the latent
    context (every rank holds z - sampling is replicated or z is broadcast by the caller; recomputing the 24-layer
    latent stack costs 1.4 ms, less than shipping queries around) and the logits are all-gathered: one collective of
    Q floats per frame.  z [B, M, C], queries [B, Q, 3] (identical on all ranks) -> logits [B, Q, 1] on every rank."""
    world = world_size()
    if world == 1:
        return vae.decode(z, queries)
    rank = dist.get_rank()
    Q = queries.shape[1]
    spans = [shard_bounds(Q, r, world) for r in range(world)]
    lo, hi = spans[rank]
    if hi > lo:
        local = vae.decode(z, queries[:, lo:hi].contiguous())
    else:
        local = torch.zeros(queries.shape[0], 0, 1, dtype=torch.float32, device=queries.device)
    return all_gather_shards(local, [b - a for a, b in spans], dim=1)
