"""Data-parallel plumbing for the DDPM train step: one process
per GPU, `torch.distributed` ("nccl" is RCCL on ROCm; "gloo" for CPU tests).
The reference has no distributed code; this layer is new.

Training shards the batch over ranks; each rank computes the mean-loss
gradient of its own shard into ONE flat fp32 buffer; a single all-reduce(SUM)
followed by a 1/world scale (folded into AdamW's grad_scale) yields exactly the
gradient of the mean loss over the global batch.  Sampling shards chains with
no collective at all."""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Initialise the process group from torchrun's env (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_batch_indices(perm: torch.Tensor, it: int, batch_size: int, rank: int, world: int) -> torch.Tensor:
    """Indices of this rank's `batch_size` samples of global iteration `it`
    (global batch = world*batch_size consecutive entries of the epoch
    permutation; ranks take consecutive slices; the tail may be short/empty)."""
    start = (it * world + rank) * batch_size
    return perm[start:start + batch_size]


def allreduce_grads_(flat_grads: torch.Tensor) -> float:
    """In-place SUM over ranks of the flat gradient; returns the scale (1/world)
    the optimiser must apply.  One collective per step: 725,892 B for the UNet."""
    rank, world = world_info()
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / world


def broadcast_params_(flat_params: torch.Tensor, src: int = 0) -> None:
    _, world = world_info()
    if world > 1:
        dist.broadcast(flat_params, src=src)


def shard_chains(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the sampling chains owned by `rank` (no collectives)."""
    per = (n_total + world - 1) // world
    return min(rank * per, n_total), min((rank + 1) * per, n_total)
