"""Data-parallel plumbing for the DDPM train step: one process
per GPU.  The reference has no distributed code; this layer is new.

Training shards the batch over ranks; each rank computes the mean-loss
gradient of its own shard into ONE flat fp32 buffer; a single all-reduce(SUM)
followed by a 1/world scale (folded into AdamW's grad_scale) yields exactly the
gradient of the mean loss over the global batch.  Sampling shards chains with
no collective at all.

The collective itself goes through the C ABI (`tdm_allreduce_sum_f32`: RCCL over
xGMI, enqueued on the compute stream by libtdm_hip.so); `torch.distributed`
is the bootstrap channel (rendezvous, the 128-byte RCCL unique id through its
store, barriers) and the collective backend of the CPU / shared-GPU rehearsals
("gloo")."""
import ctypes
import os
from typing import Optional, Tuple

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


# ---- the native communicator (C ABI: tdm_ctx / tdm_comm_init / tdm_allreduce_sum_f32) ----
class NativeComm:
    """RCCL communicator owned by libtdm_hip.so for this process's GPU."""

    def __init__(self, device_index: int, rank: int, world: int, unique_id: bytes):
        from . import _lib
        L = _lib.lib()
        self._L, self._lib = L, _lib
        self.rank, self.world = rank, world
        self.ctx = ctypes.c_void_p()
        _lib.check(L.tdm_ctx_create(device_index, ctypes.byref(self.ctx)), "ctx_create")
        buf = ctypes.create_string_buffer(unique_id, len(unique_id))
        _lib.check(L.tdm_comm_init(self.ctx, buf, rank, world), "comm_init")

    @staticmethod
    def make_unique_id() -> bytes:
        from . import _lib
        L = _lib.lib()
        n = L.tdm_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        _lib.check(L.tdm_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    def allreduce_sum_(self, t: torch.Tensor) -> None:
        if t.dtype != torch.float32:
            raise RuntimeError("the native collective reduces fp32 buffers")
        self._lib.check(self._L.tdm_allreduce_sum_f32(self.ctx, self._lib.ptr(t), t.numel(), self._lib.stream()), "allreduce")

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> None:
        if t.dtype != torch.float32:
            raise RuntimeError("the native collective broadcasts fp32 buffers")
        self._lib.check(self._L.tdm_broadcast_f32(self.ctx, self._lib.ptr(t), t.numel(), src, self._lib.stream()), "broadcast")

    def close(self) -> None:
        if self.ctx:
            self._L.tdm_ctx_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()


_native: Optional[NativeComm] = None
_native_tried = False
_native_error: Optional[str] = None


def native_comm() -> Optional[NativeComm]:
    """The process's native communicator, created on first use when the job runs one rank per GPU
    with the RCCL backend (TDM_COMM=torch keeps the collective in torch.distributed; gloo jobs
    always do).  The unique id travels through torch.distributed's rendezvous store."""
    global _native, _native_tried, _native_error
    if _native is not None or _native_tried:
        return _native
    _native_tried = True
    rank, world = world_info()
    if world == 1 or os.environ.get("TDM_COMM", "native") != "native" or dist.get_backend() != "nccl":
        return None
    try:
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("tdm_rccl_unique_id", NativeComm.make_unique_id())
        uid = bytes(store.get("tdm_rccl_unique_id"))
        comm = NativeComm(torch.cuda.current_device(), rank, world, uid)
        # one-off cross-check against torch.distributed's own RCCL all-reduce
        probe = torch.arange(1024, device="cuda", dtype=torch.float32) * (rank + 1)
        want = probe.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        comm.allreduce_sum_(probe)
        torch.cuda.synchronize()
        if not torch.equal(probe, want):
            raise RuntimeError("native all-reduce disagrees with torch.distributed")
        _native = comm
    except Exception as e:   # keep training on the torch.distributed collective; bench.py reports which one ran
        _native_error = f"{type(e).__name__}: {e}"
        _native = None
    # every rank must take the same path: agree (MIN) on whether the native communicator is usable
    ok = torch.tensor([1 if _native is not None else 0], device="cuda", dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0 and _native is not None:
        _native_error = "another rank could not initialise its native communicator"
        _native.close()
        _native = None
    return _native


def collective_name() -> str:
    _, world = world_info()
    if world == 1:
        return "none (1 rank)"
    if _native is not None:
        return "libtdm_hip tdm_allreduce_sum_f32 (RCCL)"
    return f"torch.distributed all_reduce ({dist.get_backend()})" + (f"; native comm unavailable: {_native_error}" if _native_error else "")


def shutdown() -> None:
    global _native, _native_tried
    if _native is not None:
        _native.close()
    _native, _native_tried = None, False


def shard_batch_indices(perm: torch.Tensor, it: int, batch_size: int, rank: int, world: int) -> torch.Tensor:
    """Indices of this rank's `batch_size` samples of global iteration `it`
    (global batch = world*batch_size consecutive entries of the epoch
    permutation; ranks take consecutive slices; the tail may be short/empty)."""
    start = (it * world + rank) * batch_size
    return perm[start:start + batch_size]


def global_batch_count(n: int, it: int, batch_size: int, world: int) -> int:
    """Number of samples ALL ranks hold together in global iteration `it` of an epoch over n samples
    (world * batch_size except for the ragged tail) — computed locally, identical on every rank."""
    return max(0, min(world * batch_size, n - it * world * batch_size))


def allreduce_grads_(flat_grads: torch.Tensor) -> float:
    """In-place SUM over ranks of the flat gradient; returns the scale (1/world)
    the optimiser must apply.  One collective per step: 725,892 B for the UNet."""
    rank, world = world_info()
    if world > 1:
        comm = native_comm() if flat_grads.is_cuda else None
        if comm is not None:
            comm.allreduce_sum_(flat_grads)
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / world


def allreduce_rows_(grad: torch.Tensor, ids: torch.Tensor) -> int:
    """Row-wise SUM over ranks of an embedding gradient grad[V][D] (row N1: nn.Embedding's gradient, src/shakespeare.py:60-86,
    is zero outside the token rows of the batch): only the rows some rank touched travel — the ranks exchange their
    unique token ids (two small all-gathers), reduce the union's rows as one dense [U][D] buffer, and scatter it back.
    Exact: a row no rank touched is zero everywhere.  `ids`: this rank's token ids (any shape).  Returns U.
    At V = 50,257, D = 256 the dense gradient is 51 MB per step; a 32 x 128-token batch touches at most 4,096 rows (4 MB)."""
    _, world = world_info()
    V = grad.shape[0]
    if world == 1:
        return int(torch.unique(ids).numel())
    local = torch.unique(ids.reshape(-1).to(grad.device))
    n = torch.tensor([local.numel()], dtype=torch.int64, device=grad.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    padded = torch.full((cap,), V, dtype=torch.int64, device=grad.device)     # V = "no row"
    padded[:local.numel()] = local
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    union = torch.unique(torch.cat(gathered))                                 # sorted: the same order on every rank
    union = union[union < V]
    if union.numel() == 0:
        return 0
    buf = grad.index_select(0, union).contiguous()
    allreduce_grads_(buf.view(-1))
    grad.index_copy_(0, union, buf)
    return int(union.numel())


def broadcast_params_(flat_params: torch.Tensor, src: int = 0) -> None:
    _, world = world_info()
    if world > 1:
        dist.broadcast(flat_params, src=src)


def shard_chains(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the sampling chains owned by `rank` (no collectives)."""
    per = (n_total + world - 1) // world
    return min(rank * per, n_total), min((rank + 1) * per, n_total)
