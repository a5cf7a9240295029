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
import sys
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
_native_inits = 0          # communicators this process has bootstrapped: part of the store key (a re-init must not read a stale id)


class CommInitError(RuntimeError):
    """The native (libtdm_hip / RCCL) communicator could not be brought up; the job continues on torch.distributed."""


def _all_ranks_ok(ok: bool) -> bool:
    """MIN over ranks of a local success flag: every rank takes the same branch after every bootstrap stage, so a rank
    that failed locally still joins exactly the collectives the others issue."""
    flag = torch.tensor([1 if ok else 0], device="cuda", dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def native_comm() -> Optional[NativeComm]:
    """The process's native communicator, created on first use when the job runs one rank per GPU
    with the RCCL backend (TDM_COMM=torch keeps the collective in torch.distributed; gloo jobs
    always do).  The unique id travels through torch.distributed's rendezvous store.

    Bootstrap = three stages, each followed by a MIN agreement over the ranks: (1) local: library loads, RCCL binds,
    rank 0 publishes a fresh unique id; (2) tdm_comm_init (collective); (3) a probe all-reduce cross-checked against
    torch.distributed's.  Only comm-init failures (CommInitError, the library's RuntimeError, OSError from dlopen) fall
    back — with the reason logged on rank 0 and reported by collective_name(); anything else propagates."""
    global _native, _native_tried, _native_error, _native_inits
    if _native is not None or _native_tried:
        return _native
    _native_tried = True
    rank, world = world_info()
    if world == 1 or os.environ.get("TDM_COMM", "native") != "native" or dist.get_backend() != "nccl":
        return None
    _native_inits += 1
    key = f"tdm_rccl_unique_id/{_native_inits}"
    comm, err = None, None

    def stage(fn):
        nonlocal err
        if err is None:
            try:
                fn()
            except (CommInitError, RuntimeError, OSError) as e:
                err = f"{type(e).__name__}: {e}"
        if not _all_ranks_ok(err is None) and err is None:
            err = "another rank could not initialise its native communicator"

    def local_checks():
        from . import _lib
        if _lib.lib().tdm_comm_rccl_version() < 0:
            raise CommInitError("librccl could not be loaded by libtdm_hip")
        if rank == 0:
            dist.distributed_c10d._get_default_store().set(key, NativeComm.make_unique_id())

    def init():
        nonlocal comm
        uid = bytes(dist.distributed_c10d._get_default_store().get(key))
        comm = NativeComm(torch.cuda.current_device(), rank, world, uid)

    def probe():
        buf = torch.arange(1024, device="cuda", dtype=torch.float32) * (rank + 1)
        want = buf.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        comm.allreduce_sum_(buf)
        torch.cuda.synchronize()
        if not torch.equal(buf, want):
            raise CommInitError("native all-reduce disagrees with torch.distributed")

    stage(local_checks)
    stage(init)
    stage(probe)
    if err is not None:
        if comm is not None:
            comm.close()
        _native_error, _native = err, None
        if rank == 0:
            print(f"[tdm] native RCCL communicator unavailable ({err}); gradients go through torch.distributed.all_reduce",
                  file=sys.stderr, flush=True)
    else:
        _native = comm
    return _native


_graph_ok: Optional[bool] = None
_graph_why: str = ""


def graph_collective_ok() -> bool:
    """May the train step's all-reduce be CAPTURED into the step's hipGraph (one host call per step at world > 1)?
    Default (TDM_GRAPH_COLLECTIVE unset or 0): NO — the proven three-call form (graph replay, all-reduce, AdamW).  The
    in-graph collective has never run on a multi-GPU node, and its self-check cannot catch a HANG inside a captured or
    replayed RCCL call, which would stall every multi-GPU job at its second step; it is therefore opt-in until one real
    multi-GPU run has recorded the self-check passing.  TDM_GRAPH_COLLECTIVE=1 forces it on; =auto runs the self-check,
    once per process and agreed on by all ranks (MIN): the native RCCL all-reduce of a probe buffer is captured on a side
    graph, replayed twice and compared with torch.distributed's result; anything short of a clean pass on EVERY rank keeps
    the three-call form, with the reason reported by collective_name()."""
    global _graph_ok, _graph_why
    if _graph_ok is not None:
        return _graph_ok
    rank, world = world_info()
    env = os.environ.get("TDM_GRAPH_COLLECTIVE", "0")
    comm = native_comm() if world > 1 else None
    if world == 1 or comm is None:
        _graph_ok, _graph_why = False, "no native communicator"
        return False
    if env in ("0", "1"):
        _graph_ok = env == "1"
        _graph_why = (f"TDM_GRAPH_COLLECTIVE={env}" if "TDM_GRAPH_COLLECTIVE" in os.environ else
                      "default: the in-graph collective is opt-in (TDM_GRAPH_COLLECTIVE=1 or auto) until a multi-GPU run has verified it")
        return _graph_ok
    ok, why = True, "capture / replay / compare self-check passed"
    g = None
    # stage 1: capture.  Agreed on by all ranks BEFORE anyone replays: a rank whose capture failed would leave the others
    # waiting inside the replayed collective.
    try:
        base = torch.arange(4096, device="cuda", dtype=torch.float32) * (rank + 1) + 0.5
        want = base.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        buf = base.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            comm.allreduce_sum_(buf)
    except RuntimeError as e:
        ok, why = False, f"capture failed: {e}"
    if not _all_ranks_ok(ok):
        if ok:
            ok, why = False, "another rank could not capture the all-reduce"
    else:
        # stage 2: every rank replays the same number of times, then compares
        try:
            for _ in range(2):
                buf.copy_(base)
                g.replay()
                torch.cuda.synchronize()
                if not torch.equal(buf, want):
                    ok, why = False, "captured all-reduce disagrees with torch.distributed"
        except RuntimeError as e:
            ok, why = False, f"replay failed: {e}"
        if not _all_ranks_ok(ok) and ok:
            ok, why = False, "another rank's captured all-reduce failed its self-check"
    del g
    _graph_ok, _graph_why = ok, why
    if rank == 0:
        print(f"[tdm] train-step hipGraph {'includes' if ok else 'stops before'} the RCCL all-reduce ({why})", file=sys.stderr, flush=True)
    return ok


def collective_name() -> str:
    _, world = world_info()
    if world == 1:
        return "none (1 rank)"
    if _native is not None:
        return "libtdm_hip tdm_allreduce_sum_f32 (RCCL)" + (f"; in-graph: {_graph_ok} ({_graph_why})" if _graph_ok is not None else "")
    return f"torch.distributed all_reduce ({dist.get_backend()})" + (f"; native comm unavailable: {_native_error}" if _native_error else "")


def shutdown() -> None:
    global _native, _native_tried, _graph_ok
    _graph_ok = None
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
            if _side_stream is not None and not torch.cuda.is_current_stream_capturing():
                # an asynchronous reduction may be in flight on the side stream: one communicator, ONE issue order
                allreduce_grads_async_(flat_grads).wait()
            else:
                comm.allreduce_sum_(flat_grads)
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / world


def _early_native(comm: "NativeComm", flat_grads: torch.Tensor, early_off: int, wait_early) -> None:
    """The native (RCCL behind the C ABI) form of allreduce_grads_early_: both collectives on the collective side stream, the
    first behind the library's early event, the second behind the caller's stream; the caller's stream waits for the side stream."""
    side = _comm_side_stream()
    cur = torch.cuda.current_stream()
    if not (wait_early is not None and wait_early(side)):
        side.wait_stream(cur)
    with torch.cuda.stream(side):
        comm.allreduce_sum_(flat_grads[early_off:])
    side.wait_stream(cur)                           # rb1's part: behind the backward's last launch
    with torch.cuda.stream(side):
        comm.allreduce_sum_(flat_grads[:early_off])
    flat_grads.record_stream(side)
    cur.wait_stream(side)


def allreduce_grads_early_(flat_grads: torch.Tensor, early_off: int, wait_early=None) -> float:
    """allreduce_grads_ as TWO collectives so that the first can start before the backward has ended (DESIGN section 6 "the
    gradient all-reduce under the backward"): floats [early_off:] of the flat gradient — rb2 .. out, final at the library's
    early event (tdm_set_early_grads / tdm_unet_wait_early_grads) — are summed on the collective side stream as soon as that
    event fires, i.e. under rb1's data- and weight-gradient launches; [:early_off] (rb1) follows behind the backward's last
    launch; the caller's stream then waits for both.  `wait_early(stream) -> bool` orders `stream` behind the early event
    and says whether there was one (False: the whole buffer is simply ordered behind the caller's stream, as in
    allreduce_grads_).  Element-wise SUMs over the same ranks: every rank ends with the same bits, and with the bits of the
    one-collective form whenever the reduction order per element does not depend on the message (always at world 2; gloo
    and RCCL rings at larger worlds: replicas identical, last bits may differ from the one-message sum).  Returns 1 / world."""
    rank, world = world_info()
    if world == 1:
        return 1.0
    assert flat_grads.dim() == 1 and 0 < early_off < flat_grads.numel()
    hi, lo = flat_grads[early_off:], flat_grads[:early_off]
    comm = native_comm() if flat_grads.is_cuda else None
    if comm is not None and not torch.cuda.is_current_stream_capturing():
        _early_native(comm, flat_grads, early_off, wait_early)
    elif comm is not None:
        comm.allreduce_sum_(flat_grads)             # (under capture: one collective on the captured stream)
    else:
        dist.all_reduce(hi, op=dist.ReduceOp.SUM)
        dist.all_reduce(lo, op=dist.ReduceOp.SUM)
    return 1.0 / world


def allreduce_grads_parts_(flat_grads: torch.Tensor, parts, wait_part=None) -> float:
    """allreduce_grads_ as one collective per PART, each started as soon as that part of the gradient is final:
    `parts` = [(begin, end, key), ...] in the order the backward finishes them (the denoiser: one per layer, last layer first —
    tdm_tt_layer_grad_range / tdm_tt_wait_layer_grads); `wait_part(stream, key) -> bool` orders `stream` behind part `key`'s event
    and says whether there was one (False: that part simply waits for the caller's stream).  Whatever the parts do not cover (the
    time embedding's few floats) is summed last, behind the caller's stream.  The caller's stream waits for all of it.
    Element-wise SUMs over the same ranks: replicas end bit-identical; against the one-message form the same remark as for
    allreduce_grads_early_ holds.  Every rank must pass the same parts.  Returns 1 / world."""
    rank, world = world_info()
    if world == 1:
        return 1.0
    assert flat_grads.dim() == 1
    n = flat_grads.numel()
    covered = sorted((int(b), int(e)) for b, e, _ in parts)
    rest, pos = [], 0
    for b, e in covered:
        assert pos <= b < e <= n, "parts must be disjoint ranges of the flat gradient"
        if b > pos:
            rest.append((pos, b))
        pos = e
    if pos < n:
        rest.append((pos, n))
    comm = native_comm() if flat_grads.is_cuda else None
    if comm is not None and not torch.cuda.is_current_stream_capturing():
        side = _comm_side_stream()
        cur = torch.cuda.current_stream()
        for b, e, key in parts:
            if not (wait_part is not None and wait_part(side, key)):
                side.wait_stream(cur)
            with torch.cuda.stream(side):
                comm.allreduce_sum_(flat_grads[int(b):int(e)])
        if rest:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for b, e in rest:
                    comm.allreduce_sum_(flat_grads[b:e])
        flat_grads.record_stream(side)
        cur.wait_stream(side)
    elif comm is not None:
        comm.allreduce_sum_(flat_grads)             # (under capture: one collective on the captured stream)
    else:
        for b, e, _ in parts:
            dist.all_reduce(flat_grads[int(b):int(e)], op=dist.ReduceOp.SUM)
        for b, e in rest:
            dist.all_reduce(flat_grads[b:e], op=dist.ReduceOp.SUM)
    return 1.0 / world


class _Pending:
    """A collective in flight next to the caller's stream; wait() orders the caller's CURRENT stream (torch.distributed's
    NCCL work, the native side stream) or the host (gloo) behind it."""

    def __init__(self, work=None, stream=None):
        self._work, self._stream = work, stream

    def wait(self) -> None:
        if self._work is not None:
            self._work.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self._work = self._stream = None


_side_stream = None


def _comm_side_stream() -> "torch.cuda.Stream":
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream()
    return _side_stream


def allreduce_grads_async_(flat_grads: torch.Tensor) -> _Pending:
    """allreduce_grads_ that does NOT hold up the caller's stream: the SUM is enqueued behind everything the current stream
    has been given so far and runs beside what is enqueued afterwards (RCCL moves bytes over xGMI with a few CUs; the
    kernels launched meanwhile keep the rest).  The buffer must not be touched until `.wait()`.  The text train step reduces
    its rounding-head gradient (51 MB at V = 50,257, ready before the denoiser has even started) this way, under the whole
    denoiser forward + backward.  Every rank must issue its collectives in the same order (as with the blocking form)."""
    _, world = world_info()
    if world == 1:
        return _Pending()
    comm = native_comm() if flat_grads.is_cuda else None
    if comm is not None:
        side = _comm_side_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            comm.allreduce_sum_(flat_grads)
        flat_grads.record_stream(side)
        return _Pending(stream=side)
    return _Pending(work=dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=True))


def allreduce_rows_(grad: torch.Tensor, ids: torch.Tensor) -> int:
    """Row-wise SUM over ranks of an embedding gradient grad[V][D] (row N1: nn.Embedding's gradient, src/shakespeare.py:60-86,
    is zero outside the token rows of the batch): only the rows some rank touched travel — the ranks exchange their
    unique token ids (a MAX all-reduce of their count + one all-gather), reduce the union's rows as one dense [U][D] buffer, and scatter it back.
    Exact: a row no rank touched is zero everywhere.  `ids`: this rank's token ids (any shape).  Returns U.
    At V = 50,257, D = 256 the dense gradient is 51 MB per step; a 32 x 128-token batch touches at most 4,096 rows (4 MB)."""
    _, world = world_info()
    V = grad.shape[0]
    if world == 1:
        return int(torch.unique(ids).numel())
    local = torch.unique(ids.reshape(-1).to(grad.device))
    n = torch.tensor([local.numel()], dtype=torch.int64, device=grad.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)                                   # one collective + ONE host read for the pad length
    cap = int(n.item())
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
    """Replicas start from rank `src`'s weights: `tdm_broadcast_f32` (RCCL behind the C ABI) when the native communicator
    is up and the buffer is a contiguous fp32 device tensor, torch.distributed otherwise (gloo rehearsals, CPU tests)."""
    _, world = world_info()
    if world > 1:
        comm = native_comm() if flat_params.is_cuda else None
        if comm is not None and flat_params.dtype == torch.float32 and flat_params.is_contiguous():
            comm.broadcast_(flat_params, src)
        else:
            dist.broadcast(flat_params, src=src)


def shard_chains(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the sampling chains owned by `rank` (no collectives)."""
    per = (n_total + world - 1) // world
    return min(rank * per, n_total), min((rank + 1) * per, n_total)


def sampler_stream_key() -> int:
    """Philox key of this rank's reverse-loop noise streams: a constant mixed with the rank, so identically seeded
    ranks (same torch.manual_seed, hence the same drawn offsets) still sample different chains."""
    rank, _ = world_info()
    return (0x5EED5A3B1E0FD1FF ^ (rank * 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)


def draw_stream_offset() -> int:
    """Base offset of one chain's Philox stream, from torch's CPU generator (so torch.manual_seed reproduces a chain).
    2^40 steps of headroom below 2^62; two chains overlap only if their offsets fall within a chain length of each other."""
    return int(torch.randint(0, 2 ** 62 - 2 ** 40, (1,)).item())


def barrier() -> None:
    _, world = world_info()
    if world > 1:
        dist.barrier()


def allreduce_host_(values: torch.Tensor, op: str = "sum") -> torch.Tensor:
    """In-place reduction over ranks of a small control tensor (validation sums, loader lengths): `torch.distributed`
    on whatever device the tensor lives on — control plane, not the gradient path.  No-op at world 1."""
    _, world = world_info()
    if world > 1:
        dist.all_reduce(values, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
    return values


class ShardedBatches:
    """Per-rank view of one epoch over a (N, ...) tensor of samples, the text loop's counterpart of mnist.train's
    sharding: every rank walks the SAME permutation (seeded by `seed + epoch`, advanced by each __iter__), global
    iteration `it` covers `world * batch_size` consecutive entries of it and rank r takes its consecutive slice
    (shard_batch_indices).  All ranks therefore run the same number of iterations; in the ragged tail a rank's batch may
    be short or EMPTY (shape (0, ...)) and `global_batch(it)` tells the train loop how many samples all ranks hold
    together, so it can weight its gradient by B_local / B_global instead of exchanging counts.  drop_last is never
    applied (the reference's DataLoader has none, src/shakespeare.py:589-590)."""

    def __init__(self, data: torch.Tensor, batch_size: int, rank: int, world: int, shuffle: bool = True, seed: int = 0):
        self.data, self.batch_size, self.rank, self.world, self.shuffle, self.seed = data, batch_size, rank, world, shuffle, seed
        self.epoch = 0
        self.n = int(data.shape[0])

    def __len__(self) -> int:
        per = self.world * self.batch_size
        return (self.n + per - 1) // per

    def global_batch(self, it: int) -> int:
        return global_batch_count(self.n, it, self.batch_size, self.world)

    def __iter__(self):
        if self.shuffle:
            perm = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch))
        else:
            perm = torch.arange(self.n)
        self.epoch += 1
        for it in range(len(self)):
            yield self.data[shard_batch_indices(perm, it, self.batch_size, self.rank, self.world)]
