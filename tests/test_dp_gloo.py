"""World-size-2 data-parallel logic on CPU (gloo): the collective pattern of
the train step — shard the batch, per-rank mean-loss gradient
in one flat buffer, ONE all-reduce(SUM), 1/world scale — reproduces the
global-batch gradient; replicas stay identical; sampling shards need no
collective.  Gradients come from the CPU oracle here (the HIP path needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from tinydiffusionmodels_amd import dp, unet_engine as E
    from oracle import ddpm_oracle as O
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world) and dp.world_info() == (rank, world)
    tabs = O.make_tables()
    # replicas: rank-dependent init, then broadcast from rank 0
    p = O.unet_init_params(seed=rank)
    flat = E.flat_from_state_dict(p)
    dp.broadcast_params_(flat, src=0)
    p = E.state_dict_from_flat(flat)
    # global batch of 8, this rank's shard of 4 via the sharding helper
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(8, 1, 28, 28, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (8,), generator=g)
    noise = torch.randn(8, 1, 28, 28, generator=g)
    perm = torch.arange(8)
    idx = dp.shard_batch_indices(perm, 0, 4, rank, world)
    _, grads = O.unet_loss_and_grads(p, x0[idx], t[idx], noise[idx], tabs)
    flat_g = E.flat_from_state_dict(grads)
    scale = dp.allreduce_grads_(flat_g)
    torch.save({"flat": flat, "grad": flat_g * scale, "idx": idx, "chains": dp.shard_chains(4096 + 3, rank, world)},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_gloo(tmp_path):
    from oracle import ddpm_oracle as O
    from tinydiffusionmodels_amd import unet_engine as E
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                       # identical replicas after broadcast
    assert torch.equal(r0["grad"], r1["grad"])                       # identical averaged gradient
    assert sorted(r0["idx"].tolist() + r1["idx"].tolist()) == list(range(8))   # disjoint cover
    # equals the single-process gradient of the mean loss over the global batch
    p = E.state_dict_from_flat(r0["flat"])
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(8, 1, 28, 28, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (8,), generator=g)
    noise = torch.randn(8, 1, 28, 28, generator=g)
    _, grads = O.unet_loss_and_grads(p, x0, t, noise, O.make_tables())
    ref = E.flat_from_state_dict(grads)
    assert O.rel_err(r0["grad"], ref) < 1e-5
    # sampling shards: contiguous, disjoint, complete, no collective involved
    (a0, b0), (a1, b1) = r0["chains"], r1["chains"]
    assert a0 == 0 and b0 == a1 and b1 == 4096 + 3


def test_shard_helpers_single_process():
    from tinydiffusionmodels_amd import dp
    perm = torch.arange(10)
    got = [dp.shard_batch_indices(perm, it, 2, r, 2).tolist() for it in range(3) for r in range(2)]
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], []]
    assert dp.world_info() == (0, 1)
    assert dp.allreduce_grads_(torch.ones(3)) == 1.0
    assert [dp.shard_chains(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]


# ---- train() over a ragged epoch at world 3: collectives must pair up on every rank ----
def _draws_for(x0):
    """Deterministic per-sample (t, noise) keyed on the sample's content, so any rank / any process derives the same."""
    key = (x0[:, 0, 0, 0].double() * 1e6).long().abs()
    t = key % 1000
    noise = torch.stack([torch.randn(1, 28, 28, generator=torch.Generator().manual_seed(int(k))) for k in key])
    return t, noise


def _make_oracle_stepper():
    from tinydiffusionmodels_amd import dp, unet_engine as E
    from tinydiffusionmodels_amd.mnist import DPStepper
    from oracle import ddpm_oracle as O

    class OracleStepper(DPStepper):
        """DPStepper whose local gradient comes from the CPU oracle: exercises exactly the collective / weighting /
        optimiser ordering DDPMTrainer inherits (src/mnist.py:152-159 under data parallelism)."""

        def __init__(self, flat):
            self.flat = flat
            self.g = torch.zeros_like(flat)
            self.m, self.v, self.nstep = torch.zeros_like(flat), torch.zeros_like(flat), 0
            self.tabs = O.make_tables()
            self.log = []

        def local_loss_and_grad(self, x0, t, noise):
            t, noise = _draws_for(x0)
            loss, grads = O.unet_loss_and_grads(E.state_dict_from_flat(self.flat), x0, t, noise, self.tabs)
            self.g.copy_(E.flat_from_state_dict(grads))
            return loss

        def grad_buffer(self):
            return self.g

        def optimizer_step(self, grad_scale):
            self.nstep += 1
            self.log.append((self.flat.clone(), self.g * grad_scale))
            p, self.m, self.v = O.adamw_step(self.flat, self.g * grad_scale, self.m, self.v, self.nstep)
            self.flat.copy_(p)

    return OracleStepper


def _train_worker(rank, world, port, out_dir, n, batch_size):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from tinydiffusionmodels_amd import dp, unet_engine as E
    from tinydiffusionmodels_amd import mnist as M
    from oracle import ddpm_oracle as O
    dp.init_from_env("gloo")
    flat = E.flat_from_state_dict(O.unet_init_params(seed=0))
    stepper = _make_oracle_stepper()(flat)
    data = torch.rand(n, 1, 28, 28, generator=torch.Generator().manual_seed(5)) * 2 - 1
    dummy = torch.nn.Linear(1, 1)
    M.train(dummy, "cpu", epochs=1, batch_size=batch_size, ckpt_path=os.path.join(out_dir, f"ckpt{rank}.pth"),
            sample_every_epoch=False, data=data, log_every=0, trainer=stepper)
    torch.save({"flat": stepper.flat, "log": stepper.log}, os.path.join(out_dir, f"t{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_train_ragged_epoch_world3_gloo(tmp_path):
    """n = 20 samples, batch 3 per rank, world 3: global iterations of 9, 9 and 2 samples — in the last one rank 0
    holds 2 samples and ranks 1-2 none.  Every rank must issue the same collectives (no hang / no mismatch), the
    replicas must stay identical, and each step's reduced gradient must be the gradient of the mean loss over that
    iteration's GLOBAL sample set (weights B_local / B_global, not a mean of per-rank means)."""
    from oracle import ddpm_oracle as O
    from tinydiffusionmodels_amd import unet_engine as E
    n, bs, world = 20, 3, 3
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path), n, bs), nprocs=world, join=True)
    runs = [torch.load(tmp_path / f"t{r}.pt") for r in range(world)]
    assert all(len(r["log"]) == 3 for r in runs)                               # 3 optimiser steps on every rank
    for r in runs[1:]:
        assert torch.equal(r["flat"], runs[0]["flat"])                        # identical replicas after the epoch
        for (p_a, g_a), (p_b, g_b) in zip(r["log"], runs[0]["log"]):
            assert torch.equal(p_a, p_b) and torch.equal(g_a, g_b)
    data = torch.rand(n, 1, 28, 28, generator=torch.Generator().manual_seed(5)) * 2 - 1
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    tabs = O.make_tables()
    for it, (p_before, g_used) in enumerate(runs[0]["log"]):
        idx = perm[it * bs * world:(it + 1) * bs * world]
        assert idx.numel() == (9, 9, 2)[it]
        x0 = data[idx]
        t, noise = _draws_for(x0)
        _, grads = O.unet_loss_and_grads(E.state_dict_from_flat(p_before), x0, t, noise, tabs)
        assert O.rel_err(g_used, E.flat_from_state_dict(grads)) < 1e-5, it
    assert os.path.exists(tmp_path / "ckpt0.pth") and not os.path.exists(tmp_path / "ckpt1.pth")


def test_global_batch_count():
    from tinydiffusionmodels_amd import dp
    assert [dp.global_batch_count(20, it, 3, 3) for it in range(4)] == [9, 9, 2, 0]
    assert [dp.global_batch_count(60000, it, 128, 8) for it in (0, 57, 58, 59)] == [1024, 1024, 608, 0]


def _rows_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from tinydiffusionmodels_amd import dp
    dp.init_from_env("gloo")
    V, D = 997, 16
    g = torch.Generator().manual_seed(50 + rank)
    ids = torch.randint(0, V, (3 + rank, 7), generator=g)            # ragged: the ranks hold different numbers of tokens
    if rank == 2:
        ids = ids[:0]                                                # a rank with an empty shard still joins the collectives
    grad = torch.zeros(V, D)
    grad.index_add_(0, ids.reshape(-1), torch.randn(ids.numel(), D, generator=g))   # an embedding gradient: token rows only
    dense = grad.clone()
    dist.all_reduce(dense, op=dist.ReduceOp.SUM)
    u = dp.allreduce_rows_(grad, ids)
    torch.save({"rows": grad, "dense": dense, "u": u, "ids": ids}, os.path.join(out_dir, f"rows{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_embedding_gradient_rowwise_allreduce_equals_dense(tmp_path):
    """dp.allreduce_rows_ (the embedding-table gradient of the text train step): exchanging only the
    union of the ranks' token rows gives bit-for-bit the dense all-reduce, with ragged and empty shards (world 3)."""
    mp.spawn(_rows_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    r = [torch.load(tmp_path / f"rows{k}.pt") for k in range(3)]
    union = torch.unique(torch.cat([x["ids"].reshape(-1) for x in r]))
    for x in r:
        assert torch.equal(x["rows"], x["dense"])
        assert torch.equal(x["rows"], r[0]["rows"])
        assert x["u"] == union.numel()


def _async_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from tinydiffusionmodels_amd import dp
    dp.init_from_env("gloo")
    g = torch.Generator().manual_seed(7 + rank)
    big, small, later = torch.randn(50_000, generator=g), torch.randn(33, generator=g), torch.randn(4_000, generator=g)
    want = [x.clone() for x in (big, small, later)]
    for x in want:
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    # the text step's pattern: two reductions started early, other work and a blocking reduction in between, then the waits
    pending = [dp.allreduce_grads_async_(big), dp.allreduce_grads_async_(small)]
    busy = torch.randn(256, 256, generator=g) @ torch.randn(256, 256, generator=g)     # (stands for the denoiser)
    scale = dp.allreduce_grads_(later)
    for h in pending:
        h.wait()
        h.wait()                                                                        # idempotent
    torch.save({"got": [big, small, later], "want": want, "scale": scale, "busy": float(busy.sum())},
               os.path.join(out_dir, f"async{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_async_allreduce_matches_the_blocking_one(tmp_path):
    """dp.allreduce_grads_async_ (the rounding head's gradient travelling under the denoiser in TextTrainStep, world > 1):
    started before other collectives and work, waited for afterwards — the same sums as blocking all-reduces, every rank
    issuing in the same order (world 3, gloo)."""
    mp.spawn(_async_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    r = [torch.load(tmp_path / f"async{k}.pt") for k in range(3)]
    for x in r:
        assert x["scale"] == pytest.approx(1.0 / 3)
        for got, want in zip(x["got"], x["want"]):
            assert torch.equal(got, want)
        for got, ref in zip(x["got"], r[0]["got"]):
            assert torch.equal(got, ref)


def _early_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tinydiffusionmodels_amd import dp
    dp.init_from_env("gloo")
    n, off = 181473, 9760
    g = torch.randn(n, generator=torch.Generator().manual_seed(50 + rank))
    one = g.clone()
    s1 = dp.allreduce_grads_(one)
    two = g.clone()
    s2 = dp.allreduce_grads_early_(two, off, wait_early=None)
    torch.save({"one": one, "two": two, "scales": (s1, s2), "mine": g}, os.path.join(out_dir, f"e{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_early_two_part_allreduce_gloo(tmp_path, world):
    """dp.allreduce_grads_early_ (the gradient all-reduce under the backward; none in the reference):
    two collectives over [early_off:] and [:early_off] give every rank the same bits, the global SUM, and the scale 1 / world —
    bit-identical to the one-collective form at world 2 (a + b has one order), to fp32 summation order at world 3."""
    port = _free_port()
    mp.spawn(_early_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"e{r}.pt") for r in range(world)]
    total = sum(r["mine"].double() for r in rs)
    for r in rs:
        assert r["scales"] == (1.0 / world, 1.0 / world)
        assert torch.equal(r["two"], rs[0]["two"]) and torch.equal(r["one"], rs[0]["one"])       # identical replicas
        assert (r["two"].double() - total).abs().max() < 1e-5
        if world == 2:
            assert torch.equal(r["two"], r["one"])


def _parts_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tinydiffusionmodels_amd import dp
    dp.init_from_env("gloo")
    n = 3 * 1000 + 17                                   # three "layers" + a tail no part covers (the time embedding's floats)
    parts = [(2000, 3000, 2), (1000, 2000, 1), (0, 1000, 0)]     # the order a backward finishes them: last layer first
    g = torch.randn(n, generator=torch.Generator().manual_seed(70 + rank))
    one = g.clone()
    s1 = dp.allreduce_grads_(one)
    many = g.clone()
    seen = []
    s2 = dp.allreduce_grads_parts_(many, parts, wait_part=lambda stream, key: seen.append(key) or False)
    torch.save({"one": one, "many": many, "scales": (s1, s2), "mine": g, "seen": seen}, os.path.join(out_dir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_per_layer_allreduce_parts_gloo(tmp_path, world):
    """dp.allreduce_grads_parts_ (the denoiser's gradient all-reduced layer by layer under its backward; none in the reference): one
    collective per part in the given order plus one for what the parts leave out — identical replicas, the global SUM, scale
    1 / world, bit-identical to the one-collective form at world 2."""
    mp.spawn(_parts_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"p{r}.pt") for r in range(world)]
    total = sum(r["mine"].double() for r in rs)
    for r in rs:
        assert r["scales"] == (1.0 / world, 1.0 / world) and r["seen"] == []       # (CPU tensors: no events to wait for)
        assert torch.equal(r["many"], rs[0]["many"])
        assert (r["many"].double() - total).abs().max() < 1e-5
        if world == 2:
            assert torch.equal(r["many"], r["one"])
