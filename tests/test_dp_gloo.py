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
