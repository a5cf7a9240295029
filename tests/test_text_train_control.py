"""Host-side training control of the text path (src/shakespeare.py:159-172 schedules, :174-341 train(),
:543-562 checkpoint format sniffing) on CPU.  The PRODUCT's functions are held against the fixtures the reference itself
produced (oracle/make_golden.py); train()'s control flow — cosine warm-up, rounding-weight decay, validation, `_best.pth`,
early stopping, final checkpoint — is driven with a stand-in loss closure (`losses_fn`, the hook tests use the way
mnist.train takes `trainer=`), because the native losses need a GPU; the GPU suite runs the same function with the real
kernels (tests/test_gpu_text.py::test_text_cli_train_then_sample_end_to_end, tests/test_gpu_dp.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_product_schedules_equal_reference_fixtures(golden_dir):
    """get_cosine_schedule_with_warmup(opt, 10, 100) and dynamic_rounding_weight_schedule(e, 20, 0.5) of the PRODUCT module
    against the arrays the reference's own functions produced (src/shakespeare.py:159-172), bit for bit in fp64."""
    from tinydiffusionmodels_amd import shakespeare as S
    g = _load(golden_dir, "text_denoiser.npz")
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([dummy], lr=1.0)
    sch = S.get_cosine_schedule_with_warmup(opt, 10, 100)
    lam = []
    for _ in range(100):
        lam.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert torch.equal(torch.tensor(lam, dtype=torch.float64), g["cosine_warmup_10_100"])
    assert lam[0] == 0.0 and lam[10] == 1.0                         # lr_lambda(0) = 0: the first optimiser step is a no-op
    rw = torch.tensor([S.dynamic_rounding_weight_schedule(e, 20, 0.5) for e in range(20)], dtype=torch.float64)
    assert torch.equal(rw, g["rounding_weight_e20_w0.5"])
    assert abs(S.dynamic_rounding_weight_schedule(20, 20, 0.5) - 0.1) < 1e-15   # decays to the ABSOLUTE final weight 0.1


class _Toy(torch.nn.Module):
    """Stand-in for the three modules train() drives: one weight each, state_dicts of the right kind."""

    def __init__(self, v=0.0):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([v]))


def _make_losses(model, rounding_fn, schedule):
    """losses_fn whose validation loss follows `schedule` (one value per epoch, consumed by the no-grad calls)."""
    state = {"val_calls": 0, "train_calls": 0, "rw": []}

    def losses(token_ids, rw):
        if torch.is_grad_enabled():
            state["train_calls"] += 1
            state["rw"].append(rw)
            diff = (model.w ** 2).sum() + 1.0
            rnd = (rounding_fn.w ** 2).sum() + 2.0
        else:
            diff = torch.tensor(schedule[min(state["val_calls"], len(schedule) - 1)])
            rnd = torch.tensor(0.0)
            state["val_calls"] += 1
        return diff, rnd, diff + rw * rnd
    return losses, state


def test_train_early_stop_best_and_final_checkpoints(tmp_path, capsys, monkeypatch):
    """src/shakespeare.py:305-341: a new best validation loss writes `<ckpt>_best.pth` (epoch, val_loss, three state dicts);
    `patience` epochs without improvement break the loop; the final dict carries `epoch = epochs` and `final_training`."""
    from tinydiffusionmodels_amd import shakespeare as S
    monkeypatch.delenv("AIP_MODEL_DIR", raising=False)
    model, rnd, emb = _Toy(1.0), _Toy(2.0), _Toy(3.0)
    val = [5.0, 4.0, 4.5, 4.2, 4.1, 0.1]           # improves at epochs 0, 1; then three epochs without improvement
    losses, st = _make_losses(model, rnd, val)
    data = [torch.zeros(2, 4, dtype=torch.long)] * 3
    ckpt = str(tmp_path / "t.pth")
    S.train(model, rnd, emb, data, data[:1], "cpu", ckpt_path=ckpt, epochs=10, lr=1e-2, rounding_weight=0.5, patience=3,
            use_lr_scheduling=True, warmup_steps=2, losses_fn=losses, optimizer_cls=torch.optim.AdamW)
    out = capsys.readouterr().out
    assert "Early stopping triggered after 3 epochs without improvement" in out
    assert st["val_calls"] == 5 and st["train_calls"] == 5 * 3                   # stopped after epoch index 4
    assert out.count("New best validation loss!") == 2
    # the rounding weight decays linearly from 0.5 towards the absolute 0.1 over the 10 planned epochs
    assert st["rw"][0] == 0.5 and abs(st["rw"][3] - S.dynamic_rounding_weight_schedule(1, 10, 0.5)) < 1e-15
    best = torch.load(str(tmp_path / "t_best.pth"), weights_only=True)
    assert best["epoch"] == 1 and abs(best["val_loss"] - (4.0 + st["rw"][3] * 0.0)) < 1e-6
    assert set(best) == {"diffusion_model", "rounding_fn", "embedding_fn", "epoch", "val_loss"}
    final = torch.load(ckpt, weights_only=True)
    assert final["epoch"] == 10 and final["final_training"] is True and "embedding_fn" in final
    assert not torch.equal(final["diffusion_model"]["w"], torch.tensor([1.0]))   # the optimiser really stepped
    # first step ran at lr = lr_lambda(0) * lr = 0 (reference quirk): checked through the schedule test


def test_train_without_learned_embeddings_omits_embedding_state(tmp_path, monkeypatch):
    from tinydiffusionmodels_amd import shakespeare as S
    monkeypatch.setenv("AIP_MODEL_DIR", str(tmp_path / "vertex"))
    os.makedirs(tmp_path / "vertex")
    model, rnd = _Toy(1.0), _Toy(2.0)
    losses, _ = _make_losses(model, rnd, [1.0])
    data = [torch.zeros(2, 4, dtype=torch.long)]
    S.train(model, rnd, torch.zeros(8, 4), data, data, "cpu", ckpt_path=str(tmp_path / "c.pth"), epochs=1,
            use_learned_embeddings=False, use_lr_scheduling=False, losses_fn=losses, optimizer_cls=torch.optim.AdamW)
    final = torch.load(str(tmp_path / "vertex" / "text-model.pth"), weights_only=True)      # AIP_MODEL_DIR wins for the final file
    assert "embedding_fn" not in final and final["final_training"] is True
    assert "embedding_fn" not in torch.load(str(tmp_path / "c_best.pth"), weights_only=True)


def test_load_text_checkpoint_new_and_old_formats():
    """src/shakespeare.py:543-562: the dict format restores all three modules; an old-format checkpoint is the denoiser's raw
    state_dict and restores only the denoiser."""
    from tinydiffusionmodels_amd import shakespeare as S
    model, rnd, emb = _Toy(0.0), _Toy(0.0), _Toy(0.0)
    new = {"diffusion_model": {"w": torch.tensor([1.5])}, "rounding_fn": {"w": torch.tensor([2.5])},
           "embedding_fn": {"w": torch.tensor([3.5])}, "epoch": 3, "val_loss": 0.1}
    S.load_text_checkpoint(new, model, rnd, emb)
    assert (model.w.item(), rnd.w.item(), emb.w.item()) == (1.5, 2.5, 3.5)
    model2, rnd2, emb2 = _Toy(0.0), _Toy(9.0), _Toy(9.0)
    S.load_text_checkpoint({"w": torch.tensor([7.0])}, model2, rnd2, emb2)                    # old format: raw state_dict
    assert (model2.w.item(), rnd2.w.item(), emb2.w.item()) == (7.0, 9.0, 9.0)
    model3, rnd3 = _Toy(0.0), _Toy(9.0)
    S.load_text_checkpoint({"diffusion_model": {"w": torch.tensor([4.0])}}, model3, rnd3, None)   # dict without the head
    assert (model3.w.item(), rnd3.w.item()) == (4.0, 9.0)


def test_sharded_batches_cover_every_sample_once_and_agree_on_length():
    from tinydiffusionmodels_amd import dp
    data = torch.arange(23).view(23, 1)
    world, bs = 3, 4
    loaders = [dp.ShardedBatches(data, bs, r, world, shuffle=True, seed=5) for r in range(world)]
    assert {len(l) for l in loaders} == {2}
    for epoch in range(2):
        per_rank = [list(l) for l in loaders]
        seen = torch.cat([b.reshape(-1) for batches in per_rank for b in batches])
        assert sorted(seen.tolist()) == list(range(23))
        for it in range(2):
            assert sum(per_rank[r][it].shape[0] for r in range(world)) == loaders[0].global_batch(it) == (12, 11)[it]
        assert per_rank[2][1].shape == (3, 1) and per_rank[0][1].shape == (4, 1)
    e0 = torch.cat([b.reshape(-1) for b in dp.ShardedBatches(data, bs, 0, 1, seed=5)])
    e1 = torch.cat([b.reshape(-1) for b in dp.ShardedBatches(data, bs, 0, 1, seed=6)])
    assert not torch.equal(e0, e1)
    tail = list(dp.ShardedBatches(torch.arange(4).view(4, 1), 2, 2, 3, shuffle=False))
    assert len(tail) == 1 and tail[0].shape == (0, 1)               # a rank without samples still iterates (and joins)


# ---- world 2 over gloo: replicated control flow (per-rank validation could desynchronise the ranks) -----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from tinydiffusionmodels_amd import dp
    from tinydiffusionmodels_amd import shakespeare as S
    dp.init_from_env("gloo")
    torch.manual_seed(10 + rank)                                     # rank-dependent init: train() must broadcast rank 0's
    V, D = 16, 4
    model, rnd = torch.nn.Linear(D, D), torch.nn.Linear(D, V)
    emb = torch.nn.Module()
    emb.embeddings = torch.nn.Embedding(V, D)
    emb.forward = lambda ids: emb.embeddings(ids)
    # 7 chunks of 3 tokens, 2 per rank per iteration: iterations of 4 and 3 samples (rank 1's last batch is short)
    chunks = torch.randint(0, V, (7, 3), generator=torch.Generator().manual_seed(1))
    train_dl = dp.ShardedBatches(chunks, 2, rank, world, shuffle=True, seed=3)
    # different validation shards per rank: rank 0 alone would see improvement at epoch 1, rank 1 alone would not
    val_dl = dp.ShardedBatches(torch.arange(4).view(4, 1) % V, 2, rank, world, shuffle=False)
    steps = []

    def losses(ids, rw):
        if torch.is_grad_enabled():
            x = emb(ids)
            diff = ((model(x) - 1.0) ** 2).mean()
            r = torch.nn.functional.cross_entropy(rnd(x).reshape(-1, V), ids.reshape(-1))
            return diff, r, diff + rw * r
        e = len(steps)
        table = {0: [3.0, 2.0, 9.0], 1: [3.0, 5.0, 9.0]}[rank]     # per-epoch validation loss of THIS rank's shard
        d = torch.tensor(table[min(e, 2)])
        return d, torch.tensor(0.0), d

    orig_sched = S.dynamic_rounding_weight_schedule

    def sched(epoch, total, w):                                    # (counts epochs for the stand-in validation table)
        if len(steps) < epoch:
            steps.append(epoch)
        return orig_sched(epoch, total, w)
    S.dynamic_rounding_weight_schedule = sched
    ckpt = os.path.join(out_dir, "shared.pth")                       # ONE path for both ranks: only rank 0 may write it
    S.train(model, rnd, emb, train_dl, val_dl, "cpu", ckpt_path=ckpt, epochs=6, lr=1e-2, patience=1, use_lr_scheduling=False,
            losses_fn=losses, optimizer_cls=torch.optim.AdamW)
    torch.save({"m": model.state_dict(), "r": rnd.state_dict(), "e": emb.embeddings.state_dict(), "epochs_run": len(steps) + 1},
               os.path.join(out_dir, f"dp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_text_train_world2_gloo_same_decisions_one_writer(tmp_path):
    """Two ranks, different validation shards, patience 1, one shared checkpoint path: the validation sums are all-reduced,
    so both ranks see (3+3)/2 = 3.0, then (2+5)/2 = 3.5 -> both stop after the second epoch (on its own shard rank 0 would
    have seen an improvement and gone on while rank 1 stopped — the next gradient all-reduce would hang); replicas
    stay bit-identical through a ragged tail iteration; only rank 0 writes `shared.pth` / `shared_best.pth`."""
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "dp0.pt", weights_only=True)
    b = torch.load(tmp_path / "dp1.pt", weights_only=True)
    assert a["epochs_run"] == b["epochs_run"] == 2
    for k in ("m", "r", "e"):
        for name in a[k]:
            assert torch.equal(a[k][name], b[k][name]), (k, name)
    best = torch.load(tmp_path / "shared_best.pth", weights_only=True)
    assert best["epoch"] == 0 and abs(best["val_loss"] - 3.0) < 1e-6
    final = torch.load(tmp_path / "shared.pth", weights_only=True)
    assert final["final_training"] is True
    for name in a["m"]:
        assert torch.equal(final["diffusion_model"][name], a["m"][name])


def _len_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from tinydiffusionmodels_amd import dp
    from tinydiffusionmodels_amd import shakespeare as S
    dp.init_from_env("gloo")
    m, r, e = _Toy(), _Toy(), _Toy()
    data = [torch.zeros(1, 2, dtype=torch.long)] * (3 + rank)         # rank 1's loader is longer
    msg = ""
    try:
        S.train(m, r, e, data, data[:1], "cpu", ckpt_path=os.path.join(out_dir, "x.pth"), losses_fn=lambda i, w: None,
                optimizer_cls=torch.optim.AdamW)
    except RuntimeError as ex:
        msg = str(ex)
    with open(os.path.join(out_dir, f"len{rank}.txt"), "w") as f:
        f.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_text_train_refuses_loaders_of_different_length(tmp_path):
    mp.spawn(_len_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "loaders differ in length" in (tmp_path / f"len{r}.txt").read_text()
