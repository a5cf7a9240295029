"""Two ranks running the REAL HIP train step (DDPMTrainer) on one shared GPU, collective over gloo: after one step
on the two halves of a batch the replicas hold the parameters a single process gets from the whole batch — the
per-rank mean-loss gradients, one all-reduce(SUM) and the 1/world scale folded into AdamW are exactly the
global-batch step.  On the 8-GPU node the same code runs with backend "nccl" (RCCL over xGMI)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(B):
    g = torch.Generator().manual_seed(11)
    return (torch.rand(B, 1, 28, 28, generator=g) * 2 - 1, torch.randint(0, 1000, (B,), generator=g),
            torch.randn(B, 1, 28, 28, generator=g))


def _worker(rank, world, port, out_dir, B):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from tinydiffusionmodels_amd import dp
    from tinydiffusionmodels_amd.mnist import DDPMTrainer, SimpleUNet
    dp.init_from_env("gloo")                       # both ranks use cuda:0; the gradient all-reduce goes through gloo
    dev = torch.device("cuda:0")
    torch.manual_seed(100 + rank)                  # rank-dependent init: the trainer must broadcast rank 0's weights
    model = SimpleUNet().to(dev)
    tr = DDPMTrainer(model, batch_size=B // world, lr=1e-3)
    x0, t, noise = _batch(B)
    sl = slice(rank * (B // world), (rank + 1) * (B // world))
    p0 = model.flat.detach().clone()
    loss = tr.step(x0[sl].to(dev), t=t[sl].to(dev), noise=noise[sl].to(dev))
    torch.cuda.synchronize()
    torch.save({"p0": p0.cpu(), "p1": model.flat.detach().cpu(), "loss": loss.cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_equals_global_batch_step(tmp_path):
    from tinydiffusionmodels_amd.mnist import DDPMTrainer, SimpleUNet
    B = 16
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), B), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert torch.equal(r0["p0"], r1["p0"])                          # broadcast: identical replicas before the step
    assert torch.equal(r0["p1"], r1["p1"])                          # and after it
    # single process, whole batch, same initial weights
    dev = torch.device("cuda:0")
    model = SimpleUNet().to(dev)
    with torch.no_grad():
        model.flat.copy_(r0["p0"].to(dev))
    tr = DDPMTrainer(model, batch_size=B, lr=1e-3)
    x0, t, noise = _batch(B)
    loss = tr.step(x0.to(dev), t=t.to(dev), noise=noise.to(dev))
    ref = model.flat.detach().cpu()
    # mean of the two half-batch losses = whole-batch loss; parameters agree to an lr-sized fraction (Adam's
    # g / (|g| + eps) amplifies fp32 summation-order differences on near-zero gradients)
    assert abs(0.5 * (r0["loss"].item() + r1["loss"].item()) - loss.item()) < 1e-5 * abs(loss.item())
    step = (ref - r0["p0"]).abs().max().item()
    assert step > 5e-4                                              # the step really moved the weights (lr = 1e-3)
    close = ((r0["p1"] - ref).abs() < 0.05 * 1e-3).float().mean().item()
    assert close > 0.98, close                                      # sign flips of ~zero gradients aside


def _epoch_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from tinydiffusionmodels_amd import dp
    from tinydiffusionmodels_amd import mnist as M
    dp.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.manual_seed(200 + rank)
    model = M.SimpleUNet().to(dev)
    B = 8
    data = M.synthetic_mnist(5 * B * world + 3, seed=9).to(dev)       # five whole global batches and a ragged tail
    # (1) the loop itself: two epochs, replicas identical afterwards
    M.train(model, str(dev), epochs=2, batch_size=B, lr=1e-3, ckpt_path=os.path.join(out_dir, f"ck{rank}.pth"),
            sample_every_epoch=False, data=data, log_every=0)
    torch.cuda.synchronize()
    flat = model.flat.detach().cpu()
    # (2) the positions: after k steps of an epoch this rank's batch is perm[(k * world + rank) * B : ... + B]
    tr = M.DDPMTrainer(model, B, lr=1e-3)
    perm = torch.randperm(data.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    tr.begin_epoch(data, perm)
    tr.steps_epoch(3)
    tr.steps_epoch(1)
    torch.cuda.synchronize()
    st = tr.state
    k = 3
    want = M.q_sample(data[perm[(k * world + rank) * B:(k * world + rank + 1) * B]], st.t, st.noise)
    torch.save({"flat": flat, "shard_ok": bool(torch.equal(st.x_noisy, want))}, os.path.join(out_dir, f"e{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_epoch_mode_shards_inside_the_step(tmp_path):
    """mnist.train() under torch.distributed (2 ranks sharing the GPU, gloo) with the batch gathered INSIDE the captured step:
    every rank's step reads its own shard of the global batch (stride = batch x world, offset = rank x batch, position from the
    device-side step count), the ragged tail goes through step(), and the replicas end bit-identical."""
    mp.spawn(_epoch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "e0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "e1.pt", weights_only=True)
    assert r0["shard_ok"] and r1["shard_ok"]
    assert torch.equal(r0["flat"], r1["flat"]) and torch.isfinite(r0["flat"]).all()


def _text_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from tinydiffusionmodels_amd import dp
    from tinydiffusionmodels_amd import shakespeare as S
    dp.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.manual_seed(7 + rank)                    # rank-dependent init and noise: train() must broadcast and average
    V, D, L = 211, 32, 16
    model = S.TinyTransformer(D, dropout=0.0).to(dev)
    emb, rnd = S.LearnedEmbedding(V, D).to(dev), S.LearnedRounding(D, V).to(dev)
    g = torch.Generator().manual_seed(100 + rank)
    # each rank sees only token ids of its own residue class mod 2 (so the embedding rows a rank touches differ)
    data = [(torch.randint(0, V // 2, (4, L), generator=g) * 2 + rank) for _ in range(3)]
    e0 = emb.embeddings.weight.detach().clone()
    S.train(model, rnd, emb, data, data[:1], dev, ckpt_path=os.path.join(out_dir, f"ckpt{rank}.pth"), epochs=1, lr=1e-3,
            use_lr_scheduling=False)
    torch.cuda.synchronize()
    torch.save({"flat": model.flat.detach().cpu(), "emb": emb.embeddings.weight.detach().cpu(), "emb0": e0.cpu(),
                "rw": rnd.decoder.weight.detach().cpu()},
               os.path.join(out_dir, f"t{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_text_train_keeps_replicas_identical(tmp_path):
    """shakespeare.train() under torch.distributed (2 ranks, real HIP kernels, gloo, one shared GPU): rank 0's weights are
    broadcast, every step averages the gradients — dense for the denoiser / rounding head, row-wise for the embedding table
    (dp.allreduce_rows_) — so both replicas end bit-identical although each saw different tokens and drew different noise,
    and embedding rows that only the OTHER rank's tokens touch have moved on this rank too."""
    mp.spawn(_text_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "t0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "t1.pt", weights_only=True)
    for k in ("flat", "emb", "rw"):
        assert torch.equal(r0[k], r1[k]), k
    moved = (r0["emb"] - r0["emb0"]).abs().amax(dim=1) > 0
    assert moved[0::2].any() and moved[1::2].any()       # even rows: rank 0's tokens; odd rows: rank 1's


def _main_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", TDM_DIST_BACKEND="gloo", TDM_SHARE_GPU="1")
    import io
    import contextlib
    import torch.distributed as dist
    from tinydiffusionmodels_amd import shakespeare as S
    os.chdir(out_dir)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        S.main(["--train", "--byte_tokenizer", "--embed_dim", "32", "--epochs", "2", "--batch_size", "8", "--seq_len", "16",
                "--corpus", os.path.join(out_dir, "corpus.txt"), "--ckpt", os.path.join(out_dir, "ck.pth"), "--seed", "0",
                "--warmup_steps", "2", "--lr", "1e-3"])
    torch.cuda.synchronize()
    with open(os.path.join(out_dir, f"out{rank}.txt"), "w") as f:
        f.write(buf.getvalue())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_text_cli_main(tmp_path):
    """`torchrun -m src.shakespeare --train --byte_tokenizer` at world 2 (src/shakespeare.py:473-600 with the build's DP init):
    main() reads RANK / WORLD_SIZE / LOCAL_RANK, shards the chunk list across the ranks (same permutation, disjoint slices),
    trains with averaged gradients, and only rank 0 prints the epoch lines and writes the checkpoints."""
    (tmp_path / "corpus.txt").write_text("Now is the winter of our discontent made glorious summer by this sun of York.\n" * 40,
                                         encoding="utf-8")
    mp.spawn(_main_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    out0, out1 = (tmp_path / "out0.txt").read_text(), (tmp_path / "out1.txt").read_text()
    assert "Epoch 2/2" in out0 and "rank 0 of 2" in out0
    assert "Epoch" not in out1
    assert (tmp_path / "ck.pth").exists() and (tmp_path / "ck_best.pth").exists()
    ck = torch.load(tmp_path / "ck.pth", map_location="cpu", weights_only=True)
    assert ck["final_training"] and all(torch.isfinite(v).all() for v in ck["diffusion_model"].values())


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO torchrun around it (the form the driver uses at N = 1): the script must start the
    two ranks itself — fresh child processes, before anything initialises HIP in the parent — and rank 0's one JSON line
    must say n_gpus = 2.  Rehearsed on this one-GPU box with both ranks on cuda:0 and the collectives over gloo; on a
    multi-GPU node the same command runs one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TDM_DIST_BACKEND="gloo", TDM_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--sample-steps", "16", "--text-steps", "3"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-3000:]                      # ONE JSON line at any world size
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 1024
    assert out["scaling"] == "weak" and out["value"] > 0
    assert "allreduce" in out and out["allreduce"]["bytes"] == 181_473 * 4
    assert "torch.distributed all_reduce (gloo)" in out["config"]["collective"]
    assert out["sampling"]["sharding"].startswith("chains sharded")
    # the line is the short form (it must survive in the driver's 2,000-character tail); everything measured is in the detail file
    assert len(lines[0]) < 4000 and os.path.exists(out["detail"])
    detail = json.load(open(out["detail"]))
    assert detail["value"] == out["value"] and "steady_state" in detail and "sampling" in detail


def test_bench_refuses_more_gpus_than_visible():
    """--gpus N with fewer than N devices must fail loudly, not print an `n_gpus: 1` line (VERDICT r3 'missing' #1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TDM_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert f"--gpus {n}" in r.stderr and "visible" in r.stderr
