#!/usr/bin/env python3
"""Headline benchmark: DDPM train steps/s on the MNIST UNet at batch 512 per GPU
(BASELINE.json configs[1]; data-parallel over N GPUs = configs[2]), plus the
1000-step reverse-sampling rate at batch 4096 (configs[3]) as extra keys.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is the loop body of src/mnist.py:152-159 on one synthetic 512-image
batch: draw t and noise, q_sample, UNet forward, MSE, backward, (RCCL
all-reduce of the flat gradient), AdamW.  Inputs are resident in HBM before
the timed region.  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_TRAIN = 512
B_SAMPLE = 4096
# algorithmic work of the UNet
FWD_FLOP_PER_SAMPLE = 129_002_880
TRAIN_FLOP_PER_SAMPLE = 3 * FWD_FLOP_PER_SAMPLE
TRAIN_BYTES_PER_SAMPLE = 5_901_168 + 18_816
OPT_BYTES_PER_STEP = 7 * 725_892
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0


def cpu_baseline(cores: int):
    """The CPU oracle (a port of the reference's PyTorch-CPU path) timed on the
    host: train step at B=512, 1 warm-up + 4 measured steps."""
    from oracle import ddpm_oracle as O           # checker / baseline only
    torch.set_num_threads(cores)
    p = O.unet_init_params(0)
    tabs = O.make_tables()
    g = torch.Generator().manual_seed(1234)
    x0 = torch.rand(B_TRAIN, 1, 28, 28, generator=g) * 2 - 1
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    times = []
    for step in range(1, 6):
        t0 = time.perf_counter()
        t = torch.randint(0, 1000, (B_TRAIN,), generator=g)
        noise = torch.randn(B_TRAIN, 1, 28, 28, generator=g)
        _, grads = O.unet_loss_and_grads(p, x0, t, noise, tabs)
        for k in p:
            p[k], m[k], v[k] = O.adamw_step(p[k], grads[k], m[k], v[k], step)
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(1.0 / med, 4), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "4 train steps at B=512 (1 warm-up), fp32 PyTorch-CPU oracle, median"}


def time_kernel(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters   # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sample-steps", type=int, default=100, help="reverse steps timed at B=4096 (0 = skip)")
    ap.add_argument("--text-steps", type=int, default=20, help="text-denoiser train steps timed (0 = skip)")
    ap.add_argument("--gemm-mode", type=int, default=1, choices=[0, 1, 2],
                    help="transformer linear layers: 1 = bf16x3 split MFMA (default), 2 = plain bf16 MFMA, 0 = fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-mode", type=int, default=2, choices=[0, 1, 2],
                    help="UNet conv arithmetic: 2 = bf16x3 split MFMA over pre-split tensors (default), "
                         "1 = bf16x3 splitting while staging, 0 = exact fp32 MFMA")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; there is no CPU path")
    # TDM_DIST_BACKEND=gloo + TDM_SHARE_GPU=1: rehearsal of the multi-rank path on a one-GPU box
    backend = os.environ.get("TDM_DIST_BACKEND", "nccl")
    if os.environ.get("TDM_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from tinydiffusionmodels_amd import _lib, unet_engine as E
    from tinydiffusionmodels_amd.mnist import SimpleUNet, DDPMTrainer, reverse_diffusion

    E.check_layout_against_library()
    _lib.check(_lib.lib().tdm_set_conv_mode(args.conv_mode))
    torch.manual_seed(0)                       # identical default init on every rank (+ broadcast in the trainer)
    model = SimpleUNet().to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x0 = torch.rand(B_TRAIN, 1, 28, 28, device=dev, generator=gen) * 2 - 1
    torch.manual_seed(4321 + rank)             # rank-distinct t / noise streams
    trainer = DDPMTrainer(model, batch_size=B_TRAIN, lr=1e-3)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x0)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x0)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    loss_val = float(loss.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    # the step's one collective, timed alone (725,892 B per rank per step; ring-equivalent bandwidth)
    allreduce = None
    if world > 1:
        gbuf = torch.zeros_like(trainer.state.grads)
        for _ in range(5):
            dist.all_reduce(gbuf, op=dist.ReduceOp.SUM)
        sync()
        t0 = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(gbuf, op=dist.ReduceOp.SUM)
        sync()
        tt = torch.tensor([(time.perf_counter() - t0) / 50], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        nbytes = gbuf.numel() * 4
        allreduce = {"bytes": nbytes, "us": round(tt.item() * 1e6, 1),
                     "ring_equiv_gbs": round(2.0 * (world - 1) / world * nbytes / tt.item() / 1e9, 2),
                     "share_of_step": round(tt.item() / (elapsed / args.steps), 4)}

    out = {
        "metric": "DDPM train steps/sec, MNIST UNet b=512/GPU (512-image steps summed over GPUs)",
        "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32",
        "arithmetic": "fp32 everywhere (fp32-input MFMA convs)" if args.conv_mode == 0 else
                      "fp32 tensors and accumulation; conv operands split into bf16 hi+lo, hi*hi + hi*lo + lo*hi on bf16 MFMA "
                      "(16 mantissa bits per operand, predicted noise within 2e-5 of the fp32 reference (1.3e-5 on the golden batch; bound 1e-3))",
        "data": "synthetic",
        "config": {"workload": "MNIST DDPM UNet train step (q_sample+fwd+MSE+bwd+AdamW), batch 512 per GPU, "
                               "1000-step linear beta schedule, " +
                               ("exact fp32 MFMA conv kernels" if args.conv_mode == 0 else "bf16x3 split-MFMA conv kernels"),
                   "batch_per_gpu": B_TRAIN, "global_batch": B_TRAIN * world, "parallelism": f"dp{world}"},
        "images_per_s": round(value * B_TRAIN, 1),
        "final_loss": loss_val,
    }
    if allreduce is not None:
        out["allreduce"] = allreduce

    if rank == 0:
        # ---- whole-step roofline fractions (algorithmic work / measured time) ----
        step_s = elapsed / args.steps
        flops = TRAIN_FLOP_PER_SAMPLE * B_TRAIN
        byts = TRAIN_BYTES_PER_SAMPLE * B_TRAIN + OPT_BYTES_PER_STEP
        out["step_roofline"] = {"tflops": round(flops / step_s / 1e12, 2),
                                "frac_f32_mfma": round(flops / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                "gbs": round(byts / step_s / 1e9, 1),
                                "frac_hbm": round(byts / step_s / 1e9 / PEAK_HBM_GBS, 4)}
        # ---- dominant kernel (26 % of the step): the implicit-GEMM conv kernel, timed ALONE with events on
        # the launch stream on the rb4.conv1 shape (96->32 3x3 @28x28, B=512: 43,352,064 FLOP and
        # (96+32)*4*784 = 401,408 B per sample).  In the default arithmetic (bf16x3
        # split MFMA over pre-split tensors) its binding roof is HBM (MFMA floor 26.6 us = HBM floor 25.7 us).
        L = _lib.lib()
        cin, cout, hw = 96, 32, 28
        xin = torch.randn(B_TRAIN, hw, hw, cin, device=dev)
        w = torch.randn(3, 3, cin, cout, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        yout = torch.empty(B_TRAIN, hw, hw, cout, device=dev)
        woff = (9 * cin * cout + 63) & ~63
        scratch = torch.empty(woff + B_TRAIN * hw * hw * cin + 128, device=dev)
        kname = {0: "conv_mfma_kernel<28,1,fwd>", 1: "conv_bf16x3_kernel<28,1>", 2: "conv_s16_kernel<28,1>"}[args.conv_mode]

        def conv_call(inp, flags):
            if args.conv_mode == 0:
                _lib.check(L.tdm_conv_nhwc_f32(_lib.ptr(xin), _lib.ptr(w), _lib.ptr(bias), None, None, _lib.ptr(yout),
                                               None, B_TRAIN, hw, cin, cout, 3, 1, _lib.stream()))
            elif args.conv_mode == 1:
                _lib.check(L.tdm_conv_nhwc_bf16x3_f32(_lib.ptr(xin), _lib.ptr(w), _lib.ptr(bias), None, None,
                                                      _lib.ptr(yout), None, _lib.ptr(scratch), B_TRAIN, hw, cin, cout,
                                                      3, 1, _lib.stream()))
            else:
                _lib.check(L.tdm_conv_nhwc_s16_f32(_lib.ptr(inp), _lib.ptr(w), _lib.ptr(bias), None, None,
                                                   _lib.ptr(yout), None, None, None, _lib.ptr(scratch), B_TRAIN, hw,
                                                   cin, cout, 3, flags, _lib.stream()))
        conv_call(xin, 1)                                   # mode 2: packs weights + pre-splits the input once
        xs16 = scratch[woff:woff + B_TRAIN * hw * hw * cin]
        ms = time_kernel(lambda: conv_call(xs16, 1 | 4 | 8))
        kflop = 2 * 9 * cin * cout * hw * hw * B_TRAIN
        kbytes = (cin + cout) * 4 * hw * hw * B_TRAIN
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")     # PMC FETCH_SIZE/WRITE_SIZE, see file
        if args.conv_mode == 2 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        if args.conv_mode == 0:
            ach = kflop / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic}
        else:
            ach = kbytes / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                               "mfma_tflops_bf16": round(3 * kflop / (ms * 1e-3) / 1e12, 1), "mfma_peak_bf16": 2500.0}
        out["roofline"].update({"kernel": kname + " (rb4.conv1 shape 96->32 3x3 @28x28, B=512)",
                                "ms_per_launch": round(ms, 4), "flop_per_launch": kflop,
                                "algorithmic_bytes_per_launch": kbytes})
        del xin, yout

    # ---- 1000-step sampling rate at B=4096 (configs[3]); sharded over ranks, no collectives ----
    if args.sample_steps > 0:
        xs = torch.randn(B_SAMPLE, 1, 28, 28, device=dev, generator=torch.Generator(device=dev).manual_seed(99 + rank))
        with torch.no_grad():
            reverse_diffusion(model, xs, t_start=args.sample_steps - 1)   # warm-up: captures the two-step hipGraph
            sync()
            t0 = time.perf_counter()
            reverse_diffusion(model, xs, t_start=args.sample_steps - 1)
            sync()
            el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        ms_rev = 1e3 * el / args.sample_steps
        out["sampling"] = {"batch_per_gpu": B_SAMPLE, "ms_per_reverse_step": round(ms_rev, 3),
                           "reverse_steps_timed": args.sample_steps, "hipgraph": args.sample_steps >= 16,
                           "imgs_per_s_1000_step": round(world * B_SAMPLE / (ms_rev * 1e-3 * 1000), 2),
                           "tflops": round(FWD_FLOP_PER_SAMPLE * B_SAMPLE / (ms_rev * 1e-3) / 1e12, 2)}
        del xs

    # ---- text denoiser train step (configs[4] shape: B=256/GPU, L=128, D=256; the reference's train mode,
    #      dropout 0.1 = src/shakespeare.py:490 default; the dropout-0 time is reported next to it) ----
    if args.text_steps > 0:
        from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer
        Bt, Lt, Dt = 256, 128, 256
        _lib.check(_lib.lib().tdm_set_gemm_mode(args.gemm_mode))
        xt = torch.randn(Bt, Lt, Dt, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank)) * 0.02

        def time_text(p_drop):
            torch.manual_seed(0)
            tmodel = TinyTransformer(Dt, dropout=p_drop).to(dev)
            tmodel.train()
            ttr = DenoiserTrainer(tmodel, Bt, Lt, lr=1e-4, weight_decay=1e-4)
            for _ in range(3):
                ttr.step(xt)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.text_steps):
                ttr.step(xt)
            sync()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = tt.item()
            return 1e3 * el / args.text_steps

        ms_t = time_text(0.1)
        ms_t0 = time_text(0.0)
        tflop = 3 * 8_257_536 * Bt * Lt / (ms_t * 1e-3) / 1e12      # 8,257,536 FLOP/token fwd, x3 train
        out["text_denoiser"] = {"batch_per_gpu": Bt, "seq_len": Lt, "dim": Dt, "dropout": 0.1, "ms_per_step": round(ms_t, 3),
                                "ms_per_step_dropout0": round(ms_t0, 3),
                                "steps_per_s": round(world * 1e3 / ms_t, 2),
                                "tokens_per_s": round(world * Bt * Lt * 1e3 / ms_t, 0), "tflops": round(tflop, 2),
                                "arithmetic": {0: "fp32 MFMA GEMMs (exact)", 1: "bf16x3 split-operand MFMA GEMMs, fp32 accumulate",
                                               2: "plain bf16-operand MFMA GEMMs, fp32 accumulate"}[args.gemm_mode] +
                                              "; fp32-MFMA attention, fp32 LayerNorm; counter-hash dropout masks"}
        del xt
        # rounding head of the same train step (row N1): logits + cross-entropy + the three gradients for the
        # 32,768 tokens of the batch against a GPT-2-sized vocabulary (synthetic table; V is not a multiple of 4)
        Vh = 50257
        Mh = Bt * Lt
        gh = torch.Generator(device=dev).manual_seed(11 + rank)
        xh = torch.randn(Mh, Dt, device=dev, generator=gh) * 0.5
        Wh = torch.randn(Vh, Dt, device=dev, generator=gh) * (1.0 / Dt ** 0.5)
        bh = torch.zeros(Vh, device=dev)
        idh = torch.randint(0, Vh, (Mh,), device=dev, generator=gh)
        L_ = _lib.lib()
        wsh = torch.empty(L_.tdm_round_workspace_floats(Mh, Vh, Dt), device=dev)
        lossh, dxh, dWh, dbh = torch.empty(1, device=dev), torch.empty_like(xh), torch.empty_like(Wh), torch.empty_like(bh)

        def head():
            _lib.check(L_.tdm_round_ce_loss_grad_f32(_lib.ptr(xh), _lib.ptr(Wh), _lib.ptr(bh), _lib.ptr(idh), 1.0, _lib.ptr(lossh),
                                                     _lib.ptr(dxh), _lib.ptr(dWh), _lib.ptr(dbh), _lib.ptr(wsh), Mh, Vh, Dt,
                                                     _lib.stream()), "round_ce")
        head()
        sync()
        nh = max(1, args.text_steps // 3)
        t0 = time.perf_counter()
        for _ in range(nh):
            head()
        sync()
        ms_h = 1e3 * (time.perf_counter() - t0) / nh
        out["text_denoiser"]["rounding_head"] = {"vocab": Vh, "tokens": Mh, "ms": round(ms_h, 3),
                                                 "tflops": round(3 * 2.0 * Mh * Vh * Dt / (ms_h * 1e-3) / 1e12, 1),
                                                 "logits_gb": round(Mh * 4.0 * ((Vh + 3) // 4 * 4) / 1e9, 2),
                                                 "loss": round(float(lossh.item()), 4)}
        del xh, Wh, bh, idh, wsh, dxh, dWh, dbh

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(avail, 16)             # a 1-GPU box's CPU share is 16 cores
            out["cpu_baseline"] = cpu_baseline(cores)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
