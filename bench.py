#!/usr/bin/env python3
"""Headline benchmark: DDPM train steps/s on the MNIST UNet at batch 512 per GPU
(BASELINE.json configs[1]; data-parallel over N GPUs = configs[2]), plus the
1000-step reverse-sampling rate at batch 4096 (configs[3]) and the text denoiser
(configs[4] shape) as extra keys.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is the loop body of src/mnist.py:152-159 on one synthetic 512-image
batch: draw t and noise, q_sample, UNet forward, MSE, backward, (RCCL
all-reduce of the flat gradient), AdamW — device-side Philox draws and AdamW step count,
issued eagerly with the backward's weight-gradient launches on a side stream (--graph: the
one-queue step as hipGraph replays; at N > 1: step call + all-reduce + AdamW).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

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
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16
PEAK_HBM_GBS = 8000.0

# Per-launch work of the default train step, in the order of tdm_unet_launch_name(id):
# (FLOP per sample, bytes per sample, bytes per call).  FLOP = 2 * taps * Cin * Cout * pixels of the convs the
# launch computes (fused 1x1 convs included); bytes = every tensor the launch reads or writes, counted once
# (fp32 / S16 tensors 4 B per element, byte masks 1 B per 4 elements) — the traffic floor of the launch AS BUILT.
_C3 = lambda ci, co, px: 2 * 9 * ci * co * px
_C1 = lambda ci, co, px: 2 * ci * co * px
_T28, _T14 = lambda c: 784 * c * 4, lambda c: 196 * c * 4     # one fp32 / S16 tensor per sample
_M28, _M14 = lambda c: 784 * c // 4, lambda c: 196 * c // 4   # one byte mask per sample
LAUNCH_WORK = [
    (0, 776, 181_473 * 4 + 1_451_520),                                       # pack_timebias (weights: read fp32, write bf16 hi/lo x 2 dirs; + timestep biases)
    (_C3(1, 32, 784), 3136 + _T28(32) + _M28(32), 0),                        # rb1.conv1
    (_C3(32, 32, 784) + _C1(1, 32, 784), 2 * _T28(32) + _M28(32) + 3136, 0),  # rb1.conv2 (+ rank-1 skip)
    (_C1(32, 64, 196), _T28(32) + _T14(32) + _T14(64), 0),                   # avgpool + rb2.skip
    (_C3(32, 64, 196), _T14(32) + _T14(64) + _M14(64), 0),                   # rb2.conv1
    (_C3(64, 64, 196), 4 * _T14(64) + _M14(64), 0),                          # rb2.conv2 (in, res, h2, h2s)
    (_C3(64, 64, 196), 2 * _T14(64) + _M14(64), 0),                          # rb3.conv1
    (_C3(64, 64, 196), 3 * _T14(64) + _M14(64), 0),                          # rb3.conv2 (in, res, h3s)
    (_C3(96, 32, 784) + _C1(96, 32, 784), _T14(64) + 3 * _T28(32) + _M28(32), 0),   # rb4.conv1 + skip
    (_C3(32, 32, 784) + 2 * _C1(32, 1, 784), 2 * _T28(32) + _M28(32) + 3 * 3136 + 784 * 5, 0),  # rb4.conv2 + out conv + MSE fwd/bwd + out conv grads (a1s, s4 in; mask, eps, deps, 40-float rows per 32 pixels out; noise in)
    (_C1(32, 1, 784) + _C1(96, 1, 784), 3136 + 784 * 5 + 2 * _T28(32) + _T14(64) + _M28(32), 0),   # relu mask of d x w_out + rb4.skip grads (factored): deps, rows, h1s, h3s, mask in; dc2s out
    (_C3(32, 32, 784), 2 * _T28(32), 0),                                     # rb4.conv2 wgrad
    (_C3(32, 32, 784), 2 * _T28(32) + _M28(32), 0),                          # rb4.conv2 dgrad
    (_C3(64, 32, 784), _T14(64) + _T28(32), 0),                              # rb4.conv1 wgrad (up(h3) part)
    (_C3(32, 32, 784), 2 * _T28(32), 0),                                     # rb4.conv1 wgrad (h1 part)
    (_C3(32, 32, 784) + _C1(1, 32, 784), 2 * _T28(32) + 3136, 0),            # rb4.conv1 dgrad, h1 part: dh4s in, (M,32) out; + rank-1 skip share (side queue)
    (_C3(32, 64, 784) + _C1(1, 64, 784), _T28(32) + 2 * _T14(64) + _M14(64) + 3136, 0),   # rb4.conv1 dgrad, up(h3) part at 14x14: dh4s + mask in, dout3 + its masked S16 twin out; + rank-1 skip share (FLOP: the algorithm's nine taps at 28x28; as built 16 taps at 14x14 = 0.44 x)
    (0, 0, 0),                                                               # (round 4's upsample-backward pass: fused into the launch before, not issued)
    (_C3(64, 64, 196), 2 * _T14(64), 0),                                     # rb3.conv2 wgrad
    (_C3(64, 64, 196), 2 * _T14(64) + _M14(64), 0),                          # rb3.conv2 dgrad
    (_C3(64, 64, 196), 2 * _T14(64), 0),                                     # rb3.conv1 wgrad
    (_C3(64, 64, 196), 4 * _T14(64) + _M14(64), 0),                          # rb3.conv1 dgrad + rb2's ReLU backward (in, res, mask; dout2s and its masked twin out)
    (0, 0, 0),                                                               # (relu_mask rb2: fused into the launch before, not issued)
    (_C3(64, 64, 196), 2 * _T14(64), 0),                                     # rb2.conv2 wgrad
    (_C3(64, 64, 196), 2 * _T14(64) + _M14(64), 0),                          # rb2.conv2 dgrad
    (_C3(32, 64, 196) + _C1(32, 64, 196), _T14(32) + 2 * _T14(64), 0),       # rb2.conv1 wgrad
    (_C3(64, 32, 196) + _C1(64, 32, 196), 2 * _T14(64) + _T14(32), 0),       # rb2.conv1 dgrad
    (_C1(1, 32, 784), 2 * _T28(32) + _T14(32) + _M28(32) + 3136, 0),         # combine_dh1_mask (+ rb1.skip wgrad)
    (_C3(32, 32, 784), 2 * _T28(32), 0),                                     # rb1.conv2 wgrad
    (_C3(32, 32, 784), 2 * _T28(32) + _M28(32), 0),                          # rb1.conv2 dgrad
    (0, 2 * 12_544 + 2 * 6_272, 0),                                          # group_sums (time_emb / conv1 bias grads)
    (_C3(1, 32, 784), 3136 + _T28(32), 0),                                   # rb1.conv1 wgrad
    (0, 0, 4 * (256 * (320 + 9248 + 64 + 64 + 64 + 64 + 27648 + 9248 + 32 + 33 + 384 + 160 + 1 + 4 * 96) + 128 * (18432 + 2048) + 64 * 3 * 36864) + 725_892),  # reduce
]
MFMA_LAUNCH = "conv_s16<", "wgrad2_s16<", "wgrad_s2d"    # launches whose FLOPs run on the matrix cores (bf16x3: 3 MFMA FLOP per FLOP)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(cores_all: int):
    """The CPU oracle (a port of the reference's PyTorch-CPU path) timed on this host.
    `value`: train steps/s at the metric's batch (512).  `points`: the protocol's configurations — MNIST train
    B=64 (BASELINE config 1), p_sample B=25 / B=64, text denoiser train B=32 and p_sample B=10 — at 8 threads
    and at all cores of this box's share, 3 warm-up + 10 (train) / 20 (p_sample) measured steps, median."""
    from oracle import ddpm_oracle as O           # checker / baseline only
    tabs = O.make_tables()

    def train_rate(B, threads, warm, meas):
        torch.set_num_threads(threads)
        p = O.unet_init_params(0)
        g = torch.Generator().manual_seed(1234)
        x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
        m = {k: torch.zeros_like(v) for k, v in p.items()}
        v = {k: torch.zeros_like(x) for k, x in p.items()}
        times = []
        for step in range(1, warm + meas + 1):
            t0 = time.perf_counter()
            t = torch.randint(0, 1000, (B,), generator=g)
            noise = torch.randn(B, 1, 28, 28, generator=g)
            _, grads = O.unet_loss_and_grads(p, x0, t, noise, tabs)
            for k in p:
                p[k], m[k], v[k] = O.adamw_step(p[k], grads[k], m[k], v[k], step)
            times.append(time.perf_counter() - t0)
        return _median(times[warm:])

    def psample_rate(B, threads):
        torch.set_num_threads(threads)
        p = O.unet_init_params(0)
        g = torch.Generator().manual_seed(99)
        x = torch.randn(B, 1, 28, 28, generator=g)
        t = torch.full((B,), 500, dtype=torch.long)
        times = []
        with torch.no_grad():
            for _ in range(23):
                t0 = time.perf_counter()
                x = O.p_sample(p, x, t, torch.randn(B, 1, 28, 28, generator=g), tabs)
                times.append(time.perf_counter() - t0)
        return _median(times[3:])

    def text_rates(threads):
        torch.set_num_threads(threads)
        p = O.transformer_init_params(256, seed=7)
        g = torch.Generator().manual_seed(7)
        x0 = torch.randn(32, 128, 256, generator=g) * 0.02
        times = []
        for _ in range(3 + 6):
            t0 = time.perf_counter()
            t = torch.randint(0, 1000, (32,), generator=g)
            noise = torch.randn(32, 128, 256, generator=g)
            _, grads = O.transformer_loss_and_grads(p, x0, t, noise, tabs)
            for k in p:
                O.adamw_step(p[k], grads[k], torch.zeros_like(p[k]), torch.zeros_like(p[k]), 1, lr=1e-4, weight_decay=1e-4)
            times.append(time.perf_counter() - t0)
        tr = _median(times[3:])
        x = torch.randn(10, 128, 256, generator=g)
        tt = torch.full((10,), 500, dtype=torch.long)
        times = []
        with torch.no_grad():
            for _ in range(13):
                t0 = time.perf_counter()
                x = O.text_p_sample(p, x, tt, torch.randn(10, 128, 256, generator=g), tabs)
                times.append(time.perf_counter() - t0)
        return tr, _median(times[3:])

    points = []
    for threads in sorted({min(8, cores_all), cores_all}):
        s64 = train_rate(64, threads, 3, 10)
        p25, p64 = psample_rate(25, threads), psample_rate(64, threads)
        ttr, tps = text_rates(threads)
        points.append({"threads": threads,
                       "mnist_train_b64": {"ms_per_step": round(1e3 * s64, 2), "steps_per_s": round(1 / s64, 3), "img_per_s": round(64 / s64, 1)},
                       "mnist_p_sample_b25": {"ms_per_step": round(1e3 * p25, 2), "img_per_s_1000_step": round(25 / (1000 * p25), 3)},
                       "mnist_p_sample_b64": {"ms_per_step": round(1e3 * p64, 2), "img_per_s_1000_step": round(64 / (1000 * p64), 3)},
                       "text_train_b32_l128_d256": {"ms_per_step": round(1e3 * ttr, 1), "tokens_per_s": round(32 * 128 / ttr, 0)},
                       "text_p_sample_b10": {"ms_per_step": round(1e3 * tps, 2)}})
    s512 = train_rate(B_TRAIN, cores_all, 1, 4)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(1.0 / s512, 4), "unit": "steps/s", "cores": cores_all, "kind": "port", "cpu": cpu_model,
            "sample": "4 train steps at B=512 (1 warm-up), fp32 PyTorch-CPU oracle, median; points: BASELINE.md §2 protocol "
                      "(3 warm-up + 10 train / 20 p_sample steps, median; text train 3 + 6)",
            "points": points}


def compact_line(out: dict, detail_path: str) -> dict:
    """The printed line: the contract's keys + roofline + cpu_baseline, short; per-launch tables, protocol points and texts live in
    the detail file (profiles/r05_bench_detail.json is the committed copy of the evidence run)."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    line = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {"workload": "MNIST DDPM UNet train step, batch 512 per GPU (BASELINE.json configs[1]; configs[2] at n_gpus > 1)",
                      "batch_per_gpu": cfg.get("batch_per_gpu"), "global_batch": cfg.get("global_batch"), "parallelism": cfg.get("parallelism"),
                      "collective": cfg.get("collective")}
    line["steady_state_steps_per_s"] = out.get("steady_state", {}).get("steps_per_s")
    if "allreduce" in out:
        line["allreduce"] = {k: out["allreduce"][k] for k in ("bytes", "us", "share_of_step", "via")}
    if "step_roofline" in out:
        sr = out["step_roofline"]
        line["step_roofline"] = {k: sr[k] for k in ("frac_hbm", "frac_bf16_mfma_issue", "frac_f32_mfma", "algorithmic_bytes") if k in sr}
    if "launch_table" in out:
        cb = out["launch_table"].get("conv_blocks", {})
        line["conv_blocks"] = {k: cb.get(k) for k in ("us", "hbm_frac_algorithmic", "hbm_frac")}
    if "roofline" in out:
        r = dict(out["roofline"])
        for k in ("timing", "second", "tensor_bytes", "flop", "mfma_flop_as_built_x3"):
            r.pop(k, None)
        if isinstance(r.get("in_step"), dict):
            r["in_step"] = {k: r["in_step"][k] for k in ("ms_per_launch", "frac") if k in r["in_step"]}
        line["roofline"] = r
    if "fp32_mode" in out:
        line["fp32_mode"] = {k: out["fp32_mode"][k] for k in ("steps_per_s", "frac_f32_mfma")}
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "cpu": c.get("cpu"),
                                "sample": "4 train steps at B=512 (1 warm-up) of the fp32 PyTorch-CPU oracle, median"}
    line["detail"] = detail_path
    if "sampling" in out:
        sm = out["sampling"]
        line["sampling"] = {k: sm.get(k) for k in ("batch_per_gpu", "imgs_per_s_1000_step", "ms_per_reverse_step", "measured", "frac_hbm")}
        line["sampling"]["sharding"] = "chains sharded over ranks, no collective"
    if "text_denoiser" in out:
        td = out["text_denoiser"]
        line["text_denoiser"] = {"b": td["batch_per_gpu"], "l": td["seq_len"], "d": td["dim"], "ms_per_step": td["ms_per_step"], "tokens_per_s": td["tokens_per_s"],
                                 "frac_bf16_mfma": td["frac_bf16_mfma"], "gemm_mode": td["gemm_mode"],
                                 "other_gemm_mode_ms": td["other_gemm_mode"]["ms_per_step"],
                                 "rounding_head_ms": td.get("rounding_head", {}).get("ms")}
    if "text_train_full" in out:
        tf = out["text_train_full"]
        line["text_train_full"] = {"b32_ms": tf["b32"]["ms_per_step"], "b256_ms": tf["b256"]["ms_per_step"], "vocab": tf["vocab"]}
    return line


def time_events(fn, iters, warm=3):
    """Average ms per call of fn(i) over `iters` back-to-back calls, HIP events on the launch stream."""
    for i in range(warm):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start N fresh worker processes of this script,
    one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would), let rank 0's JSON line through to this
    process's stdout and return the workers' worst exit code.  Runs BEFORE anything initialises HIP in this process
    (torch.cuda.device_count() does not): the children are spawned, never exec'ed over a process that holds the GPU."""
    import socket
    import subprocess
    share = os.environ.get("TDM_SHARE_GPU") == "1"       # rehearsal: all ranks on cuda:0 (collectives over gloo)
    visible = torch.cuda.device_count()
    if visible < 1:
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    if not share and visible < n:
        raise SystemExit(f"bench.py --gpus {n}: only {visible} GPU(s) visible on this host — refusing to report a "
                         f"{n}-GPU number from fewer devices (TDM_SHARE_GPU=1 TDM_DIST_BACKEND=gloo rehearses the "
                         f"multi-rank path on one GPU)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   TDM_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    limit = float(os.environ.get("TDM_BENCH_TIMEOUT", "1500"))
    t0, rc = time.time(), 0
    try:
        while any(p.poll() is None for p in procs):
            failed = [p for p in procs if p.poll() not in (None, 0)]
            if failed or time.time() - t0 > limit:
                rc = failed[0].returncode if failed else 124
                print(f"[bench] {'a rank exited with ' + str(rc) if failed else 'timeout'}; stopping the other ranks",
                      file=sys.stderr, flush=True)
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc or max((abs(p.returncode or 0) for p in procs), default=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sample-steps", type=int, default=100, help="reverse steps of the short (`quick`) chain timed at B=4096 (0 = skip)")
    ap.add_argument("--sample-chains", type=int, default=3, help="complete 1000-step reverse chains timed at B=4096, median reported (0 = skip)")
    ap.add_argument("--text-steps", type=int, default=20, help="text-denoiser train steps timed (0 = skip)")
    ap.add_argument("--gemm-mode", type=int, default=1, choices=[0, 1, 2],
                    help="transformer linear layers of the headline text figure: 1 = bf16x3 split MFMA (parity path, default), "
                         "2 = plain bf16 MFMA, 0 = fp32 MFMA; modes 1 and 2 are both reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-table", action="store_true", help="skip the per-launch replay timing")
    ap.add_argument("--detail-out", default=None, help="file for everything measured (default gpurun_out/bench_detail.json); the printed line is the short form")
    ap.add_argument("--graph", action="store_true", help="replay the train step as hipGraphs (one queue) instead of the default eager issue "
                                                         "with the backward's weight-gradient launches on the library's side stream")
    ap.add_argument("--no-graph", action="store_true", help="(default now) eager issue of the train step")
    ap.add_argument("--no-overlap", action="store_true", help="eager issue on ONE queue (tdm_set_bwd_overlap(0))")
    ap.add_argument("--conv-mode", type=int, default=2, choices=[0, 2],
                    help="UNet conv arithmetic: 2 = bf16x3 split MFMA over pre-split tensors (default), 0 = exact fp32 MFMA")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- N ranks: either torchrun (the driver's N > 1 form) set WORLD_SIZE, or this process starts the ranks itself ----
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={os.environ['WORLD_SIZE']}: the flag and the "
                         f"launcher disagree (torchrun --nproc-per-node N needs --gpus N)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries exactly ONE line, the JSON of rank 0: whatever libraries print on fd 1 meanwhile (gloo's "[Gloo] Rank 0 is
    # connected to ..." banner, RCCL debug output) is sent to stderr; the saved descriptor is used for the final print only
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("TDM_SHARE_GPU") != "1" and torch.cuda.device_count() < max(world, local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} of {world} needs GPU {local_rank}, but only {torch.cuda.device_count()} "
                         f"device(s) are visible")
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

    from tinydiffusionmodels_amd import _lib, dp, unet_engine as E
    from tinydiffusionmodels_amd.mnist import SimpleUNet, DDPMTrainer, reverse_diffusion

    L = _lib.lib()
    E.check_layout_against_library()
    _lib.check(L.tdm_set_conv_mode(args.conv_mode))
    torch.manual_seed(0)                       # identical default init on every rank (+ broadcast in the trainer)
    model = SimpleUNet().to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x0 = torch.rand(B_TRAIN, 1, 28, 28, device=dev, generator=gen) * 2 - 1
    torch.manual_seed(4321 + rank)             # rank-distinct t / noise streams (seeds the trainer's Philox key)
    if args.no_overlap:
        _lib.check(L.tdm_set_bwd_overlap(0))
    trainer = DDPMTrainer(model, batch_size=B_TRAIN, lr=1e-3, graph=bool(args.graph))
    xin = trainer.batch_buffer(B_TRAIN)        # (the per-launch table below replays single launches on this batch)
    xin.copy_(x0)
    x0 = xin
    # The timed loop is mnist.train()'s: a synthetic DATASET resident in HBM (32 global batches), a device-side permutation per
    # epoch, and the captured step taking its own batch (DDPMTrainer.begin_epoch / steps_epoch) — nothing is issued between
    # two graph replays, and at world 1 consecutive steps share a replay (EPOCH_UNROLL).
    its_per_epoch = 32
    n_data = its_per_epoch * B_TRAIN * world
    data = torch.rand(n_data, 1, 28, 28, device=dev, generator=torch.Generator(device=dev).manual_seed(99)) * 2 - 1   # same on every rank
    perms = [torch.randperm(n_data, generator=torch.Generator().manual_seed(e)).to(dev) for e in range(4)]
    epoch_pos = {"it": its_per_epoch, "epoch": 0}

    def run_steps(n):
        """n train steps, epochs begun as they come (what mnist.train() does per epoch: two small device copies)"""
        last = None
        while n > 0:
            if epoch_pos["it"] == its_per_epoch:
                trainer.begin_epoch(data, perms[epoch_pos["epoch"] % len(perms)])
                epoch_pos["it"], epoch_pos["epoch"] = 0, epoch_pos["epoch"] + 1
            k = min(n, its_per_epoch - epoch_pos["it"])
            last = trainer.steps_epoch(k)
            epoch_pos["it"] += k
            n -= k
        return last

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(max(args.warmup, 1 + 2 * DDPMTrainer.EPOCH_UNROLL))   # (the first step runs eagerly; then both graphs are captured and replayed once)
    sync()
    t0 = time.perf_counter()
    loss = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    loss_val = float(loss.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    # ---- steady state: the same replayed step until >= 1 s has passed.  The driver's window (--steps 20 --warmup 5) is
    #      ~20 ms of GPU time, inside the clock / power ramp of a cold device; this block shows what the step sustains ----
    ss_steps, ss_t0 = 0, time.perf_counter()
    sync()
    ss_t0 = time.perf_counter()
    while True:
        run_steps(100)
        ss_steps += 100
        torch.cuda.synchronize()
        stop = torch.tensor([1.0 if time.perf_counter() - ss_t0 >= 1.0 else 0.0], device=dev)
        if world > 1:
            dist.all_reduce(stop, op=dist.ReduceOp.MAX)      # every rank leaves the loop after the same number of steps
        if stop.item() > 0 or ss_steps >= 20000:
            break
    sync()
    ss_el = time.perf_counter() - ss_t0
    if world > 1:
        tt = torch.tensor([ss_el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ss_el = tt.item()
    steady = {"steps": ss_steps, "seconds": round(ss_el, 3), "steps_per_s": round(world * ss_steps / ss_el, 2),
              "ms_per_step": round(1e3 * ss_el / ss_steps, 4),
              "note": "the same step as `value`, run for >= 1 s after it (clock ramp of the short driver window excluded)"}

    # the step's one collective, timed alone (725,892 B per rank per step; ring-equivalent bandwidth)
    allreduce = None
    if world > 1:
        gbuf = torch.zeros_like(trainer.grads)
        for _ in range(5):
            dp.allreduce_grads_(gbuf)
        sync()
        t0 = time.perf_counter()
        for _ in range(50):
            dp.allreduce_grads_(gbuf)
        sync()
        tt = torch.tensor([(time.perf_counter() - t0) / 50], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        nbytes = gbuf.numel() * 4
        allreduce = {"bytes": nbytes, "us": round(tt.item() * 1e6, 1),
                     "ring_equiv_gbs": round(2.0 * (world - 1) / world * nbytes / tt.item() / 1e9, 2),
                     "share_of_step": round(tt.item() / (elapsed / args.steps), 4), "via": dp.collective_name()}

    graph_on = trainer._epoch_graph is not None
    unrolled = graph_on and trainer._epoch_graph[1] is not None
    out = {
        "metric": "DDPM train steps/sec, MNIST UNet b=512/GPU (512-image steps summed over GPUs)",
        "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.conv_mode == 0 else "f32 (bf16x3 operands)",
        "arithmetic": "fp32 everywhere (fp32-input MFMA convs)" if args.conv_mode == 0 else
                      "fp32 tensors and accumulation; conv operands split into bf16 hi+lo, hi*hi + hi*lo + lo*hi on bf16 MFMA "
                      "(16 mantissa bits per operand, predicted noise within 2e-5 of the fp32 reference (1.3e-5 on the golden batch; bound 1e-3))",
        "data": "synthetic",
        "config": {"workload": "MNIST DDPM UNet train step (device-drawn t/noise + q_sample + fwd + MSE + bwd + AdamW), batch 512 per GPU, "
                               "1000-step linear beta schedule, " +
                               ("exact fp32 MFMA conv kernels" if args.conv_mode == 0 else "bf16x3 split-MFMA conv kernels"),
                   "batch_per_gpu": B_TRAIN, "global_batch": B_TRAIN * world, "parallelism": f"dp{world}",
                   "step_issue": ((f"hipGraph replays of {DDPMTrainer.EPOCH_UNROLL} consecutive steps (single-step replays for remainders)" if unrolled
                                   else "one hipGraph replay per step") + ("" if trainer._epoch_whole else " + all-reduce + AdamW launches")) if graph_on
                   else ("eager launches from one C-ABI call per step (+ AdamW), " +
                         ("the backward's eight weight-gradient launches on the library's side stream next to the data-gradient chain "
                          "(4 event forks + 1 join per step)" if L.tdm_get_bwd_overlap() else "one queue")),
                   "batch_source": f"gathered inside the step from a {n_data}-image synthetic dataset resident in HBM (device-side permutation per {its_per_epoch}-step epoch)",
                   "collective": dp.collective_name()},
        "images_per_s": round(value * B_TRAIN, 1),
        "final_loss": loss_val,
        "steady_state": steady,
    }
    if allreduce is not None:
        out["allreduce"] = allreduce

    if rank == 0:
        # ---- whole-step roofline fractions (algorithmic work / measured time) ----
        step_s = elapsed / args.steps
        flops = TRAIN_FLOP_PER_SAMPLE * B_TRAIN
        byts = TRAIN_BYTES_PER_SAMPLE * B_TRAIN + OPT_BYTES_PER_STEP
        out["step_roofline"] = {"tflops": round(flops / step_s / 1e12, 2),
                                **({"frac_f32_mfma": round(flops / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)} if args.conv_mode == 0 else
                                   {"frac_bf16_mfma_issue": round(3 * flops / step_s / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}),
                                "gbs": round(byts / step_s / 1e9, 1), "algorithmic_bytes": byts,
                                "frac_hbm": round(byts / step_s / 1e9 / PEAK_HBM_GBS, 4)}

    # ---- every launch of the step timed ALONE with HIP events on the launch stream (tdm_unet_replay_launch_f32):
    #      the launch's in-pipeline arguments on two fully populated workspaces (1.3 GB each, alternated, so that the
    #      256 MB Infinity Cache cannot serve one launch's inputs to the next) ----
    if rank == 0 and args.conv_mode == 2 and not args.no_launch_table:
        nl = L.tdm_unet_launch_count()
        assert nl == len(LAUNCH_WORK), (nl, len(LAUNCH_WORK))
        sts = [trainer.state, E.TrainState(trainer.flat, B_TRAIN)]
        gn = torch.Generator(device=dev).manual_seed(5)
        tt_ = torch.randint(0, 1000, (B_TRAIN,), device=dev, generator=gn)
        nz = torch.randn(B_TRAIN, 1, 28, 28, device=dev, generator=gn)
        for st in sts:
            E.loss_and_grad(trainer.flat, st, x0, nz, tt_)
        slabs = E.slabs_for(dev)
        gscratch = torch.empty_like(trainer.grads)
        rows = []
        for lid in range(nl):
            def call(i, lid=lid):
                st = sts[i & 1]
                _lib.check(L.tdm_unet_replay_launch_f32(_lib.ptr(trainer.flat), _lib.ptr(st.x_noisy), _lib.ptr(tt_), _lib.ptr(st.eps),
                                                        _lib.ptr(st.deps), _lib.ptr(nz), _lib.ptr(gscratch), _lib.ptr(st.ws.ws), _lib.ptr(slabs),
                                                        B_TRAIN, lid, _lib.stream()), "replay")
            ms = time_events(call, 20)
            name = L.tdm_unet_launch_name(lid).decode()
            fl, bps, bpc = LAUNCH_WORK[lid]
            fl, by = fl * B_TRAIN, bps * B_TRAIN + bpc
            row = {"id": lid, "launch": name, "us": round(ms * 1e3, 2), "bytes": by, "flop": fl,
                   "hbm_frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            if any(k in name for k in MFMA_LAUNCH):
                row["mfma_frac_bf16x3"] = round(3 * fl / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
            rows.append(row)
        total_us = sum(r["us"] for r in rows)
        top = sorted(rows, key=lambda r: -r["us"])
        out["launch_table"] = {"method": "each launch replayed alone 20x (3 warm-up), HIP events on the launch stream, two alternating "
                                         "1.3 GB workspaces; bytes = tensors the launch reads + writes, once each",
                               "sum_us": round(total_us, 1), "n_launches": nl, "top": top[:8],
                               "conv_blocks": {"us": round(sum(r["us"] for r in rows if "conv_s16<" in r["launch"]), 1),
                                               "bytes": sum(r["bytes"] for r in rows if "conv_s16<" in r["launch"])},
                               "all_us": {str(r["id"]): r["us"] for r in rows}}
        cb = out["launch_table"]["conv_blocks"]
        cb["hbm_frac"] = round(cb["bytes"] / (cb["us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
        # `bytes` above are the tensors the launches AS BUILT touch — a tensor that is no longer written (h4, dout4, the
        # full-width d cat) LOWERS that fraction although the step got faster.  SURVEY.md section 8d's ALGORITHMIC bytes of
        # the convolutions these 14 launches compute do not move with the implementation: fp32 input + output elements of
        # rb1.conv2, rb2.conv1/2, rb3.conv1/2, rb4.conv1 + rb4.skip, rb4.conv2 + out, forward and data gradient.
        algo_fwd = 4 * (784 * (64 + 128 + 128 + 64 + 33) + 196 * (96 + 128 + 128 + 128))
        cb["algorithmic_bytes"] = 2 * algo_fwd * B_TRAIN
        cb["hbm_frac_algorithmic"] = round(cb["algorithmic_bytes"] / (cb["us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
        # ---- roofline of the DOMINANT launch (in-pipeline arguments, timed alone above).  What binds is decided, not assumed:
        #   mfma_us = 3 x FLOP / 2.5 PFLOP/s   (bf16x3: three bf16 MFMA products per fp32 FLOP; FLOP as the launch is BUILT)
        #   hbm_us  = traffic / 8 TB/s          (traffic: PMC bytes of the same launch when the committed fold was taken on THIS
        #                                        library - matching source digest - else the tensors the launch reads + writes)
        #   floor_us = max(mfma_us, hbm_us);  bound = the larger term, or "latency" when the launch is below 0.4 of BOTH roofs.
        # The contract's fields stay as defined: achieved = ALGORITHMIC bytes (fp32 input + output elements of
        # the convolution the launch computes, a fused second conv's shared input counted once) / duration, peak = 8 TB/s.
        ALGO_CH = {"rb4.conv1 + rb4.skip fwd": 96 + 32 + 32, "rb4.conv1 dgrad, h1 part": 32 + 32, "rb4.conv1 dgrad, up(h3) part": 32 + 16, "rb1.conv2 fwd": 32 + 32 + 1,
                   "rb4.conv2 + out conv fwd": 32 + 32 + 32 + 1, "rb4.conv2 dgrad": 64, "rb1.conv2 dgrad": 64}
        # MFMA work as built relative to the algorithm's FLOP: the phase form runs 4 of 9 taps over the 64 up-sampled channels
        phase_on = os.environ.get("TDM_RB4_PHASE", "1") != "0"
        AS_BUILT = {"rb4.conv1 + rb4.skip fwd": (4 * 64 + 9 * 32 + 96) / (9 * 96 + 96) if phase_on else 1.0,
                    "rb4.conv1 dgrad, up(h3) part": (16 * 196) / (9 * 784) if phase_on else 1.0,
                    "rb4.conv1 wgrad, up(h3) part": (16 * 196) / (9 * 784) if phase_on else 1.0}
        from tinydiffusionmodels_amd.build import source_digest
        digest = source_digest()
        traffic_by_name, traffic_src = {}, None
        tpath = os.path.join(ROOT, "profiles", "r05_conv_traffic.json")
        if os.path.exists(tpath):
            tdoc = json.load(open(tpath))
            if tdoc.get("source_digest") == digest:
                traffic_by_name = {v.get("launch", k): v for k, v in tdoc.get("by_launch_id", {}).items()}
                traffic_src = f"profiles/r05_conv_traffic.json (source digest {digest}: taken on this library; FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)"
            else:
                traffic_src = f"none: profiles/r05_conv_traffic.json was taken on other sources (digest {tdoc.get('source_digest')} != {digest})"

        def roof(r):
            kb = next((c for n, c in ALGO_CH.items() if r["launch"].startswith(n)), 0) * 4 * 784 * B_TRAIN or r["bytes"]
            us = r["us"]
            tr = traffic_by_name.get(r["launch"], {}).get("hbm_bytes_per_launch")
            built = next((c for n, c in AS_BUILT.items() if r["launch"].startswith(n)), 1.0)
            mfma_us = 3 * r["flop"] * built / (PEAK_BF16_MFMA_TFLOPS * 1e12) * 1e6
            hbm_us = (tr if tr else r["bytes"]) / (PEAK_HBM_GBS * 1e9) * 1e6
            floor_us = max(mfma_us, hbm_us)
            f_m, f_h = mfma_us / us, hbm_us / us
            bound = "latency" if max(f_m, f_h) < 0.4 else ("mfma" if mfma_us >= hbm_us else "hbm")
            ach = kb / (us * 1e-6) / 1e9
            return {"bound": bound, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                    "traffic": tr, "traffic_source": traffic_src,
                    "kernel": r["launch"] + f", B={B_TRAIN} (launch id {r['id']})", "us": us, "share_of_step": round(us / total_us, 4),
                    "floor_us": round(floor_us, 2), "time_over_floor": round(us / floor_us, 2),
                    "mfma_us": round(mfma_us, 2), "hbm_us": round(hbm_us, 2), "mfma_frac": round(f_m, 4), "hbm_traffic_frac": round(f_h, 4),
                    "algorithmic_bytes": kb, "tensor_bytes": r["bytes"], "flop": r["flop"], "mfma_flop_as_built_x3": int(3 * r["flop"] * built),
                    "timing": "HIP events on the launch stream, 20 launches alternating two 1.3 GB workspaces"}
        # dominant = the longest MFMA launch; launches within 3 % of it count as tied and the lowest id wins (stable from run to run)
        mf = [r for r in rows if any(k in r["launch"] for k in MFMA_LAUNCH)]
        top_us = max(r["us"] for r in mf)
        dom = min((r for r in mf if r["us"] >= 0.97 * top_us), key=lambda r: r["id"])
        out["roofline"] = roof(dom)
        r9 = next(r for r in rows if r["launch"].startswith("rb4.conv1 + rb4.skip fwd"))
        if dom["id"] != r9["id"]:
            out["roofline"]["second"] = roof(r9)      # the forward twin (round 2's roofline launch), for continuity
        out["launch_table"]["rooflines"] = [roof(r) for r in top[:8] if any(k in r["launch"] for k in MFMA_LAUNCH)]
        del sts, gscratch

        # ---- the same launch timed INSIDE the running step: the timed loop's steps issued eagerly (same launches, same
        #      order, ONE stream; 35 launches of host work per ~1 ms of GPU work keep the queue full) with a HIP-event pair
        #      recorded on the launch stream around that one launch (tdm_unet_mark_launch).  Its inputs are where the step's
        #      previous launches left them (L2 / Infinity Cache), unlike the replay above, which alternates two 1.3 GB
        #      workspaces so that every input comes from HBM; the kernel average of `rocprofv3 --kernel-trace` over the
        #      timed loop (profiles/r05_bench_kernel_summary.txt) is this number. ----
        def in_step(r, nsteps=96, drop=16):
            was, was_ov = trainer.use_graph, L.tdm_get_bwd_overlap()
            trainer.use_graph = False
            _lib.check(L.tdm_set_bwd_overlap(0))      # one queue: the launch alone on the GPU, as in the replay it is compared with
            try:
                _lib.check(L.tdm_unet_mark_launch(r["id"], nsteps), "mark_launch")
                run_steps(nsteps)
                buf = np.zeros(nsteps, dtype=np.float32)
                n = L.tdm_unet_mark_collect(buf.ctypes.data, nsteps)
                assert n == nsteps, n
            finally:
                L.tdm_unet_mark_launch(-1, 0)
                L.tdm_set_bwd_overlap(was_ov)
                trainer.use_graph = was
            us = float(np.median(buf[drop:]))
            kb = next((c for n_, c in ALGO_CH.items() if r["launch"].startswith(n_)), 0) * 4 * 784 * B_TRAIN or r["bytes"]
            ach = kb / (us * 1e-6) / 1e9
            return {"ms_per_launch": round(us * 1e-3, 4), "achieved": round(ach, 1), "frac": round(ach / PEAK_HBM_GBS, 4),
                    "steps": nsteps - drop, "min_us": round(float(buf[drop:].min()), 2), "max_us": round(float(buf[drop:].max()), 2),
                    "timing": f"HIP-event pair around this launch in each of {nsteps - drop} consecutive eagerly issued train steps "
                              f"(median; {drop} steps dropped), events on the launch stream, includes the event packets' own cost"}
        if world == 1:
            out["roofline"]["in_step"] = in_step(dom)
            if "second" in out["roofline"]:
                out["roofline"]["second"]["in_step"] = in_step(r9)

    # ---- the same step in the exact-fp32 arithmetic (--conv-mode 0), for the record ----
    if rank == 0 and world == 1 and args.conv_mode == 2:
        _lib.check(L.tdm_set_conv_mode(0))
        tr0 = DDPMTrainer(model, batch_size=B_TRAIN, lr=1e-3, graph=False, broadcast=False)
        for _ in range(3):
            tr0.step(x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            tr0.step(x0)
        torch.cuda.synchronize()
        el0 = (time.perf_counter() - t0) / 15
        out["fp32_mode"] = {"conv_mode": 0, "steps_per_s": round(1 / el0, 2), "ms_per_step": round(1e3 * el0, 3),
                            "tflops": round(TRAIN_FLOP_PER_SAMPLE * B_TRAIN / el0 / 1e12, 1),
                            "frac_f32_mfma": round(TRAIN_FLOP_PER_SAMPLE * B_TRAIN / el0 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        del tr0
        _lib.check(L.tdm_set_conv_mode(2))

    # ---- 1000-step sampling rate at B=4096 (configs[3]); chains sharded over ranks, no collectives.  SURVEY.md section 8d:
    #      three COMPLETE 1000-step chains, median (`--sample-chains`); the short chain of `--sample-steps` reverse steps
    #      is kept as `quick` (what rounds 1-3 reported, extrapolated) ----
    if args.sample_steps > 0 or args.sample_chains > 0:
        xs = torch.randn(B_SAMPLE, 1, 28, 28, device=dev, generator=torch.Generator(device=dev).manual_seed(99 + rank))

        def chain(nsteps):
            with torch.no_grad():
                sync()
                t0 = time.perf_counter()
                reverse_diffusion(model, xs, t_start=nsteps - 1)
                sync()
                el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = tt.item()
            return el

        sbytes = 1_967_056 * B_SAMPLE + 725_892 + 4 * 784 * 4 * B_SAMPLE
        with torch.no_grad():
            reverse_diffusion(model, xs, t_start=max(args.sample_steps, 16) - 1)   # warm-up: captures the two-step hipGraph
        samp = {"batch_per_gpu": B_SAMPLE, "hipgraph": True,
                "sharding": "chains sharded over ranks (B per rank fixed), no collective",
                "noise": "Philox4x32-10 drawn inside the update kernel; step index in device memory"}
        if args.sample_steps > 0:
            ms_q = 1e3 * chain(args.sample_steps) / args.sample_steps
            samp["quick"] = {"reverse_steps_timed": args.sample_steps, "ms_per_reverse_step": round(ms_q, 3),
                             "imgs_per_s_1000_step_extrapolated": round(world * B_SAMPLE / ms_q, 2)}
            ms_rev = ms_q
        if args.sample_chains > 0:
            els = sorted(chain(1000) for _ in range(args.sample_chains))
            el_med = els[len(els) // 2]
            ms_rev = el_med                     # seconds per 1000 steps == ms per reverse step
            samp["full_chains"] = {"chains": args.sample_chains, "reverse_steps_each": 1000,
                                   "seconds_each": [round(e, 4) for e in els], "median_s": round(el_med, 4)}
        samp.update({"ms_per_reverse_step": round(ms_rev, 3),
                     "reverse_steps_timed": 1000 if args.sample_chains > 0 else args.sample_steps,
                     "imgs_per_s_1000_step": round(world * B_SAMPLE / ms_rev, 2),
                     "measured": ("median of %d complete 1000-step chains" % args.sample_chains) if args.sample_chains > 0
                                 else "extrapolated from the short chain",
                     "tflops": round(FWD_FLOP_PER_SAMPLE * B_SAMPLE / (ms_rev * 1e-3) / 1e12, 2),
                     "frac_hbm": round(sbytes / (ms_rev * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})
        out["sampling"] = samp
        del xs
        model._samplers.clear()

    # ---- text denoiser train step (configs[4] shape: B=256/GPU, L=128, D=256; the reference's train mode,
    #      dropout 0.1 = src/shakespeare.py:490 default; the dropout-0 time is reported next to it) ----
    if args.text_steps > 0:
        from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer
        Bt, Lt, Dt = 256, 128, 256
        xt = torch.randn(Bt, Lt, Dt, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank)) * 0.02

        def time_text(p_drop, gemm_mode, steps):
            _lib.check(L.tdm_set_gemm_mode(gemm_mode))
            torch.manual_seed(0)
            tmodel = TinyTransformer(Dt, dropout=p_drop).to(dev)
            tmodel.train()
            ttr = DenoiserTrainer(tmodel, Bt, Lt, lr=1e-4, weight_decay=1e-4)
            for _ in range(3):
                ttr.step(xt)
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                ttr.step(xt)
            sync()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = tt.item()
            return 1e3 * el / steps

        tflop_step = 3 * 8_257_536 * Bt * Lt / 1e12      # 8,257,536 FLOP/token fwd, x3 train
        ms_t = time_text(0.1, args.gemm_mode, args.text_steps)
        ms_t0 = time_text(0.0, args.gemm_mode, args.text_steps)
        other = 2 if args.gemm_mode != 2 else 1
        ms_o = time_text(0.1, other, args.text_steps)
        names = {0: "fp32 MFMA GEMMs (exact)", 1: "bf16x3 split-operand MFMA GEMMs, fp32 accumulate (parity path: 3e-6 rel)",
                 2: "plain bf16-operand MFMA GEMMs, fp32 accumulate (config 5's 'bf16 MFMA'; ~3e-3 rel, above the 1e-3 bound)"}
        _lib.check(L.tdm_set_gemm_mode(args.gemm_mode))
        out["text_denoiser"] = {"batch_per_gpu": Bt, "seq_len": Lt, "dim": Dt, "dropout": 0.1, "ms_per_step": round(ms_t, 3),
                                "ms_per_step_dropout0": round(ms_t0, 3),
                                "steps_per_s": round(world * 1e3 / ms_t, 2),
                                "tokens_per_s": round(world * Bt * Lt * 1e3 / ms_t, 0), "tflops": round(tflop_step / (ms_t * 1e-3), 2),
                                "frac_bf16_mfma": round(tflop_step / (ms_t * 1e-3) / PEAK_BF16_MFMA_TFLOPS, 4),
                                "gemm_mode": args.gemm_mode,
                                "arithmetic": names[args.gemm_mode] + "; bf16x3-MFMA attention (split operands, fp32 accumulate), fp32 LayerNorm; counter-hash dropout masks",
                                "other_gemm_mode": {"gemm_mode": other, "arithmetic": names[other], "ms_per_step": round(ms_o, 3),
                                                    "tokens_per_s": round(world * Bt * Lt * 1e3 / ms_o, 0),
                                                    "tflops": round(tflop_step / (ms_o * 1e-3), 2)}}
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
        from tinydiffusionmodels_amd import shakespeare as _S
        lossh, dxh, dWh, dbh = torch.empty(1, device=dev), torch.empty_like(xh), torch.empty_like(Wh), torch.empty_like(bh)
        nh = max(1, args.text_steps // 3)

        def time_head(chunk):
            """chunk > 0: the product's form at this size — logits never held, recomputed per vocabulary chunk;
            chunk = 0: logits stored once (one GEMM pass fewer, 6.5 GiB more workspace)"""
            n = L.tdm_round_workspace_chunked_floats(Mh, Vh, Dt, chunk) if chunk else L.tdm_round_workspace_floats(Mh, Vh, Dt)
            wsh = torch.empty(n, device=dev)

            def head():
                if chunk:
                    _lib.check(L.tdm_round_ce_loss_grad_chunked_f32(_lib.ptr(xh), _lib.ptr(Wh), _lib.ptr(bh), _lib.ptr(idh), 1.0,
                                                                    _lib.ptr(lossh), _lib.ptr(dxh), _lib.ptr(dWh), _lib.ptr(dbh),
                                                                    _lib.ptr(wsh), Mh, Vh, Dt, chunk, _lib.stream()), "round_ce")
                else:
                    _lib.check(L.tdm_round_ce_loss_grad_f32(_lib.ptr(xh), _lib.ptr(Wh), _lib.ptr(bh), _lib.ptr(idh), 1.0,
                                                            _lib.ptr(lossh), _lib.ptr(dxh), _lib.ptr(dWh), _lib.ptr(dbh),
                                                            _lib.ptr(wsh), Mh, Vh, Dt, _lib.stream()), "round_ce")
            head()
            sync()
            t0 = time.perf_counter()
            for _ in range(nh):
                head()
            sync()
            ms = 1e3 * (time.perf_counter() - t0) / nh
            gb = wsh.numel() * 4 / 1e9
            del wsh
            return ms, gb

        def time_fused(nseg):
            wsh = torch.empty(L.tdm_round_workspace_fused_floats(Mh, Vh, Dt, nseg), device=dev)

            def head():
                _lib.check(L.tdm_round_ce_loss_grad_fused_f32(_lib.ptr(xh), _lib.ptr(Wh), _lib.ptr(bh), _lib.ptr(idh), 1.0, _lib.ptr(lossh),
                                                              _lib.ptr(dxh), _lib.ptr(dWh), _lib.ptr(dbh), _lib.ptr(wsh), Mh, Vh, Dt, nseg,
                                                              _lib.stream()), "round_ce_fused")
            head()
            sync()
            t0 = time.perf_counter()
            for _ in range(nh):
                head()
            sync()
            ms = 1e3 * (time.perf_counter() - t0) / nh
            gb = wsh.numel() * 4 / 1e9
            del wsh
            return ms, gb

        chunk_h = _S.round_ce_chunk(Mh, Vh)
        nseg_h = _S.round_fused_nseg(Mh, Vh, Dt)
        ms_c, gb_c = time_head(chunk_h)
        ms_s, gb_s = time_head(0) if chunk_h else (ms_c, gb_c)
        if nseg_h:
            ms_h, gb_h = time_fused(nseg_h)
            form_h = (f"logits in registers only (csrc/ce_chain.hip): token-stationary online-softmax + dX pass, vocabulary-stationary "
                      f"dW / db pass over {nseg_h} token segments")
        else:
            ms_h, gb_h, form_h = ms_c, gb_c, f"logits never held: statistics pass + {chunk_h}-entry vocabulary chunks recomputed"
        loss_h = float(lossh.item())
        out["text_denoiser"]["rounding_head"] = {"vocab": Vh, "tokens": Mh, "ms": round(ms_h, 3), "form": form_h, "gemm_passes": 4,
                                                 "tflops": round(4 * 2.0 * Mh * Vh * Dt / (ms_h * 1e-3) / 1e12, 1),
                                                 "frac_bf16_mfma_issue": round(3 * 4 * 2.0 * Mh * Vh * Dt / (ms_h * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                                                 "workspace_gb": round(gb_h, 2),
                                                 "chunked_form": {"ms": round(ms_c, 3), "workspace_gb": round(gb_c, 2)},
                                                 "stored_logits_form": {"ms": round(ms_s, 3), "workspace_gb": round(gb_s, 2)},
                                                 "loss": round(loss_h, 4)}
        del xh, Wh, bh, idh, dxh, dWh, dbh
        # the FULL text train step (src/shakespeare.py:221-250 with learned embeddings: gather, device-drawn t / noise, q_sample,
        # denoiser fwd / bwd, fused rounding CE over V = 50,257, embedding scatter-add, AdamW over all four tensors, LR schedule)
        # as ONE hipGraph replay per batch — what `python -m src.shakespeare --train` runs per iteration (TextTrainStep)
        def time_full(Bf, steps):
            torch.manual_seed(0)
            fm = TinyTransformer(Dt, dropout=0.1).to(dev)
            fm.train()
            femb, frnd = _S.LearnedEmbedding(Vh, Dt).to(dev), _S.LearnedRounding(Dt, Vh).to(dev)
            st_ = _S.TextTrainStep(fm, frnd, femb, lr=1e-4, weight_decay=1e-4, rounding_weight=1.0,
                                   lr_lambda=_S.cosine_warmup_lambda(100, 10000), total_steps=10000)
            ids_ = torch.randint(0, Vh, (Bf, Lt), device=dev, generator=torch.Generator(device=dev).manual_seed(3 + rank))
            for _ in range(3):
                st_.step(ids_)
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                st_.step(ids_)
            sync()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = tt.item()
            ls = st_.losses.tolist()
            caps = st_.captures
            del st_, fm, femb, frnd
            _S._round_ws.clear()
            return 1e3 * el / steps, ls, caps
        nfull = max(2, args.text_steps // 2)
        ms32, l32, c32 = time_full(32, nfull)
        ms256, l256, c256 = time_full(256, nfull)
        out["text_train_full"] = {
            "what": "whole text train step incl. learned embedding + rounding head over V = 50,257 + AdamW on all tensors + "
                    "warm-up / cosine LR table, ONE hipGraph replay per batch (TextTrainStep = the step of shakespeare.train())",
            "vocab": Vh, "seq_len": Lt, "dim": Dt, "dropout": 0.1,
            "b32": {"ms_per_step": round(ms32, 3), "tokens_per_s": round(world * 32 * Lt * 1e3 / ms32, 0), "graph_captures": c32,
                    "losses_diff_rnd_total": [round(x, 4) for x in l32]},
            "b256": {"ms_per_step": round(ms256, 3), "tokens_per_s": round(world * 256 * Lt * 1e3 / ms256, 0), "graph_captures": c256,
                     "losses_diff_rnd_total": [round(x, 4) for x in l256]},
            "cpu_reference_point": "cpu_baseline.points[*].text_train_b32_l128_d256 is the DENOISER part only of the B = 32 step"}

    # ---- the reference's OWN small configurations (BASELINE.md section 2's protocol; cpu_baseline.points times the CPU oracle on
    #      the same ones): what a user of the reference's CLI defaults runs.  Graph-replayed like the product loops. ----
    if rank == 0 and world == 1 and args.text_steps > 0:
        from tinydiffusionmodels_amd import shakespeare as _S2
        from tinydiffusionmodels_amd.shakespeare import TinyTransformer as _TT, DenoiserTrainer as _DT
        pp = {}
        torch.manual_seed(11)
        m64 = SimpleUNet().to(dev)
        tr64 = DDPMTrainer(m64, batch_size=64, lr=1e-3, broadcast=False)
        d64 = torch.rand(64 * 32, 1, 28, 28, device=dev) * 2 - 1
        p64 = torch.randperm(64 * 32).to(dev)

        def steps64(n):
            while n > 0:
                tr64.begin_epoch(d64, p64)
                k = min(n, 32)
                tr64.steps_epoch(k)
                n -= k
        steps64(32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps64(256)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 256
        pp["mnist_train_b64"] = {"ms_per_step": round(ms, 4), "steps_per_s": round(1e3 / ms, 1), "img_per_s": round(64e3 / ms, 0)}
        m64.eval()
        for nb_ in (25, 64):
            with torch.no_grad():
                reverse_diffusion(m64, torch.randn(nb_, 1, 28, 28, device=dev), t_start=63)      # warm-up: captures the two-step graph
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reverse_diffusion(m64, torch.randn(nb_, 1, 28, 28, device=dev))
                torch.cuda.synchronize()
            el = time.perf_counter() - t0
            pp[f"mnist_p_sample_b{nb_}"] = {"ms_per_step": round(el, 4), "img_per_s_1000_step": round(nb_ / el, 1),
                                            "measured": "one complete 1000-step chain"}
        del tr64, m64, d64
        torch.manual_seed(12)
        mt = _TT(256, dropout=0.1).to(dev)
        mt.train()
        trt = _DT(mt, 32, 128, lr=1e-4)
        xt = torch.randn(32, 128, 256, device=dev) * 0.02
        for _ in range(4):
            trt.step(xt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            trt.step(xt)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 40
        pp["text_train_b32_l128_d256"] = {"ms_per_step": round(ms, 4), "tokens_per_s": round(32 * 128e3 / ms, 0),
                                          "what": "denoiser step only (the CPU point's scope); text_train_full.b32 is the whole step"}
        mt.eval()
        with torch.no_grad():
            _S2.reverse_diffusion(mt, torch.randn(10, 128, 256, device=dev), t_start=63)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _S2.reverse_diffusion(mt, torch.randn(10, 128, 256, device=dev))
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        pp["text_p_sample_b10"] = {"ms_per_step": round(el, 4), "measured": "one complete 1000-step chain"}
        del trt, mt
        out["protocol_points"] = {"what": "the reference's own small configurations on this GPU (BASELINE.md section 2; the CPU oracle's numbers "
                                          "for the same points are cpu_baseline.points)", **pp}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(avail, 16)             # a 1-GPU box's CPU share is 16 cores
            out["cpu_baseline"] = cpu_baseline(cores)
        sys.stdout.flush()
        # The ONE line stays short (the driver keeps the parsed standard keys and a 2,000-character tail): everything measured goes
        # to the detail file, the line carries the contract's keys, then - LAST, so they survive in the tail - the secondary
        # headline numbers (sampling, text denoiser, full text step).
        detail = args.detail_out or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail), exist_ok=True)
            with open(detail, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            detail = f"not written ({e})"
        os.write(json_fd, (json.dumps(compact_line(out, detail)) + "\n").encode())
    os.close(json_fd)
    if world > 1:
        dist.barrier()
        dp.shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
