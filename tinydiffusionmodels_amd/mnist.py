"""MI355X-native DDPM on MNIST — same Python surface as the reference's
src/mnist.py, with the per-timestep work done by hand-written HIP kernels
(libtdm_hip.so) instead of ATen ops.

• Training: python -m tinydiffusionmodels_amd.mnist --train   (or: python -m src.mnist --train)
• Sampling: python -m tinydiffusionmodels_amd.mnist --sample --ckpt ckpt.pth

Reference lines are cited per function.  There is no CPU fallback: tensors
must live on a HIP device and the library must be built."""
import argparse
import gzip
import math
import os
import struct
import zlib
from collections import OrderedDict
from contextlib import contextmanager
from pathlib import Path
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from . import dp
from . import unet_engine as E
from .schedule import TIMESTEPS, cpu_tables, device_tables, linear_beta_schedule, schedule_generation  # noqa: F401
from .utils import (get_samples_dir, get_vertex_checkpoint_path, load_checkpoint, save_checkpoint,
                    save_samples)

# module-level schedule globals, as in src/mnist.py:27-33 (CPU tensors; the
# kernels read per-device copies from schedule.device_tables()).
timesteps = TIMESTEPS
_T = cpu_tables()
betas = _T["betas"]
alphas = _T["alphas"]
alphas_cumprod = _T["alphas_cumprod"]
sqrt_alphas_cumprod = _T["sqrt_alphas_cumprod"]
sqrt_one_minus_alphas_cumprod = _T["sqrt_one_minus_alphas_cumprod"]


def q_sample(x_start: torch.Tensor, t: torch.Tensor, noise=None):
    """Diffuse the data for a given timestep t (src/mnist.py:36-42): one fused
    HIP kernel (table gather + a*x + s*n), bit-exact with the reference."""
    if noise is None:
        noise = torch.randn_like(x_start)
    E._need_cuda(x_start, t, noise)
    out = torch.empty_like(x_start, memory_format=torch.contiguous_format)
    return E.q_sample_into(x_start.contiguous(), t.contiguous(), noise.contiguous(), out)


class _UNetFunction(torch.autograd.Function):
    """Autograd bridge so `loss.backward()` reaches the flat parameter, as the
    reference's training loop does with its nn.Module (src/mnist.py:157-159)."""

    @staticmethod
    def forward(ctx, x, t, flat):
        need_grad = bool(ctx.needs_input_grad[2])
        ws = E.UNetWorkspace(x.shape[0], x.device, training=need_grad)
        eps = E.unet_forward(flat.detach(), x, t, ws, save=need_grad)
        if need_grad:
            ctx.ws = ws
            ctx.arith = _lib.arithmetic()          # (backward runs on the autograd thread: _lib.use_arithmetic)
            ctx.save_for_backward(x, flat)
        return eps

    @staticmethod
    def backward(ctx, deps):
        x, flat = ctx.saved_tensors
        with _lib.use_arithmetic(ctx.arith):
            grads = E.unet_backward(flat.detach(), x, deps.contiguous(), ctx.ws)
        ctx.ws = None
        return None, None, grads


class ResidualBlock(nn.Module):
    """src/mnist.py:45-61 as a stand-alone module: the reference's submodule names and parameter shapes
    (`conv1`, `conv2`, `time_emb`, `skip`; drawn in the reference's order), forward(x, t) with x (B,in_ch,H,W)
    and t the (B,1,1,1) float tensor SimpleUNet hands its blocks, computed by ONE C-ABI call
    (tdm_resblock_fwd_f32: conv1+ReLU, time bias, conv2+ReLU, skip) in the active conv arithmetic.

    Inference surface: SimpleUNet's own forward / backward run the fused whole-network pipeline over the flat
    parameter, not four of these; a block called with autograd recording raises (no silent ATen fallback).
    Geometries: H = W in {14, 28}; in_ch in {32, 64, 96} and out_ch in {32, 64}, or the first block's (1 -> 32 @ 28)."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.time_emb = nn.Linear(1, out_ch)
        if in_ch != out_ch:
            self.skip = nn.Conv2d(in_ch, out_ch, 1)
        else:
            self.skip = nn.Identity()
        self.in_ch, self.out_ch = in_ch, out_ch

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        E._need_cuda(x, t, self.conv1.weight)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("ResidualBlock.forward is the inference entry point of the HIP path (call it under "
                               "torch.no_grad()); training runs through SimpleUNet / DDPMTrainer")
        B, C, H, W = x.shape
        if C != self.in_ch or H != W or t.numel() != B:
            raise RuntimeError(f"ResidualBlock({self.in_ch},{self.out_ch}): got x {tuple(x.shape)}, t {tuple(t.shape)}")
        hwio = lambda w: w.detach().permute(2, 3, 1, 0).contiguous()                    # noqa: E731  (OIHW -> HWIO)
        xn = x.detach().permute(0, 2, 3, 1).contiguous().float()                         # NHWC (identical when C == 1)
        that = t.detach().reshape(B).contiguous().float()
        has_skip = isinstance(self.skip, nn.Conv2d)
        args = [xn, that, hwio(self.conv1.weight), self.conv1.bias.detach(), hwio(self.conv2.weight),
                self.conv2.bias.detach(), self.time_emb.weight.detach().reshape(-1).contiguous(), self.time_emb.bias.detach(),
                hwio(self.skip.weight) if has_skip else None, self.skip.bias.detach() if has_skip else None]
        L = _lib.lib()
        out = torch.empty(B, H, W, self.out_ch, device=x.device, dtype=torch.float32)
        scratch = torch.empty(L.tdm_resblock_scratch_floats(B, H, self.in_ch, self.out_ch), device=x.device,
                              dtype=torch.float32)
        _lib.check(L.tdm_resblock_fwd_f32(*[_lib.ptr(a) for a in args], _lib.ptr(out), _lib.ptr(scratch), B, H, self.in_ch,
                                          self.out_ch, _lib.stream()), "resblock_fwd")
        return out.permute(0, 3, 1, 2).contiguous()


class SimpleUNet(nn.Module):
    """The reference's noise predictor (src/mnist.py:64-87): four residual
    blocks (1→32 @28², 32→64 @14², 64→64 @14², 96→32 @28²), avg-pool down,
    nearest up, channel-concat skip, 1×1 out conv.

    All 181,473 parameters live in ONE flat fp32 nn.Parameter (`flat`, conv
    weights HWIO) that the HIP kernels read directly; `state_dict()` /
    `load_state_dict()` speak the reference's key names and OIHW layouts, so
    checkpoints are interchangeable."""

    def __init__(self):
        super().__init__()
        # Draw the initial weights exactly as the reference's constructor does
        # (same module order => same RNG stream under torch.manual_seed).
        sd = OrderedDict()
        for name, ci, co in E.BLOCKS:
            mods = [("conv1", nn.Conv2d(ci, co, 3, padding=1)), ("conv2", nn.Conv2d(co, co, 3, padding=1)),
                    ("time_emb", nn.Linear(1, co))]
            if ci != co:
                mods.append(("skip", nn.Conv2d(ci, co, 1)))
            for mname, m in mods:
                sd[f"{name}.{mname}.weight"] = m.weight.detach()
                sd[f"{name}.{mname}.bias"] = m.bias.detach()
        out = nn.Conv2d(32, 1, kernel_size=1)
        sd["out.weight"], sd["out.bias"] = out.weight.detach(), out.bias.detach()
        self.flat = nn.Parameter(E.flat_from_state_dict(sd))
        self._infer_ws = None
        self._samplers = {}

    # ---- reference-compatible checkpoint ABI ---------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = E.state_dict_from_flat(self.flat)
        out = destination if destination is not None else OrderedDict()
        for k, v in sd.items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        flat = E.flat_from_state_dict(state_dict, device=self.flat.device)
        with torch.no_grad():
            self.flat.copy_(flat)
        return torch.nn.modules.module._IncompatibleKeys([], [])

    # ---- forward -----------------------------------------------------------
    def _workspace(self, B, device):
        ws = self._infer_ws
        if ws is None or ws.B != B or ws.ws.device != device:
            ws = self._infer_ws = E.UNetWorkspace(B, device, training=False)
        return ws

    def _graph_sampler(self, n, device):
        key = (n, str(device), self.flat.data_ptr(), schedule_generation())   # (set_tables() retires captured tables)
        if key not in self._samplers:
            self._samplers.clear()                          # one resident sampler (buffers + graph) at a time
            self._samplers[key] = _GraphSampler(self, n, device)
        return self._samplers[key]

    def forward(self, x, t):
        """x (B,1,28,28) fp32 on the GPU, t (B,) int64 raw step indices (src/mnist.py:76-87)."""
        E._need_cuda(x, t, self.flat)
        if self.flat.requires_grad and torch.is_grad_enabled():
            return _UNetFunction.apply(x, t, self.flat)
        return E.unet_forward(self.flat.detach(), x, t, self._workspace(x.shape[0], x.device), save=False)


@contextmanager
def eval_mode(module: nn.Module):
    """src/mnist.py:90-97."""
    training = module.training
    module.eval()
    try:
        yield
    finally:
        module.train(training)


def p_sample(model, x, t, noise=None):
    """Perform one reverse step (src/mnist.py:167-180).  Like the reference it
    branches on t[0] (a device→host read); the sampling loops below use the
    sync-free uniform-t path instead.  `noise` optionally supplies the z the
    reference draws internally (teacher forcing)."""
    E._need_cuda(x, t)
    tabs = device_tables(x.device)
    eps = model(x, t)
    add_noise = int(t[0]) != 0
    if add_noise and noise is None:
        noise = torch.randn_like(x)
    B = x.shape[0]
    # contiguous copies are held in locals until the launch (a temporary would be freed, and its block
    # possibly re-used by the next argument's copy, before the kernel is enqueued)
    xc, tc = x.contiguous(), t.contiguous()
    zc = noise.contiguous() if add_noise else None
    out = torch.empty_like(xc)
    _lib.check(_lib.lib().tdm_p_sample_update_pert_f32(
        _lib.ptr(xc), _lib.ptr(eps), _lib.ptr(zc),
        _lib.ptr(tabs["sqrt_recip_alphas"]), _lib.ptr(tabs["eps_coef"]), _lib.ptr(tabs["sigma"]),
        _lib.ptr(tc), 1 if add_noise else 0, _lib.ptr(out), B, xc.numel() // B, _lib.stream()),
        "p_sample_update")
    return out


def _reverse_step_device_t(flat, ws, cur, t_vec, z, eps, nxt, sigma_tab, tabs):
    """x_{t-1} from x_t with the step index held in DEVICE memory (t_vec): UNet forward, then the
    update of src/mnist.py:173-180 gathered by t; sigma_tab[0] = 0 makes the t == 0 step return the
    mean exactly as the reference's `if t[0] == 0` branch does, with no host-side branch."""
    E.unet_forward(flat, cur, t_vec, ws, save=False, out=eps)
    B = cur.shape[0]
    _lib.check(_lib.lib().tdm_p_sample_update_pert_f32(
        _lib.ptr(cur), _lib.ptr(eps), _lib.ptr(z), _lib.ptr(tabs["sqrt_recip_alphas"]), _lib.ptr(tabs["eps_coef"]),
        _lib.ptr(sigma_tab), _lib.ptr(t_vec), 1, _lib.ptr(nxt), B, cur.numel() // B, _lib.stream()), "p_sample_update")


@torch.no_grad()
def reverse_diffusion(model: "SimpleUNet", x: torch.Tensor, noises=None, t_start: int = timesteps - 1,
                      use_graph: Optional[bool] = None) -> torch.Tensor:
    """The reverse loop of src/mnist.py:190-193 (`for i in reversed(range(T))`) with no host sync.

    use_graph (default: on when the noise is drawn on the device and the chain is long): two reverse
    steps (x ping-pong) are captured ONCE into a hipGraph — noise draw, 13 UNet launches, update,
    t -= 1 each — and the graph is replayed (t_start+1)/2 times; the step index lives in device
    memory.  noises: optional sequence of z tensors (teacher forcing), noises[k] used at
    t = t_start-k; that path runs eagerly with a host-side step index."""
    E._need_cuda(x)
    n = x.shape[0]
    dev = x.device
    flat = model.flat.detach()
    ws = model._workspace(n, dev)
    eps = torch.empty_like(x)
    nsteps = t_start + 1
    if use_graph is None:
        use_graph = noises is None and nsteps >= 16
    cur, nxt = x.contiguous().clone(), torch.empty_like(x)
    if not use_graph:
        # one (n,) int64 vector per step would be 1000 tiny fills: keep all of them resident instead
        t_all = torch.arange(t_start, -1, -1, device=dev, dtype=torch.long).view(-1, 1).expand(-1, n).contiguous()
        for k, i in enumerate(range(t_start, -1, -1)):
            if i > 0:
                z = noises[k].contiguous() if noises is not None else torch.randn_like(cur)
            else:
                z = None
            E.p_sample_step(flat, ws, cur, t_all[k], i, z, eps, nxt)
            cur, nxt = nxt, cur
        return cur

    sampler = model._graph_sampler(n, dev)
    return sampler.run(cur, nsteps, t_start, noises)


class _GraphSampler:
    """Persistent buffers + one captured two-step hipGraph for a (model, batch size) pair."""

    def __init__(self, model: "SimpleUNet", n: int, dev):
        self.model, self.n, self.dev = model, n, dev
        self.tabs = device_tables(dev)
        self.sigma0 = self.tabs["sigma"].clone()
        self.sigma0[0] = 0.0                               # t == 0: x = mean (src/mnist.py:176-177)
        self.xa = torch.empty(n, 1, 28, 28, device=dev)
        self.xb = torch.empty_like(self.xa)
        self.eps = torch.empty_like(self.xa)
        self.z = torch.empty_like(self.xa)
        self.t_vec = torch.zeros(n, device=dev, dtype=torch.long)
        self.kidx = torch.zeros(1, device=dev, dtype=torch.long)
        # device-drawn noise: Philox key = a constant mixed with the rank (ranks draw distinct streams); the 64-bit stream
        # OFFSET is what torch's generator governs — run() draws a fresh base offset per chain, the device advances it
        self.seed = dp.sampler_stream_key()
        self.offset0 = 0
        self.rng_state = torch.zeros(2, device=dev, dtype=torch.long)
        self.ws = E.UNetWorkspace(n, dev, training=False)  # owned here: the graph holds its address
        self.bank = None
        self.graph = None
        self.graph_has_bank = None

    def _one(self, a, b):
        if self.bank is None:
            # one C-ABI call per reverse step: UNet forward, update with z drawn in registers, t -= 1 — no ATen launch
            _lib.check(_lib.lib().tdm_unet_p_sample_step_philox_f32(
                _lib.ptr(self.model.flat.detach()), _lib.ptr(a), _lib.ptr(self.t_vec), _lib.ptr(self.tabs["sqrt_recip_alphas"]),
                _lib.ptr(self.tabs["eps_coef"]), _lib.ptr(self.sigma0), self.seed, _lib.ptr(self.rng_state), _lib.ptr(self.eps),
                _lib.ptr(b), _lib.ptr(self.ws.ws), self.n, _lib.stream()), "p_sample_step_philox")
            return
        # teacher forcing inside the graph: z gathered from a resident bank by a device-side counter
        torch.index_select(self.bank, 0, self.kidx, out=self.z.view(1, -1))
        self.kidx.add_(1).clamp_(max=self.bank.shape[0] - 1)
        _reverse_step_device_t(self.model.flat.detach(), self.ws, a, self.t_vec, self.z, self.eps, b, self.sigma0,
                               self.tabs)
        self.t_vec.sub_(1)

    def _capture(self):
        saved = (self.xa.clone(), self.t_vec.clone(), self.kidx.clone(), self.rng_state.clone())
        side = torch.cuda.Stream(device=self.dev)          # warm-up off the capture stream, then restore
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._one(self.xa, self.xb)
            self._one(self.xb, self.xa)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.xa.copy_(saved[0]); self.t_vec.copy_(saved[1]); self.kidx.copy_(saved[2]); self.rng_state.copy_(saved[3])
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._one(self.xa, self.xb)
            self._one(self.xb, self.xa)
        self.graph_has_bank = self.bank is not None

    def run(self, x, nsteps, t_start, noises):
        bank = None
        if noises is not None:
            bank = torch.stack([zz.contiguous() for zz in noises[:nsteps]]).view(-1, x.numel())
        if self.graph is not None and (self.graph_has_bank != (bank is not None) or
                                       (bank is not None and (self.bank is None or bank.shape != self.bank.shape))):
            self.graph = None                               # the captured bank address / shape changed
        if bank is not None and self.graph is not None:
            self.bank.copy_(bank)
        else:
            self.bank = bank
        self.xa.copy_(x)
        self.t_vec.fill_(t_start)
        self.kidx.zero_()
        # a fresh stream per chain, drawn from torch's (CPU) generator like the reference's randn_like draws are:
        # `torch.manual_seed(s); reverse_diffusion(...)` twice gives the same samples although the sampler is cached
        self.offset0 = dp.draw_stream_offset()
        self.rng_state.copy_(torch.tensor([self.offset0, 0], dtype=torch.long))
        left = nsteps
        if left % 2 == 1:                                   # odd chain: first step eagerly (a -> b), then move b to a
            self._one(self.xa, self.xb)
            self.xa.copy_(self.xb)
            left -= 1
        if left > 0:
            if self.graph is None:
                self._capture()
            for _ in range(left // 2):
                self.graph.replay()
        return self.xa.clone()


def to_image_range(x: torch.Tensor):
    """(x.clamp(-1,1)+1)/2 and its uint8 quantisation (src/mnist.py:194 + save_image)."""
    E._need_cuda(x)
    x = x.contiguous()
    x01 = torch.empty_like(x)
    u8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().tdm_to_unit_u8_f32(_lib.ptr(x), _lib.ptr(x01), _lib.ptr(u8), x.numel(), _lib.stream()),
               "to_unit_u8")
    return x01, u8


# ---- PNG grid without torchvision ---------------------------
def make_grid_u8(u8: torch.Tensor, nrow: int, padding: int = 2) -> torch.Tensor:
    """uint8 (N,1,H,W) -> (3, Hgrid, Wgrid) grid laid out like
    torchvision.utils.make_grid(nrow=nrow, padding=2, pad_value=0) (restated
    from its documented behaviour; torchvision is not installed here)."""
    u8 = u8.cpu()
    n, _, h, w = u8.shape
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(n / xmaps))
    H, W = h + padding, w + padding
    grid = torch.zeros(3, H * ymaps + padding, W * xmaps + padding, dtype=torch.uint8)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[:, y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = u8[k, 0]
    return grid


def encode_png(img: torch.Tensor) -> bytes:
    """(3,H,W) or (H,W) uint8 -> PNG bytes (zlib + CRC, no external deps)."""
    if img.dim() == 2:
        img = img.unsqueeze(0)
    c, h, w = img.shape
    color_type = {1: 0, 3: 2}[c]
    rows = img.permute(1, 2, 0).contiguous().numpy().reshape(h, w * c)
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _save_grid(x: torch.Tensor, samples_dir, fname: str):
    _, u8 = to_image_range(x)
    png = encode_png(make_grid_u8(u8, nrow=int(math.sqrt(x.shape[0]))))
    if isinstance(samples_dir, str):
        sample_path = f"{samples_dir}/{fname}"
    else:
        sample_path = samples_dir / fname
    save_samples(png, sample_path, mode="wb")
    return sample_path


def sample_images(model: nn.Module, device: str, epoch: int, n_samples: int = 25, outdir: str = "samples"):
    """src/mnist.py:99-126."""
    samples_dir = get_samples_dir(outdir)
    with eval_mode(model), torch.no_grad():
        x = torch.randn(n_samples, 1, 28, 28, device=device)
        x = reverse_diffusion(model, x)
        sample_path = _save_grid(x, samples_dir, f"epoch_{epoch:03d}.png")
    print(f"[epoch {epoch}] saved samples to {sample_path}")


# ---- data ----------------------------------------------------
def load_mnist_idx(root: str = "./data") -> torch.Tensor:
    """Read the MNIST training images from torchvision's on-disk layout
    (root/MNIST/raw/train-images-idx3-ubyte[.gz]) and apply
    ToTensor + Normalize((0.5,), (0.5,)) (src/mnist.py:139-145) -> (N,1,28,28) fp32 in [-1,1]."""
    base = Path(root) / "MNIST" / "raw"
    for name, opener in (("train-images-idx3-ubyte", open), ("train-images-idx3-ubyte.gz", gzip.open)):
        p = base / name
        if p.exists():
            with opener(p, "rb") as f:
                buf = f.read()
            magic, n, h, w = struct.unpack(">IIII", buf[:16])
            if magic != 2051 or (h, w) != (28, 28):
                raise RuntimeError(f"{p}: not an MNIST image file")
            img = torch.frombuffer(bytearray(buf[16:16 + n * h * w]), dtype=torch.uint8).view(n, 1, h, w)
            return (img.float() / 255.0 - 0.5) / 0.5
    raise RuntimeError(
        f"MNIST not found under {base} and this machine has no network/torchvision to download it; "
        "place the IDX files there or pass --synthetic")


def synthetic_mnist(n: int, seed: int = 1234) -> torch.Tensor:
    """Synthetic 28×28×1 batches in the Normalize(0.5,0.5) range."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 1, 28, 28, generator=g) * 2 - 1


class DPStepper:
    """Collective + optimiser ordering of ONE data-parallel train step, independent of where the gradient
    comes from (DDPMTrainer: the HIP kernels; the CPU tests: the oracle over gloo).

    Equal shards (global_batch=None): per-rank mean-loss gradient, all-reduce(SUM), 1/world folded into AdamW.
    Ragged tail (global_batch = samples ALL ranks hold in this iteration): each rank weights its mean-loss
    gradient by B_local / global_batch before the SUM — ranks without samples contribute zeros but still join
    the collective — which is exactly the gradient of the mean loss over the global batch.  No collective is
    ever issued outside step() (construction included), so ranks cannot fall out of step with each other."""

    def local_loss_and_grad(self, x0, t, noise):      # leaves the flat gradient in the buffer all-reduced below
        raise NotImplementedError

    def grad_buffer(self) -> torch.Tensor:
        raise NotImplementedError

    def optimizer_step(self, grad_scale: float) -> None:
        raise NotImplementedError

    def allreduce_equal_shards(self, g: torch.Tensor) -> float:
        """The collective of an equal-shard step (implementers may split it: DDPMTrainer._allreduce)."""
        return dp.allreduce_grads_(g)

    def step(self, x0: Optional[torch.Tensor], t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
             global_batch: Optional[int] = None):
        b_local = 0 if x0 is None else int(x0.shape[0])
        if global_batch is None and b_local == 0:
            raise RuntimeError("a rank without samples needs global_batch (the tail iteration of an epoch)")
        loss = self.local_loss_and_grad(x0, t, noise) if b_local else None
        g = self.grad_buffer()
        if global_batch is None:
            scale = self.allreduce_equal_shards(g)    # SUM over ranks; 1/world folded into AdamW
        else:
            if b_local:
                g.mul_(b_local / float(global_batch))
            else:
                g.zero_()
            dp.allreduce_grads_(g)
            scale = 1.0
        self.optimizer_step(scale)
        return loss


class DDPMTrainer(DPStepper):
    """The loop body of src/mnist.py:152-159 as one device-side step:
    t ~ U{0..999}, noise ~ N(0,1), q_sample, UNet forward, MSE, backward,
    (data-parallel: one RCCL all-reduce of the flat 726 KB gradient), AdamW
    with torch defaults (lr passed, betas (0.9,0.999), eps 1e-8, wd 0.01).

    With t / noise left to the trainer nothing in a step is written by the host: the draws come from a device-side
    Philox stream, AdamW's step count lives in device memory.  The step's launches are issued EAGERLY by default (one C-ABI
    call + the optimiser's), with the backward's weight-gradient launches on the library's side stream next to the data-
    gradient chain (tdm_set_bwd_overlap; TDM_BWD_OVERLAP=0 keeps one queue): 5-10 % faster than the one-queue step at every
    batch size from 25 to 1024, while the same step replayed as ONE hipGraph (`graph=True` or TDM_TRAIN_GRAPH=1; captured
    with one queue, the forked graph replays slower) only ties the one-queue eager issue — the host is 2x ahead of the GPU
    even at B = 25.  At world > 1 the graph form ends before
    the collective (replay, all-reduce, AdamW = three host calls per step) unless dp.graph_collective_ok():
    opt-in (TDM_GRAPH_COLLECTIVE=1, or =auto for a capture / replay / compare self-check of the native RCCL all-reduce
    that every rank must pass) — then the all-reduce and AdamW are captured too and a step is ONE host call at any
    world size; unverified on hardware, hence not the default.  Explicit t / noise (teacher forcing, parity tests) run eagerly.
    One trainer serves every batch size (workspaces per size, optimiser state shared), and its constructor
    issues no collective unless `broadcast` (default: rank 0's weights to every replica, once)."""

    def __init__(self, model: "SimpleUNet", batch_size: int, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, graph: Optional[bool] = None, broadcast: bool = True):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.flat = model.flat.detach()
        E._need_cuda(self.flat)
        dev = self.flat.device
        self.rank, self.world = dp.world_info()
        # data parallel, eager issue: the backward finishes the gradient in two parts so that most of the all-reduce runs under it
        self._early = self.world > 1 and os.environ.get("TDM_EARLY_GRADS", "1") != "0"
        self._early_off = int(_lib.lib().tdm_unet_early_grad_offset())
        if self._early or os.environ.get("TDM_EARLY_GRADS_W1") == "1":   # (W1: the two-part reduction on one GPU, A/B timing of the step's tail)
            _lib.check(_lib.lib().tdm_set_early_grads(1), "tdm_set_early_grads")
        self.batch_size = batch_size
        self.grads = torch.zeros(E.NPARAM, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.grads)
        self.v = torch.zeros_like(self.grads)
        self.step_state = torch.zeros(4, dtype=torch.long, device=dev)     # {AdamW steps taken, scratch, beta1^t, beta2^t}
        self.rng_state = torch.zeros(2, dtype=torch.long, device=dev)      # {Philox stream offset, scratch}
        # rank-distinct draw streams: a seed from torch's generator (so torch.manual_seed governs it) mixed with the rank
        self.seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) ^ (self.rank * 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)
        self.use_graph = (os.environ.get("TDM_TRAIN_GRAPH", "0") == "1") if graph is None else bool(graph)
        if os.environ.get("TDM_BWD_OVERLAP") in ("0", "1"):     # weight-gradient launches next to the data-gradient chain (default on)
            _lib.check(_lib.lib().tdm_set_bwd_overlap(int(os.environ["TDM_BWD_OVERLAP"])), "tdm_set_bwd_overlap")
        self._states = {}
        self._cur = self._state(batch_size)
        # epoch mode (begin_epoch / step_epoch): dataset and permutation resident on the device, the batch gathered inside the step
        self._epoch_data = None
        self._epoch_perm = None
        self._epoch_issued = 0
        self._epoch_base = torch.zeros(1, dtype=torch.long, device=dev)
        self._epoch_graph = None
        self._epoch_key = None
        self._epoch_whole = False
        if broadcast:
            dp.broadcast_params_(self.flat, src=0)   # identical replicas: rank 0's weights everywhere

    # ---- per-batch-size buffers; the optimiser state is shared ----
    def _state(self, B: int) -> "E.TrainState":
        st = self._states.get(B)
        if st is None:
            st = self._states[B] = E.TrainState(self.flat, B, grads=self.grads, m=self.m, v=self.v)
        return st

    @property
    def state(self) -> "E.TrainState":
        """Buffers of the most recent (initially: the constructor's) batch size."""
        return self._cur

    def batch_buffer(self, B: int) -> torch.Tensor:
        """The fixed-address (B,1,28,28) input buffer of batch size B: a batch written here (gather with out=, copy_)
        and passed to step() is read in place by the captured step — no per-step staging copy."""
        return self._state(B).x0

    @property
    def steps_taken(self) -> int:
        return int(self.step_state[0].item())      # (host sync: checkpoints / tests only)

    # ---- DPStepper interface ----
    def grad_buffer(self):
        return self.grads

    def local_loss_and_grad(self, x0, t, noise):
        st = self._cur = self._state(int(x0.shape[0]))
        if t is None and noise is None:
            return E.loss_and_grad_philox(self.flat, st, x0, self.seed, self.rng_state)
        if t is None:
            t = torch.randint(0, timesteps, (x0.shape[0],), device=x0.device)
        if noise is None:
            noise = torch.randn_like(x0)
        return E.loss_and_grad(self.flat, st, x0, noise, t)

    def optimizer_step(self, grad_scale: float) -> None:
        E.adamw_step_dev(self.flat, self.grads, self.m, self.v, self.step_state, self.lr, self.betas, self.eps,
                         self.weight_decay, grad_scale=grad_scale)

    def allreduce_equal_shards(self, g: torch.Tensor) -> float:
        return self._allreduce()

    def _allreduce(self) -> float:
        """The step's collective behind an EAGERLY issued backward.  At world > 1 the library finishes the flat gradient in two
        parts (tdm_set_early_grads, switched on by the constructor): rb2 .. out — 95 % of the bytes — are summed over ranks while
        rb1's launches still run, rb1's 39 KB behind the last launch (dp.allreduce_grads_early_).  TDM_EARLY_GRADS=0: one
        collective behind the whole backward."""
        if self.world == 1 or not self._early:
            return dp.allreduce_grads_(self.grads)
        L = _lib.lib()

        def wait_early(stream) -> bool:
            rc = L.tdm_unet_wait_early_grads(stream.cuda_stream)
            if rc < 0:
                _lib.check(rc, "tdm_unet_wait_early_grads")
            return rc == 1
        return dp.allreduce_grads_early_(self.grads, self._early_off, wait_early)

    # ---- the hipGraph form of step() ----
    @staticmethod
    @contextmanager
    def _one_queue():
        """Captures take the one-queue backward: the forked step replays slower as a graph than the plain one (unet.hip, SideLane)."""
        L = _lib.lib()
        was = L.tdm_get_bwd_overlap()
        _lib.check(L.tdm_set_bwd_overlap(0), "tdm_set_bwd_overlap")
        try:
            yield
        finally:
            _lib.check(L.tdm_set_bwd_overlap(was), "tdm_set_bwd_overlap")

    def _capture(self, st: "E.TrainState") -> None:
        whole = self.world == 1 or dp.graph_collective_ok()      # (collective: every rank reaches this at its second step)
        g = torch.cuda.CUDAGraph()
        with self._one_queue(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            E.loss_and_grad_philox(self.flat, st, st.x0, self.seed, self.rng_state)
            if whole:
                scale = dp.allreduce_grads_(self.grads)
                self.optimizer_step(scale)
        st.graph, st.graph_whole, st.graph_gen = g, whole, (schedule_generation(), float(self.lr), float(self.weight_decay))

    def step(self, x0: Optional[torch.Tensor], t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
             global_batch: Optional[int] = None):
        """One optimisation step on batch x0 (B,1,28,28) already on the device.
        Returns the loss as a 1-element device tensor (no host sync)."""
        if not (self.use_graph and t is None and noise is None and global_batch is None and x0 is not None):
            return super().step(x0, t, noise, global_batch)
        st = self._cur = self._state(int(x0.shape[0]))
        if st.warm < 1:          # first step of a batch size runs eagerly (lazy kernel attributes, allocator warm-up)
            st.warm += 1
            return super().step(x0)
        if x0.data_ptr() != st.x0.data_ptr():
            st.x0.copy_(x0)      # the graph reads a fixed address
        # (set_tables() retires the captured table addresses; lr / weight decay are kernel arguments baked into the graph)
        if st.graph is None or st.graph_gen != (schedule_generation(), float(self.lr), float(self.weight_decay)):
            self._capture(st)
        st.graph.replay()
        if not st.graph_whole:
            self.optimizer_step(dp.allreduce_grads_(self.grads))
        return st.loss


    # ---- epoch mode: the loop `for x, _ in train_loader: step(x)` (src/mnist.py:150-158) with NOTHING issued between two
    #      graph replays — the dataset and the epoch's permutation live on the device, the captured step gathers its own batch
    #      (position = AdamW's device-side step count minus its value at begin_epoch) ----
    def begin_epoch(self, data: torch.Tensor, perm: torch.Tensor) -> None:
        """data (N,1,28,28) fp32 on the device (the same tensor every epoch keeps the captured step), perm (N,) int64: the
        epoch's sample order, identical on every rank.  Iteration k of the epoch trains on positions
        (k * world + rank) * batch_size ... + batch_size of perm (dp.shard_batch_indices)."""
        E._need_cuda(data, perm)
        if perm.numel() != data.shape[0]:
            raise ValueError(f"begin_epoch: perm has {perm.numel()} entries for {data.shape[0]} samples")
        self._epoch_issued = 0                             # host-side count of the epoch's steps (no device sync)
        if self._epoch_perm is None or self._epoch_perm.numel() != perm.numel():
            self._epoch_perm = torch.empty(perm.numel(), dtype=torch.long, device=self.flat.device)
        self._epoch_perm.copy_(perm)                       # fixed address: the captured step reads it
        self._epoch_data = data
        self._epoch_base.copy_(self.step_state[:1])        # device -> device, no host sync

    # steps per replay of the unrolled epoch graph (nothing in an epoch-mode step comes from the host); TDM_EPOCH_UNROLL=1: one per replay
    EPOCH_UNROLL = max(1, int(os.environ.get("TDM_EPOCH_UNROLL", "4")))

    def _epoch_launch(self, st) -> None:
        B = self.batch_size
        E.loss_and_grad_philox_epoch(self.flat, st, self._epoch_data, self._epoch_perm, self.step_state, self._epoch_base,
                                     B * self.world, B * self.rank, self.seed, self.rng_state)

    def _epoch_graphs(self, st):
        """(single-step graph, EPOCH_UNROLL-step graph or None, whole) for the current dataset / schedule / optimiser constants."""
        key = (schedule_generation(), float(self.lr), float(self.weight_decay), self._epoch_data.data_ptr(), self._epoch_perm.data_ptr(),
               int(self._epoch_data.shape[0]), self.batch_size)
        if self._epoch_graph is None or self._epoch_key != key:
            whole = self.world == 1 or dp.graph_collective_ok()
            g1 = torch.cuda.CUDAGraph()
            with self._one_queue(), torch.cuda.graph(g1, capture_error_mode="thread_local"):
                self._epoch_launch(st)
                if whole:
                    self.optimizer_step(dp.allreduce_grads_(self.grads))
            gn = None
            if whole and self.EPOCH_UNROLL > 1:
                # consecutive steps in ONE replay: the ~20 us between two graph launches (the queue's end-of-graph / start-of-
                # graph handshake) is paid once per EPOCH_UNROLL steps
                gn = torch.cuda.CUDAGraph()
                with self._one_queue(), torch.cuda.graph(gn, capture_error_mode="thread_local"):
                    for _ in range(self.EPOCH_UNROLL):
                        self._epoch_launch(st)
                        self.optimizer_step(dp.allreduce_grads_(self.grads))
            self._epoch_graph, self._epoch_key, self._epoch_whole = (g1, gn), key, whole
        return self._epoch_graph[0], self._epoch_graph[1], self._epoch_whole

    def steps_epoch(self, n: int = 1):
        """The next n optimisation steps of the epoch begun with begin_epoch, each on the next whole batch (every rank holds
        batch_size samples; the caller runs ragged tail iterations through step()).  Returns the last step's loss as a
        1-element device tensor (no host sync)."""
        if self._epoch_data is None:
            raise RuntimeError("steps_epoch() before begin_epoch()")
        # whole batches only: past them the gather kernel would clamp its positions and train on repeats of the last sample
        whole = int(self._epoch_data.shape[0]) // (self.batch_size * self.world)
        if n < 0 or self._epoch_issued + n > whole:
            raise ValueError(f"steps_epoch({n}): the epoch has {whole} whole iterations of {self.batch_size} x {self.world} samples, "
                             f"{self._epoch_issued} already issued (run the ragged tail through step())")
        self._epoch_issued += n
        st = self._cur = self._state(self.batch_size)
        done = 0
        while done < n:
            if not self.use_graph or st.warm < 1:          # first step of a batch size eagerly (lazy kernel attributes, allocator warm-up)
                st.warm += 1
                self._epoch_launch(st)
                self.optimizer_step(self._allreduce())
                done += 1
                continue
            g1, gn, whole = self._epoch_graphs(st)
            if gn is not None and n - done >= self.EPOCH_UNROLL:
                gn.replay()
                done += self.EPOCH_UNROLL
                continue
            g1.replay()
            if not whole:
                self.optimizer_step(dp.allreduce_grads_(self.grads))
            done += 1
        return st.loss

    def step_epoch(self):
        """One step of the epoch (steps_epoch(1))."""
        return self.steps_epoch(1)


def train(model: nn.Module,
          device: str,
          epochs: int = 5,
          batch_size: int = 128,
          lr: float = 1e-3,
          ckpt_path: str = "ckpt.pth",
          sample_every_epoch: bool = True,
          samples_per_epoch: int = 25,
          data: Optional[torch.Tensor] = None,
          log_every: int = 50,
          trainer: Optional[DPStepper] = None):
    """src/mnist.py:128-165.  `data`: optional (N,1,28,28) fp32 tensor in [-1,1]
    (default: MNIST IDX files under ./data).  The last batch of an epoch may be
    partial, as with the reference's DataLoader (no drop_last); under data parallelism
    the tail iteration weights every rank's gradient by its share of the global batch
    (DPStepper), and a rank left without samples still joins the step's one collective.
    `trainer`: a DPStepper to drive instead of the HIP DDPMTrainer (CPU tests)."""
    ckpt_path = get_vertex_checkpoint_path("image-model.pth") if "AIP_MODEL_DIR" in os.environ else ckpt_path
    if data is None:
        data = load_mnist_idx("./data")
    rank, world = dp.world_info()
    data = data.to(device)
    n = data.shape[0]
    if trainer is None:
        trainer = DDPMTrainer(model, batch_size, lr=lr)          # the ONLY broadcast of the run happens here
    nb = (n + batch_size * world - 1) // (batch_size * world)
    # (TDM_EPOCH_GATHER=0: gather every batch with a separate launch, as before round 4 — A/B and the equality test)
    epoch_mode = (hasattr(trainer, "begin_epoch") and data.is_cuda and data.dtype == torch.float32 and data.is_contiguous()
                  and os.environ.get("TDM_EPOCH_GATHER", "1") != "0")
    for epoch in range(epochs):
        g = torch.Generator().manual_seed(epoch)          # same shuffle on every rank
        perm = torch.randperm(n, generator=g).to(device)
        if epoch_mode:
            trainer.begin_epoch(data, perm)
        last = None
        it = 0
        while it < nb:
            gb = dp.global_batch_count(n, it, batch_size, world)
            if epoch_mode and gb == batch_size * world:   # whole batches everywhere: the captured step gathers its own batch
                # as many whole-batch iterations as lie before the next log line (and before the ragged tail) in one call
                run = n // (batch_size * world) - it
                if log_every:
                    run = min(run, log_every - it % log_every)
                last = trainer.steps_epoch(run)
                it += run
                if log_every and it % log_every == 0 and rank == 0:
                    print(f"Epoch {epoch + 1}/{epochs} it {it}/{nb} loss={last.item():.4f}", flush=True)
                continue
            it += 1
            idx = dp.shard_batch_indices(perm, it - 1, batch_size, rank, world)
            if idx.numel() == 0:
                x = None
            elif hasattr(trainer, "batch_buffer"):          # gather straight into the captured step's input buffer
                x = torch.index_select(data, 0, idx, out=trainer.batch_buffer(idx.numel()))
            else:
                x = data[idx]
            loss = trainer.step(x, global_batch=None if gb == batch_size * world else gb)
            last = loss if loss is not None else last
            if log_every and it % log_every == 0 and rank == 0 and last is not None:
                print(f"Epoch {epoch + 1}/{epochs} it {it}/{nb} loss={last.item():.4f}", flush=True)
        if rank == 0 and last is not None:
            print(f"Epoch {epoch + 1}/{epochs} done, loss={last.item():.4f}", flush=True)
        if sample_every_epoch and rank == 0:
            sample_images(model, device, epoch + 1, samples_per_epoch)
    if rank == 0:
        save_checkpoint(model.state_dict(), ckpt_path)


def sample(model: nn.Module, device: str, n_samples=25, ckpt_path="ckpt.pth", outdir="samples"):
    """src/mnist.py:183-212."""
    model.load_state_dict(load_checkpoint(ckpt_path, device))
    model.eval()
    samples_dir = get_samples_dir(outdir)
    with torch.no_grad():
        x = torch.randn(n_samples, 1, 28, 28, device=device)
        x = reverse_diffusion(model, x)
        sample_path = _save_grid(x, samples_dir, "samples.png")
        print(f"Saved samples to {sample_path}")


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--train", action="store_true", help="Train the model")
    parser.add_argument("--sample", action="store_true", help="Generate samples")
    parser.add_argument("--epochs", type=int, default=3)
    parser.add_argument("--batch_size", type=int, default=128)
    parser.add_argument("--ckpt", type=str,
                        default=get_vertex_checkpoint_path("image-model.pth") if "AIP_MODEL_DIR" in os.environ else "ckpt.pth")
    # build-only additions (no reference flag renamed)
    parser.add_argument("--synthetic", type=int, default=0, metavar="N",
                        help="train on N synthetic 28x28 images instead of ./data MNIST")
    parser.add_argument("--seed", type=int, default=None)
    args = parser.parse_args(argv)

    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: this build has no CPU path (the reference's CPU path is src/mnist.py)")
    _, _, local_rank = dp.init_from_env("nccl")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if args.seed is not None:
        torch.manual_seed(args.seed)
    E.check_layout_against_library()
    model = SimpleUNet().to(device)

    if args.train:
        data = synthetic_mnist(args.synthetic) if args.synthetic else None
        train(model, device, epochs=args.epochs, batch_size=args.batch_size, ckpt_path=args.ckpt, data=data)
    if args.sample:
        dp.barrier()                                       # (rank 0 wrote the checkpoint)
        if dp.world_info()[0] == 0:                        # the CLI's one 5x5 grid; bench.py shards large sampling batches
            sample(model, device, ckpt_path=args.ckpt)
    if not args.train and not args.sample:
        print("Nothing to do. Pass --train or --sample.")


if __name__ == "__main__":
    main()
