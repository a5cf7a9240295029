"""Host-side driver of the HIP SimpleUNet path (src/mnist.py:45-87, :152-160,
:167-180): flat parameter layout <-> reference state_dict, workspace
ownership, and thin wrappers over the C ABI.  PyTorch-ROCm allocates every
buffer; the library only launches kernels on torch's current stream."""
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import ctypes
import torch

from . import _lib
from .schedule import device_tables, TIMESTEPS

NPARAM = 181473
BLOCKS = (("rb1", 1, 32), ("rb2", 32, 64), ("rb3", 64, 64), ("rb4", 96, 32))


def _ref_tensors():
    """(key, reference shape) in the reference's state_dict order."""
    out = []
    for name, ci, co in BLOCKS:
        out += [(f"{name}.conv1.weight", (co, ci, 3, 3)), (f"{name}.conv1.bias", (co,)),
                (f"{name}.conv2.weight", (co, co, 3, 3)), (f"{name}.conv2.bias", (co,)),
                (f"{name}.time_emb.weight", (co, 1)), (f"{name}.time_emb.bias", (co,))]
        if ci != co:
            out += [(f"{name}.skip.weight", (co, ci, 1, 1)), (f"{name}.skip.bias", (co,))]
    out += [("out.weight", (1, 32, 1, 1)), ("out.bias", (1,))]
    return out


REF_TENSORS = _ref_tensors()
REF_KEYS = [k for k, _ in REF_TENSORS]


def param_offsets():
    """Offsets (floats) of each tensor in the flat HWIO buffer; last entry = total."""
    offs, off = [], 0
    for _, shape in REF_TENSORS:
        offs.append(off)
        n = 1
        for s in shape:
            n *= s
        off += n
    offs.append(off)
    assert off == NPARAM
    return offs


_OFFS = param_offsets()


def check_layout_against_library():
    arr = (ctypes.c_int32 * (len(REF_TENSORS) + 1))()
    _lib.check(_lib.lib().tdm_unet_param_offsets(arr), "unet_param_offsets")
    if list(arr) != _OFFS:
        raise RuntimeError("flat parameter layout mismatch between Python host and libtdm_hip")


def flat_from_state_dict(sd: Dict[str, torch.Tensor], device=None, dtype=torch.float32) -> torch.Tensor:
    """Reference OIHW state_dict -> flat HWIO buffer."""
    missing = [k for k in REF_KEYS if k not in sd]
    unexpected = [k for k in sd if k not in REF_KEYS]
    if missing or unexpected:
        raise RuntimeError(f"Error(s) in loading state_dict for SimpleUNet: missing {missing}, unexpected {unexpected}")
    parts = []
    for k, shape in REF_TENSORS:
        v = sd[k].detach().to(dtype=dtype)
        if tuple(v.shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
        if v.dim() == 4:
            v = v.permute(2, 3, 1, 0)   # OIHW -> HWIO
        parts.append(v.reshape(-1))
    flat = torch.cat(parts)
    return flat.to(device).contiguous() if device is not None else flat.contiguous()


def state_dict_from_flat(flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
    """Flat HWIO buffer -> reference-layout tensors (fresh copies)."""
    sd = OrderedDict()
    f = flat.detach()
    for i, (k, shape) in enumerate(REF_TENSORS):
        v = f[_OFFS[i]:_OFFS[i + 1]]
        if len(shape) == 4:
            co, ci, kh, kw = shape
            v = v.view(kh, kw, ci, co).permute(3, 2, 0, 1)
        else:
            v = v.view(*shape)
        sd[k] = v.contiguous().clone()
    return sd


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the HIP DDPM path needs tensors on an MI355X (HIP) device; there is no CPU fallback")


class UNetWorkspace:
    """Activation workspace for one batch size (owned by PyTorch's allocator)."""

    def __init__(self, B: int, device, training: bool):
        L = _lib.lib()
        self.B, self.training = B, training
        n = L.tdm_unet_workspace_floats(B, 1 if training else 0)
        self.ws = torch.empty(n, dtype=torch.float32, device=device)


_slabs: Dict[str, torch.Tensor] = {}


def slabs_for(device) -> torch.Tensor:
    key = str(device)
    if key not in _slabs:
        _slabs[key] = torch.empty(_lib.lib().tdm_unet_slab_floats(), dtype=torch.float32, device=device)
    return _slabs[key]


def unet_forward(flat: torch.Tensor, x: torch.Tensor, t: torch.Tensor, ws: UNetWorkspace, save: bool,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """eps = SimpleUNet(x, t) (src/mnist.py:76-87)."""
    _need_cuda(flat, x, t)
    B = x.shape[0]
    if tuple(x.shape[1:]) != (1, 28, 28):
        raise RuntimeError(f"SimpleUNet expects (B,1,28,28) input, got {tuple(x.shape)}")
    if t.shape != (B,) or t.dtype != torch.int64:
        raise RuntimeError("t must be an int64 tensor of shape (B,)")
    if ws.B != B or (save and not ws.training):
        raise RuntimeError("workspace does not match the batch")
    x, tc = x.contiguous(), t.contiguous()
    if x.dtype != torch.float32 or flat.dtype != torch.float32:
        raise RuntimeError("the HIP UNet path computes in fp32")
    eps = out if out is not None else torch.empty_like(x)
    _lib.check(_lib.lib().tdm_unet_fwd_f32(_lib.ptr(flat), _lib.ptr(x), _lib.ptr(tc), _lib.ptr(eps),
                                           _lib.ptr(ws.ws), B, 1 if save else 0, _lib.stream()), "unet_fwd")
    return eps


def unet_backward(flat: torch.Tensor, x: torch.Tensor, deps: torch.Tensor, ws: UNetWorkspace,
                  grads: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(flat, x, deps)
    B = x.shape[0]
    if grads is None:
        grads = torch.empty(NPARAM, dtype=torch.float32, device=x.device)
    xc, dc = x.contiguous(), deps.contiguous()
    _lib.check(_lib.lib().tdm_unet_bwd_f32(_lib.ptr(flat), _lib.ptr(xc), _lib.ptr(dc),
                                           _lib.ptr(grads), _lib.ptr(ws.ws), _lib.ptr(slabs_for(x.device)), B,
                                           _lib.stream()), "unet_bwd")
    return grads


def get_activation(ws: UNetWorkspace, which: str) -> torch.Tensor:
    """Saved block output as NCHW (tests): 'h1','h2','h3','h4'."""
    idx = {"h1": 0, "h2": 1, "h3": 2, "h4": 3}[which]
    shape = {0: (32, 28, 28), 1: (64, 14, 14), 2: (64, 14, 14), 3: (32, 28, 28)}[idx]
    out = torch.empty((ws.B,) + shape, dtype=torch.float32, device=ws.ws.device)
    _lib.check(_lib.lib().tdm_unet_get_activation(_lib.ptr(ws.ws), ws.B, idx, _lib.ptr(out), _lib.stream()),
               "get_activation")
    return out


class TrainState:
    """Everything one rank needs for the fused DDPM train step
    (src/mnist.py:152-159): persistent buffers, AdamW moments, step count."""

    def __init__(self, flat: torch.Tensor, B: int, grads: Optional[torch.Tensor] = None, m: Optional[torch.Tensor] = None,
                 v: Optional[torch.Tensor] = None):
        dev = flat.device
        self.B = B
        self.ws = UNetWorkspace(B, dev, training=True)
        # flat gradient / AdamW moments: shared between the batch sizes of one trainer when passed in
        self.grads = grads if grads is not None else torch.zeros(NPARAM, dtype=torch.float32, device=dev)
        self.m = m if m is not None else torch.zeros_like(self.grads)
        self.v = v if v is not None else torch.zeros_like(self.grads)
        self.x_noisy = torch.empty(B, 1, 28, 28, dtype=torch.float32, device=dev)
        self.eps = torch.empty_like(self.x_noisy)
        self.deps = torch.empty_like(self.x_noisy)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step = 0
        # device-drawn step (tdm_unet_loss_grad_philox_f32): fixed-address batch, drawn t / noise; hipGraph of the step
        self.x0 = torch.empty_like(self.x_noisy)
        self.t = torch.zeros(B, dtype=torch.long, device=dev)
        self.noise = torch.empty_like(self.x_noisy)
        self.graph = None
        self.graph_whole = False
        self.graph_gen = -1          # schedule.schedule_generation() at capture time
        self.warm = 0


def loss_and_grad(flat: torch.Tensor, st: TrainState, x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor):
    """q_sample -> forward -> mse -> backward; leaves d(loss)/d(params) in
    st.grads and the loss in st.loss (device; no host sync)."""
    _need_cuda(flat, x0, noise, t)
    tabs = device_tables(x0.device)
    B = x0.shape[0]
    if B != st.B:
        raise RuntimeError("TrainState batch mismatch")
    args = [flat, x0.contiguous(), noise.contiguous(), t.contiguous(), tabs["sqrt_alphas_cumprod"],
            tabs["sqrt_one_minus_alphas_cumprod"], st.x_noisy, st.eps, st.deps, st.loss, st.grads, st.ws.ws,
            slabs_for(x0.device)]
    _lib.check(_lib.lib().tdm_unet_loss_grad_f32(*[_lib.ptr(a) for a in args], B, _lib.stream()), "unet_loss_grad")
    return st.loss


def loss_and_grad_philox(flat: torch.Tensor, st: TrainState, x0: torch.Tensor, seed: int, rng_state: torch.Tensor):
    """loss_and_grad with t ~ U{0..999} and noise ~ N(0,1) drawn on the device (src/mnist.py:154-155) from the Philox
    stream (seed, rng_state[0]); the draws land in st.t / st.noise, the offset advances on the device."""
    _need_cuda(flat, x0, rng_state)
    tabs = device_tables(x0.device)
    B = x0.shape[0]
    if B != st.B:
        raise RuntimeError("TrainState batch mismatch")
    x0 = x0.contiguous()
    _lib.check(_lib.lib().tdm_unet_loss_grad_philox_f32(
        _lib.ptr(flat), _lib.ptr(x0), _lib.ptr(tabs["sqrt_alphas_cumprod"]), _lib.ptr(tabs["sqrt_one_minus_alphas_cumprod"]),
        seed, _lib.ptr(rng_state), _lib.ptr(st.t), _lib.ptr(st.noise), _lib.ptr(st.x_noisy), _lib.ptr(st.eps),
        _lib.ptr(st.deps), _lib.ptr(st.loss), _lib.ptr(st.grads), _lib.ptr(st.ws.ws), _lib.ptr(slabs_for(x0.device)), B,
        _lib.stream()), "unet_loss_grad_philox")
    return st.loss


def loss_and_grad_philox_epoch(flat: torch.Tensor, st: TrainState, data: torch.Tensor, perm: torch.Tensor, step_state: torch.Tensor,
                               epoch_base: torch.Tensor, stride: int, offset: int, seed: int, rng_state: torch.Tensor):
    """loss_and_grad_philox with the batch gathered on the fly from the device-resident dataset `data` (N,1,28,28): image b of
    the step is data[perm[(step_state[0] - epoch_base[0]) * stride + offset + b]] (src/mnist.py:150-152's DataLoader batch;
    dp.shard_batch_indices' positions) — no gather launch, no host-written index."""
    _need_cuda(flat, data, perm, step_state, epoch_base, rng_state)
    if data.dtype != torch.float32 or not data.is_contiguous() or data[0].numel() != 784:
        raise RuntimeError("loss_and_grad_philox_epoch: data must be a contiguous fp32 (N,1,28,28) tensor")
    if perm.dtype != torch.int64 or not perm.is_contiguous() or perm.numel() < data.shape[0]:
        raise RuntimeError("loss_and_grad_philox_epoch: perm must be a contiguous int64 tensor with one entry per image")
    tabs = device_tables(data.device)
    _lib.check(_lib.lib().tdm_unet_loss_grad_philox_epoch_f32(
        _lib.ptr(flat), _lib.ptr(data), _lib.ptr(perm), _lib.ptr(step_state), _lib.ptr(epoch_base), int(data.shape[0]), int(stride),
        int(offset), _lib.ptr(tabs["sqrt_alphas_cumprod"]), _lib.ptr(tabs["sqrt_one_minus_alphas_cumprod"]), seed, _lib.ptr(rng_state),
        _lib.ptr(st.t), _lib.ptr(st.noise), _lib.ptr(st.x_noisy), _lib.ptr(st.eps), _lib.ptr(st.deps), _lib.ptr(st.loss),
        _lib.ptr(st.grads), _lib.ptr(st.ws.ws), _lib.ptr(slabs_for(data.device)), st.B, _lib.stream()), "unet_loss_grad_philox_epoch")
    return st.loss


def adamw_step_dev(flat: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step_state: torch.Tensor,
                   lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                   weight_decay: float = 0.01, grad_scale: float = 1.0):
    """adamw_step with the step count in device memory (step_state int64[4] = {steps taken, scratch, beta1^t, beta2^t}):
    performs step steps_taken + 1 and stores it — no host-written scalar, hipGraph-replayable."""
    _need_cuda(flat, grads, m, v, step_state)
    if step_state.numel() < 4 or step_state.dtype != torch.int64:
        raise RuntimeError("adamw_step_dev: step_state must be an int64 tensor of 4 elements")
    _lib.check(_lib.lib().tdm_adamw_flat_devstep_f32(_lib.ptr(flat), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), flat.numel(),
                                                     lr, betas[0], betas[1], eps, weight_decay, _lib.ptr(step_state),
                                                     grad_scale, _lib.stream()), "adamw_devstep")


def adamw_step_devsched(p: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step_state: torch.Tensor,
                        lr_tab: torch.Tensor, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                        weight_decay: float = 0.01, grad_scale: float = 1.0, grad_scale_dev: Optional[torch.Tensor] = None,
                        bump: bool = True):
    """adamw_step_dev with the learning rate (lr_tab[min(steps taken, len - 1)], fp32 device table) and an optional gradient
    factor (grad_scale_dev, 1-element device tensor) read from DEVICE memory: nothing about the step is baked into a captured
    graph except addresses.  bump=False: another tensor of the same optimizer step follows (shared step count)."""
    _need_cuda(p, grads, m, v, step_state, lr_tab)
    if step_state.numel() < 4 or step_state.dtype != torch.int64 or lr_tab.dtype != torch.float32:
        raise RuntimeError("adamw_step_devsched: step_state int64[4], lr_tab float32")
    _lib.check(_lib.lib().tdm_adamw_flat_devsched_f32(_lib.ptr(p), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), p.numel(), _lib.ptr(lr_tab),
                                                      lr_tab.numel(), betas[0], betas[1], eps, weight_decay, _lib.ptr(step_state),
                                                      grad_scale, _lib.ptr(grad_scale_dev), 1 if bump else 0, _lib.stream()),
               "adamw_devsched")


def adamw_step(flat: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
               lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 0.01, grad_scale: float = 1.0):
    """torch.optim.AdamW (torch defaults, src/mnist.py:148) over the flat buffer, in place."""
    _need_cuda(flat, grads, m, v)
    _lib.check(_lib.lib().tdm_adamw_flat_f32(_lib.ptr(flat), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), flat.numel(),
                                             lr, betas[0], betas[1], eps, weight_decay, step, grad_scale,
                                             _lib.stream()), "adamw")


def q_sample_into(x0, t, noise, out):
    tabs = device_tables(x0.device)
    B = x0.shape[0]
    inner = x0.numel() // B
    _lib.check(_lib.lib().tdm_q_sample_f32(_lib.ptr(x0), _lib.ptr(noise), _lib.ptr(t),
                                           _lib.ptr(tabs["sqrt_alphas_cumprod"]),
                                           _lib.ptr(tabs["sqrt_one_minus_alphas_cumprod"]), _lib.ptr(out), B, inner,
                                           _lib.stream()), "q_sample")
    return out


def p_sample_step(flat: torch.Tensor, ws: UNetWorkspace, x: torch.Tensor, t_vec: torch.Tensor, t_index: int,
                  noise: Optional[torch.Tensor], eps_buf: torch.Tensor, x_out: torch.Tensor) -> torch.Tensor:
    """One reverse step for a batch that shares t = t_index (src/mnist.py:191-193, :167-180)."""
    tabs = device_tables(x.device)
    _lib.check(_lib.lib().tdm_unet_p_sample_step_f32(
        _lib.ptr(flat), _lib.ptr(x), _lib.ptr(t_vec), _lib.ptr(noise), _lib.ptr(tabs["sqrt_recip_alphas"]),
        _lib.ptr(tabs["eps_coef"]), _lib.ptr(tabs["sigma"]), int(t_index), _lib.ptr(eps_buf), _lib.ptr(x_out),
        _lib.ptr(ws.ws), x.shape[0], _lib.stream()), "p_sample_step")
    return x_out
