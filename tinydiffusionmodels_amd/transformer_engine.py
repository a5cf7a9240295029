"""Host-side driver of the HIP TinyTransformer denoiser path
(src/shakespeare.py:105-120, :230-236, :343-352): flat parameter layout <->
reference state_dict (native layouts, state_dict order), workspaces, C-ABI calls."""
import ctypes
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _lib
from .schedule import device_tables
from .unet_engine import _need_cuda

FFN = 2048            # nn.TransformerEncoderLayer default dim_feedforward
LN_EPS = 1e-5


def ref_tensors(dim: int, depth: int = 3, ffn: int = FFN):
    """(key, shape) of TinyTransformer.state_dict() in order."""
    out = []
    for l in range(depth):
        p = f"encoder.layers.{l}."
        out += [(p + "self_attn.in_proj_weight", (3 * dim, dim)), (p + "self_attn.in_proj_bias", (3 * dim,)),
                (p + "self_attn.out_proj.weight", (dim, dim)), (p + "self_attn.out_proj.bias", (dim,)),
                (p + "linear1.weight", (ffn, dim)), (p + "linear1.bias", (ffn,)),
                (p + "linear2.weight", (dim, ffn)), (p + "linear2.bias", (dim,)),
                (p + "norm1.weight", (dim,)), (p + "norm1.bias", (dim,)),
                (p + "norm2.weight", (dim,)), (p + "norm2.bias", (dim,))]
    out += [("time_emb.weight", (dim, 1)), ("time_emb.bias", (dim,))]
    return out


def param_offsets(dim: int, depth: int = 3, ffn: int = FFN):
    offs, off = [], 0
    for _, shape in ref_tensors(dim, depth, ffn):
        offs.append(off)
        n = 1
        for s in shape:
            n *= s
        off += n
    offs.append(off)
    return offs


def check_layout_against_library(dim: int, depth: int = 3, ffn: int = FFN):
    n = len(ref_tensors(dim, depth, ffn))
    arr = (ctypes.c_int64 * (n + 1))()
    _lib.check(_lib.lib().tdm_tt_param_offsets(dim, depth, ffn, arr), "tt_param_offsets")
    if list(arr) != param_offsets(dim, depth, ffn):
        raise RuntimeError("transformer flat parameter layout mismatch between Python host and libtdm_hip")


def flat_from_state_dict(sd: Dict[str, torch.Tensor], dim: int, depth: int = 3, ffn: int = FFN, device=None):
    tensors = ref_tensors(dim, depth, ffn)
    keys = [k for k, _ in tensors]
    missing = [k for k in keys if k not in sd]
    unexpected = [k for k in sd if k not in keys]
    if missing or unexpected:
        raise RuntimeError(f"Error(s) in loading state_dict for TinyTransformer: missing {missing}, unexpected {unexpected}")
    parts = []
    for k, shape in tensors:
        v = sd[k].detach().float()
        if tuple(v.shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
        parts.append(v.reshape(-1))
    flat = torch.cat(parts).contiguous()
    return flat.to(device) if device is not None else flat


def state_dict_from_flat(flat: torch.Tensor, dim: int, depth: int = 3, ffn: int = FFN):
    offs = param_offsets(dim, depth, ffn)
    sd = OrderedDict()
    f = flat.detach()
    for i, (k, shape) in enumerate(ref_tensors(dim, depth, ffn)):
        sd[k] = f[offs[i]:offs[i + 1]].view(*shape).clone()
    return sd


class TTConfig:
    def __init__(self, dim: int, n_heads: int = 4, depth: int = 3, ffn: int = FFN):
        self.dim, self.n_heads, self.depth, self.ffn = dim, n_heads, depth, ffn
        self.nparam = param_offsets(dim, depth, ffn)[-1]

    def args(self):
        return (self.dim, self.n_heads, self.depth, self.ffn)


class TTWorkspace:
    def __init__(self, cfg: TTConfig, B: int, L: int, device, training: bool):
        self.cfg, self.B, self.L, self.training = cfg, B, L, training
        n = _lib.lib().tdm_tt_workspace_floats(B, L, cfg.dim, cfg.n_heads, cfg.depth, cfg.ffn, 1 if training else 0)
        if n <= 0:
            raise RuntimeError("tdm_tt_workspace_floats failed")
        self.ws = torch.empty(n, dtype=torch.float32, device=device)


_slabs: Dict[str, torch.Tensor] = {}


_U64 = (1 << 64) - 1


def slabs_for(cfg: TTConfig, device) -> torch.Tensor:
    key = f"{device}:{cfg.dim}:{cfg.depth}:{cfg.ffn}"
    if key not in _slabs:
        n = _lib.lib().tdm_tt_slab_floats(cfg.dim, cfg.depth, cfg.ffn)
        _slabs[key] = torch.empty(n, dtype=torch.float32, device=device)
    return _slabs[key]


def _check_x(cfg, x, t):
    _need_cuda(x, t)
    if x.dim() != 3 or x.shape[2] != cfg.dim:
        raise RuntimeError(f"TinyTransformer expects (B,L,{cfg.dim}) input, got {tuple(x.shape)}")
    if t.shape != (x.shape[0],) or t.dtype != torch.int64:
        raise RuntimeError("t must be an int64 tensor of shape (B,)")
    if x.dtype != torch.float32:
        raise RuntimeError("the HIP transformer path takes fp32 tensors")


def tt_forward(cfg: TTConfig, flat, x, t, ws: TTWorkspace, save: bool, out: Optional[torch.Tensor] = None,
               p_drop: float = 0.0, seed: int = 0):
    """p_drop > 0 = train mode of the reference (dropout masks from (seed, site, index), include/tdm_hip.h)."""
    _check_x(cfg, x, t)
    B, L, D = x.shape
    if ws.B != B or ws.L != L or (save and not ws.training):
        raise RuntimeError("workspace does not match the batch")
    x, tc = x.contiguous(), t.contiguous()
    y = out if out is not None else torch.empty_like(x)
    _lib.check(_lib.lib().tdm_tt_fwd_f32(_lib.ptr(flat), _lib.ptr(x), _lib.ptr(tc), _lib.ptr(y),
                                         _lib.ptr(ws.ws), B, L, D, cfg.n_heads, cfg.depth, cfg.ffn, 1 if save else 0,
                                         float(p_drop), int(seed) & _U64, _lib.stream()), "tt_fwd")
    return y


def tt_backward(cfg: TTConfig, flat, dout, ws: TTWorkspace, grads: Optional[torch.Tensor] = None,
                dx: Optional[torch.Tensor] = None, p_drop: float = 0.0, seed: int = 0):
    _need_cuda(flat, dout)
    B, L, D = dout.shape
    if grads is None:
        grads = torch.empty(cfg.nparam, dtype=torch.float32, device=dout.device)
    doutc = dout.contiguous()
    _lib.check(_lib.lib().tdm_tt_bwd_f32(_lib.ptr(flat), _lib.ptr(doutc), _lib.ptr(grads), _lib.ptr(dx),
                                         _lib.ptr(ws.ws),
                                         _lib.ptr(slabs_for(cfg, dout.device)), B, L, D, cfg.n_heads, cfg.depth, cfg.ffn,
                                         float(p_drop), int(seed) & _U64, _lib.stream()), "tt_bwd")
    return grads


class TTTrainState:
    def __init__(self, cfg: TTConfig, flat: torch.Tensor, B: int, L: int):
        dev = flat.device
        self.cfg, self.B, self.L = cfg, B, L
        self.ws = TTWorkspace(cfg, B, L, dev, training=True)
        self.grads = torch.zeros(cfg.nparam, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.grads)
        self.v = torch.zeros_like(self.grads)
        self.x_noisy = torch.empty(B, L, cfg.dim, dtype=torch.float32, device=dev)
        self.pred = torch.empty_like(self.x_noisy)
        self.dpred = torch.empty_like(self.x_noisy)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step = 0
        # device-drawn step (tdm_tt_loss_grad_philox_f32): fixed-address batch, drawn t / noise; the hipGraph of the step
        self.x0 = torch.empty_like(self.x_noisy)
        self.t = torch.zeros(B, dtype=torch.long, device=dev)
        self.noise = torch.empty_like(self.x_noisy)
        self.graph = None
        self.graph_whole = False
        self.graph_key = None          # (schedule generation, lr, p_drop, dropout seed) the capture was made under
        self.warm = 0


def tt_loss_and_grad_philox(flat, st: TTTrainState, x0, seed: int, rng_state: torch.Tensor, p_drop: float = 0.0, drop_seed: int = 0):
    """tt_loss_and_grad with t ~ U{0..999} and noise ~ N(0,1) drawn on the device (src/shakespeare.py:228-229) from the Philox
    stream (seed, rng_state[0]) into st.t / st.noise, and the step's dropout masks salted with the advanced offset: nothing the
    host writes per step (hipGraph-replayable)."""
    _need_cuda(flat, x0, rng_state)
    cfg = st.cfg
    tabs = device_tables(x0.device)
    B, L, D = x0.shape
    if (B, L) != (st.B, st.L):
        raise RuntimeError("TTTrainState batch mismatch")
    x0 = x0.contiguous()
    _lib.check(_lib.lib().tdm_tt_loss_grad_philox_f32(
        _lib.ptr(flat), _lib.ptr(x0), _lib.ptr(tabs["sqrt_alphas_cumprod"]), _lib.ptr(tabs["sqrt_one_minus_alphas_cumprod"]),
        int(seed) & _U64, _lib.ptr(rng_state), _lib.ptr(st.t), _lib.ptr(st.noise), _lib.ptr(st.x_noisy), _lib.ptr(st.pred),
        _lib.ptr(st.dpred), _lib.ptr(st.loss), _lib.ptr(st.grads), _lib.ptr(st.ws.ws), _lib.ptr(slabs_for(cfg, x0.device)),
        B, L, D, cfg.n_heads, cfg.depth, cfg.ffn, float(p_drop), int(drop_seed) & _U64, _lib.stream()), "tt_loss_grad_philox")
    return st.loss


def tt_loss_and_grad(flat, st: TTTrainState, x0, noise, t, p_drop: float = 0.0, seed: int = 0):
    """Denoiser part of src/shakespeare.py:230-236: q_sample -> forward -> mse -> backward."""
    _need_cuda(flat, x0, noise, t)
    cfg = st.cfg
    tabs = device_tables(x0.device)
    B, L, D = x0.shape
    if (B, L) != (st.B, st.L):
        raise RuntimeError("TTTrainState batch mismatch")
    args = [flat, x0.contiguous(), noise.contiguous(), t.contiguous(), tabs["sqrt_alphas_cumprod"],
            tabs["sqrt_one_minus_alphas_cumprod"], st.x_noisy, st.pred, st.dpred, st.loss, st.grads, st.ws.ws,
            slabs_for(cfg, x0.device)]
    _lib.check(_lib.lib().tdm_tt_loss_grad_f32(*[_lib.ptr(a) for a in args], B, L, D, cfg.n_heads, cfg.depth, cfg.ffn,
                                               float(p_drop), int(seed) & _U64, _lib.stream()), "tt_loss_grad")
    return st.loss


def tt_p_sample_step(cfg: TTConfig, flat, ws: TTWorkspace, x, t_vec, t_index: int, noise, eps_buf, x_out):
    tabs = device_tables(x.device)
    B, L, D = x.shape
    _lib.check(_lib.lib().tdm_tt_p_sample_step_f32(
        _lib.ptr(flat), _lib.ptr(x), _lib.ptr(t_vec), _lib.ptr(noise), _lib.ptr(tabs["sqrt_recip_alphas"]),
        _lib.ptr(tabs["eps_coef"]), _lib.ptr(tabs["sigma"]), int(t_index), _lib.ptr(eps_buf), _lib.ptr(x_out),
        _lib.ptr(ws.ws), B, L, D, cfg.n_heads, cfg.depth, cfg.ffn, _lib.stream()), "tt_p_sample_step")
    return x_out
