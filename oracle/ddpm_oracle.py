"""CPU oracle for the DDPM train + sample hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain fp32 PyTorch-CPU / numpy) of the
algorithm in LiamConnell/TinyDiffusionModels' `src/mnist.py` and
`src/shakespeare.py`.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product path
(`tinydiffusionmodels_amd/`) never does and fails loudly when its HIP
library is missing.

Parity status: PINNED.  Every function below is checked against the imported
reference (`oracle/make_golden.py`, run in the build container, writes
`tests/golden/*.npz`; `tests/test_oracle_golden.py` replays them).  The
reference's own tests hold no vectors for this path.

Parameter dictionaries use the reference's `state_dict` key names and
layouts (OIHW conv weights etc.) so a reference checkpoint is a valid input.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

TIMESTEPS = 1000  # src/mnist.py:27, src/shakespeare.py:25


# --------------------------------------------------------------------------
# a1. noise schedule  (src/mnist.py:23-33, src/shakespeare.py:27-35)
# --------------------------------------------------------------------------
def linear_beta_schedule(timesteps: int, start: float = 1e-4, end: float = 2e-2) -> torch.Tensor:
    """src/mnist.py:23-25."""
    return torch.linspace(start, end, timesteps)


def make_tables(timesteps: int = TIMESTEPS) -> Dict[str, torch.Tensor]:
    """The five module-level fp32 tables of src/mnist.py:28-33."""
    betas = linear_beta_schedule(timesteps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
    }


def _bview(v: torch.Tensor, ndim: int) -> torch.Tensor:
    return v.view(-1, *([1] * (ndim - 1)))


# --------------------------------------------------------------------------
# a2. q_sample  (src/mnist.py:36-42, src/shakespeare.py:37-44)
# --------------------------------------------------------------------------
def q_sample(x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor,
             tables: Dict[str, torch.Tensor]) -> torch.Tensor:
    a = _bview(tables["sqrt_alphas_cumprod"][t], x_start.dim())
    s = _bview(tables["sqrt_one_minus_alphas_cumprod"][t], x_start.dim())
    return a * x_start + s * noise


# --------------------------------------------------------------------------
# a3/a4. ResidualBlock + SimpleUNet forward  (src/mnist.py:45-87)
# --------------------------------------------------------------------------
UNET_BLOCKS = (("rb1", 1, 32), ("rb2", 32, 64), ("rb3", 64, 64), ("rb4", 96, 32))


def residual_block(p: Dict[str, torch.Tensor], name: str, x: torch.Tensor, that: torch.Tensor,
                   inter: Optional[dict] = None) -> torch.Tensor:
    """src/mnist.py:56-61; `that` is (B,1,1,1) = t/1000."""
    h = F.relu(F.conv2d(x, p[f"{name}.conv1.weight"], p[f"{name}.conv1.bias"], padding=1))
    if inter is not None:
        inter[f"{name}.a1"] = h.detach()
    tb = F.linear(that, p[f"{name}.time_emb.weight"], p[f"{name}.time_emb.bias"]).view(that.shape[0], -1, 1, 1)
    h = h + tb
    h = F.relu(F.conv2d(h, p[f"{name}.conv2.weight"], p[f"{name}.conv2.bias"], padding=1))
    if inter is not None:
        inter[f"{name}.a2"] = h.detach()
    if f"{name}.skip.weight" in p:
        s = F.conv2d(x, p[f"{name}.skip.weight"], p[f"{name}.skip.bias"])
    else:
        s = x
    return h + s


def unet_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor,
                 return_intermediates: bool = False):
    """src/mnist.py:76-87.  x (B,1,28,28) fp32, t (B,) int64 raw step index."""
    that = (t.float() / TIMESTEPS).view(-1, 1, 1, 1)
    inter = {} if return_intermediates else None     # also the post-ReLU tensors rbN.a1 / rbN.a2 (their signs = ReLU masks)
    h1 = residual_block(p, "rb1", x, that, inter)
    h2 = residual_block(p, "rb2", F.avg_pool2d(h1, 2), that, inter)
    h3 = residual_block(p, "rb3", h2, that, inter)
    h4 = F.interpolate(h3, scale_factor=2, mode="nearest")
    h4 = torch.cat([h4, h1], dim=1)
    h4 = residual_block(p, "rb4", h4, that, inter)
    out = F.conv2d(h4, p["out.weight"], p["out.bias"])
    if return_intermediates:
        inter.update({"h1": h1, "h2": h2, "h3": h3, "h4": h4})
        return out, inter
    return out


def unet_init_params(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init parameters with the reference's key layout, drawn from the
    same distributions as torch's default Conv2d/Linear initialisation
    (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
    weights and biases) in the module order of src/mnist.py:65-74.  Used for
    synthetic benchmarks and property tests; golden tests use the weights
    captured from the reference's own `SimpleUNet()` instead."""
    g = torch.Generator().manual_seed(seed)

    def conv(co, ci, k):
        fan_in = ci * k * k
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt5) == U(-1/sqrt(fan_in), ..)
        w = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
        b = (torch.rand(co, generator=g) * 2 - 1) * bound
        return w, b

    def lin(co, ci):
        bound = 1.0 / math.sqrt(ci)
        w = (torch.rand(co, ci, generator=g) * 2 - 1) * bound
        b = (torch.rand(co, generator=g) * 2 - 1) * bound
        return w, b

    p: Dict[str, torch.Tensor] = {}
    for name, ci, co in UNET_BLOCKS:
        p[f"{name}.conv1.weight"], p[f"{name}.conv1.bias"] = conv(co, ci, 3)
        p[f"{name}.conv2.weight"], p[f"{name}.conv2.bias"] = conv(co, co, 3)
        p[f"{name}.time_emb.weight"], p[f"{name}.time_emb.bias"] = lin(co, 1)
        if ci != co:
            p[f"{name}.skip.weight"], p[f"{name}.skip.bias"] = conv(co, ci, 1)
    p["out.weight"], p["out.bias"] = conv(1, 32, 1)
    return p


# --------------------------------------------------------------------------
# a5. train step  (src/mnist.py:152-160, optimizer :148)
# --------------------------------------------------------------------------
def unet_loss_and_grads(p: Dict[str, torch.Tensor], x0: torch.Tensor, t: torch.Tensor,
                        noise: torch.Tensor, tables) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """q_sample -> forward -> mse_loss (mean over B*784) -> backward."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    x_noisy = q_sample(x0, t, noise, tables)
    pred = unet_forward(leaf, x_noisy, t)
    loss = F.mse_loss(pred, noise)
    loss.backward()
    return loss.detach(), {k: v.grad.detach() for k, v in leaf.items()}


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
               lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999,
               eps: float = 1e-8, weight_decay: float = 0.01):
    """One torch.optim.AdamW update (torch defaults as used by
    src/mnist.py:148), written out as torch's single-tensor implementation
    does it.  `step` is 1-based.  Returns (p, m, v) new tensors."""
    p = p * (1.0 - lr * weight_decay)
    m = torch.lerp(m, g, 1.0 - beta1)
    v = v * beta2 + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - step_size * (m / denom)
    return p, m, v


# --------------------------------------------------------------------------
# device-side draws of the build (no reference counterpart: the reference uses torch's host-seeded generator,
# src/mnist.py:154-155,178, whose CPU stream a device generator cannot reproduce — SURVEY.md §7 "RNG parity").
# Restatement of csrc/tdm_philox.h: Philox4x32-10 (Salmon et al. 2011; known answer: counter 0, key 0 ->
# 6627e8d5 e169c58d bc57ac4c 9b00dbd8), counter = (idx_lo, idx_hi | kind << 28, offset_lo, offset_hi).
# --------------------------------------------------------------------------
def philox4x32_10(ctr: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """ctr: (n,4) uint32 counters; returns (n,4) uint32 words."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c = ctr.astype(np.uint64)
    k0, k1 = key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for r in range(10):
        if r > 0:
            k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = np.stack([hi1 ^ c[:, 1] ^ np.uint64(k0), lo1, hi0 ^ c[:, 3] ^ np.uint64(k1), lo0], axis=1)
    return c.astype(np.uint32)


def philox_words(seed: int, offset: int, idx: np.ndarray, kind: int) -> np.ndarray:
    idx = np.asarray(idx, dtype=np.uint64)
    ctr = np.stack([idx & np.uint64(0xFFFFFFFF),
                    ((idx >> np.uint64(32)) & np.uint64(0x0FFFFFFF)) | np.uint64(kind << 28),
                    np.full(idx.shape, offset & 0xFFFFFFFF, dtype=np.uint64),
                    np.full(idx.shape, (offset >> 32) & 0xFFFFFFFF, dtype=np.uint64)], axis=1)
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def philox_steps(seed: int, offset: int, B: int, nsteps: int = TIMESTEPS) -> torch.Tensor:
    """t[b] = mulhi32(word0 of the step stream at counter b, nsteps)  (tdm_philox_step)."""
    w = philox_words(seed, offset, np.arange(B), 1)[:, 0].astype(np.uint64)
    return torch.from_numpy(((w * np.uint64(nsteps)) >> np.uint64(32)).astype(np.int64))


def philox_normals(seed: int, offset: int, n: int) -> torch.Tensor:
    """n (multiple of 4) N(0,1) draws: per float4 index two Box-Muller pairs over u = w * 2^-32 + 2^-33
    (tdm_philox_normal4); fp32 arithmetic, so equal to the device values up to libm rounding."""
    w = philox_words(seed, offset, np.arange(n // 4), 0).astype(np.float32)
    u = w * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)
    r0 = np.sqrt(np.float32(-2.0) * np.log(u[:, 0]))
    r1 = np.sqrt(np.float32(-2.0) * np.log(u[:, 2]))
    a0 = (np.float32(2.0) * u[:, 1]).astype(np.float64) * np.pi
    a1 = (np.float32(2.0) * u[:, 3]).astype(np.float64) * np.pi
    out = np.stack([r0 * np.cos(a0).astype(np.float32), r0 * np.sin(a0).astype(np.float32),
                    r1 * np.cos(a1).astype(np.float32), r1 * np.sin(a1).astype(np.float32)], axis=1)
    return torch.from_numpy(out.reshape(-1).astype(np.float32))


# --------------------------------------------------------------------------
# a6/a7. p_sample and the reverse loop  (src/mnist.py:167-194)
# --------------------------------------------------------------------------
def p_sample_from_eps(x: torch.Tensor, t: torch.Tensor, eps: torch.Tensor,
                      noise: Optional[torch.Tensor], tables) -> torch.Tensor:
    """The arithmetic of src/mnist.py:169-180 given the model output `eps`.
    `noise` is the z the reference draws at :178 (ignored when t[0]==0)."""
    nd = x.dim()
    betas_t = _bview(tables["betas"][t], nd)
    s1m = _bview(tables["sqrt_one_minus_alphas_cumprod"][t], nd)
    # torch.sqrt on CPU (MKL) differs by 1 ulp between hosts: when the tables
    # carry the host-pinned derived values (tests/golden/schedule.npz) use them.
    if "sqrt_recip_alphas" in tables:
        sra = _bview(tables["sqrt_recip_alphas"][t], nd)
        sig = _bview(tables["sigma"][t], nd)
    else:
        sra = _bview(1.0 / torch.sqrt(tables["alphas"][t]), nd)
        sig = torch.sqrt(betas_t)
    mean = sra * (x - betas_t / s1m * eps)
    if int(t[0]) == 0:
        return mean
    return mean + sig * noise


def p_sample(p, x, t, noise, tables):
    return p_sample_from_eps(x, t, unet_forward(p, x, t), noise, tables)


def sample_chain(p, x_T: torch.Tensor, noises: List[torch.Tensor], tables,
                 t_start: int = TIMESTEPS - 1) -> torch.Tensor:
    """src/mnist.py:190-193 with explicit noises; noises[k] is used at step
    t = t_start - k (the one for t=0 is ignored)."""
    x = x_T
    n = x.shape[0]
    for k, i in enumerate(range(t_start, -1, -1)):
        t = torch.full((n,), i, dtype=torch.long)
        x = p_sample(p, x, t, noises[k], tables)
    return x


def to_unit_range(x: torch.Tensor) -> torch.Tensor:
    """src/mnist.py:194."""
    return (x.clamp(-1, 1) + 1) / 2


def to_uint8(x01: torch.Tensor) -> torch.Tensor:
    """uint8 quantisation that `torchvision.utils.save_image` applies to the
    [0,1] grid (`mul(255).add_(0.5).clamp_(0,255).to(uint8)`); torchvision is
    not installed here, so this line is restated from its documented
    behaviour, not run against it (parity unpinned for this single step)."""
    return x01.mul(255).add(0.5).clamp(0, 255).to(torch.uint8)


# --------------------------------------------------------------------------
# a8. TinyTransformer denoiser  (src/shakespeare.py:105-120)
# --------------------------------------------------------------------------
def transformer_init_params(dim: int, depth: int = 3, ffn: int = 2048, seed: int = 0,
                            scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic parameters keyed like the reference TinyTransformer's
    state_dict.  Values come from a splitmix64 integer hash (portable across
    numpy/torch versions), scaled like torch's default init bounds, so tests
    on the GPU box regenerate exactly what make_golden.py fed the reference."""
    p: Dict[str, torch.Tensor] = {}
    ctr = [np.uint64((seed * 0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF)]

    def u(shape, bound):
        n = int(np.prod(shape))
        with np.errstate(over="ignore"):
            idx = np.arange(n, dtype=np.uint64) + ctr[0]
            ctr[0] = ctr[0] + np.uint64(n)
            z = idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        f = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
        arr = ((f * 2.0 - 1.0) * bound * scale).astype(np.float32)
        return torch.from_numpy(arr.reshape(shape))

    for l in range(depth):
        pre = f"encoder.layers.{l}."
        p[pre + "self_attn.in_proj_weight"] = u((3 * dim, dim), math.sqrt(6.0 / (4 * dim)))
        p[pre + "self_attn.in_proj_bias"] = u((3 * dim,), 0.02)
        p[pre + "self_attn.out_proj.weight"] = u((dim, dim), 1.0 / math.sqrt(dim))
        p[pre + "self_attn.out_proj.bias"] = u((dim,), 0.02)
        p[pre + "linear1.weight"] = u((ffn, dim), 1.0 / math.sqrt(dim))
        p[pre + "linear1.bias"] = u((ffn,), 1.0 / math.sqrt(dim))
        p[pre + "linear2.weight"] = u((dim, ffn), 1.0 / math.sqrt(ffn))
        p[pre + "linear2.bias"] = u((dim,), 1.0 / math.sqrt(ffn))
        p[pre + "norm1.weight"] = 1.0 + u((dim,), 0.1)
        p[pre + "norm1.bias"] = u((dim,), 0.1)
        p[pre + "norm2.weight"] = 1.0 + u((dim,), 0.1)
        p[pre + "norm2.bias"] = u((dim,), 0.1)
    p["time_emb.weight"] = u((dim, 1), 1.0)
    p["time_emb.bias"] = u((dim,), 1.0)
    return p


# Dropout (train mode, src/shakespeare.py:106-119, :210).  torch draws dropout masks from its
# generator; the product replaces that stream by a counter-based hash so that forward and
# backward (and this oracle) can regenerate a mask from (seed, site, element index) alone:
#   key  = hash32(seed_lo ^ hash32(seed_hi + 0x9E3779B9 * (site + 1)))
#   u    = hash32((idx_lo ^ key) + 0x9E3779B9 * idx_hi);  keep = u >= round(p * 2^32)
#   y    = x * (keep * (1 / (1 - p)))            -- torch's x * (mask / (1 - p))
# with hash32 = the "lowbias32" finaliser.  Sites, in the order the reference's forward reaches
# them: 0 input dropout; layer l: 1+4l attention probabilities (B,H,L,L), 2+4l dropout1 (B,L,D),
# 3+4l FFN dropout (B,L,ffn), 4+4l dropout2 (B,L,D).
def _hash32(x):
    import numpy as np
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def dropout_keep(p: float, seed: int, site: int, shape, salt=None) -> torch.Tensor:
    """Boolean keep-mask of one dropout site (all True for p = 0).  salt: None = the unsalted mask family; an int = the
    per-step salt of the graph-replayed train step, mixed through the hash (tdm_dropout.h: tdm_salted_key)."""
    import numpy as np
    n = int(np.prod(shape))
    if not p > 0.0:
        return torch.ones(shape, dtype=torch.bool)
    thr = min(4294967295, int(float(np.float32(p)) * 4294967296.0 + 0.5))
    thr = max(thr, 1)
    with np.errstate(over="ignore"):
        seed_lo = np.array([seed & 0xFFFFFFFF], dtype=np.uint32)
        seed_hi = np.array([(seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
        key = _hash32(seed_lo ^ _hash32(seed_hi + np.uint32((0x9E3779B9 * (site + 1)) & 0xFFFFFFFF)))
        if salt is not None:                           # (DropArgs::salt: the device-drawn train step salts with its Philox offset)
            key = _hash32(key ^ np.array([(int(salt) * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint32))
        idx = np.arange(n, dtype=np.uint64)
        lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (idx >> np.uint64(32)).astype(np.uint32)
        u = _hash32((lo ^ key) + hi * np.uint32(0x9E3779B9))
    return torch.from_numpy(u >= np.uint32(thr)).view(*shape)


def _dropout(x: torch.Tensor, p: float, seed: int, site: int, salt=None) -> torch.Tensor:
    if not p > 0.0:
        return x
    import numpy as np
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    return x * (dropout_keep(p, seed, site, tuple(x.shape), salt).to(x.dtype) * scale)


def transformer_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor,
                        n_heads: int = 4, depth: int = 3, eps: float = 1e-5,
                        p_drop: float = 0.0, seed: int = 0, salt=None) -> torch.Tensor:
    """src/shakespeare.py:115-120 with nn.TransformerEncoderLayer's defaults written out:
    post-LN, ReLU FFN, LayerNorm eps 1e-5, no mask, no positional encoding, no final
    norm.  p_drop = 0: eval mode; p_drop > 0: train mode with the hash-defined masks above.
    x (B,L,D) fp32, t (B,) int64."""
    B, L, D = x.shape
    hd = D // n_heads
    ts = (t.float() / TIMESTEPS).unsqueeze(-1)                               # (B,1)
    tb = F.linear(ts, p["time_emb.weight"], p["time_emb.bias"]).unsqueeze(1)  # (B,1,D)
    x = _dropout(x + tb, p_drop, seed, 0, salt)
    for l in range(depth):
        pre = f"encoder.layers.{l}."
        qkv = F.linear(x, p[pre + "self_attn.in_proj_weight"], p[pre + "self_attn.in_proj_bias"])
        q, k, v = qkv.split(D, dim=-1)
        q = q.view(B, L, n_heads, hd).transpose(1, 2)
        k = k.view(B, L, n_heads, hd).transpose(1, 2)
        v = v.view(B, L, n_heads, hd).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        att = _dropout(att, p_drop, seed, 1 + 4 * l, salt)
        o = (att @ v).transpose(1, 2).reshape(B, L, D)
        o = F.linear(o, p[pre + "self_attn.out_proj.weight"], p[pre + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + _dropout(o, p_drop, seed, 2 + 4 * l, salt), (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
        f = F.relu(F.linear(x, p[pre + "linear1.weight"], p[pre + "linear1.bias"]))
        f = F.linear(_dropout(f, p_drop, seed, 3 + 4 * l, salt), p[pre + "linear2.weight"], p[pre + "linear2.bias"])
        x = F.layer_norm(x + _dropout(f, p_drop, seed, 4 + 4 * l, salt), (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
    return x


def transformer_loss_and_grads(p, x0, t, noise, tables, n_heads: int = 4, depth: int = 3,
                               p_drop: float = 0.0, seed: int = 0, want_dx: bool = False, salt=None):
    """Denoiser part of src/shakespeare.py:230-236 (p_drop = 0: the eval-mode network)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    x_noisy = q_sample(x0, t, noise, tables)
    if want_dx:
        x_noisy = x_noisy.detach().requires_grad_(True)
    pred = transformer_forward(leaf, x_noisy, t, n_heads, depth, p_drop=p_drop, seed=seed, salt=salt)
    loss = F.mse_loss(pred, noise)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in leaf.items()}
    if want_dx:
        grads["__dx"] = x_noisy.grad.detach()
    return loss.detach(), grads


def embed(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """LearnedEmbedding.forward (src/shakespeare.py:67): rows of the table."""
    return table[ids]


def rounding_logits(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """LearnedRounding.forward (src/shakespeare.py:101)."""
    return F.linear(x, W, b)


def rounding_ce_and_grads(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, ids: torch.Tensor):
    """Rounding loss of the text train step (src/shakespeare.py:239-240) and its gradients
    w.r.t. the embeddings, the decoder weight and bias."""
    xl, Wl, bl = (v.detach().clone().requires_grad_(True) for v in (x, W, b))
    logits = F.linear(xl, Wl, bl)
    loss = F.cross_entropy(logits.reshape(-1, logits.size(-1)), ids.reshape(-1))
    loss.backward()
    return loss.detach(), xl.grad, Wl.grad, bl.grad


def cosine_decode(x: torch.Tensor, embed_matrix: torch.Tensor) -> torch.Tensor:
    """The cosine-similarity fallback decode of src/shakespeare.py:393-401: normalise both sides (F.normalize,
    eps 1e-12), similarity matrix, argmax over the vocabulary."""
    emb_norm = F.normalize(embed_matrix, dim=1)
    x_norm = F.normalize(x, dim=-1)
    return torch.matmul(x_norm, emb_norm.T).argmax(dim=-1)


def text_p_sample(p, x, t, noise, tables, n_heads: int = 4, depth: int = 3):
    """src/shakespeare.py:343-352."""
    return p_sample_from_eps(x, t, transformer_forward(p, x, t, n_heads, depth), noise, tables)


# --------------------------------------------------------------------------
# helpers shared by tests / bench
# --------------------------------------------------------------------------
def text_full_step_loss_and_grads(p, table, W, b, ids, t, noise, tables, rounding_weight: float, n_heads: int = 4, depth: int = 3,
                                  p_drop: float = 0.0, seed: int = 0, salt=None):
    """The whole text train step's losses and gradients with LEARNED embeddings (src/shakespeare.py:225-243):
    x0 = embedding_fn(ids); x_noisy = q_sample(x0, t, noise); diff = mse(model(x_noisy, t), noise);
    rnd = cross_entropy(rounding_fn(x0), ids); total = diff + rounding_weight * rnd; total.backward().
    Returns (diff, rnd, total, grads of the denoiser parameters, d table, d W, d b)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    tab, Wl, bl = (v.detach().clone().requires_grad_(True) for v in (table, W, b))
    x0 = tab[ids]
    a = tables["sqrt_alphas_cumprod"][t].view(-1, 1, 1)
    s = tables["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1)
    x_noisy = a * x0 + s * noise
    pred = transformer_forward(leaf, x_noisy, t, n_heads, depth, p_drop=p_drop, seed=seed, salt=salt)
    diff = F.mse_loss(pred, noise)
    logits = F.linear(x0, Wl, bl)
    rnd = F.cross_entropy(logits.reshape(-1, logits.size(-1)), ids.reshape(-1))
    total = diff + rounding_weight * rnd
    total.backward()
    return (diff.detach(), rnd.detach(), total.detach(), {k: v.grad for k, v in leaf.items()}, tab.grad, Wl.grad, bl.grad)


def rel_err(a: torch.Tensor, ref: torch.Tensor) -> float:
    """max|a-ref| / max|ref| — the 'rel fp32' measure of BASELINE.json."""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rel_l2(a: torch.Tensor, ref: torch.Tensor) -> float:
    """||a-ref||_2 / ||ref||_2.  Used for gradients of ReLU networks: a single
    ReLU-mask flip (a pre-activation within rounding distance of zero, which even
    the CPU reference does not reproduce run to run under multi-threaded GEMMs)
    moves a few entries by O(1/sqrt(#samples)) in max-norm but barely in L2."""
    a = a.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    return float((a - ref).norm() / ref.norm().clamp_min(1e-30))
