"""MI355X-native embedding-space text diffusion — the Python surface of the
reference's src/shakespeare.py with the denoiser hot path (q_sample,
TinyTransformer forward/backward, p_sample, reverse loop) on hand-written HIP
kernels.  Reference lines are cited per function.

Scope: the transformer DENOISER path is native.  The learned
embedding table and rounding head (src/shakespeare.py:46-102, row N1) keep their
torch parameters; lookup, rounding loss (logits + cross-entropy, fused) and argmax
decode run on native kernels (rounding.hip).  Tokenizer / Gemma / dataset loading
needs network and is out of scope (synthetic vocabularies in tests and benchmarks).
Dropout: train mode applies the reference's 1 + 4*depth dropout sites natively;
the masks come from a counter-based hash of (seed, site, element index) — torch's
Philox stream cannot be replayed by anyone else — with one 64-bit seed per forward
drawn from torch's CPU generator (so `torch.manual_seed` makes runs repeatable)."""
import ctypes
import math
import os
from collections import OrderedDict
from contextlib import contextmanager
from pathlib import Path
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import dp
from . import transformer_engine as TE
from . import unet_engine as E
from .schedule import TIMESTEPS, cpu_tables, device_tables, linear_beta_schedule, schedule_generation  # noqa: F401
from .utils import get_samples_dir, get_vertex_checkpoint_path, load_checkpoint, save_checkpoint, save_samples

T = TIMESTEPS  # number of diffusion steps (src/shakespeare.py:25)
_TB = cpu_tables()
betas = _TB["betas"]
alphas = _TB["alphas"]
alphas_cumprod = _TB["alphas_cumprod"]
sqrt_alphas_cumprod = _TB["sqrt_alphas_cumprod"]
sqrt_one_minus_alphas_cumprod = _TB["sqrt_one_minus_alphas_cumprod"]


def q_sample(x0: torch.Tensor, t: torch.Tensor, noise=None):
    """src/shakespeare.py:37-44 — same fused kernel as the image path, (B,1,1) broadcast."""
    if noise is None:
        noise = torch.randn_like(x0)
    E._need_cuda(x0, t, noise)
    out = torch.empty_like(x0, memory_format=torch.contiguous_format)
    return E.q_sample_into(x0.contiguous(), t.contiguous(), noise.contiguous(), out)


class _EmbedFunction(torch.autograd.Function):
    """x0 = table[ids] on the native gather; backward = native scatter-add (tdm_embed_*_f32)."""

    @staticmethod
    def forward(ctx, table, ids):
        E._need_cuda(table, ids)
        V, D = table.shape
        idc = ids.contiguous()       # out-of-range ids are clamped by the kernel (no host sync here to check them)
        out = torch.empty(*ids.shape, D, dtype=torch.float32, device=table.device)
        tab = table.detach().contiguous()
        _lib.check(_lib.lib().tdm_embed_gather_f32(_lib.ptr(tab), _lib.ptr(idc), _lib.ptr(out), idc.numel(), V, D,
                                                   _lib.stream()), "embed_gather")
        ctx.save_for_backward(idc)
        ctx.shape = (V, D)
        return out

    @staticmethod
    def backward(ctx, g):
        (idc,) = ctx.saved_tensors
        V, D = ctx.shape
        gc = g.contiguous()
        dtab = torch.zeros(V, D, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().tdm_embed_scatter_add_f32(_lib.ptr(gc), _lib.ptr(idc), _lib.ptr(dtab), idc.numel(), V, D, 1.0,
                                                        _lib.stream()), "embed_scatter_add")
        return dtab, None


class LearnedEmbedding(nn.Module):
    """Custom learnable embedding space for diffusion (src/shakespeare.py:46-85).  Same parameter
    (`embeddings.weight`, an nn.Embedding for checkpoint compatibility); on a HIP device the
    lookup and its gradient run on the native gather / scatter-add kernels (row N1)."""

    def __init__(self, vocab_size, embed_dim, pretrained_embeddings=None):
        super().__init__()
        self.vocab_size, self.embed_dim = vocab_size, embed_dim
        self.embeddings = nn.Embedding(vocab_size, embed_dim)
        if pretrained_embeddings is None:
            nn.init.normal_(self.embeddings.weight, mean=0.0, std=0.02)
        elif pretrained_embeddings.size(1) == embed_dim:
            self.embeddings.weight.data.copy_(pretrained_embeddings)
        else:  # random linear projection of the pre-trained matrix to embed_dim (:58-63)
            proj = nn.Linear(pretrained_embeddings.size(1), embed_dim, bias=False).to(pretrained_embeddings.device)
            with torch.no_grad():
                self.embeddings.weight.copy_(proj(pretrained_embeddings))

    def forward(self, token_ids):
        w = self.embeddings.weight
        if w.is_cuda and self.embed_dim % 4 == 0:
            return _EmbedFunction.apply(w, token_ids)
        return self.embeddings(token_ids)

    def get_embedding_matrix(self):
        return self.embeddings.weight


_round_ws = {}


def _round_workspace(M, V, D, device, chunk=0):
    key = (str(device), M, V, D, chunk)
    if key not in _round_ws:
        _round_ws.clear()          # one live workspace: the logits buffer is the big one (M x V floats, or M x chunk)
        n = (_lib.lib().tdm_round_workspace_chunked_floats(M, V, D, chunk) if chunk
             else _lib.lib().tdm_round_workspace_floats(M, V, D))
        if n < 0:
            raise RuntimeError("rounding head: bad workspace request")
        _round_ws[key] = torch.empty(n, dtype=torch.float32, device=device)
    return _round_ws[key]


ROUND_LOGITS_BYTES_MAX = 1 << 30   # above this the (M, V) logits are never held: vocabulary-chunked cross-entropy
ROUND_CHUNK = 8192                 # vocabulary entries per chunk (measured at 32,768 x 50,257: 2048 17.3 ms, 4096 17.0, 8192 16.7, 16384 17.4)


def round_ce_chunk(M: int, V: int) -> int:
    """Vocabulary chunk of the rounding cross-entropy (row N1: "never materialising the (B L, V) logits").  Small problems
    (logits <= ROUND_LOGITS_BYTES_MAX) store the logits once — one GEMM pass fewer; at real vocabulary sizes they are
    recomputed per chunk of ROUND_CHUNK entries (tdm_round_ce_loss_grad_chunked_f32): 9 % slower than storing 6.5 GiB of
    logits at 32,768 x 50,257 (16.7 vs 15.3 ms on the same box), 1.3 GiB of workspace.  TDM_ROUND_CHUNK overrides
    (0 = always store)."""
    env = os.environ.get("TDM_ROUND_CHUNK")
    if env is not None:
        c = int(env)
        return 0 if c <= 0 else max(128, (c + 127) // 128 * 128)
    return 0 if M * V * 4 <= ROUND_LOGITS_BYTES_MAX else ROUND_CHUNK


def round_fused_nseg(M: int, V: int, D: int) -> int:
    """Token segments of the fused rounding head's weight-gradient pass (csrc/ce_chain.hip), or 0 when the fused form does not
    serve the problem (D != 256, TDM_ROUND_FUSED=0).  The pass runs ceil(V / 128) vocabulary tiles x nseg segments on 256 CUs, a
    workgroup streams (M / 32) / nseg token blocks, and nseg > 1 costs nseg slabs of dW written and reduced.  Modelled time
    (measured on MI355X: ~54 us per workgroup + ~1.6 us per token block, slabs at ~5 TB/s):
        rounds(tiles * nseg / 256) * (54 + 1.6 * blocks / nseg) + (nseg > 1) * 2 * nseg * V * D * 4 B / 5 TB/s
    -> 3 segments at 32,768 tokens and V = 50,257 (393 tiles: 5 rounds of 256), 1 at 4,096 tokens (the default batch of 32:
    two rounds of long workgroups beat five rounds of short ones plus the slab reduction: 0.74 -> 0.52 ms)."""
    if os.environ.get("TDM_ROUND_FUSED", "1") == "0" or not _lib.lib().tdm_round_fused_ok(M, V, D):
        return 0
    tiles = (V + 127) // 128
    blocks = (M + 31) // 32
    nmax = max(1, min(6, blocks // 4))       # a segment should hold a few token blocks
    forced = os.environ.get("TDM_ROUND_NSEG")
    if forced:
        return max(1, min(nmax, int(forced)))

    def cost(n):
        return math.ceil(tiles * n / 256) * (54.0 + 1.6 * blocks / n) + (2.0 * n * V * D * 4 / 5e6 if n > 1 else 0.0)

    return min(range(1, nmax + 1), key=lambda n: (cost(n), n))


def round_ce_workspace(M: int, V: int, D: int, device):
    """(form, parameter, workspace) of the rounding cross-entropy at this size: ("fused", nseg, ws) — logits in registers only —
    when D = 256; else ("chunked", chunk, ws) above ROUND_LOGITS_BYTES_MAX of logits, else ("stored", 0, ws)."""
    nseg = round_fused_nseg(M, V, D)
    if nseg:
        n = _lib.lib().tdm_round_workspace_fused_floats(M, V, D, nseg)
        return "fused", nseg, torch.empty(n, dtype=torch.float32, device=device)
    chunk = round_ce_chunk(M, V)
    n = _lib.lib().tdm_round_workspace_chunked_floats(M, V, D, chunk) if chunk else _lib.lib().tdm_round_workspace_floats(M, V, D)
    return ("chunked" if chunk else "stored"), chunk, torch.empty(n, dtype=torch.float32, device=device)


def round_ce_launch(form, param, ws, x, W, b, ids, scale, loss, dx, dW, db, M, V, D):
    L_ = _lib.lib()
    args = (_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(ids), float(scale), _lib.ptr(loss), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db),
            _lib.ptr(ws), M, V, D)
    if form == "fused":
        _lib.check(L_.tdm_round_ce_loss_grad_fused_f32(*args, param, _lib.stream()), "round_ce_loss_grad_fused")
    elif form == "chunked":
        _lib.check(L_.tdm_round_ce_loss_grad_chunked_f32(*args, param, _lib.stream()), "round_ce_loss_grad_chunked")
    else:
        _lib.check(L_.tdm_round_ce_loss_grad_f32(*args, _lib.stream()), "round_ce_loss_grad")


_round_ce_ws = {}


class _RoundCEFunction(torch.autograd.Function):
    """cross_entropy(Linear(D,V)(x), ids) (src/shakespeare.py:239-240) with loss and all three
    gradients from one native call (tdm_round_ce_loss_grad_f32)."""

    @staticmethod
    def forward(ctx, x, W, b, ids):
        E._need_cuda(x, W, b, ids)
        V, D = W.shape
        xc, Wc, bc, idc = x.detach().reshape(-1, D).contiguous(), W.detach().contiguous(), b.detach().contiguous(), \
            ids.reshape(-1).contiguous()
        M = xc.shape[0]
        if idc.numel() != M:
            raise RuntimeError("rounding_cross_entropy: one target id per embedding row expected")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        dW, db = torch.empty_like(Wc), torch.empty_like(bc)
        key = (str(x.device), M, V, D, os.environ.get("TDM_ROUND_FUSED", "1"), os.environ.get("TDM_ROUND_CHUNK"))
        if key not in _round_ce_ws:
            _round_ce_ws.clear()       # one live workspace
            _round_ws.clear()
            _round_ce_ws[key] = round_ce_workspace(M, V, D, x.device)
        form, param, ws = _round_ce_ws[key]
        round_ce_launch(form, param, ws, xc, Wc, bc, idc, 1.0, loss, dx, dW, db, M, V, D)
        ctx.grads = (dx, dW, db)
        ctx.xshape = x.shape
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        dx, dW, db = ctx.grads
        ctx.grads = None
        return (dx * gl).view(ctx.xshape) if dx is not None else None, dW * gl, db * gl, None


class LearnedRounding(nn.Module):
    """Embeddings -> token logits (src/shakespeare.py:88-102); same parameters (`decoder`, an
    nn.Linear).  Row N1: `cross_entropy(x, ids)` is the fused native rounding loss of the train
    step, `argmax(x)` the native decode; `forward` under no_grad computes the logits on the native
    GEMM (with autograd enabled the unfused logits stay a torch op)."""

    def __init__(self, embed_dim, vocab_size):
        super().__init__()
        self.decoder = nn.Linear(embed_dim, vocab_size)

    def _native(self, x):
        return x.is_cuda and self.decoder.weight.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0

    def forward(self, embeddings):
        W, b = self.decoder.weight, self.decoder.bias
        if self._native(embeddings) and not (torch.is_grad_enabled() and (embeddings.requires_grad or W.requires_grad)):
            V, D = W.shape
            xc = embeddings.reshape(-1, D).contiguous()
            M, ld = xc.shape[0], (V + 3) // 4 * 4
            out = torch.empty(M, ld, dtype=torch.float32, device=xc.device)
            Wc, bc = W.detach().contiguous(), b.detach().contiguous()
            _lib.check(_lib.lib().tdm_round_logits_f32(_lib.ptr(xc), _lib.ptr(Wc), _lib.ptr(bc), _lib.ptr(out), ld, M, V, D,
                                                       _lib.stream()), "round_logits")
            return out[:, :V].reshape(*embeddings.shape[:-1], V)
        return self.decoder(embeddings)

    def cross_entropy(self, embeddings, token_ids):
        """F.cross_entropy(self(embeddings).reshape(-1, V), token_ids.reshape(-1)) (src/shakespeare.py:239-240)."""
        if self._native(embeddings):
            return _RoundCEFunction.apply(embeddings, self.decoder.weight, self.decoder.bias, token_ids)
        logits = self.decoder(embeddings)
        return F.cross_entropy(logits.reshape(-1, logits.size(-1)), token_ids.reshape(-1))

    @torch.no_grad()
    def argmax(self, embeddings):
        """self(embeddings).argmax(dim=-1) (src/shakespeare.py:389-390)."""
        if not self._native(embeddings):
            return self.decoder(embeddings).argmax(dim=-1)
        W, b = self.decoder.weight, self.decoder.bias
        V, D = W.shape
        xc = embeddings.reshape(-1, D).contiguous()
        M = xc.shape[0]
        out = torch.empty(M, dtype=torch.int64, device=xc.device)
        Wc, bc = W.detach().contiguous(), b.detach().contiguous()
        ws = _round_workspace(M, V, D, xc.device)
        _lib.check(_lib.lib().tdm_round_argmax_f32(_lib.ptr(xc), _lib.ptr(Wc), _lib.ptr(bc), _lib.ptr(out), _lib.ptr(ws), M, V, D,
                                                   _lib.stream()), "round_argmax")
        return out.view(embeddings.shape[:-1])


class _TTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, flat, cfg, p_drop=0.0, seed=0):
        need_w = bool(ctx.needs_input_grad[2])
        need_x = bool(ctx.needs_input_grad[0])
        save = need_w or need_x
        ws = TE.TTWorkspace(cfg, x.shape[0], x.shape[1], x.device, training=save)
        y = TE.tt_forward(cfg, flat.detach(), x.detach(), t, ws, save=save, p_drop=p_drop, seed=seed)
        if save:
            ctx.ws, ctx.cfg, ctx.need_x, ctx.drop = ws, cfg, need_x, (p_drop, seed)
            ctx.arith = _lib.arithmetic()          # (backward runs on the autograd thread: _lib.use_arithmetic)
            ctx.save_for_backward(flat)
        return y

    @staticmethod
    def backward(ctx, dout):
        (flat,) = ctx.saved_tensors
        dx = torch.empty_like(dout, memory_format=torch.contiguous_format) if ctx.need_x else None
        with _lib.use_arithmetic(ctx.arith):
            grads = TE.tt_backward(ctx.cfg, flat.detach(), dout.contiguous(), ctx.ws, dx=dx, p_drop=ctx.drop[0],
                                   seed=ctx.drop[1])
        ctx.ws = None
        return dx, None, grads, None, None, None


class TinyTransformer(nn.Module):
    """src/shakespeare.py:105-120: x + Linear(1,dim)(t/T) -> Dropout -> `depth`
    post-LN nn.TransformerEncoderLayer(d_model=dim, nhead=n_heads, batch_first=True)
    with torch's defaults (ReLU FFN 2048, LN eps 1e-5), no mask, no positional
    encoding.  All parameters live in one flat fp32 nn.Parameter read directly by
    the HIP kernels; state_dict()/load_state_dict() use the reference's keys."""

    def __init__(self, dim, n_heads=4, depth=3, dropout=0.1):
        super().__init__()
        self.cfg = TE.TTConfig(dim, n_heads, depth, TE.FFN)
        self.p_drop = float(dropout)
        # same construction order as the reference => same RNG stream for the default init
        layer = nn.TransformerEncoderLayer(d_model=dim, nhead=n_heads, batch_first=True, dropout=dropout)
        enc = nn.TransformerEncoder(layer, num_layers=depth)
        te = nn.Linear(1, dim)
        sd = OrderedDict(("encoder." + k, v.detach()) for k, v in enc.state_dict().items())
        sd["time_emb.weight"], sd["time_emb.bias"] = te.weight.detach(), te.bias.detach()
        self.flat = nn.Parameter(TE.flat_from_state_dict(sd, dim, depth, TE.FFN))
        self._infer_ws = None
        self._samplers = {}
        self.dropout_seed = None   # set to an int to force the seed of the next train-mode forwards (tests)
        self.last_dropout_seed = None

    def next_dropout_seed(self) -> int:
        """Seed of one train-mode forward: forced, or 62 bits from torch's CPU generator (no device sync)."""
        seed = self.dropout_seed if self.dropout_seed is not None else int(torch.randint(0, 1 << 62, (1,)).item())
        self.last_dropout_seed = seed
        return seed

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        for k, v in TE.state_dict_from_flat(self.flat, self.cfg.dim, self.cfg.depth, self.cfg.ffn).items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        flat = TE.flat_from_state_dict(state_dict, self.cfg.dim, self.cfg.depth, self.cfg.ffn, device=self.flat.device)
        with torch.no_grad():
            self.flat.copy_(flat)
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def _workspace(self, B, L, device):
        ws = self._infer_ws
        if ws is None or ws.B != B or ws.L != L or ws.ws.device != device:
            ws = self._infer_ws = TE.TTWorkspace(self.cfg, B, L, device, training=False)
        return ws

    def forward(self, x: torch.Tensor, t: torch.Tensor):
        E._need_cuda(x, t, self.flat)
        p_drop = self.p_drop if self.training else 0.0
        seed = self.next_dropout_seed() if p_drop > 0.0 else 0
        if torch.is_grad_enabled() and (self.flat.requires_grad or x.requires_grad):
            return _TTFunction.apply(x, t, self.flat, self.cfg, p_drop, seed)
        return TE.tt_forward(self.cfg, self.flat.detach(), x, t, self._workspace(x.shape[0], x.shape[1], x.device), False,
                             p_drop=p_drop, seed=seed)


class _QSampleFunction(torch.autograd.Function):
    """q_sample (src/shakespeare.py:37-44) on the fused kernel, differentiable in x0: learned embeddings get the
    diffusion-loss gradient d x_noisy / d x0 = sqrt_acp[t] through tdm_scale_by_table_f32 (:225-232)."""

    @staticmethod
    def forward(ctx, x0, t, noise):
        E._need_cuda(x0, t, noise)
        xc, tc, nc = x0.detach().contiguous(), t.contiguous(), noise.contiguous()
        out = torch.empty_like(xc)
        E.q_sample_into(xc, tc, nc, out)
        ctx.save_for_backward(tc)
        return out

    @staticmethod
    def backward(ctx, g):
        (tc,) = ctx.saved_tensors
        gc = g.contiguous()
        dx0 = torch.empty_like(gc)
        B = gc.shape[0]
        _lib.check(_lib.lib().tdm_scale_by_table_f32(_lib.ptr(gc), _lib.ptr(tc), _lib.ptr(device_tables(gc.device)["sqrt_alphas_cumprod"]),
                                                     _lib.ptr(dx0), B, gc.numel() // B, _lib.stream()), "scale_by_table")
        return dx0, None, None


class _MSEFunction(torch.autograd.Function):
    """F.mse_loss(pred, target) (src/shakespeare.py:236) forward + backward in one native pass (tdm_mse_fwd_bwd_f32)."""

    @staticmethod
    def forward(ctx, pred, target):
        E._need_cuda(pred, target)
        pc, tc = pred.detach().contiguous(), target.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pc.device)
        dpred = torch.empty_like(pc)
        scratch = torch.empty(1024, dtype=torch.float32, device=pc.device)
        _lib.check(_lib.lib().tdm_mse_fwd_bwd_f32(_lib.ptr(pc), _lib.ptr(tc), _lib.ptr(loss), _lib.ptr(dpred), _lib.ptr(scratch),
                                                  pc.numel(), _lib.stream()), "mse")
        ctx.dpred = dpred
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        d = ctx.dpred
        ctx.dpred = None
        return d * gl, None


def native_mse_loss(pred, target):
    return _MSEFunction.apply(pred, target)


class NativeAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(params, lr, weight_decay) (src/shakespeare.py:197) with the update done by
    tdm_adamw_flat_f32 — one launch per parameter tensor (the denoiser is ONE flat tensor).  A torch Optimizer
    subclass so the reference's LambdaLR warm-up / cosine schedule drives `param_groups[0]['lr']` unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.is_contiguous():
                    raise RuntimeError("NativeAdamW needs contiguous parameters")
                E.adamw_step(p.data.view(-1), p.grad.contiguous().view(-1), st["m"].view(-1), st["v"].view(-1), st["step"],
                             lr=group["lr"], betas=group["betas"], eps=group["eps"], weight_decay=group["weight_decay"])
        return None


class ByteTokenizer:
    """Offline stand-in for the HF tokenizer the reference loads (src/shakespeare.py:508): UTF-8 bytes as token ids
    (vocab 256).  Implements the two calls the reference makes: `tokenizer(text, ...).input_ids` and `batch_decode`."""
    vocab_size = 256
    bos_token_id = None
    eos_token_id = 0

    class _Enc:
        def __init__(self, ids):
            self.input_ids = ids

    def __call__(self, text, add_special_tokens=False, return_attention_mask=False, return_tensors="pt"):
        ids = torch.tensor(list(text.encode("utf-8")), dtype=torch.long).unsqueeze(0)
        return ByteTokenizer._Enc(ids)

    def batch_decode(self, tokens, skip_special_tokens=True):
        return [bytes(int(t) & 0xFF for t in row).decode("utf-8", errors="replace") for row in tokens.tolist()]


def load_text_dataset(path: Optional[str] = None) -> str:
    """The raw corpus as a single string (src/shakespeare.py:122-125).  The reference downloads
    `tiny_shakespeare` through `datasets`; there is no network here, so the corpus comes from a local text file
    (`path` or $TDM_TEXT_CORPUS)."""
    path = path or os.environ.get("TDM_TEXT_CORPUS")
    if not path or not os.path.exists(path):
        raise FileNotFoundError("load_text_dataset: pass the path of a local text corpus (or set TDM_TEXT_CORPUS); "
                                "the reference's `datasets` download needs network access")
    return Path(path).read_text(encoding="utf-8")


def tokenize_corpus(text: str, tokenizer, seq_len: int, val_split=0.1, generator=None):
    """Tokenize the full corpus once and slice it into fixed-length chunks with a random train / val split
    (src/shakespeare.py:128-156): returns two `torch.utils.data.Subset`s of a (N_chunks, seq_len) long tensor.
    `generator`: the split's RNG (data-parallel runs pass one seeded alike on every rank)."""
    ids = tokenizer(text, add_special_tokens=False, return_attention_mask=False, return_tensors="pt").input_ids.squeeze(0)
    n_chunks = ids.size(0) // seq_len
    chunks = ids[: n_chunks * seq_len].view(n_chunks, seq_len)      # drop the remainder
    n_val = int(n_chunks * val_split)
    return torch.utils.data.random_split(chunks, [n_chunks - n_val, n_val], generator=generator)


def cosine_warmup_lambda(num_warmup_steps, num_training_steps, eta_min=0):
    """The lr_lambda of src/shakespeare.py:161-166 (linear warm-up from 0, then cosine annealing)."""
    def lr_lambda(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(eta_min, 0.5 * (1.0 + math.cos(math.pi * progress)))
    return lr_lambda


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, eta_min=0):
    """Cosine annealing with linear warm-up; lr_lambda(0) = 0 (src/shakespeare.py:159-167)."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, cosine_warmup_lambda(num_warmup_steps, num_training_steps, eta_min))


def lr_schedule_table(base_lr: float, lr_lambda, n_steps: int, device) -> torch.Tensor:
    """lr_tab[i] = the value `param_groups[0]['lr']` holds during optimizer step i + 1 under LambdaLR(lr_lambda) stepped once after
    every optimizer step (src/shakespeare.py:200-202, :250): base_lr * lr_lambda(i), computed with the reference's own Python
    arithmetic and rounded to fp32 exactly as passing the float to the kernel would.  The device-scheduled AdamW
    (tdm_adamw_flat_devsched_f32) indexes it with its device-resident step count, so a captured train step never sees a
    host-written learning rate.  Steps past the table keep its last entry."""
    vals = [base_lr * lr_lambda(i) for i in range(max(1, n_steps))] if lr_lambda is not None else [base_lr]
    return torch.tensor(vals, dtype=torch.float64).to(torch.float32).to(device)


def dynamic_rounding_weight_schedule(epoch, total_epochs, initial_weight=1.0, final_weight=0.1):
    """Linear decay to an ABSOLUTE final weight (src/shakespeare.py:169-172)."""
    progress = epoch / total_epochs
    return initial_weight * (1 - progress) + final_weight * progress


# Issue mode of the text trainers.  Up to 16,384 tokens per batch the transformer backward runs its weight-gradient GEMMs on the
# library's side stream next to the data-gradient chain (tdm_set_bwd_overlap; csrc/transformer.hip) when the step is issued EAGERLY:
# 4.5 / 5.7 / 6.1 % faster than the hipGraph replay at 32 / 64 / 128 sequences of 128 tokens, bit-identical.  The forked step
# replayed as a graph is much slower, so graphs are captured with one queue, and above 16,384 tokens (where the side queue returns
# nothing) the default stays one graph replay per step.  TDM_TRAIN_GRAPH=1 / 0 or `graph=` force either form.
_EAGER_TOKENS = 16384


def _default_use_graph(graph: Optional[bool], tokens: int) -> bool:
    if graph is not None:
        return bool(graph)
    env = os.environ.get("TDM_TRAIN_GRAPH")
    if env in ("0", "1"):
        return env == "1"
    return tokens > _EAGER_TOKENS


@contextmanager
def _one_queue():
    """Captures take the one-queue backward (see above)."""
    L = _lib.lib()
    was = L.tdm_get_bwd_overlap()
    _lib.check(L.tdm_set_bwd_overlap(0), "tdm_set_bwd_overlap")
    try:
        yield
    finally:
        _lib.check(L.tdm_set_bwd_overlap(was), "tdm_set_bwd_overlap")


class DenoiserTrainer:
    """Denoiser part of the text train step (src/shakespeare.py:230-236 + AdamW :197)
    as one fused device-side step on given embeddings x0 (B,L,D): t, noise,
    q_sample, TinyTransformer forward, MSE, backward, (RCCL all-reduce), AdamW.

    With t / noise left to the trainer nothing is written by the host per step — above 16,384 tokens the step is ONE hipGraph
    replay, up to there it is issued eagerly with two launch queues in the backward (see _default_use_graph): the draws come from
    a device-side Philox stream, the dropout masks of a step are the trainer's mask family (one 64-bit seed drawn from torch's generator at
    construction) salted with that step's stream offset, and AdamW's step count lives in device memory — nothing is written
    by the host per step (`graph=False` or TDM_TRAIN_GRAPH=0 issues the same launches eagerly; a changed lr, dropout rate or
    schedule recaptures).  At world > 1 the graph includes the all-reduce when dp.graph_collective_ok().  Explicit t / noise
    (teacher forcing, parity tests) run eagerly through tdm_tt_loss_grad_f32 with host-drawn dropout seeds, as before."""

    def __init__(self, model: TinyTransformer, batch_size: int, seq_len: int, lr: float = 1e-4,
                 weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, graph: Optional[bool] = None):
        self.model, self.lr, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        self.flat = model.flat.detach()
        E._need_cuda(self.flat)
        dev = self.flat.device
        self.state = TE.TTTrainState(model.cfg, self.flat, batch_size, seq_len)
        self.rank, self.world = dp.world_info()
        self.use_graph = _default_use_graph(graph, batch_size * seq_len)
        self.step_state = torch.zeros(4, dtype=torch.long, device=dev)     # {AdamW steps taken, scratch, beta1^t, beta2^t}
        self.rng_state = torch.zeros(2, dtype=torch.long, device=dev)      # {Philox stream offset, scratch}
        # rank-distinct streams, governed by torch.manual_seed: the draw key and the dropout mask family of this trainer
        self.seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) ^ (self.rank * 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)
        self.drop_seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) ^ (self.rank * 0xC2B2AE3D27D4EB4F)) & (2 ** 64 - 1)
        # the learning rate lives in DEVICE memory (a table indexed by the device-resident step count; one entry = a constant
        # lr that step(lr=...) rewrites when it changes): an LR schedule never recaptures the step's hipGraph
        self.lr_tab = torch.tensor([lr], dtype=torch.float32, device=dev)
        self._lr_written = float(lr)
        self._scheduled = False
        dp.broadcast_params_(self.flat, src=0)
        # data parallel, eager issue: a layer's gradient is all-reduced as soon as its weight-gradient launches have retired —
        # last layer first, under the backward of the layers below (tdm_set_early_grads; TDM_EARLY_GRADS=0: one collective behind the step)
        self._early = self.world > 1 and os.environ.get("TDM_EARLY_GRADS", "1") != "0"
        self._parts = []
        if self._early:
            L = _lib.lib()
            _lib.check(L.tdm_set_early_grads(1), "tdm_set_early_grads")
            cfg = model.cfg
            for l in range(cfg.depth - 1, -1, -1):
                b, e = ctypes.c_int64(), ctypes.c_int64()
                _lib.check(L.tdm_tt_layer_grad_range(cfg.dim, cfg.depth, cfg.ffn, l, ctypes.byref(b), ctypes.byref(e)), "tt_layer_grad_range")
                self._parts.append((b.value, e.value, l))

    def _allreduce(self, grads) -> float:
        """SUM of the flat gradient over the ranks; with early gradients one collective per layer, each behind that layer's event of
        the backward just issued (dp.allreduce_grads_parts_), else one behind the step (a captured step records no events)."""
        if not self._early or torch.cuda.is_current_stream_capturing():
            return dp.allreduce_grads_(grads)
        L = _lib.lib()

        def wait_part(stream, layer) -> bool:
            rc = L.tdm_tt_wait_layer_grads(stream.cuda_stream, layer)
            if rc < 0:
                _lib.check(rc, "tdm_tt_wait_layer_grads")
            return rc == 1
        return dp.allreduce_grads_parts_(grads, self._parts, wait_part)

    def set_lr_schedule(self, lr_lambda, n_steps: int) -> None:
        """Per-step schedule of the reference's LambdaLR (src/shakespeare.py:200-202, :250) evaluated once on the host into a
        device table (lr_schedule_table): step i + 1 of this trainer uses base lr * lr_lambda(i), with no host write per step."""
        self.lr_tab = lr_schedule_table(self.lr, lr_lambda, n_steps, self.flat.device)
        self._scheduled = True
        self.state.graph = None                    # (the table's address is baked into a captured step)

    @property
    def steps_taken(self) -> int:
        return int(self.step_state[0].item())      # (host sync: checkpoints / tests only)

    def _p_drop(self) -> float:
        return self.model.p_drop if self.model.training else 0.0   # model.train() -> the reference's dropout

    def _sync_lr(self, lr: float) -> None:
        if not self._scheduled and float(lr) != self._lr_written:   # a changed constant lr: one tiny fill, no recapture
            self.lr_tab.fill_(float(lr))
            self._lr_written = float(lr)

    def _adamw(self, st, scale: float) -> None:
        E.adamw_step_devsched(self.flat, st.grads, st.m, st.v, self.step_state, self.lr_tab, self.betas, self.eps, self.wd,
                              grad_scale=scale)

    def _device_step(self, st, x0, lr: float, whole: bool):
        TE.tt_loss_and_grad_philox(self.flat, st, x0, self.seed, self.rng_state, p_drop=self._p_drop(), drop_seed=self.drop_seed)
        if whole:
            self._adamw(st, self._allreduce(st.grads))

    def step(self, x0, t=None, noise=None, lr: Optional[float] = None):
        st = self.state
        lr = self.lr if lr is None else lr
        if t is not None or noise is not None:        # teacher-forced form (parity tests)
            if t is None:
                t = torch.randint(0, T, (x0.shape[0],), device=x0.device)
            if noise is None:
                noise = torch.randn_like(x0)
            p_drop = self._p_drop()
            seed = self.model.next_dropout_seed() if p_drop > 0.0 else 0
            loss = TE.tt_loss_and_grad(self.flat, st, x0, noise, t, p_drop=p_drop, seed=seed)
            # (the same device-side step count as the graph form: a trainer may mix teacher-forced and device-drawn steps)
            self._sync_lr(lr)
            self._adamw(st, self._allreduce(st.grads))
            return loss
        self._sync_lr(lr)
        if not self.use_graph or st.warm < 1:          # first step eagerly (lazy kernel attributes, allocator warm-up)
            st.warm += 1
            self._device_step(st, x0, lr, True)
            return st.loss
        if x0.data_ptr() != st.x0.data_ptr():
            st.x0.copy_(x0)                            # the graph reads a fixed address
        key = (schedule_generation(), self._p_drop(), self.drop_seed)      # (NOT the lr: it is read from device memory)
        if st.graph is None or st.graph_key != key:
            whole = self.world == 1 or dp.graph_collective_ok()
            g = torch.cuda.CUDAGraph()
            with _one_queue(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._device_step(st, st.x0, lr, whole)
            st.graph, st.graph_whole, st.graph_key = g, whole, key
        st.graph.replay()
        if not st.graph_whole:
            self._adamw(st, dp.allreduce_grads_(st.grads))   # (a replay records no events: one collective behind the graph)
        return st.loss


class _TextStepState:
    """Fixed-address buffers (and the captured hipGraph) of TextTrainStep for one (batch, sequence length)."""

    def __init__(self, cfg, flat, B, L, V, grads, m, v):
        dev = flat.device
        D = cfg.dim
        self.B, self.L = B, L
        self.tt = TE.TTTrainState(cfg, flat, B, L)
        self.tt.grads, self.tt.m, self.tt.v = grads, m, v          # optimizer state is shared between the shapes
        self.ids = torch.zeros(B, L, dtype=torch.long, device=dev)
        self.dxn = torch.empty(B, L, D, device=dev)
        self.dxr = torch.empty(B * L, D, device=dev)
        self.dx0 = torch.empty(B, L, D, device=dev)
        self.rnd_loss = torch.zeros(1, device=dev)
        self.losses = torch.zeros(3, device=dev)
        self.graph, self.graph_key, self.warm = None, None, 0
        self.round_form, self.round_param, self.round_ws = round_ce_workspace(B * L, V, D, dev)


class TextTrainStep:
    """The FULL text train step of src/shakespeare.py:221-250 with learned embeddings, as ONE hipGraph replay per batch:

        x0 = embedding_fn(token_ids)                                    native gather
        t ~ U{0..T-1}, noise ~ N(0,1), x_noisy = q_sample(x0, t, noise)  drawn on the device (Philox), one pass
        diffusion_loss = mse(model(x_noisy, t), noise)                  denoiser forward / backward (+ d loss / d x_noisy)
        rounding_loss = cross_entropy(rounding_fn(x0), token_ids)       fused rounding head (loss, dx, dW, db)
        total = diffusion_loss + rounding_weight * rounding_loss
        backward into the embedding table (scatter-add of sqrt_acp[t] * dx_noisy + rw * dx_round), AdamW over all four tensors
        (denoiser flat vector, embedding table, rounding weight and bias), LR schedule stepped

    Nothing the host writes per step reaches a kernel argument: the draws and dropout salts come from a device-side Philox
    offset, AdamW's step count lives in device memory and indexes a device-resident LEARNING-RATE TABLE (the reference's
    LambdaLR evaluated once on the host, lr_schedule_table), and the rounding weight of the epoch is a device scalar — so the
    graph survives the per-step LR schedule and the per-epoch weight without recapture (`graph=False` / TDM_TRAIN_GRAPH=0
    issue the same launches eagerly, bit for bit).  Under torch.distributed the graph ends before the collectives
    (denoiser / rounding gradients dense, embedding gradient row-wise) and AdamW is issued after them.  Per-epoch loss sums
    accumulate on the device (`acc`); `losses` holds the last step's (diff, rnd, total)."""

    def __init__(self, model: TinyTransformer, rounding_fn: "LearnedRounding", embedding_fn: "LearnedEmbedding", lr: float = 1e-4,
                 weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, rounding_weight: float = 1.0,
                 lr_lambda=None, total_steps: int = 1, graph: Optional[bool] = None, head_first: Optional[bool] = None):
        self.model, self.rounding_fn, self.embedding_fn = model, rounding_fn, embedding_fn
        self.flat = model.flat.detach()
        self.table = embedding_fn.embeddings.weight.detach()
        self.W, self.b = rounding_fn.decoder.weight.detach(), rounding_fn.decoder.bias.detach()
        E._need_cuda(self.flat, self.table, self.W, self.b)
        for t_ in (self.table, self.W, self.b):
            if not t_.is_contiguous() or t_.dtype != torch.float32:
                raise RuntimeError("TextTrainStep: contiguous fp32 parameters expected")
        self.V, self.D = self.W.shape
        if self.table.shape != (self.V, self.D) or self.D != model.cfg.dim or self.D % 4 != 0:
            raise RuntimeError("TextTrainStep: embedding table, rounding head and denoiser must share (V, D), D % 4 == 0")
        dev = self.flat.device
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, eps
        self.rank, self.world = dp.world_info()
        self._graph_pref = graph                          # None: by batch size (_default_use_graph)
        # head_first (default: under torch.distributed): the rounding head runs BEFORE the denoiser — it needs the gathered
        # embeddings only — so that the all-reduce of its gradient (the step's largest buffer) travels under the whole
        # denoiser forward + backward (dp.allreduce_grads_async_); the step is then two launch sequences / graphs
        self.head_first = (self.world > 1) if head_first is None else bool(head_first)
        z = lambda t_: torch.zeros_like(t_)    # noqa: E731
        self.g_flat, self.m_flat, self.v_flat = z(self.flat), z(self.flat), z(self.flat)
        self.g_tab, self.m_tab, self.v_tab = z(self.table), z(self.table), z(self.table)
        self.g_W, self.m_W, self.v_W = z(self.W), z(self.W), z(self.W)
        self.g_b, self.m_b, self.v_b = z(self.b), z(self.b), z(self.b)
        self.step_state = torch.zeros(4, dtype=torch.long, device=dev)
        self.rng_state = torch.zeros(2, dtype=torch.long, device=dev)
        self.seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) ^ (self.rank * 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)
        self.drop_seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) ^ (self.rank * 0xC2B2AE3D27D4EB4F)) & (2 ** 64 - 1)
        self.lr_tab = lr_schedule_table(self.lr, lr_lambda, total_steps, dev)
        self.rw = torch.tensor([float(rounding_weight)], dtype=torch.float32, device=dev)
        self.acc = torch.zeros(4, device=dev)              # running sums of (diff, rnd, total) and the step count
        self._states = {}
        self._cur = None
        self.captures = 0                                   # graphs captured so far (tests: an LR change must not add one)
        for p_ in (self.flat, self.table, self.W, self.b):
            dp.broadcast_params_(p_, src=0)

    def set_rounding_weight(self, w: float) -> None:
        """dynamic_rounding_weight_schedule's value of the epoch (src/shakespeare.py:216): one device write per EPOCH."""
        self.rw.fill_(float(w))

    def epoch_sums(self, reset: bool = True) -> torch.Tensor:
        out = self.acc.clone()
        if reset:
            self.acc.zero_()
        return out

    @property
    def steps_taken(self) -> int:
        return int(self.step_state[0].item())

    @property
    def losses(self) -> torch.Tensor:
        return self._cur.losses

    def _state(self, B: int, L: int) -> _TextStepState:
        st = self._states.get((B, L))
        if st is None:
            st = self._states[(B, L)] = _TextStepState(self.model.cfg, self.flat, B, L, self.V, self.g_flat, self.m_flat, self.v_flat)
        return st

    def _p_drop(self) -> float:
        return self.model.p_drop if self.model.training else 0.0

    def _loss_and_grads(self, st: _TextStepState, part: str = "all") -> None:
        """part: "all" = the whole launch sequence in the reference's order; "head" = gather + rounding head, "body" = the rest
        (head_first: the same launches, the rounding head moved in front of the denoiser it does not depend on)."""
        L_, tt, cfg = _lib.lib(), st.tt, self.model.cfg
        B, L, D, V, M = st.B, st.L, self.D, self.V, st.B * st.L
        tabs = device_tables(self.flat.device)
        stream = _lib.stream()

        def gather():
            _lib.check(L_.tdm_embed_gather_f32(_lib.ptr(self.table), _lib.ptr(st.ids), _lib.ptr(tt.x0), M, V, D, stream), "embed_gather")

        def denoiser():
            _lib.check(L_.tdm_tt_loss_grad_philox_dx_f32(
                _lib.ptr(self.flat), _lib.ptr(tt.x0), _lib.ptr(tabs["sqrt_alphas_cumprod"]), _lib.ptr(tabs["sqrt_one_minus_alphas_cumprod"]),
                self.seed, _lib.ptr(self.rng_state), _lib.ptr(tt.t), _lib.ptr(tt.noise), _lib.ptr(tt.x_noisy), _lib.ptr(tt.pred),
                _lib.ptr(tt.dpred), _lib.ptr(tt.loss), _lib.ptr(self.g_flat), _lib.ptr(st.dxn), _lib.ptr(tt.ws.ws),
                _lib.ptr(TE.slabs_for(cfg, self.flat.device)), B, L, D, cfg.n_heads, cfg.depth, cfg.ffn, float(self._p_drop()),
                self.drop_seed, stream), "tt_loss_grad_philox_dx")

        def rounding():
            round_ce_launch(st.round_form, st.round_param, st.round_ws, tt.x0.view(M, D), self.W, self.b, st.ids.view(-1), 1.0, st.rnd_loss,
                            st.dxr, self.g_W, self.g_b, M, V, D)

        def tail():
            _lib.check(L_.tdm_text_combine_dx0_f32(_lib.ptr(st.dxn), _lib.ptr(tt.t), _lib.ptr(tabs["sqrt_alphas_cumprod"]), _lib.ptr(st.dxr),
                                                   _lib.ptr(self.rw), _lib.ptr(st.dx0), B, L * D, stream), "combine_dx0")
            self.g_tab.zero_()
            _lib.check(L_.tdm_embed_scatter_add_f32(_lib.ptr(st.dx0), _lib.ptr(st.ids), _lib.ptr(self.g_tab), M, V, D, 1.0, stream),
                       "embed_scatter_add")
            _lib.check(L_.tdm_text_loss_f32(_lib.ptr(tt.loss), _lib.ptr(st.rnd_loss), _lib.ptr(self.rw), _lib.ptr(st.losses),
                                            _lib.ptr(self.acc), stream), "text_loss")

        if part == "all":
            gather(); denoiser(); rounding(); tail()
        elif part == "head":
            gather(); rounding()
        else:
            denoiser(); tail()

    def _optimizer_step(self, scale: float = 1.0) -> None:
        kw = dict(betas=self.betas, eps=self.eps, weight_decay=self.wd)
        E.adamw_step_devsched(self.flat, self.g_flat, self.m_flat, self.v_flat, self.step_state, self.lr_tab, grad_scale=scale,
                              bump=False, **kw)
        E.adamw_step_devsched(self.table.view(-1), self.g_tab.view(-1), self.m_tab.view(-1), self.v_tab.view(-1), self.step_state,
                              self.lr_tab, grad_scale=scale, bump=False, **kw)
        # the rounding head's gradients are d rounding_loss: the epoch's weight multiplies them on the way into AdamW
        E.adamw_step_devsched(self.W.view(-1), self.g_W.view(-1), self.m_W.view(-1), self.v_W.view(-1), self.step_state, self.lr_tab,
                              grad_scale=scale, grad_scale_dev=self.rw, bump=False, **kw)
        E.adamw_step_devsched(self.b, self.g_b, self.m_b, self.v_b, self.step_state, self.lr_tab, grad_scale=scale,
                              grad_scale_dev=self.rw, bump=True, **kw)

    def _reduce_head_async(self, weight: float):
        """world > 1, after the "head" launches: the rounding head's gradients start travelling now (weighted by the ragged-tail
        share first, like every gradient before its SUM)."""
        if weight != 1.0:
            self.g_W.mul_(weight); self.g_b.mul_(weight)
        return [dp.allreduce_grads_async_(self.g_W.view(-1)), dp.allreduce_grads_async_(self.g_b)]

    def _sync_and_step(self, ids: torch.Tensor, weight: float, pending=None) -> None:
        """world > 1: average the gradients over the ranks, then AdamW.  weight = B_local * world / B_global (1 for equal shards;
        a ragged tail weights each rank's mean-loss gradient by its share BEFORE the sum, so every rank applies the same
        reduced gradient).  pending: the rounding head's reductions already in flight (_reduce_head_async)."""
        bufs = (self.g_flat, self.g_tab) if pending is not None else (self.g_flat, self.g_tab, self.g_W, self.g_b)
        if weight != 1.0:
            for gbuf in bufs:
                gbuf.mul_(weight)
        for gbuf in ((self.g_flat,) if pending is not None else (self.g_flat, self.g_W, self.g_b)):
            dp.allreduce_grads_(gbuf.view(-1))
        dp.allreduce_rows_(self.g_tab, ids)
        for h in pending or ():
            h.wait()
        self._optimizer_step(1.0 / self.world)

    def step(self, token_ids: torch.Tensor, global_batch: Optional[int] = None) -> torch.Tensor:
        """One optimisation step on token_ids (B, L) int64 (any device; copied into the step's fixed-address buffer).  Returns
        the device tensor (diff, rnd, total) of this step — no host sync."""
        B, L = int(token_ids.shape[0]), int(token_ids.shape[1])
        if B == 0:                                          # a rank without samples still joins the step's collectives
            if self.world == 1:
                return self._cur.losses if self._cur is not None else torch.zeros(3, device=self.flat.device)
            st = self._cur if self._cur is not None else self._state(1, L)
            for gbuf in (self.g_flat, self.g_tab, self.g_W, self.g_b):
                gbuf.zero_()
            # (the same collectives in the same order as the ranks that have samples)
            self._sync_and_step(st.ids[:0], 1.0, self._reduce_head_async(1.0) if self.head_first else None)
            return st.losses
        st = self._cur = self._state(B, L)
        st.ids.copy_(token_ids)
        weight = 1.0 if (global_batch is None or self.world == 1) else B * self.world / float(global_batch)
        whole = self.world == 1
        parts = ("head", "body") if self.head_first else ("all",)
        pending = None
        if not _default_use_graph(self._graph_pref, st.B * st.L) or st.warm < 1:   # first step of a shape eagerly (lazy kernel attributes, allocator warm-up)
            st.warm += 1
            for part in parts:
                self._loss_and_grads(st, part)
                if part == "head" and not whole:
                    pending = self._reduce_head_async(weight)
            if whole:
                self._optimizer_step(1.0)
        else:
            key = (schedule_generation(), self._p_drop(), self.lr_tab.data_ptr(), parts)
            if st.graph is None or st.graph_key != key:
                graphs = []
                for part in parts:
                    g = torch.cuda.CUDAGraph()
                    with _one_queue(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._loss_and_grads(st, part)
                        if whole and part == parts[-1]:
                            self._optimizer_step(1.0)
                    graphs.append(g)
                st.graph, st.graph_key = graphs, key
                self.captures += 1
            for part, g in zip(parts, st.graph):
                g.replay()
                if part == "head" and not whole:
                    pending = self._reduce_head_async(weight)
        if not whole:
            self._sync_and_step(st.ids, weight, pending)
        return st.losses


def p_sample(model, x, t, noise=None):
    """src/shakespeare.py:343-352 (branches on t[0] like the reference)."""
    E._need_cuda(x, t)
    tabs = device_tables(x.device)
    eps = model(x, t)
    add_noise = int(t[0]) != 0
    if add_noise and noise is None:
        noise = torch.randn_like(x)
    B = x.shape[0]
    xc, ec, tc = x.contiguous(), eps.contiguous(), t.contiguous()   # held until the launch
    zc = noise.contiguous() if add_noise else None
    out = torch.empty_like(xc)
    _lib.check(_lib.lib().tdm_p_sample_update_pert_f32(
        _lib.ptr(xc), _lib.ptr(ec), _lib.ptr(zc),
        _lib.ptr(tabs["sqrt_recip_alphas"]), _lib.ptr(tabs["eps_coef"]), _lib.ptr(tabs["sigma"]),
        _lib.ptr(tc), 1 if add_noise else 0, _lib.ptr(out), B, xc.numel() // B, _lib.stream()),
        "p_sample_update")
    return out


class _TextGraphSampler:
    """Persistent buffers + one captured two-step hipGraph of the text reverse loop for a (model, n, L) triple:
    per reverse step ONE C-ABI call (denoiser forward, update with z drawn in registers, t -= 1 on the device)."""

    def __init__(self, model: "TinyTransformer", n: int, L: int, dev):
        from .schedule import device_tables as _dt
        self.model, self.n, self.L, self.dev = model, n, L, dev
        self.tabs = _dt(dev)
        self.sigma0 = self.tabs["sigma"].clone()
        self.sigma0[0] = 0.0                               # t == 0: x = mean (src/shakespeare.py:349-350)
        D = model.cfg.dim
        self.xa = torch.empty(n, L, D, device=dev)
        self.xb = torch.empty_like(self.xa)
        self.eps = torch.empty_like(self.xa)
        self.t_vec = torch.zeros(n, device=dev, dtype=torch.long)
        self.seed = dp.sampler_stream_key()                 # constant key mixed with the rank; run() draws the stream offset
        self.offset0 = 0
        self.rng_state = torch.zeros(2, device=dev, dtype=torch.long)
        self.ws = TE.TTWorkspace(model.cfg, n, L, dev, training=False)
        self.graph = None

    def _one(self, a, b):
        cfg = self.model.cfg
        _lib.check(_lib.lib().tdm_tt_p_sample_step_philox_f32(
            _lib.ptr(self.model.flat.detach()), _lib.ptr(a), _lib.ptr(self.t_vec), _lib.ptr(self.tabs["sqrt_recip_alphas"]),
            _lib.ptr(self.tabs["eps_coef"]), _lib.ptr(self.sigma0), self.seed, _lib.ptr(self.rng_state), _lib.ptr(self.eps),
            _lib.ptr(b), _lib.ptr(self.ws.ws), self.n, self.L, cfg.dim, cfg.n_heads, cfg.depth, cfg.ffn, _lib.stream()),
            "tt_p_sample_step_philox")

    def _capture(self):
        saved = (self.xa.clone(), self.t_vec.clone(), self.rng_state.clone())
        side = torch.cuda.Stream(device=self.dev)          # warm-up off the capture stream, then restore
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._one(self.xa, self.xb)
            self._one(self.xb, self.xa)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.xa.copy_(saved[0]); self.t_vec.copy_(saved[1]); self.rng_state.copy_(saved[2])
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._one(self.xa, self.xb)
            self._one(self.xb, self.xa)

    def run(self, x, nsteps, t_start):
        self.xa.copy_(x)
        self.t_vec.fill_(t_start)
        self.offset0 = dp.draw_stream_offset()             # torch.manual_seed governs the chain's noise (see mnist._GraphSampler)
        self.rng_state.copy_(torch.tensor([self.offset0, 0], dtype=torch.long))
        left = nsteps
        if left % 2 == 1:
            self._one(self.xa, self.xb)
            self.xa.copy_(self.xb)
            left -= 1
        if left > 0:
            if self.graph is None:
                self._capture()
            for _ in range(left // 2):
                self.graph.replay()
        return self.xa.clone()


@torch.no_grad()
def reverse_diffusion(model: TinyTransformer, x: torch.Tensor, noises=None, t_start: int = T - 1,
                      use_graph: Optional[bool] = None) -> torch.Tensor:
    """`for i in reversed(range(T)): x = p_sample(model, x, full(i))` (src/shakespeare.py:382-385,
    :420-425) without per-step host syncs.  Default for chains of >= 16 steps with device-drawn noise: a
    two-step hipGraph replayed (t_start + 1) / 2 times (step index and Philox offset in device memory).
    `noises` (teacher forcing; noises[k] used at t = t_start - k) runs the eager loop."""
    E._need_cuda(x)
    n, L, _ = x.shape
    dev = x.device
    nsteps = t_start + 1
    if use_graph is None:
        use_graph = noises is None and nsteps >= 16
    if use_graph and noises is None:
        key = (n, L, str(dev), model.flat.data_ptr(), schedule_generation())
        if key not in model._samplers:
            model._samplers.clear()
            model._samplers[key] = _TextGraphSampler(model, n, L, dev)
        return model._samplers[key].run(x.contiguous(), nsteps, t_start)
    flat = model.flat.detach()
    ws = model._workspace(n, L, dev)
    eps = torch.empty_like(x)
    cur, nxt = x.contiguous().clone(), torch.empty_like(x)
    t_all = torch.arange(t_start, -1, -1, device=dev, dtype=torch.long).view(-1, 1).expand(-1, n).contiguous()
    for k, i in enumerate(range(t_start, -1, -1)):
        z = None
        if i > 0:
            z = noises[k].contiguous() if noises is not None else torch.randn_like(cur)
        TE.tt_p_sample_step(model.cfg, flat, ws, cur, t_all[k], i, z, eps, nxt)
        cur, nxt = nxt, cur
    return cur


def sample_diffusion_embeddings(model, embed_dim, device, n, seq_len):
    """Generate *pure* embeddings z using only the diffusion model (src/shakespeare.py:418-426)."""
    x = torch.randn(n, seq_len, embed_dim, device=device)
    model.eval()
    return reverse_diffusion(model, x)


def decode_tokens(x, rounding_fn, embedding_fn, use_learned_rounding=True, use_learned_embeddings=True):
    """Token ids from final embeddings: rounding-head argmax, or the cosine-similarity
    fallback (src/shakespeare.py:387-401).  Row N1: both native."""
    if use_learned_rounding:
        return rounding_fn.argmax(x) if hasattr(rounding_fn, "argmax") else rounding_fn(x).argmax(dim=-1)
    embed_matrix = embedding_fn.get_embedding_matrix() if use_learned_embeddings else embedding_fn
    return cosine_argmax(x, embed_matrix)


@torch.no_grad()
def cosine_argmax(x: torch.Tensor, embed_matrix: torch.Tensor) -> torch.Tensor:
    """argmax_v cos(x, E[v]) = (F.normalize(x, dim=2) @ F.normalize(E, dim=1).T).argmax(-1)  (src/shakespeare.py:393-401),
    native: two row-normalisation kernels, the MFMA logits GEMM, the row argmax (tdm_cosine_argmax_f32)."""
    E._need_cuda(x, embed_matrix)
    V, D = embed_matrix.shape
    if D % 4 != 0:
        raise RuntimeError("cosine_argmax: the embedding dimension must be a multiple of 4")
    xc = x.detach().reshape(-1, D).contiguous().float()
    Ec = embed_matrix.detach().contiguous().float()
    M = xc.shape[0]
    out = torch.empty(M, dtype=torch.int64, device=xc.device)
    ws = _round_workspace(M, V, D, xc.device)
    _lib.check(_lib.lib().tdm_cosine_argmax_f32(_lib.ptr(xc), _lib.ptr(Ec), _lib.ptr(out), _lib.ptr(ws), M, V, D, _lib.stream()),
               "cosine_argmax")
    return out.view(x.shape[:-1])


def sample(model, rounding_fn, embedding_fn, tokenizer, device, n_samples=4, seq_len=128,
           use_learned_rounding=True, use_learned_embeddings=True, embed_dim=None, index_offset=0):
    """src/shakespeare.py:355-415.  index_offset: first sample number of this call (`sample_{i}.txt`): data-parallel
    sampling shards the n chains over the ranks with no collective, each rank writing its own file numbers."""
    model.eval()
    rounding_fn.eval()
    if use_learned_embeddings:
        embedding_fn.eval()
    samples_dir = get_samples_dir("samples")
    with torch.no_grad():
        if embed_dim is None:
            embed_dim = embedding_fn.embed_dim if use_learned_embeddings else embedding_fn.shape[1]
        x = torch.randn(n_samples, seq_len, embed_dim, device=device)
        x = reverse_diffusion(model, x)
        tokens = decode_tokens(x, rounding_fn, embedding_fn, use_learned_rounding, use_learned_embeddings)
        texts = tokenizer.batch_decode(tokens, skip_special_tokens=True)
        for i, text in enumerate(texts, start=index_offset):
            print(text)
            if isinstance(samples_dir, str) and samples_dir.startswith("gs://"):
                sample_path = f"{samples_dir}/sample_{i}.txt"
            else:
                sample_path = Path(samples_dir) / f"sample_{i}.txt"
            save_samples(text, sample_path)
            print(f"✔ Wrote {sample_path}")
        return texts


def train(model, rounding_fn, embedding_fn, data_loader, val_loader, device, ckpt_path="text_ckpt.pth", epochs=1,
          lr=1e-4, weight_decay=1e-4, rounding_weight=1.0, use_learned_embeddings=True, patience=5,
          use_lr_scheduling=True, warmup_steps=100, losses_fn=None, optimizer_cls=None):
    """src/shakespeare.py:174-341 with the same control flow (cosine-warm-up LR,
    decaying rounding weight, validation pass, early stopping, `_best.pth`, final
    checkpoint dict).  Every per-step tensor op is native: embedding gather / scatter-add, q_sample and its
    x0-gradient, denoiser forward / backward, MSE, fused rounding cross-entropy, AdamW (NativeAdamW) — autograd only
    chains the bridges.

    Under torch.distributed (one process per GPU, each with its own shard of the loader) the replicas start from rank 0's
    weights and every step averages the gradients: dense all-reduces for the denoiser and the rounding head, a ROW-WISE
    one for the embedding table (dp.allreduce_rows_: only the batch's token rows move).  The CONTROL flow is replicated
    too: every rank must run the same number of iterations (checked up front), the validation sums are all-reduced so
    `best_val_loss`, the patience counter and the early-stop `break` are the same decision everywhere, and only rank 0
    writes checkpoints.  A loader with `global_batch(it)` (dp.ShardedBatches) may hand a rank a short or empty batch in
    the ragged tail: the rank's gradient is weighted by B_local / B_global (zeros when empty) and it still joins the
    step's collectives.

    losses_fn / optimizer_cls (tests of the host control flow on CPU): replace the native `losses(token_ids, rw) ->
    (diff, rnd, total)` closure and NativeAdamW."""
    params = list(model.parameters()) + list(rounding_fn.parameters())
    if use_learned_embeddings:
        params += list(embedding_fn.parameters())
    rank, world = dp.world_info()
    emb_w = embedding_fn.embeddings.weight if (use_learned_embeddings and hasattr(embedding_fn, "embeddings")) else None
    if world > 1:
        for p_ in params:
            dp.broadcast_params_(p_.data, src=0)
        lens = torch.tensor([len(data_loader), -len(data_loader), len(val_loader), -len(val_loader)], dtype=torch.int64,
                            device=device)
        dp.allreduce_host_(lens, "min")
        if int(lens[0]) != -int(lens[1]) or int(lens[2]) != -int(lens[3]):
            raise RuntimeError("text train(): the ranks' loaders differ in length (train "
                               f"{int(lens[0])}..{-int(lens[1])}, val {int(lens[2])}..{-int(lens[3])} iterations): a rank "
                               "that runs out of batches would leave the others blocked in the gradient all-reduce — shard "
                               "with dp.ShardedBatches")

    def sync_grads(token_ids, weight):
        """Average the gradients over the ranks (world > 1); weight = this rank's share factor B_local * world / B_global
        (1 for equal shards, 0 for a rank without samples)."""
        for p_ in params:
            if p_.grad is None:
                p_.grad = torch.zeros_like(p_)             # a rank whose batch was empty still joins every collective
            if weight != 1.0:
                p_.grad.mul_(weight)
            if p_ is emb_w:
                dp.allreduce_rows_(p_.grad, token_ids)
            else:
                dp.allreduce_grads_(p_.grad.view(-1))
            p_.grad.mul_(1.0 / world)
    total_steps = len(data_loader) * epochs
    # The train loop's step as ONE hipGraph replay (TextTrainStep: device-drawn t / noise, device-resident LR table and rounding
    # weight, AdamW over all four tensors) whenever every piece is the native one; otherwise (pre-trained embedding matrix,
    # injected loss closure / optimizer of the host-logic tests, TDM_TEXT_STEP=eager) the autograd-bridge form below.
    fused = (losses_fn is None and optimizer_cls is None and use_learned_embeddings and isinstance(model, TinyTransformer)
             and isinstance(rounding_fn, LearnedRounding) and isinstance(embedding_fn, LearnedEmbedding)
             and model.flat.is_cuda and embedding_fn.embeddings.weight.is_cuda and rounding_fn.decoder.weight.is_cuda
             and model.cfg.dim % 4 == 0 and embedding_fn.embeddings.weight.shape[1] == model.cfg.dim
             and os.environ.get("TDM_TEXT_STEP", "graph") != "eager")
    stepper = None
    if fused:
        stepper = TextTrainStep(model, rounding_fn, embedding_fn, lr=lr, weight_decay=weight_decay, rounding_weight=rounding_weight,
                                lr_lambda=cosine_warmup_lambda(warmup_steps, total_steps) if use_lr_scheduling else None,
                                total_steps=total_steps)
        optim = scheduler = None
    else:
        optim = (optimizer_cls or NativeAdamW)(params, lr=lr, weight_decay=weight_decay)   # torch.optim.AdamW's update on tdm_adamw_flat_f32
        scheduler = get_cosine_schedule_with_warmup(optim, warmup_steps, total_steps) if use_lr_scheduling else None
    best_val_loss, patience_counter = float("inf"), 0
    global_batch = getattr(data_loader, "global_batch", None)

    def losses(token_ids, rw):
        x0 = embedding_fn(token_ids) if use_learned_embeddings else embedding_fn[token_ids]
        t = torch.randint(0, T, (x0.shape[0],), device=device).long()
        noise = torch.randn_like(x0)
        # q_sample with gradient to a learned x0 (d x_noisy / d x0 = sqrt_acp[t]): native forward and backward
        x_noisy = _QSampleFunction.apply(x0, t, noise) if x0.requires_grad else q_sample(x0, t, noise)
        diff = native_mse_loss(model(x_noisy, t), noise)
        rnd = rounding_fn.cross_entropy(x0, token_ids)     # logits + cross-entropy, fused native call (row N1)
        return diff, rnd, diff + rw * rnd
    losses = losses_fn or losses

    for epoch in range(epochs):
        model.train(); rounding_fn.train()
        if use_learned_embeddings:
            embedding_fn.train()
        rw = dynamic_rounding_weight_schedule(epoch, epochs, rounding_weight)
        tr = torch.zeros(4, device=device)                 # sums of the per-batch (diff, rnd, total) and the batch count
        if stepper is not None:
            stepper.set_rounding_weight(rw)
            for it, token_ids in enumerate(data_loader):
                gb = None
                if world > 1:
                    gb = global_batch(it) if global_batch is not None else _global_count(int(token_ids.shape[0]), device)
                stepper.step(token_ids, global_batch=gb)
            tr = stepper.epoch_sums()
        for it, token_ids in enumerate(data_loader if stepper is None else ()):
            token_ids = token_ids.to(device)
            b_local = int(token_ids.shape[0])
            optim.zero_grad()
            if b_local:
                diff, rnd, total = losses(token_ids, rw)
                total.backward()
                tr += torch.stack([diff.detach(), rnd.detach(), total.detach(), torch.ones_like(diff.detach())])
            if world > 1:
                # (a loader without global_batch(): the ranks' batch sizes are summed — an empty or ragged shard must not be
                #  weighted as if all ranks held equal shares)
                gb = global_batch(it) if global_batch is not None else _global_count(b_local, device)
                sync_grads(token_ids, b_local * world / float(max(gb, 1)))
            elif not b_local:
                continue
            optim.step()
            if scheduler is not None:
                scheduler.step()
        model.eval(); rounding_fn.eval()
        if use_learned_embeddings:
            embedding_fn.eval()
        va = torch.zeros(4, device=device)
        with torch.no_grad():
            for token_ids in val_loader:
                if token_ids.shape[0] == 0:
                    continue
                diff, rnd, total = losses(token_ids.to(device), rw)
                va += torch.stack([diff, rnd, total, torch.ones_like(diff)])
        # the reference averages per-BATCH means over the batches (src/shakespeare.py:252-255, :291-298); across ranks the
        # sums and the batch counts are reduced first, so every rank holds the same numbers and takes the same branch below
        dp.allreduce_host_(tr); dp.allreduce_host_(va)
        tr = (tr[:3] / tr[3].clamp_min(1.0)).tolist()
        va = (va[:3] / va[3].clamp_min(1.0)).tolist()
        if rank == 0:
            print(f"Epoch {epoch + 1}/{epochs}:")
            print(f"  Train: diff={tr[0]:.4f}, round={tr[1]:.4f}, total={tr[2]:.4f}")
            print(f"  Val:   diff={va[0]:.4f}, round={va[1]:.4f}, total={va[2]:.4f}")
            print(f"  Rounding weight: {rw:.3f}")
        if va[2] < best_val_loss:
            best_val_loss, patience_counter = va[2], 0
            if rank == 0:
                best_ckpt_path = ckpt_path.replace(".pth", "_best.pth")
                checkpoint = {"diffusion_model": model.state_dict(), "rounding_fn": rounding_fn.state_dict(),
                              "epoch": epoch, "val_loss": best_val_loss}
                if use_learned_embeddings:
                    checkpoint["embedding_fn"] = embedding_fn.state_dict()
                save_checkpoint(checkpoint, best_ckpt_path)
                print(f"  New best validation loss! Saved to {best_ckpt_path}")
        else:
            patience_counter += 1
            if patience_counter >= patience:
                if rank == 0:
                    print(f"  Early stopping triggered after {patience} epochs without improvement")
                break
    if rank != 0:
        return
    final_ckpt_path = get_vertex_checkpoint_path("text-model.pth") if "AIP_MODEL_DIR" in os.environ else ckpt_path
    print(f"✔ Saving final checkpoint to {final_ckpt_path}...")
    final_checkpoint = {"diffusion_model": model.state_dict(), "rounding_fn": rounding_fn.state_dict(),
                        "epoch": epochs, "final_training": True}
    if use_learned_embeddings:
        final_checkpoint["embedding_fn"] = embedding_fn.state_dict()
    save_checkpoint(final_checkpoint, final_ckpt_path)


def _global_count(b_local: int, device) -> int:
    """Samples all ranks hold together in this iteration (one small SUM all-reduce; loaders without global_batch())."""
    n = torch.tensor([b_local], dtype=torch.int64, device=device)
    dp.allreduce_host_(n, "sum")
    return int(n.item())


def load_text_checkpoint(ckpt, model, rounding_fn, embedding_fn=None):
    """New dict format or the old raw-state_dict format (src/shakespeare.py:543-562)."""
    if isinstance(ckpt, dict) and "diffusion_model" in ckpt:
        model.load_state_dict(ckpt["diffusion_model"])
        if "rounding_fn" in ckpt:
            rounding_fn.load_state_dict(ckpt["rounding_fn"])
        if embedding_fn is not None and "embedding_fn" in ckpt:
            embedding_fn.load_state_dict(ckpt["embedding_fn"])
    else:
        model.load_state_dict(ckpt)


def main(argv=None):
    """CLI of src/shakespeare.py:473-600 (`python -m src.shakespeare --train | --sample`), same flags.
    The reference always loads an HF tokenizer + causal LM (`--model_id`) for the vocabulary and the pre-trained
    embedding matrix; with no network those are unavailable here, so `--byte_tokenizer` selects the offline
    stand-in (UTF-8 bytes, vocab 256, learned embeddings) and `--corpus PATH` (or $TDM_TEXT_CORPUS) the local
    text file `--train` reads.  `--guided_sample` needs the causal LM and is out of scope."""
    import argparse
    from torch.utils.data import DataLoader
    parser = argparse.ArgumentParser()
    parser.add_argument("--train", action="store_true")
    parser.add_argument("--sample", action="store_true", help="plain diffusion sample")
    parser.add_argument("--guided_sample", action="store_true", help="AR + diffusion guidance (not available offline)")
    parser.add_argument("--epochs", type=int, default=1)
    parser.add_argument("--batch_size", type=int, default=32)
    parser.add_argument("--seq_len", type=int, default=64)
    parser.add_argument("--ckpt", type=str, default="gs://text-diffusion/diffusion/outputs/model/text-model.pth"
                        if "AIP_MODEL_DIR" in os.environ else "text_ckpt.pth")
    parser.add_argument("--model_id", type=str, default="google/gemma-2b-it")
    parser.add_argument("--n", type=int, default=10)
    parser.add_argument("--alpha", type=float, default=0.3)
    parser.add_argument("--rounding_weight", type=float, default=1.0, help="Weight for learned rounding loss")
    parser.add_argument("--use_cosine_fallback", action="store_true", help="Use cosine similarity instead of learned rounding")
    parser.add_argument("--use_learned_embeddings", action="store_true", help="Use custom learned embedding space")
    parser.add_argument("--embed_dim", type=int, default=None, help="Custom embedding dimension")
    parser.add_argument("--init_from_pretrained", action="store_true", help="Initialize learned embeddings from pre-trained weights")
    parser.add_argument("--dropout", type=float, default=0.1, help="Dropout rate for regularization")
    parser.add_argument("--weight_decay", type=float, default=1e-4, help="Weight decay for regularization")
    parser.add_argument("--patience", type=int, default=5, help="Early stopping patience")
    parser.add_argument("--use_lr_scheduling", action="store_true", default=True, help="Use cosine learning rate scheduling")
    parser.add_argument("--warmup_steps", type=int, default=100, help="Number of warmup steps for learning rate scheduling")
    parser.add_argument("--val_split", type=float, default=0.1, help="Fraction of data for validation")
    parser.add_argument("--lr", type=float, default=1e-4, help="Learning rate")
    parser.add_argument("--byte_tokenizer", action="store_true", help="[build] offline UTF-8 byte vocabulary instead of --model_id")
    parser.add_argument("--corpus", type=str, default=None, help="[build] local text corpus for --train")
    parser.add_argument("--seed", type=int, default=None, help="[build] torch.manual_seed")
    args = parser.parse_args(argv)
    if args.guided_sample:
        raise SystemExit("--guided_sample needs the causal LM of --model_id (network / gated weights): out of scope here")
    if not torch.cuda.is_available():
        raise RuntimeError("src.shakespeare (HIP build) needs a GPU: there is no CPU fallback")
    # one process per GPU under torchrun (RANK / WORLD_SIZE / LOCAL_RANK); a plain `python -m` run is world 1
    rank, world, local_rank = dp.init_from_env(os.environ.get("TDM_DIST_BACKEND"))
    if os.environ.get("TDM_SHARE_GPU") == "1":             # rehearsal: several ranks on one card (gloo collectives)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    device = torch.device("cuda", local_rank)
    if rank == 0:
        print(f"Device: {device}" + (f" (rank 0 of {world})" if world > 1 else ""))

    pretrained = None
    if args.byte_tokenizer:
        tokenizer = ByteTokenizer()
        vocab_size, pretrained_dim = tokenizer.vocab_size, 256
        args.use_learned_embeddings = True
    else:   # the reference's path; needs the model files in the local HF cache
        from transformers import AutoModelForCausalLM, AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.model_id)
        lm_model = AutoModelForCausalLM.from_pretrained(args.model_id).to(device)
        pretrained = lm_model.get_input_embeddings().weight.detach().float().to(device)
        vocab_size, pretrained_dim = pretrained.size(0), pretrained.size(1)

    if args.use_learned_embeddings:
        embed_dim = args.embed_dim if args.embed_dim is not None else pretrained_dim
        embedding_fn = LearnedEmbedding(vocab_size, embed_dim, pretrained if args.init_from_pretrained else None).to(device)
        print(f"Using learned embeddings (dim={embed_dim}, init_from_pretrained={args.init_from_pretrained})")
    else:
        embed_dim = pretrained_dim
        embedding_fn = pretrained
        print(f"Using pre-trained embeddings (dim={embed_dim})")
    diff_model = TinyTransformer(embed_dim, dropout=args.dropout).to(device)
    rounding_fn = LearnedRounding(embed_dim, vocab_size).to(device)
    if args.seed is not None and world > 1:
        # identical construction on every rank (same seed above; train() broadcasts rank 0's weights anyway), but the t / noise /
        # dropout streams of the train loop must differ per rank: a global batch of world * B samples would otherwise see only
        # B distinct draws.  (The split / shuffle generator below stays rank-identical.)
        torch.manual_seed(args.seed + 1000003 * rank)

    if args.train:
        raw = load_text_dataset(args.corpus)
        if world > 1:
            # the train / val split and the epoch shuffles must be the SAME on every rank: a generator of their own
            split_gen = torch.Generator().manual_seed(0 if args.seed is None else args.seed)
            train_chunks, val_chunks = tokenize_corpus(raw, tokenizer, args.seq_len, args.val_split, generator=split_gen)
            as_tensor = lambda sub: sub.dataset[torch.as_tensor(sub.indices, dtype=torch.long)]   # noqa: E731
            # disjoint per-rank slices of every global batch (batch_size per GPU, config 5); ragged tail weighted in train()
            train_dl = dp.ShardedBatches(as_tensor(train_chunks), args.batch_size, rank, world, shuffle=True,
                                         seed=int(split_gen.initial_seed()) + 1)
            val_dl = dp.ShardedBatches(as_tensor(val_chunks), args.batch_size, rank, world, shuffle=False)
        else:
            train_chunks, val_chunks = tokenize_corpus(raw, tokenizer, args.seq_len, args.val_split)
            train_dl = DataLoader(train_chunks, batch_size=args.batch_size, shuffle=True)
            val_dl = DataLoader(val_chunks, batch_size=args.batch_size, shuffle=False)
        if rank == 0:
            print(f"Training on {len(train_chunks)} chunks, validating on {len(val_chunks)} chunks")
        train(diff_model, rounding_fn, embedding_fn, train_dl, val_dl, device, args.ckpt, epochs=args.epochs, lr=args.lr,
              weight_decay=args.weight_decay, rounding_weight=args.rounding_weight,
              use_learned_embeddings=args.use_learned_embeddings, patience=args.patience,
              use_lr_scheduling=args.use_lr_scheduling, warmup_steps=args.warmup_steps)
    texts = None
    if args.sample and world > 1:
        dp.barrier()                                       # rank 0 wrote the checkpoint the other ranks are about to read
    if args.sample:
        checkpoint = load_checkpoint(args.ckpt, device)
        if isinstance(checkpoint, dict) and "diffusion_model" in checkpoint:
            load_text_checkpoint(checkpoint, diff_model, rounding_fn, embedding_fn if args.use_learned_embeddings else None)
            if args.use_learned_embeddings and "embedding_fn" not in checkpoint:
                print("Warning: Learned embeddings requested but not found in checkpoint. Using pre-trained fallback.")
                if pretrained is None:
                    raise RuntimeError("no pre-trained embedding matrix to fall back to (--byte_tokenizer)")
                args.use_learned_embeddings, embedding_fn = False, pretrained
        else:
            diff_model.load_state_dict(checkpoint)
            print("Warning: Using old checkpoint format. Falling back to pre-trained embeddings and cosine similarity.")
            if pretrained is None:
                raise RuntimeError("an old-format checkpoint needs the pre-trained embedding matrix (--model_id)")
            args.use_cosine_fallback, args.use_learned_embeddings, embedding_fn = True, False, pretrained
        first, last = dp.shard_chains(args.n, rank, world)     # chains are independent: sharded, no collective
        texts = []
        if last > first:
            texts = sample(diff_model, rounding_fn, embedding_fn, tokenizer, device, last - first, args.seq_len,
                           use_learned_rounding=not args.use_cosine_fallback,
                           use_learned_embeddings=args.use_learned_embeddings, embed_dim=embed_dim, index_offset=first)
    return texts


if __name__ == "__main__":
    main()
