"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference).  It imports the
reference's `src.mnist` / `src.shakespeare` with inert stand-ins for three
modules that are absent from this image and carry no arithmetic (`dotenv`,
`torchvision`, `google.cloud.storage`), never writes under
/root/reference (`sys.dont_write_bytecode`), and stores inputs + outputs only
(data, no reference source).  TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden.py mnist
    python oracle/make_golden.py text      # separate process (import order)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs(with_torchvision: bool):
    _stub("dotenv", load_dotenv=lambda *a, **k: False)
    if with_torchvision:
        tv = _stub("torchvision")
        for s in ("datasets", "transforms", "utils"):
            setattr(tv, s, _stub("torchvision." + s))
    g = _stub("google")
    gc = _stub("google.cloud")
    gs = _stub("google.cloud.storage", Client=object)
    g.cloud = gc
    gc.storage = gs


def _np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def gen_mnist():
    _install_stubs(with_torchvision=True)
    sys.path.insert(0, REF)
    import src.mnist as M
    import torch.nn.functional as F

    os.makedirs(GOLD, exist_ok=True)
    # ---- a1: tables ------------------------------------------------------
    np.savez(os.path.join(GOLD, "schedule.npz"),
             betas=M.betas.numpy(), alphas=M.alphas.numpy(), alphas_cumprod=M.alphas_cumprod.numpy(),
             sqrt_alphas_cumprod=M.sqrt_alphas_cumprod.numpy(),
             sqrt_one_minus_alphas_cumprod=M.sqrt_one_minus_alphas_cumprod.numpy(),
             # the three per-step scalars p_sample derives at run time (src/mnist.py:169-171,179), as
             # computed on THIS host: torch.sqrt goes through MKL and differs by 1 ulp between hosts
             sqrt_recip_alphas=(1.0 / torch.sqrt(M.alphas)).numpy(),
             eps_coef=(M.betas / M.sqrt_one_minus_alphas_cumprod).numpy(),
             sigma=torch.sqrt(M.betas).numpy())

    # ---- weights: the reference's own default init ----------------------
    torch.manual_seed(0)
    net = M.SimpleUNet()
    net.eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}

    # ---- a2: q_sample ------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    B = 4
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    t = torch.tensor([0, 1, 500, 999])
    noise = torch.randn(B, 1, 28, 28, generator=g)
    xq = M.q_sample(x0, t, noise)

    # ---- a3/a4: forward with intermediates --------------------------------
    feats = {}
    hooks = [getattr(net, n).register_forward_hook(lambda m, i, o, n=n: feats.__setitem__(n, o.detach().clone()))
             for n in ("rb1", "rb2", "rb3", "rb4")]
    with torch.no_grad():
        eps = net(xq, t)
    for h in hooks:
        h.remove()

    # ---- a5: loss, grads, two AdamW steps ----------------------------------
    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    step_out = {}
    g2 = torch.Generator().manual_seed(12)
    tr_inputs = {}
    for step in (1, 2):
        xs = torch.rand(B, 1, 28, 28, generator=g2) * 2 - 1
        ts = torch.randint(0, M.timesteps, (B,), generator=g2)
        ns = torch.randn(B, 1, 28, 28, generator=g2)
        x_noisy = M.q_sample(xs, ts, ns)
        pred = net(x_noisy, ts)
        loss = F.mse_loss(pred, ns)
        opt.zero_grad(); loss.backward(); opt.step()
        tr_inputs[f"s{step}.x0"] = xs; tr_inputs[f"s{step}.t"] = ts; tr_inputs[f"s{step}.noise"] = ns
        step_out[f"s{step}.loss"] = loss.detach().reshape(1)
        step_out[f"s{step}.pred"] = pred.detach()
        for k, p in net.named_parameters():
            if step == 1:
                step_out[f"s{step}.grad.{k}"] = p.grad.detach().clone()
            step_out[f"s{step}.param.{k}"] = p.detach().clone()
    np.savez(os.path.join(GOLD, "unet_train.npz"), **_np(tr_inputs), **_np(step_out))

    # ---- a6: p_sample at t in {999, 500, 1, 0} with captured noise ----------
    net.load_state_dict(sd); net.eval()
    ps = {}
    g3 = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for tt in (999, 500, 1, 0):
            x = torch.randn(3, 1, 28, 28, generator=g3)
            tv = torch.full((3,), tt, dtype=torch.long)
            torch.manual_seed(1000 + tt)
            y = M.p_sample(net, x, tv)
            torch.manual_seed(1000 + tt)
            z = torch.randn_like(x)          # model forward draws no RNG in eval mode
            ps[f"t{tt}.x"] = x; ps[f"t{tt}.z"] = z; ps[f"t{tt}.y"] = y
            ps[f"t{tt}.eps"] = net(x, tv)
        # ---- a7: the last 12 steps of the reverse loop, chained ---------------
        n = 2
        x = torch.randn(n, 1, 28, 28, generator=g3)
        ps["chain.x_start"] = x.clone()
        zs = []
        for i in reversed(range(12)):
            tv = torch.full((n,), i, dtype=torch.long)
            torch.manual_seed(2000 + i)
            z = torch.randn_like(x)
            torch.manual_seed(2000 + i)
            x = M.p_sample(net, x, tv)
            zs.append(z)
        ps["chain.z"] = torch.stack(zs)
        ps["chain.x_end"] = x
        ps["chain.x01"] = (x.clamp(-1, 1) + 1) / 2
    np.savez(os.path.join(GOLD, "unet_sample.npz"), **_np(ps))

    np.savez(os.path.join(GOLD, "unet_forward.npz"),
             **{"w." + k: v.numpy() for k, v in sd.items()},
             x0=x0.numpy(), t=t.numpy(), noise=noise.numpy(), x_noisy=xq.numpy(), eps=eps.numpy(),
             h1=feats["rb1"].numpy(), h2=feats["rb2"].numpy(), h3=feats["rb3"].numpy(), h4=feats["rb4"].numpy())
    print("mnist goldens written to", GOLD)


def gen_text():
    _install_stubs(with_torchvision=False)
    sys.path.insert(0, REF)
    import src.shakespeare as S
    import torch.nn.functional as F
    from oracle import ddpm_oracle as O

    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for dim, B, L, tag in ((256, 2, 128, "d256"), (32, 3, 16, "d32")):
        params = O.transformer_init_params(dim, seed=7)
        net = S.TinyTransformer(dim, dropout=0.0)
        missing = net.load_state_dict(params, strict=True)
        g = torch.Generator().manual_seed(21 + dim)
        x0 = torch.randn(B, L, dim, generator=g) * 0.5
        t = torch.randint(0, S.T, (B,), generator=g)
        t[0] = 0
        noise = torch.randn(B, L, dim, generator=g)
        xq = S.q_sample(x0, t, noise)
        # train-mode forward (dropout 0 => deterministic, slow path) + grads
        net.train()
        pred = net(xq, t)
        loss = F.mse_loss(pred, noise)
        loss.backward()
        out[f"{tag}.x0"] = x0; out[f"{tag}.t"] = t; out[f"{tag}.noise"] = noise
        out[f"{tag}.x_noisy"] = xq; out[f"{tag}.pred"] = pred.detach(); out[f"{tag}.loss"] = loss.detach().reshape(1)
        # gradients of a few representative parameters (full set is 15.8 MB at D=256)
        gsel = ["time_emb.weight", "time_emb.bias",
                "encoder.layers.0.self_attn.in_proj_bias", "encoder.layers.2.norm2.weight",
                "encoder.layers.1.linear2.bias", "encoder.layers.0.self_attn.out_proj.weight"]
        named = dict(net.named_parameters())
        for k in gsel:
            out[f"{tag}.grad.{k}"] = named[k].grad.detach().clone()
        if dim == 32:
            for k, p in named.items():
                out[f"{tag}.gradall.{k}"] = p.grad.detach().clone()
        # eval-mode p_sample with captured noise
        net.eval()
        with torch.no_grad():
            for tt in ((999,) if dim == 256 else (999, 0)):
                x = torch.randn(B, L, dim, generator=g)
                tv = torch.full((B,), tt, dtype=torch.long)
                torch.manual_seed(3000 + tt)
                y = S.p_sample(net, x, tv)
                torch.manual_seed(3000 + tt)
                z = torch.randn_like(x)
                out[f"{tag}.ps{tt}.x"] = x; out[f"{tag}.ps{tt}.z"] = z; out[f"{tag}.ps{tt}.y"] = y
    # host-side scalar schedules (N2 rows; cheap to pin now)
    lam = []
    import torch.optim as optim
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = optim.SGD([dummy], lr=1.0)
    sch = S.get_cosine_schedule_with_warmup(opt, 10, 100)
    for _ in range(100):
        lam.append(opt.param_groups[0]["lr"]); opt.step(); sch.step()
    out["cosine_warmup_10_100"] = torch.tensor(lam, dtype=torch.float64)
    out["rounding_weight_e20_w0.5"] = torch.tensor(
        [S.dynamic_rounding_weight_schedule(e, 20, 0.5) for e in range(20)], dtype=torch.float64)
    np.savez(os.path.join(GOLD, "text_denoiser.npz"), **_np(out))
    print("text goldens written to", GOLD)


def gen_text_dropout():
    """Train-mode (dropout 0.1, the reference's default) forward + gradients of the REFERENCE
    TinyTransformer with torch's two dropout entry points replaced by the counter-hash masks
    the product uses (oracle.dropout_keep): F.dropout (nn.Dropout modules: input dropout,
    dropout1, FFN dropout, dropout2) and F.scaled_dot_product_attention (the attention-
    probability dropout inside nn.MultiheadAttention, written out as softmax -> dropout -> @V).
    Call order of the reference's own forward defines the site numbers; this pins site order,
    scaling and placement of every mask against the real module code."""
    _install_stubs(with_torchvision=False)
    sys.path.insert(0, REF)
    import math
    import src.shakespeare as S
    import torch.nn.functional as F
    from oracle import ddpm_oracle as O

    P_DROP, SEED = 0.1, 0x5EEDC0FFEE1234
    state = {"site": 0}

    def hashed_dropout(x, p=0.5, training=True, inplace=False):
        site = state["site"]; state["site"] += 1
        if not training or p == 0.0:
            return x
        return O._dropout(x, p, SEED, site)

    def hashed_sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        assert attn_mask is None and not is_causal
        site = state["site"]; state["site"] += 1
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1]), dim=-1)
        if dropout_p > 0.0:
            att = O._dropout(att, dropout_p, SEED, site)
        return att @ v

    real_dropout, real_sdpa = F.dropout, F.scaled_dot_product_attention
    out = {"p_drop": torch.tensor([P_DROP]), "seed": torch.tensor([SEED], dtype=torch.int64)}
    try:
        F.dropout, F.scaled_dot_product_attention = hashed_dropout, hashed_sdpa
        for dim, B, L, tag in ((256, 2, 128, "d256"), (32, 3, 16, "d32")):
            params = O.transformer_init_params(dim, seed=11)
            net = S.TinyTransformer(dim, dropout=P_DROP)
            net.load_state_dict(params, strict=True)
            g = torch.Generator().manual_seed(77 + dim)
            x0 = torch.randn(B, L, dim, generator=g) * 0.5
            t = torch.randint(0, S.T, (B,), generator=g)
            noise = torch.randn(B, L, dim, generator=g)
            xq = S.q_sample(x0, t, noise).requires_grad_(True)
            net.train()
            state["site"] = 0
            pred = net(xq, t)
            assert state["site"] == 13, state
            loss = F.mse_loss(pred, noise)
            loss.backward()
            out[f"{tag}.x0"] = x0; out[f"{tag}.t"] = t; out[f"{tag}.noise"] = noise
            out[f"{tag}.pred"] = pred.detach(); out[f"{tag}.loss"] = loss.detach().reshape(1)
            out[f"{tag}.dx"] = xq.grad.detach().clone()
            named = dict(net.named_parameters())
            gsel = ["time_emb.weight", "time_emb.bias", "encoder.layers.0.self_attn.in_proj_bias",
                    "encoder.layers.2.norm2.weight", "encoder.layers.1.linear2.bias", "encoder.layers.1.linear1.bias",
                    "encoder.layers.0.self_attn.out_proj.weight", "encoder.layers.2.self_attn.out_proj.bias"]
            for k in (named if dim == 32 else gsel):
                out[f"{tag}.grad.{k}"] = named[k].grad.detach().clone()
    finally:
        F.dropout, F.scaled_dot_product_attention = real_dropout, real_sdpa
    np.savez(os.path.join(GOLD, "text_dropout.npz"), **_np(out))
    print("text dropout goldens written to", GOLD)


def gen_text_head():
    """Row N1: the reference's LearnedEmbedding / LearnedRounding modules and the rounding-loss part of
    its train step (src/shakespeare.py:46-102, :225-243): x0 = embedding_fn(ids); logits = rounding_fn(x0);
    loss = cross_entropy(logits, ids); plus a diffusion-like second use of x0 so that d(loss)/d(table)
    mixes both paths, and the argmax decode (:389-390)."""
    _install_stubs(with_torchvision=False)
    sys.path.insert(0, REF)
    import src.shakespeare as S
    import torch.nn.functional as F

    out = {}
    for V, D, B, L, tag in ((1003, 32, 3, 16, "v1003"), (2048, 64, 2, 64, "v2048")):
        torch.manual_seed(100 + V)
        emb = S.LearnedEmbedding(V, D)
        rnd = S.LearnedRounding(D, V)
        with torch.no_grad():
            emb.embeddings.weight.mul_(25.0)      # std 0.5: logits with a real spread (default init 0.02 gives ~uniform softmax)
        g = torch.Generator().manual_seed(V)
        ids = torch.randint(0, V, (B, L), generator=g)
        ids[0, :4] = ids[0, 0]                    # repeated ids: the embedding gradient accumulates
        target = torch.randn(B, L, D, generator=g)
        x0 = emb(ids)
        logits = rnd(x0)
        ce = F.cross_entropy(logits.reshape(-1, V), ids.reshape(-1))
        total = F.mse_loss(x0, target) + 0.7 * ce
        total.backward()
        out[f"{tag}.table"] = emb.embeddings.weight.detach().clone()
        out[f"{tag}.W"] = rnd.decoder.weight.detach().clone(); out[f"{tag}.b"] = rnd.decoder.bias.detach().clone()
        out[f"{tag}.ids"] = ids; out[f"{tag}.target"] = target
        out[f"{tag}.x0"] = x0.detach(); out[f"{tag}.ce"] = ce.detach().reshape(1)
        out[f"{tag}.argmax"] = logits.detach().argmax(dim=-1)
        out[f"{tag}.logits_head"] = logits.detach()[0, :2].clone()
        out[f"{tag}.dtable"] = emb.embeddings.weight.grad.clone()
        out[f"{tag}.dW"] = rnd.decoder.weight.grad.clone(); out[f"{tag}.db"] = rnd.decoder.bias.grad.clone()
    # Cosine-similarity fallback decode: the lines live inside the reference's `sample()` (src/shakespeare.py:393-401), so
    # that function itself is run, with its 1000-step loop short-circuited (p_sample patched to hand back a prepared
    # final x at t = 0), a tokenizer stand-in that captures the token ids it is asked to decode, and the file writers
    # patched out.  Both variants: learned embedding module, and a raw pre-trained matrix.
    import contextlib
    import io
    for V, D, n, L, tag in ((1003, 32, 3, 16, "cos1003"), (2048, 64, 2, 64, "cos2048")):
        torch.manual_seed(200 + V)
        emb = S.LearnedEmbedding(V, D)
        with torch.no_grad():
            emb.embeddings.weight.mul_(25.0)
        g = torch.Generator().manual_seed(V + 1)
        ids = torch.randint(0, V, (n, L), generator=g)
        x_final = emb.embeddings.weight.detach()[ids] * (0.5 + torch.rand(n, L, 1, generator=g)) + \
            0.6 * torch.randn(n, L, D, generator=g)          # noisy, rescaled embeddings: not every token decodes to its id
        captured = {}

        class Tok:
            def batch_decode(self, tokens, skip_special_tokens=True):
                captured["tokens"] = tokens.clone()
                return ["" for _ in range(tokens.shape[0])]

        def fake_p_sample(model, x, t):
            return x_final.clone() if int(t[0]) == 0 else x

        saved = (S.p_sample, S.save_samples, S.get_samples_dir)
        S.p_sample, S.save_samples, S.get_samples_dir = fake_p_sample, (lambda *a, **k: None), (lambda *a, **k: "/tmp/tdm_golden_samples")
        try:
            dummy = torch.nn.Linear(1, 1)
            for variant, efn, learned in (("learned", emb, True), ("matrix", emb.embeddings.weight.detach().clone(), False)):
                with contextlib.redirect_stdout(io.StringIO()):
                    S.sample(dummy, dummy, efn, Tok(), "cpu", n_samples=n, seq_len=L, use_learned_rounding=False,
                             use_learned_embeddings=learned, embed_dim=D)
                out[f"{tag}.{variant}.tokens"] = captured["tokens"]
        finally:
            S.p_sample, S.save_samples, S.get_samples_dir = saved
        assert torch.equal(out[f"{tag}.learned.tokens"], out[f"{tag}.matrix.tokens"])
        out[f"{tag}.E"] = emb.embeddings.weight.detach().clone()
        out[f"{tag}.x"] = x_final
        out[f"{tag}.ids"] = ids
    np.savez(os.path.join(GOLD, "text_head.npz"), **_np(out))
    print("text head goldens written to", GOLD)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "mnist"
    torch.set_num_threads(8)
    {"mnist": gen_mnist, "text": gen_text, "text_dropout": gen_text_dropout, "text_head": gen_text_head}[which]()
