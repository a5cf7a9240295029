"""The CPU oracle (oracle/ddpm_oracle.py) against vectors captured from the
reference itself (oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as O


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _weights(d, prefix="w."):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


# SURVEY.md §8(a1) known answers (sha256[:16] of the raw fp32 bytes)
TABLE_SHA = {
    "betas": "a455de8584c2913e", "alphas": "b32d9a7d2718af05", "alphas_cumprod": "b5555536933367c4",
    "sqrt_alphas_cumprod": "f03fa1d930d2b0da", "sqrt_one_minus_alphas_cumprod": "6f29eaa972842493",
}


def test_tables_bit_exact(golden_dir):
    g = _load(golden_dir, "schedule.npz")
    tabs = O.make_tables()
    for k, sha in TABLE_SHA.items():
        assert hashlib.sha256(g[k].numpy().tobytes()).hexdigest()[:16] == sha   # fixture == SURVEY known answers
        if k.startswith("sqrt"):
            # torch.sqrt (MKL) is host-dependent by 1 ulp (Intel vs AMD EPYC hosts)
            ulp = (tabs[k].view(torch.int32) - g[k].view(torch.int32)).abs().max().item()
            assert ulp <= 1, (k, ulp)
        else:
            assert torch.equal(tabs[k], g[k]), k
    assert tabs["betas"][0].view(torch.int32).item() == 0x38D1B717
    assert tabs["alphas_cumprod"][999].view(torch.int32).item() == 0x38294666


def test_q_sample_bit_exact(golden_dir, golden_tables):
    g = _load(golden_dir, "unet_forward.npz")
    out = O.q_sample(g["x0"], g["t"], g["noise"], golden_tables)
    assert torch.equal(out, g["x_noisy"])


def test_unet_forward(golden_dir):
    g = _load(golden_dir, "unet_forward.npz")
    eps, inter = O.unet_forward(_weights(g), g["x_noisy"], g["t"], return_intermediates=True)
    assert O.rel_err(eps, g["eps"]) < 1e-6
    for k in ("h1", "h2", "h3", "h4"):
        assert O.rel_err(inter[k], g[k]) < 1e-6, k


def test_unet_train_two_steps(golden_dir, golden_tables):
    g = _load(golden_dir, "unet_train.npz")
    p = _weights(_load(golden_dir, "unet_forward.npz"))
    tabs = golden_tables
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    for step in (1, 2):
        loss, grads = O.unet_loss_and_grads(p, g[f"s{step}.x0"], g[f"s{step}.t"], g[f"s{step}.noise"], tabs)
        assert abs(loss.item() - g[f"s{step}.loss"].item()) < 1e-6 * abs(g[f"s{step}.loss"].item()) + 1e-7
        for k in p:
            if step == 1:
                assert O.rel_err(grads[k], g[f"s1.grad.{k}"]) < 2e-5, k
            p[k], m[k], v[k] = O.adamw_step(p[k], grads[k], m[k], v[k], step)
            assert O.rel_err(p[k], g[f"s{step}.param.{k}"]) < 2e-6, (step, k)


def test_p_sample_and_chain(golden_dir, golden_tables):
    g = _load(golden_dir, "unet_sample.npz")
    p = _weights(_load(golden_dir, "unet_forward.npz"))
    tabs = golden_tables
    for tt in (999, 500, 1, 0):
        x = g[f"t{tt}.x"]
        t = torch.full((x.shape[0],), tt, dtype=torch.long)
        # arithmetic only (teacher-forced eps): bit exact
        y = O.p_sample_from_eps(x, t, g[f"t{tt}.eps"], g[f"t{tt}.z"], tabs)
        assert torch.equal(y, g[f"t{tt}.y"]), tt
        assert O.rel_err(O.p_sample(p, x, t, g[f"t{tt}.z"], tabs), g[f"t{tt}.y"]) < 1e-6
    x_end = O.sample_chain(p, g["chain.x_start"], list(g["chain.z"]), tabs, t_start=11)
    assert O.rel_err(x_end, g["chain.x_end"]) < 1e-5
    assert torch.equal(O.to_unit_range(g["chain.x_end"]), g["chain.x01"])
    u8 = O.to_uint8(O.to_unit_range(x_end))
    ref = O.to_uint8(g["chain.x01"])
    assert (u8 != ref).sum().item() == 0


@pytest.mark.parametrize("tag,dim", [("d32", 32), ("d256", 256)])
def test_transformer(golden_dir, golden_tables, tag, dim):
    g = _load(golden_dir, "text_denoiser.npz")
    p = O.transformer_init_params(dim, seed=7)
    tabs = golden_tables
    assert torch.equal(O.q_sample(g[f"{tag}.x0"], g[f"{tag}.t"], g[f"{tag}.noise"], tabs), g[f"{tag}.x_noisy"])
    pred = O.transformer_forward(p, g[f"{tag}.x_noisy"], g[f"{tag}.t"])
    assert O.rel_err(pred, g[f"{tag}.pred"]) < 2e-6
    loss, grads = O.transformer_loss_and_grads(p, g[f"{tag}.x0"], g[f"{tag}.t"], g[f"{tag}.noise"], tabs)
    assert abs(loss.item() - g[f"{tag}.loss"].item()) < 1e-5 * abs(g[f"{tag}.loss"].item())
    for k, v in g.items():
        if k.startswith(f"{tag}.grad.") or k.startswith(f"{tag}.gradall."):
            name = k.split(".", 2)[2]
            assert O.rel_err(grads[name], v) < 5e-5, name
    for tt in (999, 0):
        if f"{tag}.ps{tt}.x" not in g:
            continue
        x = g[f"{tag}.ps{tt}.x"]
        t = torch.full((x.shape[0],), tt, dtype=torch.long)
        y = O.text_p_sample(p, x, t, g[f"{tag}.ps{tt}.z"], tabs)
        assert O.rel_err(y, g[f"{tag}.ps{tt}.y"]) < 2e-6


@pytest.mark.parametrize("tag,dim", [("d32", 32), ("d256", 256)])
def test_transformer_train_mode_dropout(golden_dir, golden_tables, tag, dim):
    """Train mode (dropout 0.1): the oracle's placement / scaling / site numbering of the 13
    dropout masks against the REFERENCE module run with the same hash-defined masks
    (oracle/make_golden.py:gen_text_dropout)."""
    g = _load(golden_dir, "text_dropout.npz")
    p_drop, seed = float(g["p_drop"][0]), int(g["seed"][0])
    p = O.transformer_init_params(dim, seed=11)
    x0, t, noise = g[f"{tag}.x0"], g[f"{tag}.t"], g[f"{tag}.noise"]
    pred = O.transformer_forward(p, O.q_sample(x0, t, noise, golden_tables), t, p_drop=p_drop, seed=seed)
    assert O.rel_err(pred, g[f"{tag}.pred"]) < 2e-6
    # the masks matter: eval-mode output is far away
    assert O.rel_err(O.transformer_forward(p, O.q_sample(x0, t, noise, golden_tables), t), g[f"{tag}.pred"]) > 1e-2
    loss, grads = O.transformer_loss_and_grads(p, x0, t, noise, golden_tables, p_drop=p_drop, seed=seed, want_dx=True)
    assert abs(loss.item() - g[f"{tag}.loss"].item()) < 1e-5 * abs(g[f"{tag}.loss"].item())
    assert O.rel_err(grads["__dx"], g[f"{tag}.dx"]) < 5e-5
    n = 0
    for k, v in g.items():
        if k.startswith(f"{tag}.grad."):
            assert O.rel_err(grads[k.split(".", 2)[2]], v) < 5e-5, k
            n += 1
    assert n >= 8
    keep = O.dropout_keep(p_drop, seed, 3, (200000,))
    assert abs(keep.float().mean().item() - (1 - p_drop)) < 5e-3


def test_salted_dropout_masks_of_consecutive_steps_are_independent():
    """the graph-replayed text step salts its dropout masks with the Philox offset, which advances by 1 per step.
    XORed into the key that made mask[s+1][i] == mask[s][i ^ (s ^ (s+1))] — one co-drop pattern, permuted, for the whole
    run.  The salt now goes through the hash (tdm_dropout.h: tdm_salted_key; same integers here): consecutive steps' masks
    are not XOR-translates of each other, per-block drop counts vary from step to step, the unsalted family (the golden
    masks) is unchanged, and the library's host evaluation agrees with the oracle bit for bit."""
    import ctypes
    import numpy as np
    p_drop, seed, site, n = 0.1, 0x1234ABCD5678EF01, 3, 1 << 14
    masks = [O.dropout_keep(p_drop, seed, site, (n,), salt=s).numpy() for s in range(1000, 1009)]
    idx = np.arange(n)
    for a in range(len(masks) - 1):
        sa, sb = 1000 + a, 1001 + a
        assert not np.array_equal(masks[a + 1], masks[a][idx ^ (sa ^ sb)])          # the old failure, exactly
        # ... and no other small XOR translation maps one step's mask to the next
        assert all(not np.array_equal(masks[a + 1], masks[a][idx ^ c]) for c in range(64))
        # agreement between consecutive steps is what independent Bernoulli(0.9) masks give: 0.9^2 + 0.1^2 = 0.82
        assert abs((masks[a] == masks[a + 1]).mean() - 0.82) < 0.015
    counts = np.stack([(~m).reshape(-1, 256).sum(axis=1) for m in masks])                # drops per 256-element block, per step
    assert (counts.std(axis=0) > 0).mean() > 0.95                                        # (frozen under the XOR salt)
    assert abs(counts.mean() - 25.6) < 1.0
    # a salt of 0 is a salted key too; None is the unsalted family the goldens hold
    assert not np.array_equal(O.dropout_keep(p_drop, seed, site, (n,), salt=0).numpy(), O.dropout_keep(p_drop, seed, site, (n,)).numpy())
    from tinydiffusionmodels_amd import _lib
    L = _lib.lib()
    for salt in (0, 1, 1003, 0xFFFFFFFF):
        keep = np.empty(4096, dtype=np.uint8)
        _lib.check(L.tdm_dropout_keep_salted_u8(p_drop, seed, salt, site, 0, keep.size, keep.ctypes.data), "keep")
        assert np.array_equal(keep.astype(bool), O.dropout_keep(p_drop, seed, site, (4096,), salt=salt).numpy()), salt
    keep = np.empty(4096, dtype=np.uint8)
    _lib.check(L.tdm_dropout_keep_u8(p_drop, seed, site, 0, keep.size, keep.ctypes.data), "keep")
    assert np.array_equal(keep.astype(bool), O.dropout_keep(p_drop, seed, site, (4096,)).numpy())


@pytest.mark.parametrize("tag", ["v1003", "v2048"])
def test_text_head_oracle(golden_dir, tag):
    """Row N1: embedding lookup, rounding logits / cross-entropy / gradients / argmax of the oracle against
    the reference's own LearnedEmbedding + LearnedRounding modules (oracle/make_golden.py:gen_text_head)."""
    g = _load(golden_dir, "text_head.npz")
    table, W, b, ids, target = (g[f"{tag}.{k}"] for k in ("table", "W", "b", "ids", "target"))
    x0 = O.embed(table, ids)
    assert torch.equal(x0, g[f"{tag}.x0"])
    logits = O.rounding_logits(x0, W, b)
    assert O.rel_err(logits[0, :2], g[f"{tag}.logits_head"]) < 2e-6
    assert torch.equal(logits.argmax(dim=-1), g[f"{tag}.argmax"])
    ce, dx, dW, db = O.rounding_ce_and_grads(x0, W, b, ids)
    assert abs(ce.item() - g[f"{tag}.ce"].item()) < 1e-6 * abs(g[f"{tag}.ce"].item())
    assert O.rel_err(0.7 * dW, g[f"{tag}.dW"]) < 1e-5 and O.rel_err(0.7 * db, g[f"{tag}.db"]) < 1e-5
    # d(total)/d(table) = scatter-add over ids of (mse part + 0.7 * rounding part)
    dx_tot = 2.0 * (x0 - target) / x0.numel() + 0.7 * dx
    dtab = torch.zeros_like(table).index_add_(0, ids.reshape(-1), dx_tot.reshape(-1, table.shape[1]))
    assert O.rel_err(dtab, g[f"{tag}.dtable"]) < 1e-5


def test_host_schedules(golden_dir):
    g = _load(golden_dir, "text_denoiser.npz")
    assert g["cosine_warmup_10_100"].shape[0] == 100 and g["cosine_warmup_10_100"][0] == 0.0
    assert abs(g["rounding_weight_e20_w0.5"][0].item() - 0.5) < 1e-12


@pytest.mark.parametrize("tag", ["cos1003", "cos2048"])
def test_cosine_decode_oracle(golden_dir, tag):
    """The oracle's cosine-similarity decode against the token ids the reference's own `sample()` produced through its
    fallback branch (src/shakespeare.py:393-401; oracle/make_golden.py:gen_text_head runs that function)."""
    g = _load(golden_dir, "text_head.npz")
    tok = O.cosine_decode(g[f"{tag}.x"], g[f"{tag}.E"])
    assert torch.equal(tok, g[f"{tag}.learned.tokens"]) and torch.equal(tok, g[f"{tag}.matrix.tokens"])
    assert 0.5 < (tok == g[f"{tag}.ids"]).float().mean().item() < 1.0      # a real decode: most, not all, tokens recovered


@pytest.mark.parametrize("which", ["mnist", "text"])
def test_oracle_matches_the_live_reference(which):
    """Where the reference itself is present (the build container; never the GPU box) the oracle is also checked against it
    LIVE on random cases — six UNets / four denoisers with the reference's own default init: tables and q_sample bit for
    bit, forward, one ResidualBlock, loss, every gradient, p_sample with the reference's own draw captured, embedding,
    rounding cross-entropy and its gradients, and the PRODUCT's host schedules against the reference's.  Own process: the
    reference's `src` package and this repository's `src` aliases cannot share an interpreter (oracle/check_live_reference.py)."""
    import json
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference is not present on this machine (fixtures under tests/golden pin the oracle)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HF_HUB_OFFLINE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "check_live_reference.py"), which], capture_output=True,
                       text=True, timeout=600, env=env, cwd=os.path.join(root, "oracle"))
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["cases"] >= 4
    tol = 1e-6 if which == "mnist" else 2e-5
    for k, v in res["worst"].items():
        assert v <= tol, (k, v)
    assert not os.path.exists("/root/reference/src/__pycache__")      # nothing was written under the reference
