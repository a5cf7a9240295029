"""GPU parity tests of the HIP MNIST-UNet path, called through the C ABI
(ctypes -> libtdm_hip.so).  Checker = CPU oracle (oracle/ddpm_oracle.py) and
the golden vectors captured from the reference (tests/golden/*.npz).

Tolerances: q_sample / p_sample arithmetic / uint8 pixels: bit-exact.
UNet eps and gradients: <= 1e-3 relative (BASELINE.json north_star), measured
as max|a-ref| / max|ref| per tensor; fp32 MFMA is an exact-fp32 fmaf chain so
the achieved error is ~1e-6 and the asserts below use 2e-5."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ddpm_oracle as O

TOL = 2e-5          # asserted for the exact-fp32 MFMA kernels; the north-star bound is 1e-3
NORTH_STAR_TOL = 1e-3
# bf16x3 split-operand kernels keep 16 mantissa bits per operand: ~1e-5 per layer
TOL_BF16X3 = 2e-4


def _tol(base=TOL):
    """Tolerance for whole-network checks under the active conv arithmetic."""
    from tinydiffusionmodels_amd import _lib
    return base if _lib.lib().tdm_get_conv_mode() == 0 else max(base * 10, TOL_BF16X3)


def _gtol():
    """Gradient tolerance.  Exact-fp32 mode: 5e-5.  bf16x3 mode: the ~1e-5 forward
    difference flips the sign of a few near-zero ReLU pre-activations relative to
    the reference; every flipped mask entry moves weight gradients by
    O(1/sqrt(#pixels)) (measured 1.8e-3 at B=37, falling as 1/sqrt(B)), so
    gradients are held to 2.5e-3 there (round 4: what is measured at B >= 37 — 5.9e-4 at B = 37, 6e-4 at B = 512,
    profiles/r04_parity.json — plus margin; it was 5e-3) — the kernels themselves are pinned at 5e-5
    by the per-layer tests and by the fp32 mode of this same test, and at 2e-4 with the oracle's masks teacher-forced."""
    from tinydiffusionmodels_amd import _lib
    return 5e-5 if _lib.lib().tdm_get_conv_mode() == 0 else 2.5e-3


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _weights(d, prefix="w."):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def pinned_tables(golden_tables):
    """Teacher-force the schedule: product and oracle both use the tables of
    the host that produced the golden vectors (torch.sqrt differs by 1 ulp
    between hosts), so bit-exact comparisons are meaningful on any box."""
    from tinydiffusionmodels_amd import schedule
    schedule.set_tables(golden_tables)
    yield golden_tables
    schedule.set_tables(None)


@pytest.fixture(scope="module", autouse=True, params=[2, 0], ids=["bf16x3-s16", "fp32"])
def conv_mode(request):
    """Run every test of this module under both conv arithmetics (include/tdm_hip.h: tdm_set_conv_mode)."""
    from tinydiffusionmodels_amd import _lib
    L = _lib.lib()
    _lib.check(L.tdm_set_conv_mode(request.param))
    yield request.param
    _lib.check(L.tdm_set_conv_mode(2))


def _s16_decode(t):
    """S16 tensor (bytes of a float32 tensor [..., C]) -> float32 hi+lo values [..., C]."""
    C = t.shape[-1]
    u = t.contiguous().view(torch.int16).view(*t.shape[:-1], C // 16, 2, 16).to(torch.int32)   # [.., group, hi/lo, 16]
    f = (u.to(torch.int64) << 16).to(torch.int32).view(torch.float32)     # bf16 bits -> fp32
    return (f[..., 0, :] + f[..., 1, :]).reshape(*t.shape[:-1], C)


@pytest.fixture(scope="module")
def lib():
    from tinydiffusionmodels_amd import _lib, unet_engine
    L = _lib.lib()          # raises if the HIP library is missing: no fallback
    unet_engine.check_layout_against_library()
    return L


@pytest.fixture(scope="module")
def model(dev, lib, golden_dir):
    from tinydiffusionmodels_amd.mnist import SimpleUNet
    m = SimpleUNet()
    m.load_state_dict(_weights(_load(golden_dir, "unet_forward.npz")))
    return m.to(dev)


# ------------------------------------------------------------------ q_sample
def test_q_sample_bit_exact_golden(dev, lib, golden_dir):
    from tinydiffusionmodels_amd.mnist import q_sample
    g = _load(golden_dir, "unet_forward.npz")
    out = q_sample(g["x0"].to(dev), g["t"].to(dev), g["noise"].to(dev)).cpu()
    assert torch.equal(out, g["x_noisy"])


@pytest.mark.parametrize("shape", [(512, 1, 28, 28), (7, 1, 28, 28), (5, 3), (256, 128, 256)])
def test_q_sample_bit_exact_oracle(dev, lib, shape, golden_tables):
    from tinydiffusionmodels_amd.mnist import q_sample
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(*shape, generator=g)
    noise = torch.randn(*shape, generator=g)
    t = torch.randint(0, 1000, (shape[0],), generator=g)
    ref = O.q_sample(x0, t, noise, golden_tables)
    out = q_sample(x0.to(dev), t.to(dev), noise.to(dev)).cpu()
    assert torch.equal(out, ref)


# ------------------------------------------------------------ per-layer convs
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _hwio(w):
    return w.permute(2, 3, 1, 0).contiguous()


CONV_CASES = [  # (hw, Cin, Cout, k, B)
    (28, 32, 32, 3, 3), (14, 32, 64, 3, 5), (14, 32, 64, 1, 5), (14, 64, 64, 3, 2), (28, 32, 32, 3, 1),
    (28, 96, 32, 3, 2), (28, 96, 32, 1, 2), (14, 64, 64, 3, 70),
]


def _run_conv(lib, conv_mode, dev, args, scratch_floats, B, hw, cin, cout, k, flags):
    from tinydiffusionmodels_amd import _lib
    if conv_mode == 0:
        _lib.check(lib.tdm_conv_nhwc_f32(*[_lib.ptr(a) for a in args], B, hw, cin, cout, k, flags, _lib.stream()))
    else:   # S16 pipeline: also returns the pre-split copy of (result + tb_out)
        scratch = torch.empty(scratch_floats + B * hw * hw * cin + 128, device=dev)
        out_s16 = torch.zeros(B, hw, hw, cout, device=dev)
        tb_out = torch.linspace(-1, 1, B * cout, device=dev).view(B, cout).contiguous()
        _lib.check(lib.tdm_conv_nhwc_s16_f32(*[_lib.ptr(a) for a in args], _lib.ptr(out_s16), _lib.ptr(tb_out),
                                             _lib.ptr(scratch), B, hw, cin, cout, k, flags, _lib.stream()))
        torch.cuda.synchronize()
        dec = _s16_decode(out_s16.cpu())
        want = args[5].cpu() + tb_out.cpu()[:, None, None, :]
        assert O.rel_err(dec, want) < 2e-5          # hi+lo keeps 16 mantissa bits: 2^-17 relative
    torch.cuda.synchronize()


@pytest.mark.parametrize("hw,cin,cout,k,B", CONV_CASES)
def test_conv_forward_layer(dev, lib, conv_mode, hw, cin, cout, k, B):
    from tinydiffusionmodels_amd import _lib
    g = torch.Generator().manual_seed(hw * 1000 + cin + cout + k + B)
    x = torch.randn(B, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    tb = torch.randn(B, cin, generator=g)
    res = torch.randn(B, cout, hw, hw, generator=g)
    a_ref = F.relu(F.conv2d(x + tb[:, :, None, None], w, bias, padding=k // 2))
    ref = a_ref + res
    out = torch.empty(B, hw, hw, cout, device=dev)
    aux = torch.empty_like(out)
    args = [_nhwc(x).to(dev), _hwio(w).to(dev), bias.to(dev), _nhwc(res).to(dev), tb.to(dev), out, aux]
    _run_conv(lib, conv_mode, dev, args, k * k * cin * cout, B, hw, cin, cout, k, 1)
    tol = TOL if conv_mode == 0 else 5e-5
    assert O.rel_err(_nchw(out.cpu()), ref) < tol
    assert O.rel_err(_nchw(aux.cpu()), a_ref) < tol


DGRAD_CASES = [(28, 32, 32, 3, 3), (14, 64, 64, 3, 5), (14, 32, 64, 3, 2), (28, 96, 32, 3, 2), (14, 32, 64, 1, 3)]


@pytest.mark.parametrize("hw,cin,cout,k,B", DGRAD_CASES)
def test_conv_dgrad_layer(dev, lib, conv_mode, hw, cin, cout, k, B):
    """dx of y = conv(x, w): the dgrad call gets dy (cout channels) and the forward HWIO weight."""
    from tinydiffusionmodels_amd import _lib
    g = torch.Generator().manual_seed(7 + hw + cin + cout + k)
    dy = torch.randn(B, cout, hw, hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cout * k * k) ** 0.5
    res = torch.randn(B, cin, hw, hw, generator=g)
    ref = F.conv_transpose2d(dy, w, padding=k // 2) + res
    out = torch.empty(B, hw, hw, cin, device=dev)
    args = [_nhwc(dy).to(dev), _hwio(w).to(dev), None, _nhwc(res).to(dev), None, out, None]
    # dgrad: "Cin" of the call = channels of dy (K), "Cout" of the call = channels of dx (N)
    _run_conv(lib, conv_mode, dev, args, k * k * cin * cout, B, hw, cout, cin, k, 2)
    assert O.rel_err(_nchw(out.cpu()), ref) < (TOL if conv_mode == 0 else 5e-5)


WGRAD_CASES = [(28, 32, 32, 3, 3), (14, 64, 64, 3, 5), (14, 32, 64, 3, 2), (14, 32, 64, 1, 3), (28, 32, 32, 3, 40),
               (28, 64, 32, 3, 2)]


@pytest.mark.parametrize("hw,cin,cout,k,B", WGRAD_CASES)
def test_conv_wgrad_layer(dev, lib, conv_mode, hw, cin, cout, k, B):
    from tinydiffusionmodels_amd import _lib
    g = torch.Generator().manual_seed(11 + hw + cin + cout + k + B)
    x = torch.randn(B, cin, hw, hw, generator=g)
    tb = torch.randn(B, cin, generator=g)
    dy = torch.randn(B, cout, hw, hw, generator=g)
    w = torch.zeros(cout, cin, k, k, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    y = F.conv2d(x + tb[:, :, None, None], w, b, padding=k // 2)
    y.backward(dy)
    dw = torch.empty(k, k, cin, cout, device=dev)
    db = torch.empty(cout, device=dev)
    if conv_mode == 2:
        scratch = torch.empty(B * hw * hw * (cin + cout) + 65 * k * k * cin * cout + 256, device=dev)
        args = [_nhwc(x).to(dev), tb.to(dev), _nhwc(dy).to(dev), dw, scratch]
        _lib.check(lib.tdm_conv_wgrad_nhwc_s16_f32(*[_lib.ptr(a) for a in args], B, hw, cin, cout, k, _lib.stream()))
        torch.cuda.synchronize()
        assert O.rel_err(dw.cpu().permute(3, 2, 0, 1), w.grad) < 5e-5
        return
    slabs = torch.empty(65 * (k * k * cin * cout + cout), device=dev)
    args = [_nhwc(x).to(dev), tb.to(dev), _nhwc(dy).to(dev), dw, db, slabs]
    _lib.check(lib.tdm_conv_wgrad_nhwc_f32(*[_lib.ptr(a) for a in args], B, hw, cin, cout, k, _lib.stream()))
    torch.cuda.synchronize()
    assert O.rel_err(dw.cpu().permute(3, 2, 0, 1), w.grad) < (TOL if conv_mode == 0 else 5e-5)
    assert O.rel_err(db.cpu(), b.grad) < TOL          # bias gradient is summed in exact fp32 in both modes


# ------------------------------------------------------------- whole network
def test_unet_forward_golden(dev, model, golden_dir):
    from tinydiffusionmodels_amd import unet_engine as E
    g = _load(golden_dir, "unet_forward.npz")
    x, t = g["x_noisy"].to(dev), g["t"].to(dev)
    ws = E.UNetWorkspace(x.shape[0], dev, training=True)
    eps = E.unet_forward(model.flat.detach(), x, t, ws, save=True)
    for k in ("h1", "h2", "h3", "h4"):
        assert O.rel_err(E.get_activation(ws, k).cpu(), g[k]) < _tol(), k
    assert O.rel_err(eps.cpu(), g["eps"]) < _tol()
    with torch.no_grad():
        assert torch.equal(model(x, t), eps)      # module call == engine call


def test_residual_block_module_golden(dev, lib, golden_dir):
    """`from src.mnist import ResidualBlock` (src/mnist.py:45-61 of the reference): each of the four blocks, loaded with the
    golden weights under the reference's own parameter names, maps the golden block input to the golden block output
    (h1..h4 captured from the imported reference) through tdm_resblock_fwd_f32."""
    from src.mnist import ResidualBlock
    g = _load(golden_dir, "unet_forward.npz")
    w = _weights(g)
    that = (g["t"].float() / 1000).view(-1, 1, 1, 1)
    h1, h2, h3 = g["h1"], g["h2"], g["h3"]
    ins = {"rb1": g["x_noisy"], "rb2": F.avg_pool2d(h1, 2), "rb3": h2,
           "rb4": torch.cat([F.interpolate(h3, scale_factor=2, mode="nearest"), h1], dim=1)}
    outs = {"rb1": h1, "rb2": h2, "rb3": h3, "rb4": g["h4"]}
    for name, ci, co in (("rb1", 1, 32), ("rb2", 32, 64), ("rb3", 64, 64), ("rb4", 96, 32)):
        blk = ResidualBlock(ci, co)
        missing = blk.load_state_dict({k[len(name) + 1:]: v for k, v in w.items() if k.startswith(name + ".")}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        blk = blk.to(dev)
        with torch.no_grad():
            y = blk(ins[name].to(dev), that.to(dev)).cpu()
        assert y.shape == outs[name].shape
        assert O.rel_err(y, outs[name]) < _tol(), name
    with pytest.raises(RuntimeError, match="inference entry point"):
        blk(ins["rb4"].to(dev), that.to(dev))          # autograd recording: refused loudly, no ATen fallback


@pytest.mark.parametrize("B", [1, 9, 33])
def test_unet_forward_oracle_ragged_batches(dev, model, golden_dir, B):
    p = _weights(_load(golden_dir, "unet_forward.npz"))
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 1, 28, 28, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ref = O.unet_forward(p, x, t)
    with torch.no_grad():
        out = model(x.to(dev), t.to(dev)).cpu()
    assert O.rel_err(out, ref) < _tol()


def test_adamw_teacher_forced_golden(dev, lib, golden_dir, golden_tables):
    """The AdamW kernel alone, fed the reference's own gradients: must reproduce
    the reference's parameters after opt.step() (src/mnist.py:148,159)."""
    from tinydiffusionmodels_amd import unet_engine as E
    g = _load(golden_dir, "unet_train.npz")
    p0 = _weights(_load(golden_dir, "unet_forward.npz"))
    tabs = golden_tables
    flat = E.flat_from_state_dict(p0, device=dev)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    g1 = {k: g[f"s1.grad.{k}"] for k in p0}
    E.adamw_step(flat, E.flat_from_state_dict(g1, device=dev), m, v, 1)
    for k, val in E.state_dict_from_flat(flat).items():
        assert O.rel_err(val.cpu(), g[f"s1.param.{k}"]) < 2e-6, (1, k)
    # step 2: gradients from the oracle at the reference's step-1 parameters
    p1 = {k: g[f"s1.param.{k}"] for k in p0}
    _, g2 = O.unet_loss_and_grads(p1, g["s2.x0"], g["s2.t"], g["s2.noise"], tabs)
    flat.copy_(E.flat_from_state_dict(p1, device=dev))
    E.adamw_step(flat, E.flat_from_state_dict(g2, device=dev), m, v, 2)
    for k, val in E.state_dict_from_flat(flat).items():
        assert O.rel_err(val.cpu(), g[f"s2.param.{k}"]) < 2e-6, (2, k)


def test_unet_train_two_steps_golden(dev, lib, conv_mode, golden_dir):
    """loss, every gradient and two full train steps against the reference's own
    `loss.backward(); opt.step()` (src/mnist.py:152-159).  Adam's normalised
    update g/(|g|+eps) amplifies ~1e-7 gradient differences on near-zero
    gradients, so end-to-end parameters are held to 5 % of one lr-sized step
    (the kernel itself is pinned to 2e-6 by the teacher-forced test above)."""
    from tinydiffusionmodels_amd import unet_engine as E
    from tinydiffusionmodels_amd.mnist import SimpleUNet, DDPMTrainer
    g = _load(golden_dir, "unet_train.npz")
    m = SimpleUNet()
    m.load_state_dict(_weights(_load(golden_dir, "unet_forward.npz")))
    m = m.to(dev)
    lr = 1e-3
    tr = DDPMTrainer(m, batch_size=4, lr=lr)
    for step in (1, 2):
        loss = tr.step(g[f"s{step}.x0"].to(dev), t=g[f"s{step}.t"].to(dev), noise=g[f"s{step}.noise"].to(dev))
        assert abs(loss.item() - g[f"s{step}.loss"].item()) < 1e-4 * abs(g[f"s{step}.loss"].item())
        assert O.rel_err(tr.state.eps.cpu(), g[f"s{step}.pred"]) < 1e-3
        if step == 1:
            grads = E.state_dict_from_flat(tr.state.grads)
            for k, v in grads.items():
                assert O.rel_err(v.cpu(), g[f"s1.grad.{k}"]) < _gtol(), k
        sd = m.state_dict()
        got = torch.cat([v.cpu().reshape(-1) for v in sd.values()])
        want = torch.cat([g[f"s{step}.param.{k}"].reshape(-1) for k in sd])
        err = (got - want).abs()
        if conv_mode == 0:
            assert err.max().item() < 0.05 * lr * step, step
        else:
            # bf16x3 at B=4: the ~1e-5 forward difference flips a handful of near-zero ReLU pre-activations, and
            # Adam's g / (|g| + eps) turns the resulting gradient differences on near-zero gradient ELEMENTS into
            # O(lr) parameter differences there.  Asserted: all but a small fraction of the 181,473 parameters
            # within 5 % of one lr-sized update, and the update as a whole agrees (relative L2 over what moved).
            frac = (err < 0.05 * lr * step).float().mean().item()
            assert frac > 0.98, (step, frac)
            p0 = torch.cat([_weights(_load(golden_dir, "unet_forward.npz"))[k].reshape(-1) for k in sd])
            rel_l2 = ((got - want).norm() / (want - p0).norm()).item()
            assert rel_l2 < 0.10, (step, rel_l2)


@pytest.mark.parametrize("B", [3, 37])
def test_unet_grads_oracle_and_autograd_bridge(dev, model, golden_dir, golden_tables, B):
    """Engine gradients == oracle autograd; and the nn.Module/autograd surface
    (`loss = mse(model(x,t), noise); loss.backward()`) gives the same numbers."""
    from tinydiffusionmodels_amd import unet_engine as E
    from tinydiffusionmodels_amd.mnist import q_sample
    p = _weights(_load(golden_dir, "unet_forward.npz"))
    g = torch.Generator().manual_seed(100 + B)
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 1, 28, 28, generator=g)
    loss_ref, grads_ref = O.unet_loss_and_grads(p, x0, t, noise, golden_tables)
    model.zero_grad()
    xq = q_sample(x0.to(dev), t.to(dev), noise.to(dev))
    loss = F.mse_loss(model(xq, t.to(dev)), noise.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    got = E.state_dict_from_flat(model.flat.grad)
    for k, v in grads_ref.items():
        assert O.rel_err(got[k].cpu(), v) < _gtol(), k
    model.zero_grad()


@pytest.mark.parametrize("B", [5, 37])
def test_train_step_form_equals_forward_plus_backward(dev, model, conv_mode, B):
    """The train step takes F.mse_loss's backward and the output conv's gradients in the epilogue of the forward's last
    launch and never writes h4 (default arithmetic; ConvArgs::o1_tgt); tdm_unet_fwd_f32(save) + tdm_unet_bwd_f32 over a given
    d(loss)/d(eps) is the stand-alone form that reads h4.  Same d, same kernels otherwise: every gradient except the output
    conv's is BIT-identical, the output conv's and the loss agree to summation order (src/mnist.py:156-159)."""
    from tinydiffusionmodels_amd import unet_engine as E
    flat = model.flat.detach()
    g = torch.Generator().manual_seed(900 + B)
    x0 = (torch.rand(B, 1, 28, 28, generator=g) * 2 - 1).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    noise = torch.randn(B, 1, 28, 28, generator=g).to(dev)
    st = E.TrainState(flat, B)
    loss = E.loss_and_grad(flat, st, x0, noise, t).clone()
    ws = E.UNetWorkspace(B, dev, training=True)
    eps = E.unet_forward(flat, st.x_noisy, t, ws, save=True)
    assert torch.equal(eps, st.eps)
    deps = (eps - noise) * (2.0 / (B * 784))
    assert torch.equal(deps, st.deps)                                    # the epilogue's d is F.mse_loss's backward, bit for bit
    grads = E.unet_backward(flat, st.x_noisy, deps, ws)
    a, b = E.state_dict_from_flat(st.grads), E.state_dict_from_flat(grads)
    for k in a:
        if k.startswith("out."):
            assert O.rel_err(a[k], b[k]) < 2e-6, k
        else:
            assert torch.equal(a[k], b[k]), k
    ref_loss = ((eps - noise).double() ** 2).mean().item()
    assert abs(loss.item() - ref_loss) < 2e-6 * ref_loss


def _mask_io(ws, B, block, which, mask=None):
    """Read (mask=None) or install the ReLU sign mask of block / conv `which` as a (B,C,H,W) uint8 tensor."""
    from tinydiffusionmodels_amd import _lib
    C, hw = ((32, 28), (64, 14), (64, 14), (32, 28))[block]
    buf = torch.empty(B, C, hw, hw, dtype=torch.uint8, device=ws.ws.device) if mask is None else mask.contiguous()
    _lib.check(_lib.lib().tdm_unet_relu_mask_io(_lib.ptr(ws.ws), B, block, which, _lib.ptr(buf), 0 if mask is None else 1,
                                                _lib.stream()), "relu_mask_io")
    return buf


@pytest.mark.parametrize("B", [4, 37])
def test_unet_grads_teacher_forced_relu_masks(dev, model, conv_mode, golden_dir, golden_tables, B):
    """The claim behind _gtol(): in the default (bf16x3) arithmetic the end-to-end gradient differs from the
    reference's by up to ~2e-3 ONLY because a few near-zero ReLU pre-activations change sign (a 1e-5 forward
    difference).  Tested here: (1) the product's own masks differ from the oracle's in a handful of entries,
    every one of them at a pre-activation within 1e-4 of zero; (2) with the ORACLE's masks installed in the
    workspace (teacher forcing through the C ABI) the same backward kernels reproduce the reference gradients
    to 2e-4 — 25x tighter than the end-to-end bound (src/mnist.py:158-159)."""
    if conv_mode != 2:
        pytest.skip("byte masks exist in the default (S16) pipeline; fp32 mode is held to 5e-5 end to end")
    from tinydiffusionmodels_amd import _lib, unet_engine as E
    from tinydiffusionmodels_amd.mnist import q_sample
    p = _weights(_load(golden_dir, "unet_forward.npz"))
    g = torch.Generator().manual_seed(100 + B)
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 1, 28, 28, generator=g)
    _, grads_ref = O.unet_loss_and_grads(p, x0, t, noise, golden_tables)
    xq_ref = O.q_sample(x0, t, noise, golden_tables)
    _, inter = O.unet_forward(p, xq_ref, t, return_intermediates=True)
    flat = model.flat.detach()
    xq = q_sample(x0.to(dev), t.to(dev), noise.to(dev))
    ws = E.UNetWorkspace(B, dev, training=True)
    eps = E.unet_forward(flat, xq, t.to(dev), ws, save=True)
    deps = (2.0 / eps.numel()) * (eps - noise.to(dev))
    # (1) the product's masks vs the oracle's
    flipped = total = 0
    for blk, name in enumerate(("rb1", "rb2", "rb3", "rb4")):
        for which in (1, 2):
            a = inter[f"{name}.a{which}"]
            mine = _mask_io(ws, B, blk, which).cpu().bool()
            diff = mine != (a > 0)
            flipped += int(diff.sum())
            total += diff.numel()
            assert (a[diff].abs() < 1e-4).all(), (name, which)      # only at the ReLU's kink
    assert flipped <= max(8, total // 20000), (flipped, total)        # measured: a few per 10^5
    e2e = E.state_dict_from_flat(E.unet_backward(flat, xq, deps, ws))
    worst_e2e = max(O.rel_err(e2e[k].cpu(), v) for k, v in grads_ref.items())
    # (2) teacher-forced masks
    for blk, name in enumerate(("rb1", "rb2", "rb3", "rb4")):
        for which in (1, 2):
            _mask_io(ws, B, blk, which, (inter[f"{name}.a{which}"] > 0).to(torch.uint8).to(dev))
    tf = E.state_dict_from_flat(E.unet_backward(flat, xq, deps, ws))
    worst_tf = max(O.rel_err(tf[k].cpu(), v) for k, v in grads_ref.items())
    print(f"B={B}: {flipped} of {total} mask entries flipped; grad rel err end-to-end {worst_e2e:.2e}, teacher-forced {worst_tf:.2e}")
    assert worst_tf < 2e-4, worst_tf
    assert worst_e2e < _gtol()
    if flipped == 0:
        assert worst_e2e < 2e-4


def test_schedule_tables_of_this_host(lib, golden_dir):
    """a1 on the box the tests run on, with the schedule NOT pinned: betas / alphas / alphas_cumprod are
    host-independent and must be bit-equal to the reference's (src/mnist.py:28-31);
    the two sqrt tables go through the host's libm / MKL and may differ from the golden host's by 1 ulp
    (the reference's own module globals would, too) — the count is printed."""
    import hashlib
    from tinydiffusionmodels_amd import schedule
    g = _load(golden_dir, "schedule.npz")
    own = schedule.make_tables()
    want_sha = {"betas": "a455de8584c2913e", "alphas": "b32d9a7d2718af05", "alphas_cumprod": "b5555536933367c4"}
    for k, h in want_sha.items():
        assert torch.equal(own[k], g[k]), k
        assert hashlib.sha256(own[k].numpy().tobytes()).hexdigest()[:16] == h, k
    for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        a, b = own[k].view(torch.int32), g[k].view(torch.int32)
        ulp = (a - b).abs()
        print(f"{k}: {int((ulp != 0).sum())} of 1000 entries differ from the golden host's, max {int(ulp.max())} ulp")
        assert int(ulp.max()) <= 1, k


def test_full_size_properties_b4096_sampler(dev, model, conv_mode):
    """BASELINE config 4 size (B=4096 reverse steps): one reverse step is independent of the rest of the batch
    (bitwise), the hipGraph-captured chain equals the eager chain bitwise (teacher-forced noise bank), and the
    device-noise graph path stays finite (src/mnist.py:190-193)."""
    if conv_mode != 2:
        pytest.skip("full-size sampler properties run once, in the default arithmetic")
    from tinydiffusionmodels_amd.mnist import reverse_diffusion, p_sample
    B = 4096
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(B, 1, 28, 28, device=dev, generator=g)
    zs = [torch.randn(B, 1, 28, 28, device=dev, generator=g) for _ in range(4)]
    with torch.no_grad():
        t = torch.full((B,), 321, dtype=torch.long, device=dev)
        full = p_sample(model, x, t, noise=zs[0])
        sl = slice(2000, 2016)
        part = p_sample(model, x[sl].contiguous(), t[sl].contiguous(), noise=zs[0][sl].contiguous())
        assert torch.equal(full[sl], part)
        a = reverse_diffusion(model, x, noises=zs, t_start=3, use_graph=False)
        b = reverse_diffusion(model, x, noises=zs, t_start=3, use_graph=True)
        assert torch.equal(a, b)
        c = reverse_diffusion(model, x, t_start=15, use_graph=True)
        assert torch.isfinite(c).all() and c.shape == x.shape
    model._samplers.clear()


# ------------------------------------------------------------------ sampling
def test_p_sample_golden(dev, model, golden_dir):
    from tinydiffusionmodels_amd import _lib
    from tinydiffusionmodels_amd.mnist import p_sample
    from tinydiffusionmodels_amd.schedule import device_tables
    g = _load(golden_dir, "unet_sample.npz")
    tabs = device_tables(dev)
    with torch.no_grad():
        for tt in (999, 500, 1, 0):
            x = g[f"t{tt}.x"].to(dev)
            t = torch.full((x.shape[0],), tt, dtype=torch.long, device=dev)
            y = p_sample(model, x, t, noise=g[f"t{tt}.z"].to(dev))
            assert O.rel_err(y.cpu(), g[f"t{tt}.y"]) < _tol(), tt
            # update arithmetic alone, teacher-forced with the reference's eps: bit exact
            out = torch.empty_like(x)
            z = g[f"t{tt}.z"].to(dev) if tt > 0 else None
            eps_ref = g[f"t{tt}.eps"].to(dev)
            _lib.check(_lib.lib().tdm_p_sample_update_f32(
                _lib.ptr(x), _lib.ptr(eps_ref), _lib.ptr(z), _lib.ptr(tabs["sqrt_recip_alphas"]),
                _lib.ptr(tabs["eps_coef"]), _lib.ptr(tabs["sigma"]), tt, _lib.ptr(out), x.numel(), _lib.stream()))
            assert torch.equal(out.cpu(), g[f"t{tt}.y"]), tt


def test_reverse_chain_and_uint8_golden(dev, model, golden_dir):
    from tinydiffusionmodels_amd.mnist import reverse_diffusion, to_image_range
    g = _load(golden_dir, "unet_sample.npz")
    zs = [z.to(dev) for z in g["chain.z"]]
    x_end = reverse_diffusion(model, g["chain.x_start"].to(dev), noises=zs, t_start=11)
    assert O.rel_err(x_end.cpu(), g["chain.x_end"]) < _tol(1e-4)
    # integer outputs, teacher-forced on an identical final x: bit exact
    x01, u8 = to_image_range(g["chain.x_end"].to(dev))
    assert torch.equal(x01.cpu(), g["chain.x01"])
    assert torch.equal(u8.cpu(), O.to_uint8(g["chain.x01"]))
    # end to end from shared noise: count differing pixels
    _, u8_e2e = to_image_range(x_end)
    assert (u8_e2e.cpu() != O.to_uint8(g["chain.x01"])).sum().item() == 0


def test_eps_vs_oracle_at_the_benchmarked_sizes(dev, model, conv_mode, golden_tables):
    """north_star's parity clause at the sizes bench.py times, against the CPU oracle itself (not the HIP path against itself):
    predicted noise of the TRAIN forward at B = 512 (config 2; x_t = q_sample(x0, t, noise), src/mnist.py:156-157), and of one
    REVERSE step at B = 4096 (config 4; src/mnist.py:174) on a 64-image slice — the oracle runs those 64 images alone, the
    HIP path runs all 4096 (images are independent).  Bound 1e-3 relative; asserted at the usual 2e-4 / 2e-5."""
    from tinydiffusionmodels_amd.mnist import q_sample
    p = {k: v.cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(77)
    B = 512
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 1, 28, 28, generator=g)
    with torch.no_grad():
        xq = q_sample(x0.to(dev), t.to(dev), noise.to(dev))
        eps = model(xq, t.to(dev)).cpu()
    ref = O.unet_forward(p, O.q_sample(x0, t, noise, golden_tables), t)
    e512 = O.rel_err(eps, ref)
    B = 4096
    x = torch.randn(B, 1, 28, 28, generator=g)
    tt = torch.full((B,), 417, dtype=torch.long)
    sl = slice(2000, 2064)
    with torch.no_grad():
        eps4096 = model(x.to(dev), tt.to(dev))[sl].cpu()
    e4096 = O.rel_err(eps4096, O.unet_forward(p, x[sl], tt[sl]))
    print(f"eps vs oracle: B=512 train forward {e512:.2e}, 64-image slice of a B=4096 reverse step {e4096:.2e}")
    assert e512 < _tol() and e4096 < _tol()
    assert max(e512, e4096) < NORTH_STAR_TOL


def test_full_size_properties_b512(dev, model):
    """BASELINE config 2 size (B=512): batch independence (bitwise), and the
    batch gradient equals the mean of per-chunk gradients (linearity of the
    mean loss in the batch)."""
    from tinydiffusionmodels_amd import unet_engine as E
    g = torch.Generator().manual_seed(5)
    B = 512
    x0 = (torch.rand(B, 1, 28, 28, generator=g) * 2 - 1).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    noise = torch.randn(B, 1, 28, 28, generator=g).to(dev)
    flat = model.flat.detach()
    with torch.no_grad():
        full = model(x0, t)
        part = model(x0[100:108].contiguous(), t[100:108].contiguous())
    assert torch.equal(full[100:108], part)
    st = E.TrainState(flat, B)
    E.loss_and_grad(flat, st, x0, noise, t)
    g_full = st.grads.clone()
    acc = torch.zeros_like(g_full)
    st64 = E.TrainState(flat, 64)
    for c in range(8):
        s = slice(c * 64, (c + 1) * 64)
        E.loss_and_grad(flat, st64, x0[s].contiguous(), noise[s].contiguous(), t[s].contiguous())
        acc += st64.grads
    assert O.rel_err(g_full, acc / 8) < 5e-5          # same arithmetic both sides: no mask flips
    assert torch.isfinite(g_full).all()


def test_reverse_diffusion_hipgraph_matches_eager(dev, model):
    """The hipGraph-captured reverse loop (BASELINE config 4: device-resident step index, two-step
    ping-pong graph replayed) against the eager loop, teacher-forced with the same noise bank;
    even and odd step counts; also that the device-RNG graph path runs and stays finite."""
    from tinydiffusionmodels_amd.mnist import reverse_diffusion
    g = torch.Generator().manual_seed(77)
    x = torch.randn(5, 1, 28, 28, generator=g).to(dev)
    for t_start in (15, 16):
        zs = [torch.randn(5, 1, 28, 28, generator=g).to(dev) for _ in range(t_start + 1)]
        a = reverse_diffusion(model, x, noises=zs, t_start=t_start, use_graph=False)
        b = reverse_diffusion(model, x, noises=zs, t_start=t_start, use_graph=True)
        assert torch.equal(a, b), t_start          # same kernels, same inputs: bitwise
    c = reverse_diffusion(model, x, t_start=31, use_graph=True)
    assert torch.isfinite(c).all() and c.shape == x.shape


def test_mnist_cli_train_then_sample_end_to_end(dev, conv_mode, tmp_path, monkeypatch):
    """`python -m src.mnist --train` then `--sample` (src/mnist.py:214-241) on synthetic images: the whole loop
    (epoch shuffling, DDPMTrainer steps, per-epoch sample grid, raw-state_dict checkpoint with the reference's 32 keys,
    then 1000 reverse steps and the 4x4 PNG grid)."""
    if conv_mode != 2:
        pytest.skip("one arithmetic is enough for the end-to-end smoke")
    from tinydiffusionmodels_amd import mnist as M
    monkeypatch.chdir(tmp_path)
    ckpt = str(tmp_path / "ckpt.pth")
    M.main(["--train", "--synthetic", "256", "--epochs", "1", "--batch_size", "64", "--ckpt", ckpt, "--seed", "0"])
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    assert len(sd) == 32 and sd["rb4.conv1.weight"].shape == (32, 96, 3, 3) and sd["out.weight"].shape == (1, 32, 1, 1)
    assert all(torch.isfinite(v).all() for v in sd.values())
    M.main(["--sample", "--ckpt", ckpt, "--seed", "1"])
    pngs = sorted(p.name for p in (tmp_path / "samples").glob("*.png"))
    assert "samples.png" in pngs and "epoch_001.png" in pngs
    png = (tmp_path / "samples" / "samples.png").read_bytes()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"


def test_epoch_mode_gathers_inside_the_step_and_equals_the_gather_per_step_loop(dev, conv_mode, tmp_path, monkeypatch):
    """mnist.train() with the batch gathered INSIDE the captured step (DDPMTrainer.begin_epoch / step_epoch: dataset and
    permutation resident on the device, position = AdamW's device-side step count minus its value at the start of the epoch —
    one hipGraph replay per batch and nothing between two replays) against the same loop with a gather launch per batch
    (TDM_EPOCH_GATHER=0): same draws, same batches, so the same weights bit for bit — two epochs of five whole batches and a
    ragged tail of three samples each (the tail runs through step() in both forms)."""
    if conv_mode != 2:
        pytest.skip("host-side loop logic; the default arithmetic is enough")
    from tinydiffusionmodels_amd import mnist as M
    B = 16
    data = M.synthetic_mnist(5 * B + 3, seed=77)
    finals = []
    for graph in ("0", "1"):      # the default eager issue (side stream in the backward) and the hipGraph form (TDM_TRAIN_GRAPH=1)
        monkeypatch.setenv("TDM_TRAIN_GRAPH", graph)
        for flag in ("1", "0"):
            monkeypatch.setenv("TDM_EPOCH_GATHER", flag)
            torch.manual_seed(5)
            m = M.SimpleUNet().to(dev)
            M.train(m, str(dev), epochs=2, batch_size=B, lr=1e-3, ckpt_path=str(tmp_path / f"ck{graph}{flag}.pth"), sample_every_epoch=False,
                    data=data, log_every=0)
            torch.cuda.synchronize()
            finals.append(m.flat.detach().clone())
    monkeypatch.delenv("TDM_TRAIN_GRAPH")
    assert all(torch.equal(finals[0], f) for f in finals[1:])
    # the epoch position really comes from the device-side step count: a fresh trainer, three steps, then the batch the
    # fourth step would read is perm[3 B : 4 B]
    torch.manual_seed(6)
    m = M.SimpleUNet().to(dev)
    tr = M.DDPMTrainer(m, B, lr=1e-3)
    d = data.to(dev)
    perm = torch.randperm(d.shape[0], generator=torch.Generator().manual_seed(1)).to(dev)
    tr.begin_epoch(d, perm)
    for _ in range(3):
        tr.step_epoch()
    torch.cuda.synchronize()
    assert tr.steps_taken == 3
    # (x_noisy of the next step = q_sample of exactly that batch with the step's own draws)
    tr.step_epoch()
    torch.cuda.synchronize()
    st = tr.state
    want = M.q_sample(d[perm[3 * B:4 * B]], st.t, st.noise)
    assert torch.equal(st.x_noisy, want)
    # an epoch holds whole batches only: asking for more is refused on the host (the gather kernel would clamp its positions and
    # silently train on repeats of the last sample), and so is a permutation of the wrong length
    whole = d.shape[0] // B
    tr.steps_epoch(whole - 4)
    with pytest.raises(ValueError, match="whole iterations"):
        tr.steps_epoch(1)
    with pytest.raises(ValueError, match="perm has"):
        tr.begin_epoch(d, perm[:-1])
    tr.begin_epoch(d, perm)
    with pytest.raises(ValueError, match="whole iterations"):
        tr.steps_epoch(whole + 1)
    tr.steps_epoch(whole)


def test_in_step_launch_marks_time_one_launch_of_every_eager_step(dev, conv_mode):
    """tdm_unet_mark_launch / tdm_unet_mark_collect (bench.py's in-step roofline timing): with marks on launch id k, every
    eagerly issued train step records one event pair; collect returns one positive duration per step and forgets them; marks
    change no result (same weights as the unmarked run, bit for bit); id < 0 switches them off; ids out of range are refused."""
    if conv_mode != 2:
        pytest.skip("the launch ids describe the default pipeline")
    import numpy as np
    from tinydiffusionmodels_amd import _lib, mnist as M
    L = _lib.lib()
    B, steps = 16, 6
    data = M.synthetic_mnist(steps * B, seed=3).to(dev)
    perm = torch.randperm(data.shape[0], generator=torch.Generator().manual_seed(2)).to(dev)
    nl = L.tdm_unet_launch_count()
    lid = next(i for i in range(nl) if L.tdm_unet_launch_name(i).decode().startswith("rb4.conv1 dgrad, up(h3) part"))
    finals = []
    for marked in (False, True):
        torch.manual_seed(11)
        m = M.SimpleUNet().to(dev)
        tr = M.DDPMTrainer(m, B, lr=1e-3, graph=False)
        tr.begin_epoch(data, perm)
        if marked:
            assert L.tdm_unet_mark_launch(lid, steps + 2) == 0
        try:
            tr.steps_epoch(steps)
            if marked:
                buf = np.full(steps + 2, -1.0, dtype=np.float32)
                assert L.tdm_unet_mark_collect(buf.ctypes.data, steps + 2) == steps
                assert (buf[:steps] > 0).all() and (buf[:steps] < 1e5).all() and (buf[steps:] == -1).all()
                assert L.tdm_unet_mark_collect(buf.ctypes.data, steps + 2) == 0          # forgotten after a collect
                tr.steps_epoch(0)
        finally:
            assert L.tdm_unet_mark_launch(-1, 0) == 0
        torch.cuda.synchronize()
        finals.append(m.flat.detach().clone())
    assert torch.equal(finals[0], finals[1])
    assert L.tdm_unet_mark_launch(nl, 4) != 0 and L.tdm_unet_mark_launch(0, 0) != 0
    buf = np.zeros(4, dtype=np.float32)
    assert L.tdm_unet_mark_collect(buf.ctypes.data, 4) == 0


def test_backward_overlap_side_stream_is_bit_identical_eager_and_captured(dev, conv_mode):
    """tdm_set_bwd_overlap: the weight-gradient launches on the library's side stream (fork / join by events; a parallel
    branch of the graph under capture) give the same weights bit for bit as everything on one stream — eagerly and as
    graph replays (epoch mode, unrolled graphs included), 9 steps at B = 37 (ragged tiles)."""
    from tinydiffusionmodels_amd import _lib, mnist as M
    L = _lib.lib()
    B, steps = 37, 9
    data = M.synthetic_mnist(steps * B, seed=8).to(dev)
    perm = torch.randperm(data.shape[0], generator=torch.Generator().manual_seed(4)).to(dev)
    finals = {}
    try:
        for overlap in (0, 1):
            for graph in (False, True):
                assert L.tdm_set_bwd_overlap(overlap) == 0 and L.tdm_get_bwd_overlap() == overlap
                torch.manual_seed(21)
                m = M.SimpleUNet().to(dev)
                tr = M.DDPMTrainer(m, B, lr=1e-3, graph=graph)
                tr.begin_epoch(data, perm)
                loss = tr.steps_epoch(steps)
                torch.cuda.synchronize()
                finals[(overlap, graph)] = (m.flat.detach().clone(), float(loss.item()))
    finally:
        L.tdm_set_bwd_overlap(1)
    ref = finals[(0, False)]
    for k, v in finals.items():
        assert torch.equal(v[0], ref[0]) and v[1] == ref[1], k
    assert L.tdm_set_bwd_overlap(2) != 0
    # the benchmarked size, eager: twice with the side stream (run-to-run determinism) and once without
    B, steps = 512, 12
    data = M.synthetic_mnist(4 * B, seed=9).to(dev)
    perm = torch.randperm(data.shape[0], generator=torch.Generator().manual_seed(5)).to(dev)
    big = []
    try:
        for overlap in (1, 1, 0):
            assert L.tdm_set_bwd_overlap(overlap) == 0
            torch.manual_seed(22)
            m = M.SimpleUNet().to(dev)
            tr = M.DDPMTrainer(m, B, lr=1e-3, graph=False)
            for _ in range(steps // 4):
                tr.begin_epoch(data, perm)
                tr.steps_epoch(4)
            torch.cuda.synchronize()
            big.append(m.flat.detach().clone())
    finally:
        L.tdm_set_bwd_overlap(1)
    assert torch.equal(big[0], big[1]) and torch.equal(big[0], big[2])
