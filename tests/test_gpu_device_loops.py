"""GPU tests of the host-free loops (through the C ABI): device-drawn Philox noise fused into q_sample and
the p_sample update, the device-resident AdamW step count, the hipGraph-captured train step and reverse
step, and the RCCL collective behind tdm_allreduce_sum_f32 (world 1 on a one-GPU box).

The reference draws from torch's host generator (src/mnist.py:154-155,178), which no device stream can
reproduce, so these paths are checked (a) bit-for-bit against the teacher-forced kernels of the parity
tests fed the SAME draws, (b) against the numpy restatement of the generator in oracle/ddpm_oracle.py
(integers bit-exact, normals to libm rounding), (c) statistically."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ddpm_oracle as O


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def L():
    from tinydiffusionmodels_amd import _lib, unet_engine
    lib = _lib.lib()
    unet_engine.check_layout_against_library()
    return lib


@pytest.fixture(scope="module", autouse=True)
def pinned_tables(golden_tables):
    from tinydiffusionmodels_amd import schedule
    schedule.set_tables(golden_tables)
    yield golden_tables
    schedule.set_tables(None)


def _state(dev, offset=0):
    return torch.tensor([offset, 0], dtype=torch.long, device=dev)


# ---------------------------------------------------------------- generator
@pytest.mark.parametrize("seed,offset", [(0, 0), (0x1234567890ABCDEF, 5), (2 ** 63 + 11, 2 ** 33 + 3)])
def test_philox_words_device_host_oracle(dev, L, seed, offset):
    from tinydiffusionmodels_amd import _lib
    n = 4096
    for kind in (0, 1):
        out = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(L.tdm_philox_u32(seed, offset, kind, _lib.ptr(out), n, _lib.stream()))
        got = out.cpu().numpy().view(np.uint32).reshape(-1, 4)
        want = O.philox_words(seed, offset, np.arange(n // 4), kind)
        assert np.array_equal(got, want)
        h = (ctypes.c_uint32 * 4)()
        L.tdm_philox_u32_host(seed, offset, kind, 777, h)
        assert list(h) == want[777].tolist()
        if seed == 0 and offset == 0 and kind == 0:   # Random123 known answer (counter 0, key 0)
            assert [hex(v) for v in want[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


def test_philox_normals_match_oracle_and_are_standard_normal(dev, L):
    from tinydiffusionmodels_amd import _lib
    n = 1 << 20
    out = torch.empty(n, device=dev)
    _lib.check(L.tdm_philox_normal_f32(99, 3, _lib.ptr(out), n, _lib.stream()))
    z = out.cpu()
    ref = O.philox_normals(99, 3, n)
    assert (z - ref).abs().max().item() < 2e-5         # same integers, libm-level differences in log / sincos
    zz = z.double()
    assert abs(zz.mean().item()) < 4e-3 and abs(zz.var().item() - 1) < 6e-3
    assert abs((zz ** 3).mean().item()) < 2e-2 and abs((zz ** 4).mean().item() - 3) < 5e-2
    assert 4.0 < z.abs().max().item() < 7.0
    # different offsets / seeds are different streams
    out2 = torch.empty(n, device=dev)
    _lib.check(L.tdm_philox_normal_f32(99, 4, _lib.ptr(out2), n, _lib.stream()))
    assert abs(torch.corrcoef(torch.stack([out, out2]))[0, 1].item()) < 5e-3


# ------------------------------------------------- fused draw + q_sample (train)
@pytest.mark.parametrize("B", [1, 37, 512])
def test_draw_q_sample_equals_teacher_forced_q_sample(dev, L, golden_tables, B):
    from tinydiffusionmodels_amd import _lib
    from tinydiffusionmodels_amd.mnist import q_sample
    from tinydiffusionmodels_amd.schedule import device_tables
    tabs = device_tables(dev)
    seed, offset = 0xC0FFEE, 41
    x0 = (torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(B)) * 2 - 1).to(dev)
    st = _state(dev, offset)
    t = torch.empty(B, dtype=torch.long, device=dev)
    noise, xn = torch.empty_like(x0), torch.empty_like(x0)
    _lib.check(L.tdm_ddpm_draw_q_sample_f32(_lib.ptr(x0), _lib.ptr(tabs["sqrt_alphas_cumprod"]),
                                            _lib.ptr(tabs["sqrt_one_minus_alphas_cumprod"]), seed, _lib.ptr(st), _lib.ptr(t),
                                            _lib.ptr(noise), _lib.ptr(xn), B, 784, _lib.stream()))
    assert st.cpu().tolist() == [offset + 1, 0]                               # offset advanced on the device
    assert torch.equal(t.cpu(), O.philox_steps(seed, offset, B))              # integer draws: bit-exact vs numpy
    assert (noise.cpu().reshape(-1) - O.philox_normals(seed, offset, B * 784)).abs().max().item() < 2e-5
    assert torch.equal(xn, q_sample(x0, t, noise))                            # same arithmetic as the parity kernel
    assert torch.equal(xn.cpu(), O.q_sample(x0.cpu(), t.cpu(), noise.cpu(), golden_tables))
    ref_noise = torch.empty_like(noise)
    _lib.check(L.tdm_philox_normal_f32(seed, offset, _lib.ptr(ref_noise), B * 784, _lib.stream()))
    assert torch.equal(noise, ref_noise)


def test_step_index_draws_are_uniform(dev, L):
    t = O.philox_steps(5, 0, 200000)
    assert t.min().item() == 0 and t.max().item() == 999
    counts = torch.bincount(t, minlength=1000).double()
    chi2 = ((counts - 200.0) ** 2 / 200.0).sum().item()
    assert 800 < chi2 < 1200                                                  # 999 dof: mean 999, sd 44.7


# ------------------------------------------- fused draw + p_sample update (sampling)
def test_p_update_philox_equals_teacher_forced_update(dev, L):
    from tinydiffusionmodels_amd import _lib
    from tinydiffusionmodels_amd.schedule import device_tables
    tabs = device_tables(dev)
    sigma0 = tabs["sigma"].clone()
    sigma0[0] = 0.0
    B, seed, offset = 9, 77, 1000
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 1, 28, 28, generator=g).to(dev)
    eps = torch.randn(B, 1, 28, 28, generator=g).to(dev)
    t0 = torch.tensor([999, 500, 2, 1, 0, 0, 17, 1, 333], dtype=torch.long, device=dev)
    t = t0.clone()
    st = _state(dev, offset)
    out = torch.empty_like(x)
    _lib.check(L.tdm_p_sample_update_philox_f32(_lib.ptr(x), _lib.ptr(eps), _lib.ptr(tabs["sqrt_recip_alphas"]),
                                                _lib.ptr(tabs["eps_coef"]), _lib.ptr(sigma0), _lib.ptr(t), seed, _lib.ptr(st),
                                                _lib.ptr(out), B, 784, _lib.stream()))
    z = torch.empty_like(x)
    _lib.check(L.tdm_philox_normal_f32(seed, offset, _lib.ptr(z), B * 784, _lib.stream()))
    want = torch.empty_like(x)
    _lib.check(L.tdm_p_sample_update_pert_f32(_lib.ptr(x), _lib.ptr(eps), _lib.ptr(z), _lib.ptr(tabs["sqrt_recip_alphas"]),
                                              _lib.ptr(tabs["eps_coef"]), _lib.ptr(sigma0), _lib.ptr(t0), 1, _lib.ptr(want), B, 784,
                                              _lib.stream()))
    assert torch.equal(out, want)
    assert torch.equal(t.cpu(), (t0.cpu() - 1).clamp(min=0)) and st.cpu().tolist() == [offset + 1, 0]
    # rows at t == 0 are the posterior mean (the reference's `if t[0] == 0` branch): no noise term
    mean = torch.empty_like(x)
    _lib.check(L.tdm_p_sample_update_pert_f32(_lib.ptr(x), _lib.ptr(eps), None, _lib.ptr(tabs["sqrt_recip_alphas"]),
                                              _lib.ptr(tabs["eps_coef"]), _lib.ptr(sigma0), _lib.ptr(t0), 0, _lib.ptr(mean), B, 784,
                                              _lib.stream()))
    assert torch.equal(out[4:6], mean[4:6])


# --------------------------------------------------------- AdamW, device step count
def test_adamw_device_step_count_equals_host_step(dev, L):
    from tinydiffusionmodels_amd import unet_engine as E
    g = torch.Generator().manual_seed(8)
    n = E.NPARAM
    p0 = torch.randn(n, generator=g).to(dev)
    a, b = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    state = torch.zeros(4, dtype=torch.long, device=dev)
    for step in range(1, 6):
        grad = (torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))).to(dev)
        E.adamw_step(a, grad, ma, va, step, lr=1e-3, grad_scale=0.5)
        E.adamw_step_dev(b, grad, mb, vb, state, lr=1e-3, grad_scale=0.5)
        assert state[:2].cpu().tolist() == [step, 0]
        assert (a - b).abs().max().item() <= 1e-9 + 2e-7 * a.abs().max().item(), step   # (device pow vs host libm pow)
        assert torch.equal(ma, mb) and torch.equal(va, vb)
    assert not torch.equal(a, p0)


# ------------------------------------------------------------- captured train step
def _fresh_trainer(dev, golden_dir, B, graph):
    from tinydiffusionmodels_amd.mnist import SimpleUNet, DDPMTrainer
    z = np.load(f"{golden_dir}/unet_forward.npz")
    m = SimpleUNet()
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")})
    m = m.to(dev)
    torch.manual_seed(2024)                                     # governs the trainer's Philox seed
    return m, DDPMTrainer(m, batch_size=B, lr=1e-3, graph=graph)


def test_train_step_hipgraph_equals_eager_and_teacher_forced(dev, L, golden_dir, golden_tables):
    """Six steps: graph replay vs the same launches issued eagerly -> bitwise equal parameters, losses and AdamW
    step count; and the draws the device made (t, noise), fed to the teacher-forced entry point the parity tests
    pin, give the same loss and gradient bitwise — so the captured step IS the parity-tested step
    (src/mnist.py:152-159)."""
    from tinydiffusionmodels_amd import unet_engine as E
    B = 24
    x0 = (torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
    mg, tg = _fresh_trainer(dev, golden_dir, B, True)
    me, te = _fresh_trainer(dev, golden_dir, B, False)
    assert tg.seed == te.seed
    for k in range(6):
        before = me.flat.detach().clone()
        lg = tg.step(x0).clone()
        le = te.step(x0).clone()
        assert torch.equal(lg, le), k
        assert torch.equal(mg.flat, me.flat), k
        # teacher-forced replay of the device's draws at the pre-step weights
        st = E.TrainState(before, B)
        E.loss_and_grad(before, st, x0, te.state.noise, te.state.t)
        assert torch.equal(st.loss, le) and torch.equal(st.grads, te.grads), k
    assert tg.state.graph is not None and te.state.graph is None
    assert tg.steps_taken == te.steps_taken == 6
    # the drawn t of the last step are the generator's integers (offset 5 = sixth call)
    assert torch.equal(te.state.t.cpu(), O.philox_steps(te.seed, 5, B))
    # and the oracle agrees with the device-drawn step end to end
    p = {k: v.cpu() for k, v in E.state_dict_from_flat(before).items()}
    loss_ref, _ = O.unet_loss_and_grads(p, x0.cpu(), te.state.t.cpu(), te.state.noise.cpu(), golden_tables)
    assert abs(le.item() - loss_ref.item()) < 1e-4 * abs(loss_ref.item())


def test_trainer_serves_ragged_batch_sizes_with_one_optimizer_state(dev, L, golden_dir):
    """src/mnist.py:146-147 has no drop_last: the last batch of an epoch is smaller.  One trainer, two batch sizes,
    shared AdamW moments / step count; the tail step weights the local gradient by B_local / global_batch."""
    m, tr = _fresh_trainer(dev, golden_dir, 16, True)
    g = torch.Generator().manual_seed(4)
    xa = (torch.rand(16, 1, 28, 28, generator=g) * 2 - 1).to(dev)
    xb = (torch.rand(5, 1, 28, 28, generator=g) * 2 - 1).to(dev)
    for _ in range(3):
        tr.step(xa)
    p_before = m.flat.detach().clone()
    tr.step(xb, global_batch=5)                                  # world 1: weight 5/5
    assert tr.steps_taken == 4 and not torch.equal(m.flat, p_before)
    assert tr.state.B == 5 and set(tr._states) == {16, 5}
    assert tr._states[16].m is tr._states[5].m is tr.m
    tr.step(xa)
    assert tr.steps_taken == 5 and torch.isfinite(m.flat).all()


# ------------------------------------------------------------ captured reverse step
def test_graph_sampler_device_noise_equals_eager_chain_with_same_draws(dev, L, golden_dir):
    """The hipGraph reverse loop with device-drawn noise (one C-ABI call per step) against the eager,
    teacher-forced loop fed the SAME Philox draws: bitwise equal after 20 steps (src/mnist.py:190-193)."""
    from tinydiffusionmodels_amd import _lib
    from tinydiffusionmodels_amd.mnist import reverse_diffusion
    m, _ = _fresh_trainer(dev, golden_dir, 4, False)
    n, steps = 6, 20
    x = torch.randn(n, 1, 28, 28, generator=torch.Generator().manual_seed(12)).to(dev)
    torch.manual_seed(7)
    with torch.no_grad():
        got = reverse_diffusion(m, x, t_start=steps - 1, use_graph=True)
    sampler = next(iter(m._samplers.values()))
    assert sampler.graph is not None and sampler.rng_state.cpu().tolist() == [sampler.offset0 + steps, 0]
    assert sampler.t_vec.cpu().tolist() == [0] * n
    zs = []
    for k in range(steps):
        z = torch.empty_like(x)
        _lib.check(L.tdm_philox_normal_f32(sampler.seed, sampler.offset0 + k, _lib.ptr(z), z.numel(), _lib.stream()))
        zs.append(z)
    with torch.no_grad():
        want = reverse_diffusion(m, x, noises=zs, t_start=steps - 1, use_graph=False)
    assert torch.equal(got, want)
    # the cached sampler honours torch.manual_seed: reseeding reproduces the chain bit for bit, not reseeding
    # draws a different one
    with torch.no_grad():
        torch.manual_seed(7)
        again = reverse_diffusion(m, x, t_start=steps - 1, use_graph=True)
        other = reverse_diffusion(m, x, t_start=steps - 1, use_graph=True)
    assert next(iter(m._samplers.values())) is sampler and torch.equal(again, got) and not torch.equal(other, got)


# ---------------------------------------------------------------- RCCL, world 1
def test_native_rccl_world1_allreduce_broadcast_and_graph_capture(dev, L):
    """tdm_ctx_create / tdm_comm_init / tdm_allreduce_sum_f32 over a real RCCL communicator of one rank (all a
    one-GPU box allows): sum over one rank = identity, enqueued on the caller's stream, and capturable into a
    hipGraph together with AdamW (the N-rank form of the train step).  Multi-rank arithmetic is covered on CPU
    (gloo) by tests/test_dp_gloo.py and with two processes on this GPU by tests/test_gpu_dp.py."""
    from tinydiffusionmodels_amd import dp, unet_engine as E
    assert L.tdm_comm_rccl_version() >= 20000
    uid = dp.NativeComm.make_unique_id()
    assert len(uid) == L.tdm_comm_unique_id_bytes() == 128
    comm = dp.NativeComm(0, 0, 1, uid)
    assert L.tdm_comm_world(comm.ctx) == 1 and L.tdm_comm_rank(comm.ctx) == 0
    buf = torch.randn(E.NPARAM, generator=torch.Generator().manual_seed(0)).to(dev)
    want = buf.clone()
    comm.allreduce_sum_(buf)
    comm.broadcast_(buf, 0)
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    # inside a hipGraph, followed by the device-step AdamW
    p = torch.zeros(E.NPARAM, device=dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.zeros(4, dtype=torch.long, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.allreduce_sum_(buf)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        comm.allreduce_sum_(buf)
        E.adamw_step_dev(p, buf, m, v, state, lr=1e-3)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert state[:2].cpu().tolist() == [3, 0] and torch.equal(buf, want) and torch.isfinite(p).all() and p.abs().max() > 0
    comm.close()
    # a context without a communicator refuses the collective with rc != 0 and a message (no abort, no hang)
    from tinydiffusionmodels_amd import _lib
    ctx = ctypes.c_void_p()
    _lib.check(L.tdm_ctx_create(0, ctypes.byref(ctx)))
    assert L.tdm_allreduce_sum_f32(ctx, _lib.ptr(buf), buf.numel(), _lib.stream()) != 0
    assert b"not initialised" in L.tdm_last_error()
    assert L.tdm_ctx_destroy(ctx) == 0


def test_graph_replayed_training_converges(dev):
    """Soak of the train steps in both issue forms (src/mnist.py:150-160, src/shakespeare.py:228-236 as the trainers run them: device draws,
    MSE backward in the forward's last epilogue, AdamW with its step count on the device): 1500 UNet steps at B = 512 on a
    fixed synthetic image set and 300 denoiser steps on fixed embeddings — losses stay finite and fall far below their start."""
    from tinydiffusionmodels_amd.mnist import SimpleUNet, DDPMTrainer
    from tinydiffusionmodels_amd.shakespeare import TinyTransformer, DenoiserTrainer
    g = torch.Generator(device=dev).manual_seed(1)
    data = torch.nn.functional.interpolate(torch.rand(2048, 1, 7, 7, device=dev, generator=g), size=28, mode="bilinear") * 2 - 1
    for graph in (True, None):     # hipGraph replays (one queue), then the default: eager issue, weight gradients on the side stream
        torch.manual_seed(0)
        m = SimpleUNet().to(dev)
        tr = DDPMTrainer(m, batch_size=512, lr=1e-3, graph=graph)
        first = last = None
        for step in range(1500):
            loss = tr.step(data[torch.randint(0, 2048, (512,), device=dev, generator=g)])
            if step == 0:
                first = loss.item()
        last = loss.item()
        assert np.isfinite(first) and np.isfinite(last) and last < 0.1 * first, (graph, first, last)
        assert torch.isfinite(m.flat).all() and tr.steps_taken == 1500
    t = TinyTransformer(64, dropout=0.1).to(dev)
    t.train()
    tt = DenoiserTrainer(t, 16, 32, lr=1e-3)
    x0 = torch.randn(16, 32, 64, device=dev, generator=g) * 0.5
    l0 = tt.step(x0).item()
    for _ in range(299):
        l1 = tt.step(x0)
    l1 = l1.item()
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < 0.7 * l0, (l0, l1)
    assert torch.isfinite(t.flat).all() and tt.steps_taken == 300


def test_early_gradient_parts_are_bit_identical_and_the_event_orders_a_collective_stream(dev, L):
    """tdm_set_early_grads / tdm_unet_wait_early_grads (data-parallel training: the all-reduce under the backward; no
    counterpart in the reference, deployment/configs/mnist-training.yaml:5-6 is one GPU).  The slab reduction in two parts
    gives the same flat gradient bit for bit as the one reduction, with the backward on two queues and on one; the wait call
    reports an event exactly once per backward call and none for a captured call; and the native early all-reduce — a real
    RCCL communicator of one rank, the collective side stream behind the library's event — leaves the gradient as it is and
    the caller's stream ordered behind both collectives (AdamW right after it sees the same bits as without)."""
    from tinydiffusionmodels_amd import _lib, dp, unet_engine as E
    from tinydiffusionmodels_amd.mnist import SimpleUNet
    B = 96
    torch.manual_seed(4)
    model = SimpleUNet().to(dev)
    flat = model.flat.detach()
    g = torch.Generator(device=dev).manual_seed(8)
    x0 = torch.rand(B, 1, 28, 28, device=dev, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    noise = torch.randn(B, 1, 28, 28, device=dev, generator=g)
    off = int(L.tdm_unet_early_grad_offset())
    assert off == 288 + 32 + 9216 + 32 + 32 + 32 + 32 + 32 and 0 < off < E.NPARAM      # rb1's tensors come first in state_dict order
    was_ov = L.tdm_get_bwd_overlap()
    grads = {}
    try:
        for ov in (1, 0):
            for early in (0, 1):
                _lib.check(L.tdm_set_bwd_overlap(ov))
                _lib.check(L.tdm_set_early_grads(early))
                assert L.tdm_get_early_grads() == early
                st = E.TrainState(flat, B)
                st.grads.fill_(float("nan"))
                loss = E.loss_and_grad(flat, st, x0, noise, t)
                side = torch.cuda.Stream()
                rc = L.tdm_unet_wait_early_grads(side.cuda_stream)
                assert rc == early and L.tdm_unet_wait_early_grads(side.cuda_stream) == 0      # one event per call, consumed
                if early:   # what the side stream sees behind the event: the early part final (the rest may still be in flight)
                    with torch.cuda.stream(side):
                        hi = st.grads[off:].clone()
                    torch.cuda.synchronize()
                    assert torch.equal(hi, st.grads[off:]) and torch.isfinite(hi).all()
                torch.cuda.synchronize()
                grads[(ov, early)] = (st.grads.clone(), loss.clone())
        ref = grads[(0, 0)]
        for k, v in grads.items():
            assert torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1]), k
        # a captured call keeps the one reduction: no event
        _lib.check(L.tdm_set_early_grads(1))
        _lib.check(L.tdm_set_bwd_overlap(0))
        st = E.TrainState(flat, B)
        E.loss_and_grad(flat, st, x0, noise, t)                  # warm: lazily set kernel attributes are not capturable
        L.tdm_unet_wait_early_grads(torch.cuda.current_stream().cuda_stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            E.loss_and_grad(flat, st, x0, noise, t)
        assert L.tdm_unet_wait_early_grads(torch.cuda.current_stream().cuda_stream) == 0
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(st.grads, ref[0])
        # the native early all-reduce (RCCL, one rank) behind the library's event
        _lib.check(L.tdm_set_bwd_overlap(1))
        comm = dp.NativeComm(0, 0, 1, dp.NativeComm.make_unique_id())
        st = E.TrainState(flat, B)
        p = flat.clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        state = torch.zeros(4, dtype=torch.long, device=dev)
        E.loss_and_grad(flat, st, x0, noise, t)
        dp._early_native(comm, st.grads, off, lambda s: L.tdm_unet_wait_early_grads(s.cuda_stream) == 1)
        E.adamw_step_dev(p, st.grads, m, v, state, lr=1e-3)
        torch.cuda.synchronize()
        assert torch.equal(st.grads, ref[0])
        p2, m2, v2, state2 = flat.clone(), torch.zeros_like(p), torch.zeros_like(p), torch.zeros(4, dtype=torch.long, device=dev)
        E.adamw_step_dev(p2, ref[0], m2, v2, state2, lr=1e-3)
        torch.cuda.synchronize()
        assert torch.equal(p, p2)
        comm.close()
    finally:
        L.tdm_set_early_grads(0)
        L.tdm_set_bwd_overlap(was_ov)


@pytest.mark.gpu
def test_two_threads_two_contexts_keep_their_own_arithmetic_and_side_queue(dev):
    """include/tdm_hip.h "Contexts": selector state and the side queue live in explicit tdm_ctx objects.  Two host threads bind
    two contexts — exact fp32 and the default bf16x3 — and run UNet loss + gradient calls concurrently on their own streams: each
    gets the bits of its arithmetic run alone, the main thread's default context is untouched, a context that is current on one
    thread cannot be bound or destroyed by another, and destroying a context releases the side queue it created."""
    import threading
    from tinydiffusionmodels_amd import _lib, unet_engine as E
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(0)
    B = 37
    params = torch.randn(E.NPARAM, device=dev, generator=g) * 0.05
    x = torch.rand(B, 1, 28, 28, device=dev, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    deps = torch.randn(B, 1, 28, 28, device=dev, generator=g) / B

    def fwd_bwd():
        """predicted noise + flat gradient in the calling thread's current context (its own workspace and slabs)"""
        ws = E.UNetWorkspace(B, dev, training=True)
        eps = E.unet_forward(params, x, t, ws, save=True)
        slabs = torch.empty(L.tdm_unet_slab_floats(), device=dev)
        grads = torch.empty(E.NPARAM, device=dev)
        _lib.check(L.tdm_unet_bwd_f32(_lib.ptr(params), _lib.ptr(x), _lib.ptr(deps), _lib.ptr(grads), _lib.ptr(ws.ws), _lib.ptr(slabs), B,
                                      _lib.stream()), "unet_bwd")
        return eps, grads

    def run(modes):
        with _lib.use_arithmetic(modes):
            return fwd_bwd()

    ref = {m: run(m) for m in ((0, 1, 2), (2, 1, 2))}          # alone, in the main thread's default context
    assert _lib.arithmetic() == (2, 1, 2)
    ctxs = {(0, 1, 2): _lib.Context(0, arithmetic=(0, 1, 2)), (2, 1, 2): _lib.Context(0, arithmetic=(2, 1, 2))}
    out, errs = {}, []
    gate = threading.Barrier(2)

    def worker(modes):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                with ctxs[modes] as c:
                    assert _lib.arithmetic() == modes and L.tdm_ctx_current() == c.handle.value
                    gate.wait()
                    if modes == (0, 1, 2):                      # the other thread has its context current: refused here
                        assert L.tdm_ctx_make_current(ctxs[(2, 1, 2)].handle) != 0
                        assert L.tdm_ctx_destroy(ctxs[(2, 1, 2)].handle) != 0
                        assert L.tdm_ctx_make_current(c.handle) == 0
                    gate.wait()
                    for _ in range(3):
                        out[modes] = fwd_bwd()
                    torch.cuda.current_stream().synchronize()
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            try:
                gate.abort()
            except Exception:
                pass

    th = [threading.Thread(target=worker, args=(m,)) for m in ctxs]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    assert _lib.arithmetic() == (2, 1, 2) and L.tdm_ctx_current() is None      # the main thread never bound anything
    for m in ctxs:
        assert torch.equal(out[m][0], ref[m][0]) and torch.equal(out[m][1], ref[m][1]), m
    assert not torch.equal(out[(0, 1, 2)][1], out[(2, 1, 2)][1])               # (the two arithmetics do differ)
    for c in ctxs.values():
        c.close()                                                              # joins nothing: idle between calls; frees the side queue
