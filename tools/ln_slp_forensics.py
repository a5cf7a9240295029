#!/usr/bin/env python3
"""Forensics of the round-4 LayerNorm-backward miscompute: ONE run of the SLP-vectorised build of ln_bwd_kernel<1>
(a diagnostic library, tools/micro/libtdm_slp.so = the shipped objects with transformer.hip compiled under round 4's early flags)
next to the foreign token-major GEMM stream, with the wrong rows decomposed on the host.

    TDM_HIP_LIB=tools/micro/libtdm_slp.so python tools/ln_slp_forensics.py

For a row, ds[c] = rs * (dd[c] * gamma[c] - c1 - xhat[c] * c2) with c1 = mean(g), c2 = mean(g * xhat), g = dd * gamma.  A wrong
row sum shifts the whole row: delta ds[c] = -rs * (dc1 + xhat[c] * dc2).  A least-squares fit of (dc1, dc2) per wrong row and the
residual say whether the row sums or single elements were wrong; the un-normalised error D * dc2 is then compared with what
specific stale-register hypotheses predict from the inputs (lane L of the wave holds columns 4L .. 4L+3 = x, y, z, w)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinydiffusionmodels_amd import _lib   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    print("library:", _lib.LIB_PATH)
    _lib.check(L.tdm_set_gemm_mode(1))
    M, D = 32768, 256
    g = torch.Generator(device=dev).manual_seed(1)
    Ms = 32768
    dy_s = torch.randn(Ms, 2048, device=dev, generator=g) * 0.01
    x_s = torch.randn(Ms, 256, device=dev, generator=g)

    def s16(t):
        o = torch.empty_like(t)
        _lib.check(L.tdm_split_s16_f32(_lib.ptr(t), _lib.ptr(o), t.numel(), _lib.stream()), "split")
        return o
    dy16_s, x16_s = s16(dy_s), s16(x_s)
    slab = torch.empty(8, 2048, 256, device=dev)
    side = torch.cuda.Stream()
    dy = torch.randn(M, D, device=dev, generator=g)
    s = torch.randn(M, D, device=dev, generator=g)
    mean = s.mean(1).contiguous()
    rstd = (1.0 / torch.sqrt(s.var(1, unbiased=False) + 1e-5)).contiguous()
    gamma = torch.randn(D, device=dev, generator=g)
    ds = torch.empty(M, D, device=dev)
    dgb = torch.empty(2, D, device=dev)
    scratch = torch.empty(L.tdm_layernorm_scratch_floats(D), device=dev)

    def ln():
        _lib.check(L.tdm_layernorm_residual_bwd_f32(_lib.ptr(dy), _lib.ptr(s), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(ds),
                                                    _lib.ptr(dgb), _lib.ptr(scratch), M, D, _lib.stream()), "ln_bwd")

    def run(n):
        side.wait_stream(torch.cuda.current_stream())
        if n:
            with torch.cuda.stream(side):
                for _ in range(n):
                    _lib.check(L.tdm_gemm_f32(_lib.ptr(dy16_s), 1, 2048, _lib.ptr(x16_s), 256, 1, _lib.ptr(slab), 256, None, None, 2048, 256, Ms, 2, 8,
                                              2048 * 256, side.cuda_stream), "tn gemm")
        ln()
        torch.cuda.synchronize()
    run(0)
    ref = ds.clone()
    run(0)
    print("quiet repeat bit-identical:", torch.equal(ds, ref))
    # fp64 host quantities of every row
    dd = dy.double().cpu(); sv = s.double().cpu(); mu = mean.double().cpu()[:, None]; rs = rstd.double().cpu()[:, None]
    gm = gamma.double().cpu()[None, :]
    xh = (sv - mu) * rs
    gg = dd * gm
    ref_c = ref.double().cpu()
    dgb_ref = dgb.clone()
    c1_all = gg.mean(1, keepdim=True)
    c2_all = (gg * xh).mean(1, keepdim=True)
    seen = 0
    clean = slice(0, 192)                                            # lanes 0-47: taken to be right (checked by the residual there)
    for rep in range(6):
        ds.zero_()
        run(6)
        got = ds.double().cpu()
        bad_rows = (got != ref_c).any(1).nonzero().flatten().tolist()
        print(f"rep {rep}: {len(bad_rows)} wrong rows; row % 4 histogram {[sum(1 for r in bad_rows if r % 4 == k) for k in range(4)]}; "
              f"dgamma equal {torch.equal(dgb[0], dgb_ref[0])} (max rel diff {((dgb[0] - dgb_ref[0]).abs().max() / dgb_ref[0].abs().max()).item():.2e}), "
              f"dbeta equal {torch.equal(dgb[1], dgb_ref[1])} (max rel diff {((dgb[1] - dgb_ref[1]).abs().max() / dgb_ref[1].abs().max()).item():.2e})", flush=True)
        for r in bad_rows[:10]:
            seen += 1
            dlt = got[r] - ref_c[r]
            A = torch.stack([-rs[r] * torch.ones(D, dtype=torch.float64), -rs[r] * xh[r]], 1)
            sol = torch.linalg.lstsq(A[clean], dlt[clean, None]).solution.flatten()      # row-sum errors seen by the clean lanes
            res = dlt - A @ sol
            dc1, dc2 = sol[0].item(), sol[1].item()
            c1p, c2p = c1_all[r].item() + dc1, c2_all[r].item() + dc2
            tol = 50 * res[clean].abs().max().item() + 1e-7
            S = (res.abs() > tol).nonzero().flatten().tolist()
            print(f"  row {r} (rr {r % 4}, wave-iteration row0 {r - r % 4}): D*dc1 {dc1 * D:+.5e} D*dc2 {dc2 * D:+.5e}; residual in lanes 0-47 {res[clean].abs().max():.1e}; "
                  f"{len(S)} element errors at columns {S[:24]} -> lanes {sorted(set(c // 4 for c in S))[:16]}, components {sorted(set('xyzw'[c % 4] for c in S))}")
            # under E2 (dd wrong in S): e_c = rs * ddelta_c * gamma_c; consistency: D*dc1 = sum ddelta*gamma, D*dc2 = sum ddelta*gamma*xhat
            e = res[S]
            dgam = e / rs[r]                                           # = delta(dd*gamma) if dd was wrong
            chk1, chk2 = dgam.sum().item(), (dgam * xh[r][S]).sum().item()
            # under E1 (xhat wrong in S): e_c = -rs * dx_c * c2'; consistency: D*dc2 = sum g * dx, dc1 = 0
            dxh = -e / (rs[r] * c2p)
            chk3 = (gg[r][S] * dxh).sum().item()
            print(f"      if dd was wrong:   sum d(g) {chk1:+.5e} (D*dc1 {dc1 * D:+.5e})   sum d(g)*xhat {chk2:+.5e} (D*dc2 {dc2 * D:+.5e})")
            print(f"      if xhat was wrong: sum g*d(xhat) {chk3:+.5e} (D*dc2 {dc2 * D:+.5e}); D*dc1 would be 0")
            if S:
                # the element errors are one register's lanes 48-63: a CONSTANT offset of xhat = a different (uniform) mean subtracted
                off = dxh
                mu_imp = mu[r, 0].item() - (off.mean() / rs[r, 0]).item()
                r0 = r - r % 4
                print(f"      d(xhat) over the 16 lanes: mean {off.mean().item():+.6e} spread {(off.max() - off.min()).item():.1e} -> implied mean' {mu_imp:+.6e}; "
                      f"means of the wave's rows {[round(mu[r0 + k, 0].item(), 6) for k in range(4)]} rstd {[round(rs[r0 + k, 0].item(), 6) for k in range(4)]}; "
                      f"previous sweep's rows (row0 - 16384) means {[round(mu[r0 - 16384 + k, 0].item(), 6) for k in range(4)] if r0 >= 16384 else None}; "
                      f"c1 {c1_all[r].item():+.6e} c2 {c2_all[r].item():+.6e} c1' {c1p:+.6e} c2' {c2p:+.6e}; e/rs mean {(e / rs[r]).mean().item():+.6e} spread {((e / rs[r]).max() - (e / rs[r]).min()).item():.1e}; "
                      f"1/D {1.0 / D:.6e}")
            for c in S[:2]:
                l, k = c // 4, c % 4
                ddp = (dd[r, c] * gm[0, c] + dgam[S.index(c)]) / gm[0, c]         # implied dd'
                cands = {"dd same lane rr-1": dd[r - 1, c].item(), "dd rr+1": dd[min(r + 1, M - 1), c].item(), "dd comp+1": dd[r, c + 1].item(),
                         "dd prev sweep (row-16384)": dd[r - 16384, c].item() if r >= 16384 else float("nan"), "s (raw) same": sv[r, c].item(),
                         "s rr-1": sv[r - 1, c].item(), "xhat same": xh[r, c].item(), "zero": 0.0}
                best = min(cands.items(), key=lambda kv: abs(kv[1] - ddp.item()) if kv[1] == kv[1] else 1e9)
                xp = xh[r, c] + dxh[S.index(c)]
                candx = {"xhat rr-1": xh[r - 1, c].item(), "s-mean (no rstd)": (sv[r, c] - mu[r, 0]).item(), "s raw": sv[r, c].item(),
                         "(s-mean_{rr-1})*rstd_{rr-1} of row rr-1's s": xh[r - 1, c].item(), "(s - mean_rr-1)*rs_rr": ((sv[r, c] - mu[r - 1, 0]) * rs[r, 0]).item(),
                         "(s - mean_rr)*rs_rr-1": ((sv[r, c] - mu[r, 0]) * rs[r - 1, 0]).item(), "g.x+g.y": (gg[r, c - k] + gg[r, c - k + 1]).item(),
                         "g.z+g.w": (gg[r, c - k + 2] + gg[r, c - k + 3]).item(), "zero": 0.0}
                bestx = min(candx.items(), key=lambda kv: abs(kv[1] - xp.item()))
                print(f"        col {c} (lane {l} {'xyzw'[k]}): true dd {dd[r, c].item():+.6f} implied dd' {ddp.item():+.6f} (closest: {best[0]} = {best[1]:+.6f}); "
                      f"true xhat {xh[r, c].item():+.6f} implied xhat' {xp.item():+.6f} (closest: {bestx[0]} = {bestx[1]:+.6f})")
        if seen >= 16:
            break


if __name__ == "__main__":
    main()
