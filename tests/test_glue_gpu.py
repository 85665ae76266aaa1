"""GPU, round 6: the stage boundaries as one launch each (csrc/glue.hip) -- tulip_merge_fwd, tulip_merge_bwd, tulip_unmerge_skip_fwd,
tulip_skip_unmerge_bwd -- each against
  (a) the oracle's functions for the same boundary (oracle/tulip_oracle.py patch_merging / patch_unmerging + linear(cat), tulip.py:
      101-106, 117-123, 713-716) with the bf16 rounding model, forward and autograd;
  (b) the launch sequences they replace (tulip_layernorm_fwd(merge) + tulip_gemm_bf16, ...): same rounding points, so the results
      agree to fp32 summation order (bf16 tensors: a vanishing fraction of 1-ulp flips).
Tolerances: fp32 outputs of bf16 GEMMs 1e-3 of the scale vs the oracle, 2e-5 vs the sequence; bf16 outputs 1 ulp."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tulip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from tulip_amd import ops as _ops
    from tulip_amd import _lib
    _lib.load()
    return _ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + int(np.prod(shape)) % 1000)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def bf(t):
    return t.to(torch.bfloat16)


def close(a, b, rtol, atol_scale, what="", frac=0.0):
    a, b = a.float().reshape(-1), b.float().reshape(-1).to(a.device)
    assert torch.isfinite(a).all(), what
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs()
    bad = err > rtol * b.abs() + atol_scale * scale
    assert bad.float().mean().item() <= frac, (f"{what}: {bad.sum().item()}/{bad.numel()} out of tolerance; max err "
                                               f"{err.max().item():.4e} (scale {scale:.3e})")


def packed(ops, w, transpose=False):
    """fragment-major copy of the bf16 matrix w (or of its transpose), tulip_pack_bf16_multi"""
    w = w.contiguous()
    dst = torch.zeros(w.numel(), dtype=torch.bfloat16, device=DEV)
    items, n = ops.pack_items([(w, dst, w.shape[0], w.shape[1], int(transpose))])
    ops.pack_bf16_multi(items, n)
    return dst


def merged(x):
    return torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)      # tulip.py:94-98


# (B, H, W, Cin): KITTI grids at batch 2 / 8, the 16x2048 / 32x2048 grids of CARLA / DurLAR (W = 512 tokens at level 0), an odd batch
# (sample / row arithmetic off the power-of-two path), and the row counts at which the launcher changes its (row block, slices) choice
MERGE_CASES = [(2, 16, 256, 96), (8, 16, 256, 96), (2, 8, 128, 192), (8, 32, 512, 192), (2, 4, 64, 384), (8, 4, 64, 384),
               (16, 8, 256, 384), (3, 16, 256, 96), (2, 32, 512, 96), (3, 8, 128, 192), (6, 4, 64, 384), (2, 8, 256, 384)]


@pytest.mark.parametrize("B,H,W,Cin", MERGE_CASES)
def test_merge_fwd(ops, B, H, W, Cin):
    assert ops.merge_fwd_supported(Cin, B, H, W)
    K, N, rows = 4 * Cin, 2 * Cin, B * (H // 2) * (W // 2)
    x = rnd(B, H, W, Cin, seed=1)
    gamma, beta = 1 + 0.1 * rnd(K, seed=2), 0.1 * rnd(K, seed=3)
    w = bf(rnd(N, K, scale=K ** -0.5, seed=4))
    wp = packed(ops, w)
    xm = torch.empty(rows, K, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    y = torch.empty(rows, N, device=DEV)
    cat = torch.zeros(rows, 2 * N, dtype=torch.bfloat16, device=DEV)
    ops.merge_fwd(x=x, gamma=gamma, beta=beta, w_packed=wp, xm=xm, mean=mean, rstd=rstd, y=y,
                  y_bf16=cat.data_ptr() + 2 * N, ld_bf16=2 * N, B=B, H=H, W=W, Cin=Cin, eps=1e-6)
    torch.cuda.synchronize()
    # (b) the two-launch form
    xm2 = torch.empty_like(xm); m2, r2 = torch.empty_like(mean), torch.empty_like(rstd); y2 = torch.empty_like(y)
    cat2 = torch.zeros_like(cat)
    ops.layernorm_fwd(x, gamma, beta, xm2, m2, r2, rows, K, 1e-6, merge=True, B=B, H=H, W=W)
    ops.gemm(xm2, w, rows, N, K, lda=K, ldb=K, epi=ops.EPI_F32, out=y2, out2=cat2.data_ptr() + 2 * N, ldo2=2 * N)
    torch.cuda.synchronize()
    close(mean, m2, 1e-5, 2e-6, "mean"); close(rstd, r2, 1e-5, 1e-6, "rstd")        # (fp32 summation order: 16 lanes x K/64 chunks here)
    close(xm, xm2, 2 ** -7, 1e-6, "xm vs sequence", frac=1e-4)
    # the GEMM itself, from THIS launch's normalised rows (isolated 1-ulp flips of xm between the two forms move whole rows of y)
    close(y, xm.float() @ w.float().t(), 1e-4, 2e-5, "y vs fp32 matmul of its own xm")
    close(y, y2, 1e-3, 1e-3, "y vs sequence")
    close(cat[:, N:], y.bfloat16(), 0, 0, "bf16 copy = bf16(y)")
    close(cat, cat2, 2 ** -6, 1e-3, "bf16 copy vs sequence")
    assert torch.equal(cat[:, :N], torch.zeros_like(cat[:, :N]))          # the other half of the concat rows is not touched
    # (a) the oracle
    sd = {"d.norm.weight": gamma.cpu(), "d.norm.bias": beta.cpu(), "d.reduction.weight": w.float().cpu()}
    ref = O.patch_merging(O._Prec(True), sd, "d", O.TulipConfig(), x.cpu()).reshape(rows, N)
    close(y, ref, 1e-3, 1e-3, "y vs oracle")


@pytest.mark.parametrize("B,H,W,Cp,skip", [(2, 16, 256, 96, True), (8, 16, 256, 96, True), (2, 8, 128, 192, True),
                                            (8, 8, 128, 192, True), (2, 16, 256, 96, False), (4, 8, 128, 192, False),
                                            (3, 16, 256, 96, True), (2, 32, 512, 96, True), (3, 16, 256, 192, True)])
def test_merge_bwd(ops, B, H, W, Cp, skip):
    """(B,H,W,Cp): the finer stage.  skip: the x_save half of the skip Linear's input gradient is added in front."""
    assert ops.merge_bwd_supported(Cp, B, H, W)
    Cs, K4, rows = 2 * Cp, 4 * Cp, B * (H // 2) * (W // 2)
    xprev = rnd(B, H, W, Cp, seed=1)
    gamma, beta = 1 + 0.1 * rnd(K4, seed=2), 0.1 * rnd(K4, seed=3)
    wred = bf(rnd(Cs, K4, scale=K4 ** -0.5, seed=4))
    wskip = bf(rnd(Cs, 2 * Cs, scale=(2 * Cs) ** -0.5, seed=5))
    dx_in = rnd(rows, Cs, seed=6)
    dys = bf(rnd(rows, Cs, seed=7))
    scale = (1 + torch.arange(B, device=DEV).float() * 0.25)          # a per-sample DropPath scale for the cast output
    tok = H * W
    # forward statistics
    xm = torch.empty(rows, K4, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_fwd(xprev, gamma, beta, xm, mean, rstd, rows, K4, 1e-6, merge=True, B=B, H=H, W=W)
    # ---- the sequence
    dx2 = dx_in.clone()
    dyb2 = torch.empty(rows, Cs, dtype=torch.bfloat16, device=DEV)
    if skip:
        ops.gemm(dys, wskip.data_ptr() + 2 * Cs, rows, Cs, Cs, lda=Cs, ldb=2 * Cs, b_trans=True, epi=ops.EPI_F32, out=dx2, ldo=Cs,
                 accumulate=True, out2=dyb2, ldo2=Cs)
    else:
        dyb2.copy_(dx_in)
    dxm = torch.empty(rows, K4, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dyb2, wred, rows, K4, Cs, lda=Cs, ldb=K4, b_trans=True, epi=ops.EPI_BF16, out=dxm, ldo=K4)
    R2 = ops.layernorm_bwd_partial_rows(rows, K4)
    part2 = torch.zeros(R2, 2 * K4, device=DEV)
    dxp2, cast2 = torch.empty_like(xprev), torch.empty(B, H, W, Cp, dtype=torch.bfloat16, device=DEV)
    ops.layernorm_bwd(dxm, xprev, mean, rstd, gamma, None, dxp2, rows, K4, merge=True, B=B, H=H, W=W, param_partials=part2,
                      dx_bf16=cast2, cast_rowscale=scale, cast_rows_per_sample=tok)
    # ---- one launch
    R = ops.merge_bwd_partial_rows(Cp, B, H, W)
    part = torch.full((R, 2 * K4), float("nan"), device=DEV)
    dxp, cast = torch.full_like(xprev, float("nan")), torch.zeros(B, H, W, Cp, dtype=torch.bfloat16, device=DEV)
    dyb = torch.zeros(rows, Cs, dtype=torch.bfloat16, device=DEV) if skip else dyb2.clone()
    ops.merge_bwd(dx_in=dx_in if skip else None, dy_skip=dys if skip else None,
                  w_skip_t_packed=packed(ops, wskip, True) if skip else None, dyb=dyb, w_red_t_packed=packed(ops, wred, True),
                  x_prev=xprev, mean=mean, rstd=rstd, gamma=gamma, dx_prev=dxp, param_partials=part, dx_bf16=cast,
                  cast_rowscale=scale, cast_rows_per_sample=tok, B=B, H=H, W=W, Cp=Cp)
    torch.cuda.synchronize()
    close(dyb, dyb2, 2 ** -7, 1e-6, "dyb vs sequence", frac=1e-4)
    close(dxp, dxp2, 1e-3, 2e-4, "dx_prev vs sequence", frac=2e-3)            # (isolated bf16 flips of d(norm out) move a row's sums)
    close(cast, cast2, 2 ** -6, 1e-3, "cast vs sequence", frac=2e-3)
    close(part.sum(0), part2.sum(0), 2e-3, 1e-3, "[dgamma | dbeta] vs sequence")
    # ---- the oracle's autograd of the same boundary
    xr = xprev.cpu().clone().requires_grad_(True)
    gr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    pr = O._Prec(True)
    sd = {"d.norm.weight": gr, "d.norm.bias": br, "d.reduction.weight": wred.float().cpu()}
    y = O.patch_merging(pr, sd, "d", O.TulipConfig(), xr).reshape(rows, Cs)
    if skip:
        g_up = (dx_in.cpu() + dys.float().cpu() @ wskip.float().cpu()[:, Cs:]).bfloat16().float()
    else:
        g_up = dyb2.float().cpu()
    y.backward(g_up)
    close(dxp, xr.grad, 2e-2, 4e-3, "dx_prev vs oracle autograd", frac=1e-3)
    rel = ((dxp.cpu() - xr.grad).norm() / xr.grad.norm()).item()
    assert rel <= 4e-3, rel
    for got, want, nm in ((part[:, :K4].sum(0), gr.grad, "dgamma"), (part[:, K4:].sum(0), br.grad, "dbeta")):
        r = ((got.cpu() - want).norm() / want.norm()).item()
        assert r <= 4e-3, (nm, r)


UNMERGE_CASES = [(2, 8, 128, 192), (8, 8, 128, 192), (2, 4, 64, 384), (8, 4, 64, 384), (3, 8, 128, 192), (2, 16, 256, 192), (3, 8, 128, 384)]


@pytest.mark.parametrize("B,H,W,C", UNMERGE_CASES)
def test_unmerge_skip_fwd(ops, B, H, W, C):
    """(B,H,W,C): the COARSE stage.  expand C -> 2C, PixelShuffle(2), cat with x_save, skip Linear C -> C/2."""
    assert ops.unmerge_skip_supported(C, B, H, W)
    Cf, M = C // 2, B * H * W
    xb = bf(rnd(M, C, seed=1))
    wexp, bexp = bf(rnd(2 * C, C, scale=C ** -0.5, seed=2)), 0.1 * rnd(2 * C, seed=3)
    wskip, bskip = bf(rnd(Cf, C, scale=C ** -0.5, seed=4)), 0.1 * rnd(Cf, seed=5)
    xsave = bf(rnd(4 * M, Cf, seed=6))
    cat = torch.zeros(4 * M, C, dtype=torch.bfloat16, device=DEV); cat[:, Cf:] = xsave
    cat2 = cat.clone()
    out, out2 = torch.full((4 * M, Cf), float("nan"), device=DEV), torch.empty(4 * M, Cf, device=DEV)
    ops.unmerge_skip_fwd(x_bf16=xb, w_expand_packed=packed(ops, wexp), b_expand=bexp, cat=cat, w_skip_packed=packed(ops, wskip),
                         b_skip=bskip, out=out, B=B, H=H, W=W, C=C)
    ops.gemm(xb, wexp, M, 2 * C, C, lda=C, ldb=C, epi=ops.EPI_PIXSHUF2_F32, bias=bexp, out=None, out2=cat2, ldo2=C, psH=H, psW=W)
    ops.gemm(cat2, wskip, 4 * M, Cf, C, lda=C, ldb=C, epi=ops.EPI_F32, bias=bskip, out=out2)
    torch.cuda.synchronize()
    assert torch.equal(cat[:, Cf:], xsave)
    close(cat, cat2, 2 ** -7, 1e-6, "concat rows vs sequence", frac=1e-4)
    close(out, out2, 1e-3, 1e-4, "out vs sequence", frac=1e-3)
    # the oracle
    pr = O._Prec(True)
    sd = {"u.expand.weight": wexp.float().cpu().reshape(2 * C, C, 1, 1), "u.expand.bias": bexp.cpu()}
    z = O.patch_unmerging(pr, sd, "u", xb.float().cpu().reshape(B, H, W, C))
    ref = O.linear(pr, torch.cat([z, xsave.float().cpu().reshape(B, 2 * H, 2 * W, Cf)], -1), wskip.float().cpu(), bskip.cpu())
    close(out, ref.reshape(4 * M, Cf), 2e-3, 2e-3, "out vs oracle")


@pytest.mark.parametrize("B,H,W,C", UNMERGE_CASES)
def test_skip_unmerge_bwd(ops, B, H, W, C):
    Cf, M = C // 2, B * H * W
    dys = bf(rnd(4 * M, Cf, seed=1))
    wexp = bf(rnd(2 * C, C, scale=C ** -0.5, seed=2))
    wskip = bf(rnd(Cf, C, scale=C ** -0.5, seed=4))
    scale = (1 + torch.arange(B, device=DEV).float() * 0.25)
    dz, dz2 = torch.zeros(M, 2 * C, dtype=torch.bfloat16, device=DEV), torch.zeros(M, 2 * C, dtype=torch.bfloat16, device=DEV)
    dx, dx2 = torch.full((M, C), float("nan"), device=DEV), torch.empty(M, C, device=DEV)
    cast, cast2 = torch.zeros(M, C, dtype=torch.bfloat16, device=DEV), torch.zeros(M, C, dtype=torch.bfloat16, device=DEV)
    ops.skip_unmerge_bwd(dy_skip=dys, w_skip_t_packed=packed(ops, wskip, True), dz=dz, w_expand_t_packed=packed(ops, wexp, True),
                         dx=dx, dx_bf16=cast, cast_rowscale=scale, cast_rows_per_sample=H * W, B=B, H=H, W=W, C=C)
    ops.gemm(dys, wskip, 4 * M, Cf, Cf, lda=Cf, ldb=C, b_trans=True, epi=ops.EPI_UNSHUF2_BF16, out=dz2, ldo=2 * C, psH=H, psW=W)
    ops.gemm(dz2, wexp, M, C, 2 * C, lda=2 * C, ldb=C, b_trans=True, epi=ops.EPI_F32, out=dx2, ldo=C, out2=cast2, ldo2=C,
             rowscale=scale, rows_per_sample=H * W)
    torch.cuda.synchronize()
    close(dz, dz2, 2 ** -7, 1e-6, "dz vs sequence", frac=1e-4)
    close(dx, dx2, 1e-3, 1e-4, "dx vs sequence", frac=1e-3)
    close(cast, cast2, 2 ** -6, 1e-3, "cast vs sequence", frac=1e-3)
    # the oracle's autograd: d/dx of <dys, linear(cat[unmerge(x), .])> restricted to the unmerged half
    pr = O._Prec(True)
    xr = torch.zeros(B, H, W, C, requires_grad=True)
    sd = {"u.expand.weight": wexp.float().cpu().reshape(2 * C, C, 1, 1), "u.expand.bias": torch.zeros(2 * C)}
    z = O.patch_unmerging(pr, sd, "u", xr)
    g_z = (dys.float().cpu() @ wskip.float().cpu()[:, :Cf]).bfloat16().float().reshape(B, 2 * H, 2 * W, Cf)
    z.backward(g_z)
    close(dx, xr.grad.reshape(M, C), 2e-3, 2e-3, "dx vs oracle autograd")
