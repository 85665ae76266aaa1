"""GPU tests added in round 2: the PatchExpanding / FinalPatchExpanding kernels against plain PyTorch, checkpoint /
resume of the fused trainer in the reference's order (optimizer built, then misc.load_model: main_lidar_upsampling.py:
283,288), and the configurations of BASELINE.json that round 1 left untested at full size."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tulip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from tulip_amd import ops as o
    return o


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize("B,H,W,P,Cn", [(2, 4, 16, 2, 48), (1, 2, 32, 2, 384), (2, 8, 64, 4, 96), (1, 1, 8, 2, 768),
                                        (2, 4, 16, 4, 48)])
def test_expand_norm_fwd_bwd_vs_torch(ops, B, H, W, P, Cn):
    """tulip.py:134-139 / :152-158 after the Linear: rearrange 'B H W (P1 P2 C) -> B (H P1) (W P2) C' + LayerNorm(C),
    and for the final layer the 1x1 decoder_pred conv (tulip.py:731) on the bf16-rounded LayerNorm output."""
    g = torch.Generator().manual_seed(B * 1000 + Cn + P)
    M, PP = B * H * W, P * P
    y = torch.randn(M, PP * Cn, generator=g) * 1.5 + 0.3
    gamma, beta = 1 + 0.2 * torch.randn(Cn, generator=g), 0.1 * torch.randn(Cn, generator=g)
    dotw = torch.randn(Cn, generator=g) * 0.3
    fine = B * H * W * PP
    # ---- reference in fp32 torch
    yt = y.clone().requires_grad_(True)
    gt, bt, wt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), dotw.clone().requires_grad_(True)
    z = O.expand_rearrange(yt.reshape(B, H, W, PP * Cn), P)                   # (B, HP, WP, Cn)
    zn = F.layer_norm(z, (Cn,), gt, bt, 1e-6)
    dy_fine = torch.randn(fine, Cn, generator=g).bfloat16().float()
    (zn.reshape(fine, Cn) * dy_fine).sum().backward()
    ref_dy, ref_dg, ref_db = yt.grad.clone(), gt.grad.clone(), bt.grad.clone()
    # ---- kernel, rows path: bf16 rows with a row pitch (first half of a concat buffer)
    yd, gd, bd = y.to(DEV), gamma.to(DEV), beta.to(DEV)
    ld = 2 * Cn
    out = torch.zeros(fine, ld, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(fine, device=DEV), torch.empty(fine, device=DEV)
    ops.expand_norm_fwd(yd, gd, bd, mean, rstd, B, H, W, P, Cn, 1e-6, out_bf16=out, ld=ld)
    torch.cuda.synchronize()
    got = out[:, :Cn].float().cpu()
    assert (got - zn.detach().reshape(fine, Cn)).abs().max().item() <= 2e-2          # bf16 rounding of O(1..4) values
    assert _rel(got, zn.detach().reshape(fine, Cn)) <= 4e-3
    assert out[:, Cn:].abs().max().item() == 0                                       # nothing written past the pitch
    R = ops.expand_norm_bwd_partial_rows(B, H, W, P)
    part = torch.full((R, 3 * Cn), float("nan"), device=DEV)
    dy_nat = torch.empty(M, PP * Cn, dtype=torch.bfloat16, device=DEV)
    dfine = torch.zeros(fine, ld, dtype=torch.bfloat16, device=DEV)
    dfine[:, :Cn] = dy_fine.to(DEV).bfloat16()
    ops.expand_norm_bwd(yd, mean, rstd, gd, dy_nat, part, B, H, W, P, Cn, dy_fine=dfine, ld=ld)
    torch.cuda.synchronize()
    assert _rel(dy_nat.float().cpu(), ref_dy) <= 6e-3
    tot = part.sum(0).cpu()
    assert _rel(tot[:Cn], ref_dg) <= 1e-4 and _rel(tot[Cn:2 * Cn], ref_db) <= 1e-4
    assert tot[2 * Cn:].abs().max().item() == 0
    # ---- kernel, decoder path: pred = sum_c w[c] * bf16(LN out)[c]; gradient from d(pred)
    for t in (yt, gt, bt, wt):
        t.grad = None
    zn = F.layer_norm(O.expand_rearrange(yt.reshape(B, H, W, PP * Cn), P), (Cn,), gt, bt, 1e-6)
    znr = O._BF16Round.apply(zn)
    pred_ref = (znr * wt).sum(-1)                                             # (B, HP, WP)
    dpred = torch.randn(pred_ref.shape, generator=g)
    (pred_ref * dpred).sum().backward()
    pred = torch.empty(fine, device=DEV)
    wd = dotw.to(DEV)
    ops.expand_norm_fwd(yd, gd, bd, mean, rstd, B, H, W, P, Cn, 1e-6, dotw=wd, pred=pred)
    torch.cuda.synchronize()
    assert (pred.cpu() - pred_ref.detach().reshape(-1)).abs().max().item() <= 2e-2 * max(1.0, pred_ref.abs().max().item())
    assert _rel(pred.cpu(), pred_ref.detach().reshape(-1)) <= 3e-3
    part.fill_(float("nan"))
    ops.expand_norm_bwd(yd, mean, rstd, gd, dy_nat, part, B, H, W, P, Cn, dpred=dpred.reshape(-1).to(DEV), dotw=wd,
                        beta=bd)
    torch.cuda.synchronize()
    tot = part.sum(0).cpu()
    assert _rel(dy_nat.float().cpu(), yt.grad) <= 6e-3
    assert _rel(tot[:Cn], gt.grad) <= 1e-4 and _rel(tot[Cn:2 * Cn], bt.grad) <= 1e-4
    assert _rel(tot[2 * Cn:], wt.grad) <= 3e-3


def test_trainer_checkpoint_resume_in_the_reference_order():
    """misc.save_model / load_model (misc.py:332-349, :361-470): {'model', 'optimizer'} saved after k steps; a NEW
    process builds model + optimizer first (main:283) and loads afterwards (main:288).  The continued run must equal
    the uninterrupted one bit for bit (AdamW moments, bias-correction step, DropPath counter, bf16 GEMM operands)."""
    from tests.test_model_gpu import build
    from tulip_amd.trainer import Trainer
    cfg = O.tiny_config()
    lo, hi = O.synthetic_batch(cfg, 4, seed=5)
    torch.manual_seed(21)
    a = Trainer(build(cfg, O.key_seeded_state_dict(cfg, seed=1), train=True), 4, lr=1e-3)
    a.load_batch(lo.to(DEV), hi.to(DEV))
    for _ in range(3):
        a.step()
    ckpt = {"model": {k: v.clone() for k, v in a.model.state_dict().items()}, "optimizer": a.state_dict()}
    la = [a.step().clone() for _ in range(3)]
    torch.cuda.synchronize()
    torch.manual_seed(99)                                                       # a different process: other seeds,
    mb = build(cfg, O.key_seeded_state_dict(cfg, seed=2), train=True)           # other initial weights
    b = Trainer(mb, 4, lr=7e-4)
    b.load_batch(lo.to(DEV), hi.to(DEV))
    b.step()                                                                    # graphs captured on the wrong weights
    mb.load_state_dict(ckpt["model"], strict=True)                              # AFTER construction
    b.load_state_dict(ckpt["optimizer"])
    lb = [b.step().clone() for _ in range(3)]
    torch.cuda.synchronize()
    assert b.t == a.t == 6 and b.lr == 1e-3
    assert torch.equal(torch.stack(lb), torch.stack(la))
    assert torch.equal(b.eng.params.flat, a.eng.params.flat)
    assert torch.equal(b.eng.params.shadow, a.eng.params.shadow)
    assert torch.equal(b.m, a.m) and torch.equal(b.v, a.v)


def test_pixel_loss_is_marked_non_differentiable():
    from tests.test_model_gpu import build
    cfg = O.tiny_config()
    m = build(cfg, O.key_seeded_state_dict(cfg, seed=1), train=True)
    lo, hi = O.synthetic_batch(cfg, 2, seed=5)
    pred, loss, pix = m(lo.to(DEV), hi.to(DEV))
    assert loss.requires_grad and not pix.requires_grad and not pred.requires_grad
    with pytest.raises(RuntimeError):
        pix.backward()
