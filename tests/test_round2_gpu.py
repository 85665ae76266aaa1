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
    from tests.conftest import describe_flat_diff
    assert torch.equal(b.eng.params.flat, a.eng.params.flat), describe_flat_diff(a.eng, b.eng.params.flat, a.eng.params.flat)
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


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2..4] at their full sizes
def _grads_vs_oracle(cfg, sd, lo, hi, B, tol=2e-2):
    from tests.test_model_gpu import build, rel_l2
    m = build(cfg, sd, train=False)
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    P = eng.plan(B)
    P.x_in.copy_(lo.to(DEV)); P.target.copy_(hi.to(DEV))
    eng.draw_drop_scales(P, False)
    eng.run_forward(P)
    g1 = torch.zeros(eng.params.total, device=DEV)
    eng.run_backward(P, g1)
    g2 = torch.zeros(eng.params.total, device=DEV)
    eng.run_backward(P, g2, gscale=0.5)
    torch.cuda.synchronize()
    _, oloss, _, og = O.tulip_loss_and_grads(sd, cfg, lo, hi)
    assert abs(P.losses[0].item() - oloss.item()) <= 1e-3 * oloss.item()
    W_ = eng.params
    worst = 0.0
    for n in W_.names:
        g = g1[W_.offset[n]:W_.offset[n] + W_.numel[n]].view(W_.shape[n])
        table = n.endswith("relative_position_bias_table")
        e = rel_l2(g, og[n])
        assert e <= (1.5e-1 if table else tol), (n, e)
        worst = max(worst, 0.0 if table else e)
    assert rel_l2(g2 * 2, g1) <= 4e-3              # linear in the upstream loss scale
    return worst, eng, P


def test_carla_large_16x2048_training_gradients_vs_oracle():
    """BASELINE.json configs[2], train half (bash_scripts/tulip_upsampling_carla.sh:10,26-27: tulip_large, 16x2048 ->
    64x2048; stage 4 runs the (1,16) backup window): loss and EVERY parameter gradient of the HIP path at full size against
    the oracle's fp32 autograd (B=1, DropPath off)."""
    cfg = O.tulip_large_config(img_size=(16, 2048), target_img_size=(64, 2048))
    sd = O.key_seeded_state_dict(cfg, seed=13)
    lo, hi = O.synthetic_batch(cfg, 1, seed=31)
    worst, _, _ = _grads_vs_oracle(cfg, sd, lo, hi, 1)
    print(f"CARLA tulip_large 16x2048: worst per-tensor relative L2 gradient error vs fp32 oracle (non-table) {worst:.3e}")


def test_carla_large_16x2048_training_step_and_eval_full_size():
    """configs[2] continued: a fused training step at that size stays finite and learns (B=2, 30 steps), and the
    evaluation post-processing (expm1, gate, low-res rows restored, range image -> CARLA point cloud, voxel IoU,
    Chamfer; engine_upsampling.py:176-271, util/evaluation.py:90-175) of one 64x2048 image equals the oracle's."""
    from oracle import eval_oracle as EO
    from tulip_amd import evaluation as EV
    from tulip_amd.model.tulip import tulip_large
    from tulip_amd.trainer import Trainer
    torch.manual_seed(0)
    m = tulip_large(img_size=(16, 2048), target_img_size=(64, 2048), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                    pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    cfg = O.tulip_large_config(img_size=(16, 2048), target_img_size=(64, 2048))
    lo, hi = O.synthetic_batch(cfg, 2, seed=5)
    tr = Trainer(m, 2, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
    tr.load_batch(lo.to(DEV), hi.to(DEV))
    hist = torch.stack([tr.step().clone() for _ in range(30)])[:, 0].cpu()
    assert torch.isfinite(hist).all() and hist[-3:].mean().item() < 0.8 * hist[:3].mean().item(), hist
    m.eval()
    with torch.no_grad():
        pred, _, _ = m(lo.to(DEV), hi.to(DEV))
    # evaluation of the model's own prediction for image 0 (the reference-pinned fixture for this size is the
    # "carla_full" case of tests/test_eval_gpu.py); stage by stage against the oracle as there
    ev = EV.RangeEvaluator("carla", (16, 2048), (64, 2048), True, 0.1, False, False, DEV)
    res = ev(pred[:1].contiguous(), lo[:1].to(DEV), hi[:1].to(DEV)).cpu().numpy()
    mae, mae_low, p_img, t_img = EO.postprocess(pred[:1].cpu(), hi[:1], lo[:1], "carla", True)
    d_p, d_t = ev.pred_img.cpu().numpy(), ev.hi_img.cpu().numpy()
    assert np.array_equal(d_p == 0, p_img == 0) and np.abs(d_p - p_img).max() <= 3e-7 and np.abs(d_t - t_img).max() <= 3e-7
    assert abs(res[0] - mae) <= 1e-6 * mae and abs(res[1] - mae_low) <= 1e-6 * max(mae_low, 1e-9)
    tab = EO.carla_tables(64, 2048)
    op, ot = EO.spherical_pcd(d_p, tab, 80), EO.spherical_pcd(d_t, tab, 80)
    assert np.array_equal(ev.pcd_pred.cpu().numpy(), op) and np.array_equal(ev.pcd_gt.cpu().numpy(), ot)
    iou, prec, rec, f1, _ = EO.voxel_metrics(op, ot, 0.1)
    assert (res[3], res[4], res[5], res[6]) == (iou, prec, rec, f1)
    assert np.isfinite(res[2]) and res[2] > 0


def test_durlar_large_32x2048_forward_vs_golden_and_backward_properties(golden_dir):
    """BASELINE.json configs[3] geometry: tulip_large, DurLAR 32x2048 -> 128x2048.  B=1 eval forward against the
    reference's fp32 forward (fixture g6, inside the reference's own bf16-autocast band); B=8 backward: finite, and
    linear in the upstream loss scale."""
    import json, os
    from tests.test_model_gpu import _load, rel_l2
    from tulip_amd.model import tulip as T
    z, meta, cfg = _load(golden_dir, "g6_durlar_large")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = T.tulip_large(img_size=(32, 2048), target_img_size=(128, 2048), patch_size=(1, 4), in_chans=1,
                      window_size=[2, 8], pixel_shuffle=True, circular_padding=True, log_transform=True,
                      patch_unmerging=True)
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"]) == 108_621_156
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        pred, loss, pix = m(lo.to(DEV), hi.to(DEV))
    sub = pred.cpu().reshape(-1)[::257]
    d = (sub - torch.from_numpy(z["pred_sub257"])).abs()
    assert d.max().item() <= float(z["autocast_bf16_vs_fp32_maxabs"]) * 1.25, d.max().item()
    assert d.mean().item() <= float(z["autocast_bf16_vs_fp32_meanabs"]) * 1.25, d.mean().item()
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * float(z["loss"])
    assert abs(pix.item() - float(z["pixel_loss"])) <= 1e-3 * float(z["pixel_loss"])
    # ---- B = 8 (the per-GPU batch of the DDP run): backward properties at full size
    m.train()
    eng = m.engine()
    P = eng.plan(8)
    lo8, hi8 = O.synthetic_batch(cfg, 8, seed=77)
    P.x_in.copy_(lo8.to(DEV)); P.target.copy_(hi8.to(DEV))
    eng.draw_drop_scales(P, True)
    eng.run_forward(P)
    g1 = torch.zeros(eng.params.total, device=DEV)
    eng.run_backward(P, g1)
    g2 = torch.zeros(eng.params.total, device=DEV)
    eng.run_backward(P, g2, gscale=0.25)
    torch.cuda.synchronize()
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    assert rel_l2(g2 * 4, g1) <= 4e-3


@pytest.mark.parametrize("attn_fp8", [False, True])
def test_kitti_batch64_training_is_stable_and_learns(attn_fp8):
    """BASELINE.json configs[4]: per-GPU batch 64, 200 fused steps (HIP-graph replay, DropPath, AdamW) -- in bf16 and with
    the fp8 attention scores the configuration names; the two loss curves stay close."""
    from tulip_amd.model.tulip import tulip_base
    from tulip_amd.trainer import Trainer, cosine_lr
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    lo, hi = O.synthetic_batch(O.tulip_base_config(), 64, seed=5)
    tr = Trainer(m, 64, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, attn_fp8=attn_fp8)
    assert tr.eng.attn_fp8 == attn_fp8
    tr.load_batch(lo.to(DEV), hi.to(DEV))
    hist = torch.stack([tr.step(lr=cosine_lr(it / 20, 5e-4, 1e-5, 1.0, 20.0)).clone() for it in range(200)])[:, 0].cpu()
    assert torch.isfinite(hist).all()
    assert hist[-10:].mean().item() < 0.7 * hist[:5].mean().item(), (hist[:5], hist[-10:])
    assert all(torch.isfinite(p).all() for p in m.parameters())
    _B64_CURVES[attn_fp8] = hist
    if len(_B64_CURVES) == 2:       # same seed, same data: e4m3 scores move the trajectory by a few per cent at most
        a, b = _B64_CURVES[False], _B64_CURVES[True]
        rel = ((a - b).abs() / a).max().item()
        print(f"batch 64, 200 steps: max relative difference of the loss curves bf16 vs fp8 scores {rel:.3e}; "
              f"final loss {a[-1].item():.5f} vs {b[-1].item():.5f}")
        assert rel <= 0.1, rel


_B64_CURVES = {}


def test_fp8_attention_scores_whole_model_vs_oracle():
    """BASELINE.json configs[4] (KITTI 16x1024 -> 64x1024, "fp8 MFMA attention"): with engine.attn_fp8 every block --
    fused C = 96 / 192 / 384 kernels and the unfused C = 768 sequence -- takes its attention scores from e4m3 q, k.  Against
    the oracle run with the SAME rounding model (bf16 operands + q, k through torch.float8_e4m3fn, straight-through
    gradient): prediction and loss to the bf16 bounds of the bf16 tests, every parameter gradient <= 2e-2 relative L2
    (tables 1.5e-1); and the fp8 prediction differs from the bf16 one by what the oracle says it should."""
    from tests.test_model_gpu import build, rel_l2
    cfg = O.tulip_base_config()
    sd = O.key_seeded_state_dict(cfg, seed=5)
    lo, hi = O.synthetic_batch(cfg, 2, seed=9)
    m = build(cfg, sd, train=False)
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    res = {}
    for fp8 in (False, True):
        eng.attn_fp8 = fp8
        P = eng.plan(2)
        P.x_in.copy_(lo.to(DEV)); P.target.copy_(hi.to(DEV))
        eng.draw_drop_scales(P, False)
        eng.run_forward(P)
        g = torch.zeros(eng.params.total, device=DEV)
        eng.run_backward(P, g)
        torch.cuda.synchronize()
        res[fp8] = (P.pred.detach().cpu().clone(), P.losses[0].item(), g.cpu())
    opred, oloss, _, og = O.tulip_loss_and_grads(sd, cfg, lo, hi, lowp=True, attn_fp8=True)
    with torch.no_grad():
        bpred = O.tulip_forward(sd, cfg, lo, hi, lowp=True)[0]
    pred, loss, g = res[True]
    d = (pred.reshape(-1) - opred.reshape(-1)).abs()
    print(f"fp8 scores: max|d pred| vs fp8 oracle {d.max().item():.3e}, mean {d.mean().item():.3e}; loss {loss:.6f} vs {oloss.item():.6f}")
    assert d.max().item() <= 8e-3 and d.mean().item() <= 6e-4
    assert abs(loss - oloss.item()) <= 3e-4 * oloss.item()
    W_ = eng.params
    worst = 0.0
    for n in W_.names:
        gn = g[W_.offset[n]:W_.offset[n] + W_.numel[n]].view(W_.shape[n])
        table = n.endswith("relative_position_bias_table")
        e = rel_l2(gn, og[n])
        assert e <= (1.5e-1 if table else 2e-2), (n, e)
        worst = max(worst, 0.0 if table else e)
    print(f"fp8 scores: worst per-tensor gradient error vs the fp8 oracle (non-table) {worst:.3e}")
    # the flag does something, and about as much as the oracle predicts
    dev_hip = (res[True][0] - res[False][0]).abs().max().item()
    dev_orc = (opred - bpred).abs().max().item()
    print(f"fp8 vs bf16 prediction: HIP max {dev_hip:.3e}, oracle max {dev_orc:.3e}")
    assert dev_hip > 0 and dev_hip <= 4 * dev_orc + 8e-3


@pytest.mark.parametrize("M,N,K,splits", [(512, 768, 768, 3), (512, 768, 3072, 4), (128, 1536, 1536, 4), (2048, 256, 1024, 2)])
def test_splitk_fold_with_layernorm_fused_matches_two_launches(M, N, K, splits):
    """tulip_splitk_resid_ln (fold of a GEMM's raw split-K slabs + bias + DropPath residual + the LayerNorm that follows, one
    launch) against tulip_gemm_bf16's own fold launch followed by tulip_layernorm_fwd: the residual output bit for bit, the
    normalised rows to bf16 rounding flips; and tulip_layernorm_bwd_splitk against fold + tulip_layernorm_bwd: bit for bit."""
    from tulip_amd import ops as o
    from tulip_amd._lib import EPI_RESID_F32, EPI_SPLIT_F32, EPI_BF16
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    rnd = lambda *s, scale=1.0: torch.randn(*s, device=DEV, generator=g) * scale
    A, Wt = rnd(M, K, scale=0.5).bfloat16(), rnd(N, K, scale=0.05).bfloat16()
    bias, aux = rnd(N, scale=0.1), rnd(M, N)
    rows_per_sample = M // 4
    rowscale = torch.tensor([1.0, 0.0, 1.1111, 1.1111], device=DEV)
    gamma, beta = 1 + 0.1 * rnd(N), 0.1 * rnd(N)
    ws = torch.empty(splits * M * N, device=DEV)
    eff = o.gemm_effective_splits(K, splits)
    # ---- two launches + LayerNorm
    out_ref, ob_ref = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o.gemm(A, Wt, M, N, K, lda=K, ldb=K, epi=EPI_RESID_F32, bias=bias, out=out_ref, aux=aux, ldaux=N, rowscale=rowscale,
           rows_per_sample=rows_per_sample, out2=ob_ref, ldo2=N, splits=splits, workspace=ws, workspace_bytes=ws.numel() * 4)
    xn_ref = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    mu_ref, rs_ref = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    o.layernorm_fwd(out_ref, gamma, beta, xn_ref, mu_ref, rs_ref, M, N, 1e-6)
    # ---- fused
    assert o.splitk_resid_ln_supported(N)
    o.gemm(A, Wt, M, N, K, lda=K, ldb=K, epi=EPI_SPLIT_F32, out=ws, ldo=N, splits=splits)
    out, ob = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    xn = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    mu, rs = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    o.splitk_resid_ln(ws, eff, M, N, bias, aux, N, rowscale, rows_per_sample, out, N, ob, N, gamma, beta, xn, mu, rs, 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(out, out_ref) and torch.equal(ob, ob_ref)
    assert torch.allclose(mu, mu_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rs, rs_ref, rtol=1e-5, atol=1e-6)
    d = (xn.float() - xn_ref.float()).abs()
    assert (d > 2 ** -7 * (0.05 + xn_ref.float().abs())).float().mean().item() <= 1e-3
    # ---- backward: the data gradient's slabs straight into the LayerNorm backward
    dY = rnd(M, N, scale=0.5).bfloat16()                 # dxn = dY . W  with W [N][K]: C = K columns here
    if K % 256 == 0 and o.layernorm_bwd_partial_rows(M, K) > 0:
        x = rnd(M, K) + 0.2
        xn2 = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
        m2, r2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        gam = 1 + 0.1 * rnd(K)
        o.layernorm_fwd(x, gam, torch.zeros(K, device=DEV), xn2, m2, r2, M, K, 1e-6)
        sp2 = o.gemm_effective_splits(N, 3)
        ws2 = torch.empty(max(sp2, 1) * M * K, device=DEV)
        dxn = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
        o.gemm(dY, Wt, M, K, N, lda=N, ldb=K, b_trans=True, epi=EPI_BF16, out=dxn, ldo=K, splits=3, workspace=ws2,
               workspace_bytes=ws2.numel() * 4)
        R = o.layernorm_bwd_partial_rows(M, K)
        res = []
        for fused in (False, True):
            dres = rnd(M, K)
            g.manual_seed(7)
            dres = torch.randn(M, K, device=DEV, generator=g)
            dx = torch.empty(M, K, device=DEV)
            part = torch.full((R, 2 * K), float("nan"), device=DEV)
            cast = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
            if fused:
                o.gemm(dY, Wt, M, K, N, lda=N, ldb=K, b_trans=True, epi=EPI_SPLIT_F32, out=ws2, ldo=K, splits=3)
                o.layernorm_bwd_splitk(ws2, sp2, x, m2, r2, gam, dres, dx, M, K, param_partials=part, dx_bf16=cast,
                                       cast_rowscale=rowscale, cast_rows_per_sample=rows_per_sample)
            else:
                o.layernorm_bwd(dxn, x, m2, r2, gam, dres, dx, M, K, param_partials=part, dx_bf16=cast,
                                cast_rowscale=rowscale, cast_rows_per_sample=rows_per_sample)
            torch.cuda.synchronize()
            res.append((dx.clone(), part.clone(), cast.clone()))
        for a, b in zip(res[0], res[1]):
            assert torch.equal(a, b)
