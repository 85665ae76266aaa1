"""GPU: the drop-in TULIP module (HIP engine) against the reference-derived golden vectors and the
oracle, whole-model forward + backward.

Tolerances (bf16 GEMM operands, fp32 residual stream / softmax / LayerNorm / loss):
  * vs the oracle run with the SAME rounding model (lowp=True): differences are only accumulation
    order + rare 1-ulp bf16 rounding flips -> max|d pred| <= 4e-3, mean <= 4e-4, loss rel <= 2e-4.
  * vs the reference's fp32 forward (golden): must sit inside the reference's OWN bf16-autocast
    self-consistency band recorded in the fixture (mean / max abs error), loss rel <= 1e-3
    (the tolerance BASELINE.json states).
  * gradients, per tensor, relative L2 error, checked on EVERY tensor:
      vs the reference's fp32 autograd   <= 1.5e-2   (measured: median 0.7e-2, worst 1.0e-2)
      vs the same-rounding oracle (lowp) <= 1.0e-2
    except the 45 x nh relative-position-bias tables: their gradient is a sum of softmax-gradient
    terms over all windows with heavy cancellation (|g| ~ 1e-6 against ~1e-3 for weights), so bf16
    rounding of P / dS is a few-% relative effect -- the oracle's own bf16-rounded run is 4-6.5 %
    away from fp32 there.  Bounds for the tables: <= 1e-1 vs fp32, <= 5e-2 vs the lowp oracle.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tulip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    with open(os.path.join(golden_dir, name + ".json")) as f:
        meta = json.load(f)
    cfg = O.TulipConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()})
    return z, meta, cfg


def build(cfg: O.TulipConfig, sd, train=False):
    from functools import partial
    import torch.nn as nn
    from tulip_amd.model import tulip as T
    m = T.TULIP(img_size=cfg.img_size, target_img_size=cfg.target_img_size, patch_size=cfg.patch_size,
                in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, window_size=list(cfg.window_size), depths=cfg.depths,
                num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, drop_path_rate=cfg.drop_path_rate,
                norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps), pixel_shuffle=cfg.pixel_shuffle,
                circular_padding=cfg.circular_padding, log_transform=cfg.log_transform,
                patch_unmerging=cfg.patch_unmerging)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    m.train(train)
    return m


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_tiny_eval_forward_vs_golden_and_oracle(golden_dir):
    z, meta, cfg = _load(golden_dir, "g3_tiny_fp32")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = build(cfg, sd)
    with torch.no_grad():
        pred, loss, pix = m(lo.to(DEV), hi.to(DEV))
    assert pred.shape == (meta["batch"], 1, *cfg.target_img_size) and loss.dim() == 0 and pix.dim() == 0
    pred = pred.cpu()
    with torch.no_grad():
        lp, ll, lpix = O.tulip_forward(sd, cfg, lo, hi, lowp=True)
    d = (pred - lp).abs()
    assert d.max().item() <= 4e-3 and d.mean().item() <= 4e-4, (d.max().item(), d.mean().item())
    assert abs(loss.item() - ll.item()) <= 2e-4 * ll.item()
    assert abs(pix.item() - lpix.item()) <= 2e-4 * lpix.item()
    ref = torch.from_numpy(z["pred"])
    d = (pred - ref).abs()
    assert d.max().item() <= float(z["autocast_bf16_vs_fp32_maxabs"]) * 1.25
    assert d.mean().item() <= float(z["autocast_bf16_vs_fp32_meanabs"]) * 1.25
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * float(z["loss"])
    assert abs(pix.item() - float(z["pixel_loss"])) <= 1e-3 * float(z["pixel_loss"])
    # determinism of the eval forward
    with torch.no_grad():
        pred2, _, _ = m(lo.to(DEV), hi.to(DEV))
    assert torch.equal(pred2.cpu(), pred)
    # mc_drop path returns pred only (engine_upsampling.py:417-419)
    with torch.no_grad():
        p3 = m(lo.to(DEV), hi[:1].to(DEV), mc_drop=True)
    assert torch.equal(p3.cpu(), pred)


@pytest.mark.parametrize("name", ["g3_tiny_fp32", "g3_tiny_droppath", "g12_tiny3_expanding", "g12_tiny_patch_expanding",
                                  "g12_tiny_final_expanding"])
def test_tiny_gradients_vs_reference(golden_dir, name):
    """g12_*: the reference's non-default decoder alternates, PatchExpanding / FinalPatchExpanding (tulip.py:126-159):
    forward against the reference's fp32 prediction (fixture) and the same-rounding oracle, then every gradient."""
    z, meta, cfg = _load(golden_dir, name)
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = build(cfg, sd, train=True)
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    drop_u = None
    if meta["drop_path"]:
        du = torch.from_numpy(z["drop_u"])                       # (nblocks, 2, B) in block order
        drop_u = du.reshape(-1, du.shape[-1]).to(DEV)
        assert drop_u.shape[0] == eng.n_drop_slots
    else:
        m.eval()                                                  # rate 0 in the fixture <=> identity
    P = eng.plan(meta["batch"])
    P.x_in.copy_(lo.to(DEV)); P.target.copy_(hi.to(DEV))
    eng.draw_drop_scales(P, meta["drop_path"], drop_u)
    eng.run_forward(P)
    torch.cuda.synchronize()
    assert abs(P.losses[0].item() - float(z["loss"])) <= 1e-3 * float(z["loss"])
    if name.startswith("g12"):
        with torch.no_grad():
            lp, ll, _ = O.tulip_forward(sd, cfg, lo, hi, lowp=True)
        # A LayerNorm right before the output (FinalPatchExpanding) amplifies isolated bf16 rounding flips, so the
        # yardstick is the oracle's own bf16-vs-fp32 band for this model (max 1.3e-2 / mean 2.2e-3 on the 3-level one,
        # against 4e-3 / 7e-4 for the default head): the kernels must sit inside it against both oracle runs.
        ref32 = torch.from_numpy(z["pred"])
        band = (lp - ref32).abs()
        d = (P.pred.cpu() - lp).abs()
        assert d.max().item() <= 1.25 * band.max().item() and d.mean().item() <= 1.25 * band.mean().item(), \
            (d.max().item(), d.mean().item(), band.max().item(), band.mean().item())
        assert abs(P.losses[0].item() - ll.item()) <= 5e-4 * ll.item()
        d = (P.pred.cpu() - ref32).abs()                            # vs the reference's fp32 forward
        assert d.max().item() <= 1.5 * band.max().item() and d.mean().item() <= 1.5 * band.mean().item(), \
            (d.max().item(), d.mean().item(), band.max().item(), band.mean().item())
    gflat = torch.zeros(eng.params.total, device=DEV)
    eng.run_backward(P, gflat)
    torch.cuda.synchronize()
    W_ = eng.params
    grads = {n: gflat[W_.offset[n]:W_.offset[n] + W_.numel[n]].view(W_.shape[n]).cpu() for n in W_.names}
    assert all(torch.isfinite(g).all() for g in grads.values())
    # oracle gradients in fp32 (== reference autograd to 2e-5, pinned by make_golden.py)
    du_dict = None
    if meta["drop_path"]:
        du_dict = {k: torch.from_numpy(u) for k, u in zip(z["drop_u_keys"].tolist(), z["drop_u"])}
    _, _, _, og = O.tulip_loss_and_grads(sd, cfg, lo, hi, drop_u=du_dict)
    _, _, _, ol = O.tulip_loss_and_grads(sd, cfg, lo, hi, drop_u=du_dict, lowp=True)
    worst = [0.0, 0.0]
    for k, l2 in zip(z["grad_keys"].tolist(), z["grad_l2"]):
        assert abs(og[k].double().norm().item() - l2) <= 1e-4 * l2 + 1e-9    # oracle == fixture
        table = k.endswith("relative_position_bias_table")
        e32, elp = rel_l2(grads[k], og[k]), rel_l2(grads[k], ol[k])
        if not table:
            worst = [max(worst[0], e32), max(worst[1], elp)]
        # g12 (PatchExpanding / FinalPatchExpanding): the extra LayerNorms of the alternates sit behind bf16-rounded
        # gradients the oracle keeps in fp32 -> bounds x1.6 (measured worst 1.7e-2 / 1.4e-2, tables 5.6e-2)
        k32, klp = (2.4e-2, 1.6e-2) if name.startswith("g12") else (1.5e-2, 1.0e-2)
        # bias tables: 1e-1 / 5e-2 for the default model (module docstring); the wider 1.5e-1 only for the g12 alternates
        t32, tlp = (1.5e-1, 1.5e-1) if name.startswith("g12") else (1e-1, 5e-2)
        assert e32 <= (t32 if table else k32), (k, e32)
        assert elp <= (tlp if table else klp), (k, elp)
    print(f"worst per-tensor relative L2 gradient error (non-table): vs fp32 {worst[0]:.3e}, vs lowp {worst[1]:.3e}")
    for k in z.files:
        if k.startswith("grad::"):
            table = k.endswith("relative_position_bias_table")
            assert rel_l2(grads[k[6:]], torch.from_numpy(z[k])) <= (1e-1 if table else 1.5e-2), k


def test_autograd_bridge_matches_engine(golden_dir):
    z, meta, cfg = _load(golden_dir, "g3_tiny_fp32")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = build(cfg, sd)
    pred, loss, pix = m(lo.to(DEV), hi.to(DEV))
    assert loss.requires_grad and not pred.requires_grad
    (loss * 3.0).backward()                                       # upstream scale as GradScaler would
    eng = m.engine()
    P = eng.plan(meta["batch"])
    gflat = torch.zeros(eng.params.total, device=DEV)
    eng.run_backward(P, gflat, gscale=3.0)
    for n, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape
        ref = gflat[eng.params.offset[n]:eng.params.offset[n] + p.numel()].view(p.shape)
        assert rel_l2(p.grad, ref) <= 1e-3, n                      # atomics reorder only
    # optimizer step through the ordinary PyTorch API still reaches the HIP path's weights
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    opt.step()
    with torch.no_grad():
        _, loss2, _ = m(lo.to(DEV), hi.to(DEV))
    assert loss2.item() != loss.item()
    # state_dict round trip keeps the reference's keys
    sd2 = m.state_dict()
    assert list(sd2.keys()) == list(O.state_dict_spec(cfg).keys())


def test_noncircular_forward(golden_dir):
    z, meta, cfg = _load(golden_dir, "g3_tiny_noncircular")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = build(cfg, sd)
    with torch.no_grad():
        pred, loss, _ = m(lo.to(DEV), hi.to(DEV))
    d = (pred.cpu() - torch.from_numpy(z["pred"])).abs()
    assert d.max().item() <= 1.2e-2 and d.mean().item() <= 2e-3
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * float(z["loss"])


def test_kitti_base_forward_vs_golden(golden_dir):
    """BASELINE config 2 geometry: tulip_base 16x1024 -> 64x1024 (B=2, eval)."""
    from tulip_amd.model import tulip as T
    z, meta, cfg = _load(golden_dir, "g4_kitti_base")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = T.tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1,
                     window_size=[2, 8], swin_v2=False, pixel_shuffle=True, circular_padding=True,
                     log_transform=True, patch_unmerging=True)          # main_lidar_upsampling.py:221-230
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"]) == 27_149_076
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        pred, loss, pix = m(lo.to(DEV), hi.to(DEV))
    sub = pred.cpu().reshape(-1)[::257]
    d = (sub - torch.from_numpy(z["pred_sub257"])).abs()
    assert d.max().item() <= float(z["autocast_bf16_vs_fp32_maxabs"]) * 1.25
    assert d.mean().item() <= float(z["autocast_bf16_vs_fp32_meanabs"]) * 1.25
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * float(z["loss"])
    assert abs(pix.item() - float(z["pixel_loss"])) <= 1e-3 * float(z["pixel_loss"])


def test_large_backup_window_forward_vs_golden(golden_dir):
    """tulip_large at 16x2048: stage 4 has H=1 -> backup (1,16) window, shift (0,8) (tulip.py:284-287)."""
    from tulip_amd.model import tulip as T
    z, meta, cfg = _load(golden_dir, "g5_large_16x2048")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    m = T.tulip_large(img_size=(16, 2048), target_img_size=(64, 2048), patch_size=(1, 4), in_chans=1,
                      window_size=[2, 8], pixel_shuffle=True, circular_padding=True, log_transform=True,
                      patch_unmerging=True)
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"]) == 108_621_156
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    with torch.no_grad():
        pred, loss, pix = m(lo.to(DEV), hi.to(DEV))
    sub = pred.cpu().reshape(-1)[::257]
    d = (sub - torch.from_numpy(z["pred_sub257"])).abs()
    assert d.max().item() <= 1.2e-2 and d.mean().item() <= 2e-3
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * float(z["loss"])


def test_backup_window_gradients_vs_oracle():
    """Gradients through the (1,16) backup window + (0,8) shift (tulip.py:284-287): a 3-stage model whose
    last stage has H=1.  The oracle's forward for this path is pinned by the g5 fixture; its autograd is the
    reference for the backward."""
    cfg = O.TulipConfig(img_size=(4, 256), target_img_size=(16, 256), depths=(2, 2, 2), embed_dim=48,
                        num_heads=(3, 6, 12))
    sd = O.key_seeded_state_dict(cfg, seed=7)
    lo, hi = O.synthetic_batch(cfg, 3, seed=99)
    m = build(cfg, sd)
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    deep = eng.enc_blocks[2]
    assert deep[0].win == (1, 16) and deep[1].sft == (0, 8)
    pred, loss, _ = m(lo.to(DEV), hi.to(DEV))
    loss.backward()
    op, ol, _, og = O.tulip_loss_and_grads(sd, cfg, lo, hi)
    _, _, _, olp = O.tulip_loss_and_grads(sd, cfg, lo, hi, lowp=True)
    assert abs(loss.item() - ol.item()) <= 1e-3 * ol.item()
    assert (pred.detach().cpu() - op).abs().max().item() <= 1.2e-2
    for n, p in m.named_parameters():
        table = n.endswith("relative_position_bias_table")
        assert rel_l2(p.grad, og[n]) <= (1e-1 if table else 1.5e-2), n
        assert rel_l2(p.grad, olp[n]) <= (5e-2 if table else 1.0e-2), n


def test_fused_training_step_matches_reference_trajectory(golden_dir):
    """Trainer (HIP-graph replay of fwd+loss+bwd+fused AdamW with timm's decay grouping) against
    the loss trajectory of the REFERENCE model + torch.optim.AdamW (fixture g7).  bf16 GEMMs vs the fp32
    reference: loss agrees to 2e-3 relative at every step, and must actually descend."""
    from tulip_amd.trainer import Trainer
    z = np.load(os.path.join(golden_dir, "g7_train_trajectory.npz"))
    cfg = O.tiny_config(drop_path_rate=0.0)
    sd = O.key_seeded_state_dict(cfg, seed=int(z["seed"]))
    lo, hi = O.synthetic_batch(cfg, int(z["batch"]), seed=int(z["data_seed"]))
    for use_graph in (True, False):
        m = build(cfg, sd, train=True)
        tr = Trainer(m, int(z["batch"]), lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, use_graph=use_graph)
        tr.load_batch(lo.to(DEV), hi.to(DEV))
        got = []
        for _ in range(4):
            got.append(tr.step().clone())
        with torch.no_grad():
            _, l, px = m(lo.to(DEV), hi.to(DEV))                 # parameters updated in place are what the module sees
        got = [g[0].item() for g in got] + [l.item()]
        ref = z["loss"].tolist()
        assert ref[-1] < ref[0]
        for a, b in zip(got, ref):
            assert abs(a - b) <= 2e-3 * b, (use_graph, got, ref)


def test_train_one_epoch_matches_reference_loop(golden_dir):
    """train_one_epoch (per-window LR, accum_iter=2, grad-norm read-out, fused AdamW) against the
    REFERENCE loop semantics captured in g9 (reference model + imported lr_sched + get_grad_norm_), with HIP
    graphs and without.  bf16 GEMMs vs the fp32 reference: losses to 3e-3, gradient norms to 2e-2 relative."""
    from types import SimpleNamespace
    from tulip_amd.trainer import Trainer, train_one_epoch
    z = np.load(os.path.join(golden_dir, "g9_train_loop.npz"))
    cfg = O.tiny_config(drop_path_rate=0.0)
    sd = O.key_seeded_state_dict(cfg, seed=int(z["seed"]))
    nb, accum = int(z["n_batches"]), int(z["accum_iter"])
    batches = [O.synthetic_batch(cfg, int(z["batch"]), seed=int(z["data_seed0"]) + i) for i in range(nb)]
    lr, min_lr, warm, epochs = z["sched"].tolist()
    args = SimpleNamespace(lr=lr, min_lr=min_lr, warmup_epochs=warm, epochs=epochs)
    for use_graph in (True, False):
        m = build(cfg, sd, train=True)
        tr = Trainer(m, int(z["batch"]), lr=lr, betas=(0.9, 0.95), weight_decay=0.01, use_graph=use_graph,
                     accum_iter=accum, track_grad_norm=True)
        losses, norms, lrs = [], [], []

        class Loader(list):                       # records what the loop did at every micro-step
            def __iter__(self_inner):
                for i, b in enumerate(list.__iter__(self_inner)):
                    yield b
                    losses.append(tr.P.losses[0].item())
                    lrs.append(tr.lr)
                    if (i + 1) % accum == 0:
                        norms.append(tr.grad_norm.item())

        for epoch in range(int(z["epochs_run"])):
            stats = train_one_epoch(tr, Loader(batches), epoch, args)
        np.testing.assert_allclose(lrs, z["lr"], rtol=1e-12)
        np.testing.assert_allclose(losses, z["loss"], rtol=3e-3, err_msg=str(use_graph))
        np.testing.assert_allclose(norms, z["grad_norm"], rtol=2e-2, err_msg=str(use_graph))
        assert abs(stats["loss"] - z["loss"][nb:].mean()) <= 3e-3 * z["loss"][nb:].mean()
        assert tr.t == int(z["epochs_run"]) * nb // accum


def test_non_finite_loss_aborts_like_reference():
    """engine_upsampling.py:85-88: a non-finite loss stops training with exit status 1."""
    from types import SimpleNamespace
    from tulip_amd.trainer import Trainer, train_one_epoch
    cfg = O.tiny_config(drop_path_rate=0.0)
    sd = O.key_seeded_state_dict(cfg, seed=1)
    m = build(cfg, sd, train=True)
    tr = Trainer(m, 2, use_graph=False)
    lo, hi = O.synthetic_batch(cfg, 2, seed=3)
    hi = hi.clone()
    hi[0, 0, 0, 0] = float("nan")
    args = SimpleNamespace(lr=5e-4, min_lr=0.0, warmup_epochs=0, epochs=2)
    with pytest.raises(SystemExit) as e:
        train_one_epoch(tr, [(lo, hi)], 0, args)
    assert e.value.code == 1


def test_graphed_inference_matches_module_forward():
    """HIP-graph replay of the eval forward == module forward (mc_drop=True), bit for bit, also after
    the weights change under it."""
    from tulip_amd.infer import GraphedForward
    cfg = O.tiny_config(drop_path_rate=0.1)
    sd = O.key_seeded_state_dict(cfg, seed=2)
    m = build(cfg, sd, train=False)
    lo, hi = O.synthetic_batch(cfg, 8, seed=9)
    gf = GraphedForward(m, 8)
    with torch.no_grad():
        ref = m(lo.to(DEV), hi.to(DEV), mc_drop=True)
    for _ in range(2):
        assert torch.equal(gf(lo.to(DEV)), ref)
    # tiled input of MCdrop (engine_upsampling.py:414): all 8 predictions identical
    out = gf(lo[:1].tile(8, 1, 1, 1).to(DEV))
    assert all(torch.equal(out[0], out[i]) for i in range(1, 8))
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)
        ref2 = m(lo.to(DEV), hi.to(DEV), mc_drop=True)
    gf.weights_changed()
    assert torch.equal(gf(lo.to(DEV)), ref2) and not torch.equal(ref, ref2)


@pytest.mark.parametrize("B", [2, 8])
def test_inference_form_of_the_fused_blocks(B):
    """run_forward(with_loss=False) (eval / MC-dropout inference) launches the fused block kernels without
    the activations a backward would read (all saved-tensor pointers NULL: csrc/swin96.hip SAVE = false, csrc/swinw.hip
    MODE bit 2).  Same arithmetic as the training form, but a separate instantiation: the compiler contracts one fp32
    expression of a LayerNorm differently around the removed stores, which flips the bf16 rounding of ~1 token in 8 000 per
    C = 96 block (one block alone: a single row differs, by 7e-5; the C = 192 / 384 kernels are bit-identical).  Through
    the network that stays a fraction of the bf16-vs-fp32 noise the whole-model tests bound against the oracle: rel L2
    <= 3e-3 of the prediction (measured 1.5e-3), deterministic from run to run.  KITTI size; batch 8 also runs stage 2 fused."""
    cfg = O.tulip_base_config()
    m = build(cfg, O.key_seeded_state_dict(cfg, seed=4), train=False)
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    P = eng.plan(B)
    lo, hi = O.synthetic_batch(cfg, B, seed=6)
    P.x_in.copy_(lo.to(DEV))
    eng.draw_drop_scales(P, False)
    preds = []
    for no_save in (False, True):
        eng.infer_no_save = no_save
        P.pred.fill_(float("nan"))
        for sp in eng.blocks:                                  # the saved tensors of the fused blocks: poisoned
            if eng._fusable96(sp) or eng._fusable_wide(sp, B):
                P[sp.prefix + ".xn1"].fill_(7.0)             # (xn1: every training form writes it; the C = 96 one no longer writes qkv)
        eng.run_forward(P, with_loss=False)
        torch.cuda.synchronize()
        assert eng._no_save == no_save
        touched = [bool((P[sp.prefix + ".xn1"] != 7.0).any()) for sp in eng.blocks
                   if eng._fusable96(sp) or eng._fusable_wide(sp, B)]
        assert touched and all(t != no_save for t in touched)  # written in the training form only
        preds.append(P.pred.clone())
    eng.run_forward(P, with_loss=False)                    # the inference form again: same bits
    torch.cuda.synchronize()
    assert torch.equal(P.pred, preds[1])
    assert torch.isfinite(preds[0]).all() and torch.isfinite(preds[1]).all()
    rel = ((preds[0] - preds[1]).norm() / preds[0].norm()).item()
    print(f"inference form vs training form: rel L2 {rel:.2e}, max {(preds[0] - preds[1]).abs().max().item():.2e}")
    assert rel <= 3e-3


def test_kitti_base_full_size_gradients_vs_oracle():
    """BASELINE.json configs[1] at its full size (tulip_base, 16x1024 -> 64x1024): loss and every parameter gradient of
    the HIP path against the oracle's fp32 autograd on the same seeded weights / inputs (B=2, DropPath off), plus the
    size-independent property that the gradient is linear in the upstream loss scale."""
    cfg = O.tulip_base_config()
    sd = O.key_seeded_state_dict(cfg, seed=11)
    lo, hi = O.synthetic_batch(cfg, 2, seed=21)
    m = build(cfg, sd, train=False)
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    P = eng.plan(2)
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
        assert e <= (1e-1 if table else 2e-2), (n, e)
        worst = max(worst, 0.0 if table else e)
    print(f"KITTI base: worst per-tensor relative L2 gradient error vs fp32 oracle (non-table) {worst:.3e}")
    # d(0.5*loss) = 0.5*d(loss) up to the bf16 rounding of the scaled upstream gradient
    assert rel_l2(g2 * 2, g1) <= 4e-3


def test_kitti_base_training_is_stable_and_learns():
    """BASELINE.json configs[1] end to end: 200 fused steps (HIP-graph replay, DropPath on, AdamW) on one synthetic
    batch of 8: finite throughout, and the L1 loss falls well below its starting value."""
    from tulip_amd.model.tulip import tulip_base
    from tulip_amd.trainer import Trainer, cosine_lr
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    cfg = O.tulip_base_config()
    lo, hi = O.synthetic_batch(cfg, 8, seed=5)
    tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
    tr.load_batch(lo.to(DEV), hi.to(DEV))
    hist = []
    for it in range(200):
        hist.append(tr.step(lr=cosine_lr(it / 20, 5e-4, 1e-5, 1.0, 20.0)))      # 20 "epochs" of 20 steps, 1 warm-up
        hist[-1] = hist[-1].clone()
    L = torch.stack(hist)[:, 0].cpu()
    assert torch.isfinite(L).all()
    assert L[-10:].mean().item() < 0.6 * L[:5].mean().item(), (L[:5], L[-10:])
    for p in m.parameters():
        assert torch.isfinite(p).all()
