#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE implementation (build container only).

Imports /root/reference/tulip/model/tulip.py with the two import stubs of SURVEY.md Appendix B
(neither touches hot-path arithmetic), runs it on key-seeded weights / seeded inputs, asserts the
oracle (oracle/tulip_oracle.py) reproduces it, and stores inputs-by-seed + expected outputs as
plain arrays.  Nothing from /root/reference is written into the repo: fixtures are data only.

Usage:  python tests/golden/make_golden.py            (needs /root/reference; ~1-2 min on 8 cores)
"""
import json
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import tulip_oracle as O  # noqa: E402

REF = "/root/reference/tulip"


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")

    class _DP(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    timm_layers.DropPath = _DP
    timm_layers.to_2tuple = lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)
    timm_layers.trunc_normal_ = nn.init.trunc_normal_
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})
    cd = types.ModuleType("chamfer_distance")
    cd.ChamferDistance = object
    sys.modules["chamfer_distance"] = cd
    sys.path.insert(0, REF)
    import model.tulip as T  # noqa
    return T


def ref_model(T, cfg: O.TulipConfig, drop_path_rate: float):
    m = T.TULIP(img_size=cfg.img_size, target_img_size=cfg.target_img_size, patch_size=cfg.patch_size,
                in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, window_size=list(cfg.window_size),
                depths=cfg.depths, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
                drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=drop_path_rate,
                norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps), pixel_shuffle=cfg.pixel_shuffle,
                circular_padding=cfg.circular_padding, log_transform=cfg.log_transform,
                patch_unmerging=cfg.patch_unmerging)
    return m


def sub(t: torch.Tensor, stride: int = 257) -> np.ndarray:
    return t.detach().reshape(-1)[::stride].numpy().copy()


def check(name, a, b, tol):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-30
    print(f"  {name:48s} max|d|={err:.3e}  rel={err / ref:.3e}")
    assert err <= tol * max(1.0, ref), f"{name}: oracle != reference ({err} > {tol})"
    return err


# ------------------------------------------------------------------ G1: index ops (bit exact)
def golden_index(T):
    from einops import rearrange
    out = {}
    wa = T.WindowAttention(dim=96, window_size=[2, 8], num_heads=3, shift=False)
    out["rel_pos_index_2x8"] = wa.relative_position_index.numpy().copy()
    assert np.array_equal(out["rel_pos_index_2x8"], O.relative_position_index(2, 8))
    grids = [(8, 64), (4, 32), (16, 256), (2, 32), (1, 32), (32, 512)]
    for (H, W) in grids:
        for shift in (False, True):
            a = T.WindowAttention(dim=96, window_size=[2, 8], num_heads=3, shift=shift)
            ids = torch.arange(H * W, dtype=torch.float32).reshape(1, H, W, 1)
            # replay the reference's own index manipulations on a token-id image
            if H < a.window_size[0]:
                a.window_size = a.backup_window_size
                if a.shift:
                    a.shift_size = a.backup_shift_size
            x = ids
            if a.shift:
                x = torch.roll(x, shifts=(-a.shift_size[0], -a.shift_size[1]), dims=(1, 2))
                mask = a.create_mask(x).numpy().copy()
            else:
                mask = None
            part = a.window_partition(x)
            tok = part.reshape(part.shape[0], -1).long().numpy().copy()
            win, sft = O.effective_window(H, (2, 8), shift)
            assert win == tuple(a.window_size)
            mine = O.window_token_index(H, W, win, sft)
            assert np.array_equal(tok, mine), (H, W, shift)
            tag = f"{H}x{W}_{'s' if shift else 'n'}"
            out[f"win_tok_{tag}"] = tok.astype(np.int32)
            # inverse: partition -> reverse -> unroll must give identity
            back = rearrange(part, '(B Nh Nw) Mh Mw C -> B (Nh Mh) (Nw Mw) C', Nh=H // win[0], Nw=W // win[1])
            if a.shift_size != 0:
                back = torch.roll(back, shifts=(a.shift_size[0], a.shift_size[1]), dims=(1, 2))
            assert torch.equal(back, ids)
            if mask is not None:
                mm = O.shift_attention_mask(H, W, win, sft)
                assert np.array_equal(mask, mm), (H, W)
                out[f"mask_{tag}"] = (mask != 0).astype(np.uint8)
                lab = O.shift_region_labels(H, W, win, sft)
                out[f"labels_{tag}"] = lab.astype(np.uint8)
    # patch merging gather order (tulip.py:92-99) on a token-id image
    for (H, W) in [(4, 8), (16, 256)]:
        ids = torch.arange(H * W, dtype=torch.float32).reshape(1, H, W, 1)
        g = T.PatchMerging.merging(ids).reshape(-1, 4).long().numpy()
        assert np.array_equal(g, O.patch_merge_gather_index(H, W))
        out[f"merge_gather_{H}x{W}"] = g.astype(np.int32)
    # pixel shuffle permutation for r=2,4
    for r in (2, 4):
        C, H, W = 3, 2, 3
        ids = torch.arange(C * r * r * H * W, dtype=torch.float32).reshape(1, C * r * r, H, W)
        ps = nn.PixelShuffle(r)(ids)
        mine = torch.empty_like(ps)
        for c in range(C):
            for i in range(r):
                for j in range(r):
                    mine[0, c, i::r, j::r] = ids[0, O.pixel_shuffle_source_channel(c, i, j, r)]
        assert torch.equal(ps, mine)
        out[f"pixel_shuffle_r{r}"] = ps.long().numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "g1_index.npz"), **out)
    print(f"G1 index fixtures: {len(out)} arrays")


# ------------------------------------------------------------------ whole-model fixtures
def golden_model(T, name, cfg: O.TulipConfig, batch, seed, with_grads, drop_path=False, bf16_ref=True):
    print(f"== {name}")
    sd = O.key_seeded_state_dict(cfg, seed=seed)
    lo, hi = O.synthetic_batch(cfg, batch, seed=1234 + seed)
    m = ref_model(T, cfg, drop_path_rate=cfg.drop_path_rate if drop_path else 0.0)
    ref_sd = m.state_dict()
    spec = O.state_dict_spec(cfg)
    assert list(ref_sd.keys()) == list(spec.keys()), "state_dict key order/name mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == spec[k][0], (k, v.shape, spec[k][0])
    m.load_state_dict(sd, strict=True)
    out = {"n_params": np.int64(sum(p.numel() for p in m.parameters()))}
    meta = {"name": name, "batch": batch, "seed": seed, "cfg": cfg.__dict__, "drop_path": drop_path}

    drop_u = None
    if drop_path:
        m.train()
        torch.manual_seed(4321)
        enc, dec = O.drop_path_rates(cfg)
        # reference draws torch.rand((B,1,1,1)) twice per block with rate>0, in execution order
        g = torch.Generator().manual_seed(4321)
        drop_u = {}
        order = [(f"layers.{s}.blocks.{b}", enc[s][b]) for s in range(cfg.num_layers) for b in range(cfg.depths[s])]
        order += [(f"layers_up.{i}.blocks.{b}", dec[i][b]) for i in range(cfg.num_layers - 1)
                  for b in range(cfg.depths[cfg.num_layers - i - 2])]
        for p, rate in order:
            if rate > 0:
                drop_u[p] = torch.stack([torch.rand(batch, generator=g), torch.rand(batch, generator=g)])
        out["drop_u_keys"] = np.array(list(drop_u.keys()))
        out["drop_u"] = torch.stack([drop_u[k] for k in drop_u]).numpy()
    else:
        m.eval()

    if with_grads:
        m.zero_grad()
        pred, loss, pix = m(lo, hi)
        loss.backward()
        ref_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        opred, oloss, opix, ograds = O.tulip_loss_and_grads(sd, cfg, lo, hi, drop_u=drop_u)
    else:
        with torch.no_grad():
            pred, loss, pix = m(lo, hi)
            opred, oloss, opix = O.tulip_forward(sd, cfg, lo, hi, drop_u=drop_u)
        ref_grads = ograds = None

    check("pred", opred, pred, 2e-5)
    check("loss", oloss, loss, 1e-6)
    check("pixel_loss", opix, pix, 1e-6)
    if with_grads:
        worst = 0.0
        for k in ref_grads:
            a, b = ograds[k].double(), ref_grads[k].double()
            e = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
            worst = max(worst, e)
            assert e < 2e-3, (k, e)
        print(f"  grads: {len(ref_grads)} tensors, worst rel-to-max err {worst:.3e}")
        out["grad_keys"] = np.array(list(ref_grads.keys()))
        out["grad_l2"] = np.array([ref_grads[k].double().norm().item() for k in ref_grads])
        out["grad_absmax"] = np.array([ref_grads[k].abs().max().item() for k in ref_grads])
        for k in ("patch_embed.proj.weight", "layers.0.blocks.1.attn.relative_position_bias_table",
                  "layers.0.blocks.0.attn.qkv.bias", "decoder_pred.weight", "norm_up.weight",
                  "layers.0.downsample.reduction.weight", "skip_connection_layers.0.weight",
                  "first_patch_expanding.expand.bias", "first_patch_expanding.expand.weight",
                  "first_patch_expanding.norm.weight", "layers_up.0.upsample.norm.bias",
                  "final_patch_expanding.expand.weight", "final_patch_expanding.norm.weight",
                  "final_patch_expanding.norm.bias"):
            if k in ref_grads:
                out["grad::" + k] = ref_grads[k].numpy().copy()

    out["loss"] = np.float64(loss.item())
    out["pixel_loss"] = np.float64(pix.item())
    if pred.numel() <= 1 << 17:
        out["pred"] = pred.detach().numpy().copy()
    out["pred_sub257"] = sub(pred)
    out["pred_abs_mean"] = np.float64(pred.detach().abs().mean().item())

    # per-stage activation checksums via the oracle taps (already proven equal at the output)
    taps = {}
    with torch.no_grad():
        O.tulip_forward(sd, cfg, lo, hi, drop_u=drop_u, taps=taps)
    out["tap_keys"] = np.array(list(taps.keys()))
    out["tap_abs_mean"] = np.array([t.abs().double().mean().item() for t in taps.values()])
    out["tap_sum"] = np.array([t.double().sum().item() for t in taps.values()])

    if bf16_ref and not drop_path:
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            pb, lb, xb = m(lo, hi)
        out["autocast_bf16_pred_sub257"] = sub(pb.float())
        out["autocast_bf16_loss"] = np.float64(lb.item())
        d = (pb.float() - pred).abs()
        out["autocast_bf16_vs_fp32_maxabs"] = np.float64(d.max().item())
        out["autocast_bf16_vs_fp32_meanabs"] = np.float64(d.mean().item())
        print(f"  reference self-consistency bf16-autocast vs fp32: max {d.max().item():.3e} mean {d.mean().item():.3e}"
              f"  loss rel {(lb.item() - loss.item()) / loss.item():+.3e}")
        with torch.no_grad():
            lp, ll, _ = O.tulip_forward(sd, cfg, lo, hi, lowp=True)
        d2 = (lp - pred).abs()
        print(f"  oracle lowp(bf16 operands, fp32 stream) vs fp32: max {d2.max().item():.3e} mean {d2.mean().item():.3e}"
              f"  loss rel {(ll.item() - loss.item()) / loss.item():+.3e}")
        out["oracle_lowp_vs_fp32_maxabs"] = np.float64(d2.max().item())
        out["oracle_lowp_vs_fp32_meanabs"] = np.float64(d2.mean().item())

    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(meta, f, indent=1, default=list)


def golden_lr(T):
    sys.path.insert(0, REF)
    import util.lr_sched as L  # dependency-free
    rows = []
    for (step_epoch, lr, min_lr, warm, epochs) in [(0.0, 5e-4, 0.0, 60, 600), (0.5, 5e-4, 0.0, 60, 600),
                                                  (59.99, 5e-4, 0.0, 60, 600), (60.0, 5e-4, 0.0, 60, 600),
                                                  (300.25, 5e-4, 1e-6, 60, 600), (599.9, 5e-4, 0.0, 60, 600),
                                                  (3.0, 1e-3, 1e-5, 5, 10)]:
        args = types.SimpleNamespace(lr=lr, min_lr=min_lr, warmup_epochs=warm, epochs=epochs)

        class Opt:
            param_groups = [{"lr": 0.0}]
        got = L.adjust_learning_rate(Opt(), step_epoch, args)
        mine = O.cosine_lr(step_epoch, lr, min_lr, warm, epochs)
        assert abs(got - mine) <= 1e-12 * max(1, abs(got)), (got, mine)
        rows.append([step_epoch, lr, min_lr, warm, epochs, got])
    np.savez_compressed(os.path.join(HERE, "g_lr_sched.npz"), table=np.array(rows))
    print("LR schedule fixture:", len(rows), "rows")


def golden_alternates(T):
    """G12: the non-default decoder alternates (tulip.py:126-159): PatchExpanding (patch_unmerging=False) and
    FinalPatchExpanding (pixel_shuffle=False), alone and together; a 3-level model so that a decoder stage's own
    upsample (layers_up.0.upsample) is a PatchExpanding too."""
    tiny3 = dict(depths=(2, 2, 2), num_heads=(3, 6, 12))
    golden_model(T, "g12_tiny3_expanding", O.tiny_config(pixel_shuffle=False, patch_unmerging=False, **tiny3), batch=2,
                 seed=5, with_grads=True, bf16_ref=False)
    golden_model(T, "g12_tiny_patch_expanding", O.tiny_config(patch_unmerging=False), batch=2, seed=6, with_grads=True,
                 bf16_ref=False)
    golden_model(T, "g12_tiny_final_expanding", O.tiny_config(pixel_shuffle=False), batch=2, seed=7, with_grads=True,
                 bf16_ref=False)


def golden_durlar(T):
    """G6: BASELINE config 4 geometry -- tulip_large on DurLAR 32x2048 -> 128x2048 (bash_scripts/
    tulip_upsampling_durlar.sh:11,26-27 runs tulip_base there; tulip_upsampling_carla.sh:10 runs tulip_large at this
    size), B=1, eval forward incl. the bf16-autocast self-consistency band."""
    large = O.tulip_large_config(img_size=(32, 2048), target_img_size=(128, 2048))
    golden_model(T, "g6_durlar_large", large, batch=1, seed=0, with_grads=False, bf16_ref=True)


def golden_base_2048(T):
    """G13: tulip_base on the 2048-wide grids -- the reference's own DurLAR recipe (bash_scripts/tulip_upsampling_durlar.sh:
    11,26-27: tulip_base, 32x2048 -> 128x2048) and the CARLA geometry with the base model (SURVEY 8(d) config 3 "run base
    and large"): B=1, eval forward, sub-sampled prediction + both losses + the bf16-autocast self-consistency band."""
    golden_model(T, "g13_base_16x2048", O.tulip_base_config(img_size=(16, 2048), target_img_size=(64, 2048)), batch=1, seed=3,
                 with_grads=False, bf16_ref=True)
    golden_model(T, "g13_base_32x2048", O.tulip_base_config(img_size=(32, 2048), target_img_size=(128, 2048)), batch=1, seed=4,
                 with_grads=False, bf16_ref=True)


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    T = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "base2048":
        golden_base_2048(T)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "alternates":
        golden_alternates(T)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "eval":
        golden_eval()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "durlar":
        golden_durlar(T)
        return
    golden_index(T)
    golden_lr(T)
    golden_init(T)
    golden_train_trajectory(T)
    golden_train_loop(T)
    golden_transforms()
    golden_eval()
    golden_kitti_projection()
    tiny = O.tiny_config()
    golden_model(T, "g3_tiny_fp32", tiny, batch=2, seed=0, with_grads=True)
    golden_model(T, "g3_tiny_droppath", tiny, batch=4, seed=1, with_grads=True, drop_path=True)
    tiny_nc = O.tiny_config(circular_padding=False)
    golden_model(T, "g3_tiny_noncircular", tiny_nc, batch=2, seed=2, with_grads=False, bf16_ref=False)
    golden_model(T, "g4_kitti_base", O.tulip_base_config(), batch=2, seed=0, with_grads=False)
    large = O.tulip_large_config(img_size=(16, 2048), target_img_size=(64, 2048))
    golden_model(T, "g5_large_16x2048", large, batch=1, seed=0, with_grads=False, bf16_ref=False)
    golden_alternates(T)
    golden_durlar(T)
    golden_base_2048(T)
    print("done")




def golden_init(T):
    """Seeded-initialisation fingerprints of the reference modules (torch.manual_seed(0))."""
    out = {}
    for name, cfg, fac in [("tiny", O.tiny_config(), None), ("base", O.tulip_base_config(), "tulip_base")]:
        torch.manual_seed(0)
        if fac:
            m = getattr(T, fac)(img_size=cfg.img_size, target_img_size=cfg.target_img_size, patch_size=cfg.patch_size,
                                in_chans=1, window_size=[2, 8], pixel_shuffle=True, circular_padding=True,
                                log_transform=True, patch_unmerging=True)
        else:
            m = ref_model(T, cfg, drop_path_rate=0.1)
        sd = m.state_dict()
        out[f"{name}_keys"] = np.array(list(sd.keys()))
        out[f"{name}_sum"] = np.array([v.double().sum().item() for v in sd.values()])
        out[f"{name}_abssum"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        out[f"{name}_numel"] = np.array([v.numel() for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, "g0_init.npz"), **out)
    print("init fingerprints written")


def golden_train_trajectory(T):
    """f-1: the reference model + torch.optim.AdamW (main_lidar_upsampling.py:282-283 grouping and
    hyper-parameters) for 4 steps on one tiny batch, fp32, DropPath off: loss before each step."""
    cfg = O.tiny_config()
    sd = O.key_seeded_state_dict(cfg, seed=3)
    lo, hi = O.synthetic_batch(cfg, 4, seed=77)
    m = ref_model(T, cfg, drop_path_rate=0.0)
    m.load_state_dict(sd, strict=True)
    m.train()
    decay = [p for p in m.parameters() if p.ndim > 1]
    nodecay = [p for p in m.parameters() if p.ndim <= 1]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}],
                            lr=5e-4, betas=(0.9, 0.95))
    losses, pix = [], []
    for step in range(4):
        opt.zero_grad()
        _, loss, px = m(lo, hi)
        loss.backward()
        opt.step()
        losses.append(loss.item()); pix.append(px.item())
    with torch.no_grad():
        _, loss, px = m(lo, hi)
    losses.append(loss.item()); pix.append(px.item())
    np.savez_compressed(os.path.join(HERE, "g7_train_trajectory.npz"), loss=np.array(losses), pixel_loss=np.array(pix),
                        seed=np.int64(3), batch=np.int64(4), data_seed=np.int64(77))
    print("train trajectory:", [round(l, 6) for l in losses])


def golden_train_loop(T):
    """f-1: the reference's train_one_epoch semantics (engine_upsampling.py:46-124) with accum_iter=2 over two
    epochs of 4 micro-batches: util.lr_sched.adjust_learning_rate (imported) at every window start, loss/accum
    backward, util.misc.get_grad_norm_ (imported) before each optimizer step, zero_grad after it."""
    import math
    six = types.ModuleType("torch._six")          # util/misc.py:21 imports `inf` from a module torch no longer has
    six.inf = math.inf
    sys.modules["torch._six"] = six
    import util.lr_sched as lr_sched
    import util.misc as misc
    cfg = O.tiny_config()
    sd = O.key_seeded_state_dict(cfg, seed=5)
    batches = [O.synthetic_batch(cfg, 2, seed=100 + i) for i in range(4)]
    m = ref_model(T, cfg, drop_path_rate=0.0)
    m.load_state_dict(sd, strict=True)
    m.train()
    decay = [p for p in m.parameters() if p.ndim > 1]
    nodecay = [p for p in m.parameters() if p.ndim <= 1]
    args = types.SimpleNamespace(lr=5e-4, min_lr=1e-5, warmup_epochs=1, epochs=3, accum_iter=2)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}],
                            lr=args.lr, betas=(0.9, 0.95))
    losses, lrs, norms = [], [], []
    for epoch in range(2):
        opt.zero_grad()
        for it, (lo, hi) in enumerate(batches):
            if it % args.accum_iter == 0:
                lr_sched.adjust_learning_rate(opt, it / len(batches) + epoch, args)
            _, loss, _ = m(lo, hi)
            losses.append(loss.item())
            (loss / args.accum_iter).backward()
            if (it + 1) % args.accum_iter == 0:
                norms.append(misc.get_grad_norm_(m.parameters()).item())
                opt.step()
                opt.zero_grad()
            lrs.append(opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(HERE, "g9_train_loop.npz"), loss=np.array(losses), lr=np.array(lrs),
                        grad_norm=np.array(norms), seed=np.int64(5), batch=np.int64(2), data_seed0=np.int64(100),
                        n_batches=np.int64(4), accum_iter=np.int64(2), epochs_run=np.int64(2),
                        sched=np.array([args.lr, args.min_lr, args.warmup_epochs, args.epochs]))
    print("train loop:", [round(l, 6) for l in losses], [round(n_, 5) for n_ in norms], lrs)


def import_reference_datasets():
    """util/datasets.py needs torchvision and timm.data at import time (DatasetFolder base class, ImageNet
    helpers); neither is used by the transform classes or the two loaders captured here."""
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvd = types.ModuleType("torchvision.datasets")
    tvv = types.ModuleType("torchvision.datasets.vision")
    tvd.ImageFolder = tvd.DatasetFolder = tvv.VisionDataset = type("_Absent", (), {})
    tv.transforms, tv.datasets = tvt, tvd
    td = types.ModuleType("timm.data")
    td.create_transform = None
    tdc = types.ModuleType("timm.data.constants")
    tdc.IMAGENET_DEFAULT_MEAN = tdc.IMAGENET_DEFAULT_STD = None
    tdd = types.ModuleType("timm.data.dataset")
    tdd.ImageDataset = object
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.datasets": tvd,
                        "torchvision.datasets.vision": tvv, "timm.data": td, "timm.data.constants": tdc,
                        "timm.data.dataset": tdd})
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import util.datasets as D
    return D


def golden_transforms():
    """f-3: the reference's transform classes composed in the order of build_{durlar,kitti,carla}_upsampling_dataset
    (datasets.py:244-340), and its npy/rimg loaders on files written here.  ToTensor is absent (torchvision):
    for float32 (H, W) input it is from_numpy(a)[None]."""
    import tempfile
    from oracle import data_oracle as DO
    D = import_reference_datasets()
    out = {}
    cases = [("kitti", (16, 64), (64, 64), True, None), ("kitti_w", (16, 32), (64, 64), True, None),
             ("durlar", (32, 128), (128, 128), True, 37), ("carla", (8, 96), (32, 96), False, None),
             ("durlar_lin", (16, 64), (64, 64), False, 5)]
    for i, (name, lo_size, hi_size, log_t, shift) in enumerate(cases):
        ds = name.split("_")[0]
        raw = DO.synthetic_raw(2, hi_size[0], hi_size[1], seed=40 + i)
        if ds == "durlar":
            t_lo = [D.ScaleTensor(1 / 120), D.FilterInvalidPixels(min_range=0.3 / 120, max_range=1)]
            t_hi = [D.ScaleTensor(1 / 120), D.FilterInvalidPixels(min_range=0.3 / 120, max_range=1)]
        elif ds == "kitti":
            t_lo, t_hi = [D.ScaleTensor(1 / 80)], [D.ScaleTensor(1 / 80)]
        else:
            t_lo = [D.ScaleTensor(1 / 80), D.FilterInvalidPixels(min_range=2 / 80, max_range=1)]
            t_hi = [D.ScaleTensor(1 / 80), D.FilterInvalidPixels(min_range=2 / 80, max_range=1)]
        t_lo.append(D.DownsampleTensor(h_high_res=hi_size[0], downsample_factor=hi_size[0] // lo_size[0]))
        if hi_size[1] // lo_size[1] > 1:
            t_lo.append(D.DownsampleTensorWidth(w_high_res=hi_size[1], downsample_factor=hi_size[1] // lo_size[1]))
        if log_t:
            t_lo.append(D.LogTransform()); t_hi.append(D.LogTransform())
        if shift is not None:
            t_lo.append(D.RandomRollRangeMap(shift=shift)); t_hi.append(D.RandomRollRangeMap(shift=shift))
        los, his = [], []
        for b in range(raw.shape[0]):
            lo = hi = torch.from_numpy(raw[b].numpy())[None]
            for t in t_lo:
                lo = t(lo)
            for t in t_hi:
                hi = t(hi)
            los.append(lo); his.append(hi)
        lo, hi = torch.stack(los), torch.stack(his)
        olo, ohi = DO.range_prep(raw, DO.DATASETS[ds], lo_size, hi_size, log_t, shift)
        assert torch.equal(olo, lo) and torch.equal(ohi, hi), name
        out[f"{name}_lo"], out[f"{name}_hi"] = lo.numpy(), hi.numpy()
        out[f"{name}_meta"] = np.array([40 + i, *lo_size, *hi_size, int(log_t), -1 if shift is None else shift])
    # loaders
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory() as tmp:
        a = (rng.random((16, 48, 2)) * 100).astype(np.float32)
        p = os.path.join(tmp, "a.npy")
        np.save(p, a)
        out["npy_bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        out["npy_expected"] = D.npy_loader(p)
        pay = (rng.random((40, 12)) * 80).astype(np.float16)          # stored (s1, s0) with header (s0, s1)
        p = os.path.join(tmp, "a.rimg")
        with open(p, "wb") as f:
            np.array([12, 40], dtype=np.uint).tofile(f)
            pay.tofile(f)
        out["rimg_bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        out["rimg_expected"] = D.rimg_loader(p)
    assert np.array_equal(DO.npy_range(out["npy_bytes"].tobytes()), out["npy_expected"])
    assert np.array_equal(DO.rimg_range(out["rimg_bytes"].tobytes()), out["rimg_expected"])
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "g8_transforms.npz"), **out)
    print("transform fixtures written:", [c[0] for c in cases], out["rimg_expected"].shape)


def import_reference_engine():
    """engine_upsampling.py imports plotting/export helpers and the Chamfer CUDA extension at module level.
    Stand-ins that do not touch the evaluated arithmetic: torchvision.utils.make_grid (tensorboard image),
    trimesh (.ply export, off), torch._six.inf.  The Chamfer extension is absent: the class below restates it
    (squared float32 nearest-neighbour distances, oracle/eval_oracle.py) so the loop can run; chamfer values
    in the fixture are therefore NOT reference outputs."""
    import math
    from oracle import eval_oracle as EO
    import_reference_datasets()                       # torchvision stubs
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda imgs, nrow=1: torch.zeros(3, 1, 1)
    sys.modules["torchvision.utils"] = tvu
    sys.modules["torchvision"].utils = tvu
    sys.modules["trimesh"] = types.ModuleType("trimesh")
    six = types.ModuleType("torch._six")
    six.inf = math.inf
    sys.modules["torch._six"] = six

    class ChamferDistance:
        def __call__(self, a, b):
            a, b = a[0].float(), b[0].float()

            def nn_sq(p, q):
                out = torch.empty(p.shape[0])
                for s0 in range(0, p.shape[0], 2048):
                    d = p[s0:s0 + 2048, None, :] - q[None, :, :]
                    out[s0:s0 + 2048] = (d * d).sum(-1).min(dim=1).values
                return out
            return nn_sq(a, b)[None], nn_sq(b, a)[None], None, None

    cd = types.ModuleType("chamfer_distance")
    cd.ChamferDistance = ChamferDistance
    sys.modules["chamfer_distance"] = cd
    for k in [k for k in sys.modules if k == "util.evaluation"]:
        del sys.modules[k]                             # re-import against the stand-in above
    torch.Tensor.cuda = lambda self, *a, **k: self     # evaluation.py:126-127 moves the clouds to the GPU
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import engine_upsampling as E
    return E


def golden_eval():
    """f-2 / f-4: the reference's evaluate() and MCdrop() run end to end on CPU over seeded synthetic images with a
    stand-in model whose outputs are seeded tensors; results.txt metrics and the intermediate point clouds
    (recorded by wrapping the img_to_pcd_* names the loops call) are stored, and oracle/eval_oracle.py is
    asserted against them."""
    import json
    import tempfile
    import warnings
    from oracle import eval_oracle as EO
    warnings.filterwarnings("ignore")
    E = import_reference_engine()
    import util.evaluation as RE
    out = {"durlar_elevation_lut": np.array(RE.elevation_lut), "durlar_offset_lut": np.array(RE.offset_lut),
           "durlar_azimuth_lut": np.array(RE.azimuth_lut)}
    recorded = []
    for fn in ("img_to_pcd_kitti", "img_to_pcd_carla", "img_to_pcd_durlar"):
        def wrap(f):
            def g(img, *a, **k):
                r = f(img, *a, **k)
                recorded.append((np.array(img, copy=True), r))
                return r
            return g
        setattr(E, fn, wrap(getattr(RE, fn)))

    class Writer:
        log_dir = "none"
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass

    class Model(nn.Module):
        def __init__(self, preds, mc_sigma=0.0):
            super().__init__()
            self.preds, self.i, self.calls, self.mc_sigma = preds, 0, 0, mc_sigma
        def forward(self, lo, hi, eval=False, mc_drop=False):
            if not mc_drop:
                p = self.preds[self.i]; self.i += 1
                return p, None, None
            n = lo.shape[0]
            base = self.preds[self.i]
            outs = []
            for k in range(n):
                g = torch.Generator().manual_seed(9000 + self.calls); self.calls += 1
                outs.append(base[0] + self.mc_sigma * torch.randn(base.shape[1:], generator=g) *
                            (torch.rand(base.shape[1:], generator=g) < 0.3))
            if self.calls % 12 == 0:
                self.i += 1
            return torch.stack(outs)

    cases = [  # name, dataset, (H,W), (h,w), log, mc, keep_close, n_images
        ("kitti", "kitti", (64, 1024), (16, 1024), True, False, False, 2),
        ("carla", "carla", (32, 256), (8, 256), False, False, False, 2),
        ("carla_w", "carla", (32, 256), (8, 128), True, False, False, 1),
        ("durlar", "durlar", (128, 256), (32, 256), True, False, True, 2),
        ("kitti_mc", "kitti", (64, 1024), (16, 1024), True, True, True, 1),
        ("durlar_mc", "durlar", (128, 256), (32, 256), True, True, False, 1),
        # BASELINE.json configs[2] at its full size (bash_scripts/tulip_evaluate_carla.sh: 16x2048 -> 64x2048)
        ("carla_full", "carla", (64, 2048), (16, 2048), True, False, False, 1),
    ]
    for ci, (name, ds, HW, hw, log_t, mc, keep, n_img) in enumerate(cases):
        data = [EO.synthetic_eval_case(ds, *HW, *hw, seed=500 + 10 * ci + k, log_transform=log_t) for k in range(n_img)]
        loader = [({"sample": lo}, {"sample": hi}) for _, hi, lo in data]
        thr = 0.0005 if ds == "durlar" else 0.03
        with tempfile.TemporaryDirectory() as tmp:
            args = types.SimpleNamespace(img_size_low_res=hw, img_size_high_res=HW, grid_size=0.1, log_transform=log_t,
                                         dataset_select=ds, output_dir=tmp, save_pcd=False, keep_close_scan=keep,
                                         num_mcdropout_iterations=12, noise_threshold=thr)
            model = Model([p for p, _, _ in data], mc_sigma=0.01)
            recorded.clear()
            if mc:
                E.MCdrop(loader, model, "cpu", Writer(), args)
                res = json.load(open(os.path.join(tmp, "results_mcdrop.txt")))
            else:
                E.evaluate(loader, model, "cpu", Writer(), args)
                res = json.load(open(os.path.join(tmp, "results.txt")))
        for k in ("mae", "chamfer_dist", "iou", "precision", "recall", "f1"):
            out[f"{name}_{k}"] = np.array(res[k], dtype=np.float64)
        # ---- oracle vs reference, image by image
        el = out["durlar_elevation_lut"]
        for k, (pred, hi, lo) in enumerate(data):
            if mc:
                stack = []
                for c in range(12):
                    g = torch.Generator().manual_seed(9000 + c)
                    stack.append(pred[0] + 0.01 * torch.randn(pred.shape[1:], generator=g) *
                                 (torch.rand(pred.shape[1:], generator=g) < 0.3))
                pred_in = EO.mc_aggregate(torch.stack(stack), thr)
            else:
                pred_in = pred
            mae, mae_low, p_img, t_img = EO.postprocess(pred_in, hi, lo, ds, log_t, mc_drop=mc, keep_close_scan=keep)
            (rp_img, rp_pcd), (rt_img, rt_pcd) = recorded[2 * k], recorded[2 * k + 1]
            assert np.array_equal(p_img, rp_img) and np.array_equal(t_img, rt_img), name
            if ds == "kitti":
                op, ot = (EO.spherical_pcd(im, EO.kitti_tables(), 80) for im in (p_img, t_img))
            elif ds == "carla":
                op, ot = (EO.spherical_pcd(im, EO.carla_tables(*HW), 80) for im in (p_img, t_img))
            else:
                op, ot = (EO.durlar_pcd(im, el, 120) for im in (p_img, t_img))
            assert op.dtype == rp_pcd.dtype and np.array_equal(op, rp_pcd) and np.array_equal(ot, rt_pcd), name
            iou, prec, rec, f1, dims = EO.voxel_metrics(op, ot, 0.1)
            assert abs(mae - res["mae"][k]) <= 1e-7 * max(1.0, abs(mae)), (name, mae, res["mae"][k])
            if "iou" in res and len(res["iou"]) > k:
                assert (iou, prec, rec, f1) == (res["iou"][k], res["precision"][k], res["recall"][k], res["f1"][k]), name
            cd = EO.chamfer_sq(ot, op)
            assert abs(cd - res["chamfer_dist"][k]) <= 1e-6 * cd, (name, cd, res["chamfer_dist"][k])
            out[f"{name}_{k}_pcd_pred_sample"] = rp_pcd[::97].copy()
            out[f"{name}_{k}_pcd_gt_sample"] = rt_pcd[::97].copy()
            out[f"{name}_{k}_voxel"] = np.array([iou, prec, rec, f1], dtype=np.float64)
            out[f"{name}_{k}_dims"] = np.array(dims)
            out[f"{name}_{k}_mae_low"] = np.float64(mae_low)
            print(f"  {name}[{k}] mae={mae:.6f} cd={cd:.6f} iou={iou:.4f} p={prec:.4f} r={rec:.4f} dims={dims}")
        out[f"{name}_meta"] = np.array([ci, *HW, *hw, int(log_t), int(mc), int(keep), n_img])
    out["cases"] = np.array([c[0] for c in cases])
    out["case_dataset"] = np.array([c[1] for c in cases])
    np.savez_compressed(os.path.join(HERE, "g10_eval.npz"), **out)
    print("eval fixtures written")


def golden_kitti_projection():
    """f-3 (producer side): kitti_utils/sample_kitti_dataset.py create_range_map on a seeded synthetic scan.
    The module imports cv2 at the top (unused by the function: stubbed) and calls np.round_, an alias NumPy 2.0
    removed (restored as np.round, the same function)."""
    from oracle import data_oracle as DO
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    if not hasattr(np, "round_"):
        np.round_ = np.round
    sys.path.insert(0, os.path.join(os.path.dirname(REF), "kitti_utils"))
    import sample_kitti_dataset as K
    pts = DO.synthetic_kitti_scan(120000, seed=3)
    ref = K.create_range_map(pts.copy(), image_rows_full=64, image_cols=1024, ang_start_y=24.8, ang_res_y=26.8 / 63,
                             ang_res_x=360 / 1024, max_range=120, min_range=0)
    assert np.array_equal(ref, DO.kitti_range_map(pts))
    np.savez_compressed(os.path.join(HERE, "g11_kitti_projection.npz"), n=np.int64(120000), seed=np.int64(3),
                        every_third_column=ref[:, ::3, :].copy(), sums=ref.astype(np.float64).sum(axis=(0, 1)),
                        nonzero=np.array([(ref[..., 0] > 0).sum(), (ref[..., 1] > 0).sum()]))
    print("kitti projection fixture written:", ref.shape, (ref[..., 0] > 0).mean())


if __name__ == "__main__":
    if "--kitti-projection-only" in sys.argv:
        golden_kitti_projection()
    elif "--eval-only" in sys.argv:
        golden_eval()
    elif "--transforms-only" in sys.argv:
        golden_transforms()
    elif "--loop-only" in sys.argv:
        golden_train_loop(import_reference())
    elif "--traj-only" in sys.argv:
        golden_train_trajectory(import_reference())
    elif "--init-only" in sys.argv:
        golden_init(import_reference())
    else:
        main()
