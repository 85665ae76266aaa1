"""CPU: the oracle (oracle/tulip_oracle.py) against the committed golden vectors that
tests/golden/make_golden.py produced from the reference implementation."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tulip_oracle as O


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    with open(os.path.join(golden_dir, name + ".json")) as f:
        meta = json.load(f)
    c = meta["cfg"]
    cfg = O.TulipConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in c.items()})
    return z, meta, cfg


def test_relative_position_index_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_index.npz"))
    idx = O.relative_position_index(2, 8)
    assert np.array_equal(idx, z["rel_pos_index_2x8"])
    # SURVEY 8(a) a2 known answers
    assert idx[0, 0] == 22 and idx[0, 15] == 0 and idx[15, 0] == 44
    assert idx.min() == 0 and idx.max() == 44


@pytest.mark.parametrize("grid", [(8, 64), (4, 32), (16, 256), (2, 32), (1, 32), (32, 512)])
@pytest.mark.parametrize("shift", [False, True])
def test_window_index_and_mask_bit_exact(golden_dir, grid, shift):
    z = np.load(os.path.join(golden_dir, "g1_index.npz"))
    H, W = grid
    tag = f"{H}x{W}_{'s' if shift else 'n'}"
    win, sft = O.effective_window(H, (2, 8), shift)
    tok = O.window_token_index(H, W, win, sft)
    assert np.array_equal(tok, z[f"win_tok_{tag}"])
    # a permutation of all tokens
    assert np.array_equal(np.sort(tok.reshape(-1)), np.arange(H * W))
    if shift:
        m = O.shift_attention_mask(H, W, win, sft)
        assert np.array_equal((m != 0).astype(np.uint8), z[f"mask_{tag}"])
        assert set(np.unique(m)).issubset({0.0, -100.0})
        assert np.array_equal(O.shift_region_labels(H, W, win, sft).astype(np.uint8), z[f"labels_{tag}"])


def test_backup_window_geometry():
    # tulip_large stage 4 at 16x2048: H=1 < 2 -> window (1,16), shift (0,8)  (tulip.py:284-287)
    assert O.effective_window(1, (2, 8), True) == ((1, 16), (0, 8))
    assert O.effective_window(1, (2, 8), False) == ((1, 16), (0, 0))
    assert O.effective_window(2, (2, 8), True) == ((2, 8), (1, 4))
    lab = O.shift_region_labels(1, 32, (1, 16), (0, 8))
    # '-0' slice semantics: last h-slice covers every row -> labels 6,7,8 only
    assert set(np.unique(lab)) == {6, 7, 8}


def test_merge_gather_and_pixel_shuffle(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_index.npz"))
    for (H, W) in [(4, 8), (16, 256)]:
        assert np.array_equal(O.patch_merge_gather_index(H, W), z[f"merge_gather_{H}x{W}"])
    for r in (2, 4):
        ps = z[f"pixel_shuffle_r{r}"]
        C, H, W = 3, 2, 3
        ids = np.arange(C * r * r * H * W).reshape(C * r * r, H, W)
        for c in range(C):
            for i in range(r):
                for j in range(r):
                    assert np.array_equal(ps[0, c, i::r, j::r], ids[O.pixel_shuffle_source_channel(c, i, j, r)])


def test_lr_schedule_table(golden_dir):
    t = np.load(os.path.join(golden_dir, "g_lr_sched.npz"))["table"]
    for row in t:
        got = O.cosine_lr(row[0], row[1], row[2], row[3], row[4])
        assert abs(got - row[5]) <= 1e-12 * max(1.0, abs(row[5]))


def test_param_counts():
    n = lambda cfg: sum(int(np.prod(s)) for s, k in O.state_dict_spec(cfg).values() if k != "index")
    assert n(O.tulip_base_config()) == 27_149_076
    assert n(O.tulip_large_config(img_size=(16, 2048), target_img_size=(64, 2048))) == 108_621_156
    assert n(O.tiny_config()) == 418_248
    assert len(O.state_dict_spec(O.tulip_base_config())) == 226
    assert O.tulip_base_config().upscale_factor == 4


def test_drop_path_rates():
    enc, dec = O.drop_path_rates(O.tulip_base_config())
    flat = [r for s in enc for r in s]
    assert flat[0] == 0.0 and abs(flat[-1] - 0.1) < 1e-7 and abs(flat[1] - 0.1 / 7) < 1e-7
    assert dec[0] == enc[2] and dec[1] == enc[1] and dec[2] == enc[0]


@pytest.mark.parametrize("name", ["g3_tiny_fp32", "g3_tiny_droppath", "g12_tiny3_expanding", "g12_tiny_patch_expanding",
                                  "g12_tiny_final_expanding"])
def test_tiny_model_forward_backward_vs_reference(golden_dir, name):
    """g12_*: the PatchExpanding / FinalPatchExpanding alternates (patch_unmerging=False / pixel_shuffle=False)."""
    z, meta, cfg = _load(golden_dir, name)
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    drop_u = None
    if meta["drop_path"]:
        drop_u = {k: torch.from_numpy(u) for k, u in zip(z["drop_u_keys"].tolist(), z["drop_u"])}
    pred, loss, pix, grads = O.tulip_loss_and_grads(sd, cfg, lo, hi, drop_u=drop_u)
    assert int(z["n_params"]) == sum(v.numel() for v in sd.values() if v.is_floating_point())
    np.testing.assert_allclose(pred.numpy(), z["pred"], rtol=0, atol=5e-6)
    assert abs(loss.item() - float(z["loss"])) <= 1e-6
    assert abs(pix.item() - float(z["pixel_loss"])) <= 1e-6
    for k, l2, amax in zip(z["grad_keys"].tolist(), z["grad_l2"], z["grad_absmax"]):
        assert abs(grads[k].double().norm().item() - l2) <= 1e-4 * l2 + 1e-9, k
    for k in z.files:
        if k.startswith("grad::"):
            g = grads[k[6:]].numpy()
            np.testing.assert_allclose(g, z[k], rtol=0, atol=2e-5 * np.abs(z[k]).max() + 1e-9)


def test_tiny_noncircular_forward(golden_dir):
    z, meta, cfg = _load(golden_dir, "g3_tiny_noncircular")
    assert cfg.circular_padding is False
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    with torch.no_grad():
        pred, loss, pix = O.tulip_forward(sd, cfg, lo, hi)
    np.testing.assert_allclose(pred.numpy(), z["pred"], rtol=0, atol=5e-6)
    assert abs(loss.item() - float(z["loss"])) <= 1e-6


def test_kitti_base_forward_vs_reference(golden_dir):
    """BASELINE config 2 geometry (tulip_base, 16x1024 -> 64x1024), B=2, eval forward."""
    z, meta, cfg = _load(golden_dir, "g4_kitti_base")
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    taps = {}
    with torch.no_grad():
        pred, loss, pix = O.tulip_forward(sd, cfg, lo, hi, taps=taps)
    np.testing.assert_allclose(pred.reshape(-1)[::257].numpy(), z["pred_sub257"], rtol=0, atol=5e-6)
    assert abs(loss.item() - float(z["loss"])) <= 1e-6
    assert abs(pix.item() - float(z["pixel_loss"])) <= 1e-6
    for k, am in zip(z["tap_keys"].tolist(), z["tap_abs_mean"]):
        assert abs(taps[k].abs().double().mean().item() - am) <= 1e-5 * am, k
    # lowp rounding model stays within the reference's own bf16-autocast self-consistency band
    with torch.no_grad():
        lp, ll, _ = O.tulip_forward(sd, cfg, lo, hi, lowp=True)
    d = (lp - pred).abs()
    assert d.max().item() <= float(z["autocast_bf16_vs_fp32_maxabs"]) * 1.5
    assert d.mean().item() <= float(z["autocast_bf16_vs_fp32_meanabs"]) * 1.5
    assert abs(ll.item() - loss.item()) / loss.item() <= 1e-3


@pytest.mark.parametrize("name", ["g13_base_16x2048", "g13_base_32x2048"])
def test_base_2048_forward_vs_reference(golden_dir, name):
    """tulip_base on the 2048-wide grids (bash_scripts/tulip_upsampling_durlar.sh:11,26-27 trains base at 32x2048; CARLA's
    16x2048 geometry with the base model): the oracle against the reference's fp32 forward, B=1, sub-sampled."""
    z, meta, cfg = _load(golden_dir, name)
    sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
    lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
    with torch.no_grad():
        pred, loss, pix = O.tulip_forward(sd, cfg, lo, hi)
    np.testing.assert_allclose(pred.reshape(-1)[::257].numpy(), z["pred_sub257"], rtol=0, atol=5e-6)
    assert abs(loss.item() - float(z["loss"])) <= 1e-6 and abs(pix.item() - float(z["pixel_loss"])) <= 1e-6


def test_adamw_step_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(7, 5)
    g = torch.randn(7, 5)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
    m = torch.zeros_like(p0)
    v = torch.zeros_like(p0)
    q = p0.clone()
    for step in range(1, 4):
        p.grad = g.clone() * step
        opt.step()
        q, m, v = O.adamw_reference_step(q, g * step, m, v, step, 5e-4)
        assert torch.allclose(q, p.detach(), rtol=1e-6, atol=1e-7)


def test_train_loop_vs_reference(golden_dir):
    """8(f)-1: the oracle's restatement of train_one_epoch (accum_iter=2, warm-up + cosine LR, grad norm,
    AdamW with timm's decay grouping) against the REFERENCE model + imported lr_sched/get_grad_norm_ (g9)."""
    z = np.load(os.path.join(golden_dir, "g9_train_loop.npz"))
    cfg = O.tiny_config(drop_path_rate=0.0)
    sd = O.key_seeded_state_dict(cfg, seed=int(z["seed"]))
    batches = [O.synthetic_batch(cfg, int(z["batch"]), seed=int(z["data_seed0"]) + i) for i in range(int(z["n_batches"]))]
    lr, min_lr, warm, epochs = z["sched"].tolist()
    losses, lrs, norms = O.train_loop_reference(sd, cfg, batches, int(z["epochs_run"]), lr, min_lr, warm, epochs,
                                                accum_iter=int(z["accum_iter"]))
    np.testing.assert_allclose(lrs, z["lr"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(losses, z["loss"], rtol=2e-5)
    np.testing.assert_allclose(norms, z["grad_norm"], rtol=2e-4)
