"""CPU: host-side logic of the drop-in module and engine (no kernels are launched)."""
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import tulip_oracle as O
from tulip_amd.ddp import plan_buckets
from tulip_amd.engine import ALIGN, FlatParams, TulipEngine, effective_window
from tulip_amd.model import tulip as T
from tulip_amd.trainer import cosine_lr

KW = dict(patch_size=(1, 4), in_chans=1, window_size=[2, 8], pixel_shuffle=True, circular_padding=True,
          log_transform=True, patch_unmerging=True)


def tiny():
    cfg = O.tiny_config()
    return cfg, T.TULIP(img_size=cfg.img_size, target_img_size=cfg.target_img_size, depths=cfg.depths,
                        embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, norm_layer=partial(nn.LayerNorm, eps=1e-6), **KW)


def test_factories_and_state_dict_match_reference_contract(golden_dir):
    z = np.load(os.path.join(golden_dir, "g0_init.npz"))
    for name, build, cfg in [
        ("tiny", lambda: tiny()[1], O.tiny_config()),
        ("base", lambda: T.tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), **KW), O.tulip_base_config()),
    ]:
        torch.manual_seed(0)
        m = build()
        sd = m.state_dict()
        assert list(sd.keys()) == z[f"{name}_keys"].tolist() == list(O.state_dict_spec(cfg).keys())
        for k, s, a, n in zip(sd, z[f"{name}_sum"], z[f"{name}_abssum"], z[f"{name}_numel"]):
            v = sd[k]
            assert v.numel() == n and tuple(v.shape) == O.state_dict_spec(cfg)[k][0]
            # same registration + init order as the reference => same seeded weights
            assert abs(v.double().sum().item() - s) <= 1e-9 * max(1.0, a), k
            assert abs(v.double().abs().sum().item() - a) <= 1e-9 * max(1.0, a), k
        assert sd["layers.0.blocks.0.attn.relative_position_index"].dtype == torch.int64
        # engine_upsampling.enable_dropout relies on nn.Dropout* modules being present (engine:39-43)
        assert any(type(x).__name__.startswith("Dropout") for x in m.modules())
    assert sum(p.numel() for p in T.tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), **KW).parameters()) \
        == 27_149_076


def test_constructor_quirks():
    with pytest.raises(AttributeError):          # the reference's swin_v2 branch crashes the same way
        T.TULIP(swin_v2=True, **{k: v for k, v in KW.items()})
    with pytest.raises(NotImplementedError):
        T.TULIP(patch_norm=False, **KW)
    m = T.tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), **KW)
    assert m.upscale_factor == 4 and m.drop_path == 0.1
    rates = [b.drop_path_rate for s in m.layers for b in s.blocks]
    enc, _ = O.drop_path_rates(O.tulip_base_config())
    assert rates == [r for s in enc for r in s]
    assert [b.drop_path_rate for b in m.layers_up[0].blocks] == enc[2]   # decoder reuses encoder slices


def test_reference_default_flags_construct_with_the_reference_state_dict():
    """TULIP() with the reference's own constructor defaults (tulip.py:531-535: pixel_shuffle=False,
    patch_unmerging=False -> PatchExpanding / FinalPatchExpanding, tulip.py:126-159) and the two mixed flag sets: keys,
    order and shapes of the state_dict are the reference's (the spec is pinned to the reference by the g12 fixtures)."""
    ref_defaults = O.TulipConfig(img_size=(32, 2048), target_img_size=(128, 2048), patch_size=(4, 4), window_size=(4, 4),
                                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), ln_eps=1e-5, pixel_shuffle=False,
                                 circular_padding=False, log_transform=False, patch_unmerging=False)
    m = T.TULIP()
    spec = O.state_dict_spec(ref_defaults)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == spec[k][0] for k in sd)
    assert "final_patch_expanding.norm.weight" in sd and "first_patch_expanding.norm.bias" in sd
    assert "layers_up.0.upsample.expand.weight" in sd and sd["layers_up.0.upsample.expand.weight"].ndim == 2
    for ps, pu in [(True, False), (False, True), (False, False)]:
        cfg = O.tiny_config(pixel_shuffle=ps, patch_unmerging=pu, depths=(2, 2, 2), num_heads=(3, 6, 12))
        kw = dict(KW, pixel_shuffle=ps, patch_unmerging=pu)
        m = T.TULIP(img_size=cfg.img_size, target_img_size=cfg.target_img_size, depths=cfg.depths, embed_dim=cfg.embed_dim,
                    num_heads=cfg.num_heads, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw)
        spec = O.state_dict_spec(cfg)
        assert list(m.state_dict().keys()) == list(spec.keys())
        assert all(tuple(v.shape) == spec[k][0] for k, v in m.state_dict().items())
        # the flat layout orders every parameter of the alternates too
        from tulip_amd.engine import FlatParams
        order, marks = FlatParams._completion_order(m, dict(m.named_parameters()))
        assert sorted(order) == sorted(dict(m.named_parameters()))


def test_cpu_forward_is_refused():
    cfg, m = tiny()
    lo, hi = O.synthetic_batch(cfg, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(lo, hi)


def test_effective_window_matches_oracle():
    for H in (1, 2, 4, 16):
        for shift in (False, True):
            assert effective_window(H, (2, 8), shift) == O.effective_window(H, (2, 8), shift)


def test_flat_params_layout_groups_and_decay_mask():
    cfg, m = tiny()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m, torch.device("cpu"))
    # parameters are now views of the flat buffer, values unchanged, state_dict intact
    for k, p in m.named_parameters():
        assert p.data_ptr() == fp.p32(k) and torch.equal(p.detach(), before[k])
        assert fp.offset[k] % ALIGN == 0
    assert fp.still_bound(m)
    assert list(m.state_dict().keys()) == list(before.keys())
    # completion order: head first, patch embedding last, groups end on parameter boundaries and cover all
    assert fp.names[0].startswith("decoder_pred") and fp.names[-1].startswith("patch_embed")
    tags = [t for t, _ in fp.groups]
    assert tags[0] == "head" and tags[-1] == "embed" and fp.groups[-1][1] == fp.total
    ends = [e for _, e in fp.groups]
    assert ends == sorted(ends)
    # decay mask: ndim>1 parameters only (timm grouping, main_lidar_upsampling.py:282)
    mask = fp.decay_mask
    for k, p in m.named_parameters():
        blk = mask[fp.offset[k] // ALIGN:(fp.offset[k] + p.numel() + ALIGN - 1) // ALIGN]
        assert bool(blk.all()) == (p.ndim > 1) and bool(blk.any()) == (p.ndim > 1), k
    # moving the module invalidates the binding
    m.double()
    assert not fp.still_bound(m)


def test_bucket_plan_covers_buffer_once():
    m = T.tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), **KW)
    fp = FlatParams(m, torch.device("cpu"))
    for mb in (0.001, 1.0, 16.0, 1000.0):
        b = plan_buckets(fp.groups, fp.total, int(mb * (1 << 20) / 4))
        assert b[0][1] == 0 and b[-1][2] == fp.total
        for (_, _, e0), (_, s1, _) in zip(b, b[1:]):
            assert e0 == s1
        assert all(e > s for _, s, e in b)
    assert len(plan_buckets(fp.groups, fp.total, 1 << 40)) == 1
    # a group may be handed to the optimizer when its hook fires, so every parameter must sit in the group where the
    # backward READS it last: the skip Linear of level s is read again at encoder stage s (x_save half of its input
    # gradient, tulip.py:715 backward), blocks / merges / unmerges / head only inside their own stage
    nl = m.num_layers
    starts = dict(zip([t for t, _ in fp.groups], [0] + [e for _, e in fp.groups][:-1]))
    ends = dict(fp.groups)
    group_of = lambda name: next(t for t in starts if starts[t] <= fp.offset[name] < ends[t])
    for i in range(nl - 1):
        for suffix in ("weight", "bias"):
            assert group_of(f"skip_connection_layers.{i}.{suffix}") == f"enc{nl - i - 2}"
    assert group_of("first_patch_expanding.expand.weight") == f"enc{nl - 1}"
    assert group_of("layers.0.downsample.reduction.weight") == "enc1"
    assert group_of("layers_up.0.blocks.0.attn.qkv.weight") == "dec0"
    assert group_of("decoder_pred.weight") == "head" and group_of("patch_embed.proj.weight") == "embed"


def test_engine_block_specs():
    m = T.tulip_large(img_size=(16, 2048), target_img_size=(64, 2048), **KW)
    eng = TulipEngine(m)
    assert len(eng.blocks) == 10 + 8
    deep = eng.enc_blocks[4]
    assert (deep[0].H, deep[0].W, deep[0].C, deep[0].nh) == (1, 32, 1536, 48)
    assert deep[0].win == (1, 16) and deep[0].sft == (0, 0) and deep[1].sft == (0, 8)     # backup window
    assert eng.enc_blocks[0][1].sft == (1, 4)
    assert eng.blocks[0].slot == -1 and eng.n_drop_slots == 2 * (len(eng.blocks) - 2)     # rate 0 twice (enc0.0, dec last.0)


def test_cosine_lr_table(golden_dir):
    t = np.load(os.path.join(golden_dir, "g_lr_sched.npz"))["table"]
    for row in t:
        assert abs(cosine_lr(row[0], row[1], row[2], row[3], row[4]) - row[5]) <= 1e-12 * max(1.0, abs(row[5]))


def test_oracle_e4m3_rounding_known_answers():
    """The rounding model behind the fp8 attention scores (oracle._FP8Round = torch.float8_e4m3fn, the OCP e4m3 of
    gfx950): round to nearest even on the 3-bit mantissa, subnormals down to 2^-9, largest finite value 448."""
    import torch
    from oracle import tulip_oracle as O
    x = torch.tensor([1.0, 1.0625, 1.07, 1.1875, 17.0, 19.0, 448.0, 2.0 ** -9, 0.0009, 0.0015, -3.3, 0.0])
    want = torch.tensor([1.0, 1.0, 1.125, 1.25, 16.0, 20.0, 448.0, 2.0 ** -9, 0.0, 2.0 ** -9, -3.25, 0.0])
    got = O._FP8Round.apply(x)
    assert torch.equal(got, want), (got, want)
    # straight-through gradient, and every e4m3 value is a bf16 value (what the HIP backward relies on)
    y = x.clone().requires_grad_(True)
    O._FP8Round.apply(y).sum().backward()
    assert torch.equal(y.grad, torch.ones_like(x))
    grid = torch.arange(-448, 449, dtype=torch.float32) / 7.0
    r = O._FP8Round.apply(grid)
    assert torch.equal(r.to(torch.bfloat16).float(), r)


def test_weight_copies_of_the_fused_stage_boundaries_and_the_pack_mark():
    """Round 6 (csrc/glue.hip): which boundary weights get fragment-major copies -- only those a fused form can stream (widths 96 /
    192 / 384) --, in which refresh piece each sits, that the deep widths are NOT maintained until a plan streams them, and which
    completion group carries the end-of-step refresh's mark once the skip Linears are rewritten late (TulipEngine.bind, CPU: no launch)."""
    m = T.tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), **KW)
    eng = m.engine()
    eng.bind(torch.device("cpu"))
    W = eng.params
    glue = sorted(n for n, w in W.pk_width.items() if w == 0)
    assert glue == sorted([f"layers.{s}.downsample.reduction.weight" for s in range(3)] + [f"skip_connection_layers.{i}.weight" for i in range(3)]
                          + ["layers_up.0.upsample.expand.weight", "layers_up.1.upsample.expand.weight",
                             "first_patch_expanding.expand.weight"])        # (C = 768: no fused form, but its GEMM launches read the copies)
    # the small-K GEMM form reads boundary copies only (pseudo-width 0), at a row-tile offset for a sub-block of the transpose
    assert eng._pk("first_patch_expanding.expand.weight") == W.p16p("first_patch_expanding.expand.weight")
    assert eng._pk("layers.1.blocks.0.attn.qkv.weight") is None
    assert eng._pk("skip_connection_layers.2.weight", transposed=True, row0=96, K=96) == W.p16t("skip_connection_layers.2.weight") + 2 * 6 * 3 * 512
    assert W.pk_active == {0, 192, 384} and 768 in {w for w in W.pk_width.values()}
    names = lambda part: {n for (w, q), ent in W._pk_entries.items() if q == part for n, _ in ent}
    assert "layers.0.downsample.reduction.weight" in names(0) and "skip_connection_layers.0.weight" in names(1)
    assert "layers_up.1.upsample.expand.weight" in names(1) and all(W.pk_width[n] == 768 for n in names(2))
    assert eng._pack_mark_tag == "enc0"                  # the level-0 skip Linear sits in the last encoder group ...
    W.pk_late = frozenset(n for n in W.packed_names() if n.startswith("skip_connection_layers."))
    eng._pack_mark_tag_ = None
    assert eng._pack_mark_tag == "enc1"                  # ... until the Trainer rewrites the skip Linears' copies behind its AdamW launch
    # tulip_large: the two deepest boundaries (768 -> 1536, 1536 -> 3072 wide) get no copies
    ml = T.tulip_large(img_size=(16, 2048), target_img_size=(64, 2048), **KW)
    el = ml.engine()
    el.bind(torch.device("cpu"))
    gl = {n for n, w in el.params.pk_width.items() if w == 0}
    assert "layers.3.downsample.reduction.weight" not in gl and "layers.2.downsample.reduction.weight" in gl
    assert "skip_connection_layers.0.weight" not in gl and "skip_connection_layers.1.weight" in gl        # [768][1536] / [384][768]
    assert {"layers_up.0.upsample.expand.weight", "layers_up.1.upsample.expand.weight", "layers_up.2.upsample.expand.weight"} <= gl
    assert "first_patch_expanding.expand.weight" not in gl                   # C = 1536
