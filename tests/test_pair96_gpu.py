"""Two consecutive C = 96 blocks in one launch (tulip_swin96_pair_fwd, csrc/swin96.hip) against the two launches it replaces
(tulip.py:399-436 runs the blocks of a stage back to back): every tensor either form writes, bit for bit, with both arrival
orders of neighbouring tiles forced, across repeated launches (epoch-stamped flags) and under graph replay."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from tulip_amd import ops as o
    return o


def _engine(B, fp8=False):
    from tulip_amd.model.tulip import tulip_base
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 1 or "relative_position_bias_table" in n:
                p.add_(0.2 * torch.randn_like(p))
    eng = m.engine()
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    eng.params.refresh_shadow()
    eng.attn_fp8 = fp8
    return eng


NAMES = ["xn1", "mean1", "rstd1", "qkv", "o", "x1", "xn2", "mean2", "rstd2", "h", "g"]


def _descs(eng, P, specs, xin, bufs, save=True):
    out = []
    x = xin
    for sp, b in zip(specs, bufs):
        d = eng._desc96(P, sp, x, b["out"])
        if save:
            d.update(x1=b["x1"], xn1=b["xn1"], qkv=b["qkv"], attn_out=b["o"], xn2=b["xn2"], fc1_pre=b["h"], fc1_act=b["g"],
                     mean1=b["mean1"], rstd1=b["rstd1"], mean2=b["mean2"], rstd2=b["rstd2"])
        else:
            for k in ("x1", "xn1", "qkv", "attn_out", "xn2", "fc1_pre", "fc1_act", "mean1", "rstd1", "mean2", "rstd2"):
                d[k] = None
        out.append(d)
        x = b["out"]
    return out


def _fresh(P, specs):
    bufs = []
    for sp in specs:
        b = {k: torch.full_like(P[sp.prefix + "." + k], float("nan") if P[sp.prefix + "." + k].dtype == torch.float32 else 0)
             for k in NAMES}
        b["out"] = torch.full_like(P[sp.prefix + ".out"], float("nan"))
        bufs.append(b)
    return bufs


@pytest.mark.parametrize("B,fp8,save", [(2, False, True), (8, False, True), (8, True, True), (8, False, False), (3, False, True)])
def test_pair_launch_is_the_two_launches_bit_for_bit(ops, B, fp8, save):
    eng = _engine(B, fp8)
    P = eng.plan(B)
    specs = eng.enc_blocks[0][:2]
    assert [sp.shift for sp in specs] == [False, True] and specs[0].C == 96
    xin = P["enc0.in"]
    xin.copy_((torch.randn(B * specs[0].H * specs[0].W, 96, device=DEV) * 1.5 + 0.2).view_as(xin))
    du = torch.rand(eng.n_drop_slots, B, device=DEV)
    du[:, 0] = 0.01
    eng.draw_drop_scales(P, True, du)
    ref = _fresh(P, specs)
    for d in _descs(eng, P, specs, xin, ref, save):
        ops.swin96_block_fwd(**d)
    torch.cuda.synchronize()
    assert torch.isfinite(ref[1]["out"]).all()
    sync = torch.zeros(ops.swin96_pair_sync_bytes(B, specs[0].H, specs[0].W) // 4, dtype=torch.int32, device=DEV)
    launches = 0
    # arrival orders: as they come, the even tiles late, the odd tiles late (a tile of the second block reads both parities)
    for hold, quanta in [(0, 0), (1, 400), (2, 400), (0, 0), (1, 50)]:
        sync[2], sync[3] = hold, quanta
        got = _fresh(P, specs)
        d0, d1 = _descs(eng, P, specs, xin, got, save)
        ops.swin96_pair_fwd(d0, d1, sync)
        torch.cuda.synchronize()
        launches += 1
        assert sync[0].item() == launches and sync[1].item() == 0          # the epoch advanced once, everybody left
        assert sync[4].item() == 0                                          # no poll gave up
        for r, g in zip(ref, got):
            for k in (NAMES + ["out"]) if save else ["out"]:
                assert torch.equal(r[k], g[k]), (hold, k)
    # graph replay: no clearing between replays
    sync[2] = 0
    got = _fresh(P, specs)
    d0, d1 = _descs(eng, P, specs, xin, got, save)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.swin96_pair_fwd(d0, d1, sync)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ops.swin96_pair_fwd(d0, d1, sync)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(6):
        got[1]["out"].fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ref[1]["out"], got[1]["out"])
    assert sync[0].item() == launches + 1 + 6


def test_pair_argument_errors(ops):
    eng = _engine(2)
    P = eng.plan(2)
    specs = eng.enc_blocks[0][:2]
    xin = P["enc0.in"]
    bufs = _fresh(P, specs)
    d0, d1 = _descs(eng, P, specs, xin, bufs)
    sync = torch.zeros(ops.swin96_pair_sync_bytes(2, specs[0].H, specs[0].W) // 4, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        ops.swin96_pair_fwd(d0, d1, sync[:8])                      # too small
    bad = dict(d1, x_in=bufs[1]["x1"])
    with pytest.raises(RuntimeError):
        ops.swin96_pair_fwd(d0, bad, sync)                         # the second block must read the first block's output
    bad = dict(d1, qkv=None, fc1_pre=None)
    with pytest.raises(RuntimeError):
        ops.swin96_pair_fwd(d0, bad, sync)                         # both in one form
    big = torch.zeros(ops.swin96_pair_sync_bytes(16, specs[0].H, specs[0].W) // 4, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        ops.swin96_pair_fwd(dict(d0, B=16), dict(d1, B=16), big)   # 512 tiles: more than one round of the chip


@pytest.mark.parametrize("train", [True, False])
def test_engine_forward_with_pairs_is_bit_identical(ops, train):
    """TulipEngine.pair96: the stage-0 pairs of the encoder and the decoder through the one-launch form -- same loss, same saved
    tensors as the launch per block."""
    res = []
    for pair in (False, True):
        eng = _engine(8)
        eng.pair96 = pair
        P = eng.plan(8)
        g = torch.Generator(device="cpu").manual_seed(5)
        x = torch.rand(8, 1, 16, 1024, generator=g).to(DEV)
        y = torch.rand(8, 1, 64, 1024, generator=g).to(DEV)
        du = torch.rand(eng.n_drop_slots, 8, device=DEV)
        P.x_in.copy_(x)
        P.target.copy_(y)
        eng.draw_drop_scales(P, train, du)
        eng.run_forward(P, with_loss=train)
        torch.cuda.synchronize()
        # (what the forward writes: the backward's buffers are still uninitialised memory here)
        fwd = (".xn1", ".qkv", ".o", ".x1", ".xn2", ".h", ".g", ".out", ".mean1", ".rstd1", ".mean2", ".rstd2", ".cat", ".in")
        if not train:                        # the inference form saves nothing
            fwd = (".out", ".in")
        keep = {k: v.clone() for k, v in P.bufs.items() if k.endswith(fwd) and v.is_floating_point()}
        res.append((keep, P.pred.clone()))
        if pair:
            assert any(k.startswith("xchg.pair.") for k in P.bufs)
    assert torch.equal(res[0][1], res[1][1])
    common = set(res[0][0]) & set(res[1][0])
    assert len(common) > (50 if train else 10)
    for k in sorted(common):
        a, b = res[0][0][k], res[1][0][k]
        assert torch.equal(a.nan_to_num(), b.nan_to_num()), k
