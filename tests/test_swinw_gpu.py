"""GPU: the one-launch Swin block for the wider stages (csrc/swinw.hip, C = 192 / 384) -- forward and backward
  (a) against the 7-kernel sequences they replace, on every tensor either path writes, and
  (b) DIRECTLY against the oracle's restatement of SwinTransformerBlock.forward (tulip.py:338-352) run with the same
      rounding model, forward output and every gradient via the oracle's autograd (no unfused kernel in the loop)."""
import pytest
import torch

from oracle import tulip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (stage, shifted, batch).  Windows per workgroup: C = 192: 2 (4 at B = 16); C = 384: 1 below 256 windows, 2 from B = 16
CASES = [(1, False, 2), (1, True, 2), (2, False, 2), (2, True, 2), (1, True, 16), (2, True, 16)]


def _model(seed):
    from tulip_amd.model.tulip import tulip_base
    torch.manual_seed(seed)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    with torch.no_grad():                                     # non-trivial norms / biases / bias tables
        for n, p in m.named_parameters():
            if p.ndim == 1 or "relative_position_bias_table" in n:
                p.add_(0.2 * torch.randn_like(p))
    eng = m.engine()
    eng.wide_min_windows = 0                                  # C = 384 fused at any batch size (the engine waits for 128 windows)
    eng.bind(torch.device(DEV, torch.cuda.current_device()))
    eng.params.refresh_shadow()
    return m, eng


def _setup(stage, shifted, B, seed, fp8=False):
    m, eng = _model(seed)
    eng.attn_fp8 = fp8                                        # BASELINE configs[4]: attention scores from e4m3 q, k
    P = eng.plan(B)
    sp = eng.enc_blocks[stage][1 if shifted else 0]
    assert sp.shift == shifted and sp.C == 96 << stage and eng._fusable_wide(sp)
    M = B * sp.H * sp.W
    x = torch.randn(M, sp.C, device=DEV) * 1.5 + 0.2
    xin = P[f"enc{stage}.in"]
    xin.copy_(x.view_as(xin))
    du = 0.5 + 0.5 * torch.rand(eng.n_drop_slots, B, device=DEV)   # every other sample kept (rates <= 0.1)
    du[:, 0] = 0.01                                           # sample 0: both branches of every block dropped
    eng.draw_drop_scales(P, True, du)
    return m, eng, P, sp, M, x, xin


def _oracle_block(m, eng, P, sp, x, B, need_grad=False):
    """SwinTransformerBlock.forward of the oracle (bf16 rounding model) on the same input, weights and DropPath draws."""
    sd = {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu())
          for k, v in m.state_dict().items() if k.startswith(sp.prefix + ".")}
    if need_grad:
        for k, v in sd.items():
            if v.is_floating_point():
                v.requires_grad_(True)
    cfg = O.tulip_base_config()
    keep = None
    if sp.slot >= 0:
        keep = torch.stack([P.drop_scale[sp.slot], P.drop_scale[sp.slot + 1]]).cpu()
    xc = x.detach().cpu().reshape(B, sp.H, sp.W, sp.C).clone().requires_grad_(need_grad)
    out = O.swin_block(O._Prec(True, attn_fp8=eng.attn_fp8), sd, sp.prefix, cfg, xc, sp.nh, sp.shift, keep)
    return out, xc, sd


@pytest.mark.parametrize("stage,shifted,B", CASES)
def test_wide_block_forward(stage, shifted, B):
    _forward(stage, shifted, B, False)


@pytest.mark.parametrize("stage,shifted,B", [(1, True, 2), (2, False, 2), (2, True, 16)])
def test_wide_block_forward_fp8_scores(stage, shifted, B):
    """engine.attn_fp8: fused == unfused kernel sequence and == the oracle with e4m3-rounded q, k, to the bf16 bounds."""
    _forward(stage, shifted, B, True)


def _forward(stage, shifted, B, fp8):
    m, eng, P, sp, M, x, xin = _setup(stage, shifted, B, seed=stage, fp8=fp8)
    C, p = sp.C, sp.prefix
    names = ["xn1", "mean1", "rstd1", "qkv", "o", "x1", "xn2", "mean2", "rstd2", "h", "g"]
    res = {}
    for fused in (False, True):
        eng.fuse_wide = fused
        for k in names:
            P[p + "." + k].fill_(float("nan") if P[p + "." + k].dtype == torch.float32 else 0)
        out = torch.full((M, C), float("nan"), device=DEV)
        ob = torch.zeros(M, C, device=DEV, dtype=torch.bfloat16)
        eng._block_fwd(P, sp, xin, out, out_bf16=ob)
        torch.cuda.synchronize()
        res[fused] = {k: P[p + "." + k].float().clone() for k in names}
        res[fused]["out"], res[fused]["out_bf16"] = out.clone(), ob.float().clone()
    eng.fuse_wide = True
    if eng._hgrad_wide(sp, B):      # round 4: the fused forward hands gelu'(h) to the fused backward in the fc1_pre buffer
        h = res[False]["h"]
        res[False]["h"] = (0.5 * (1 + torch.erf(h * 0.7071067811865476)) + h * torch.exp(-0.5 * h * h) * 0.3989422804014327
                           ).bfloat16().float()
    for k, ref in res[False].items():
        a, b = res[True][k].reshape(-1), ref.reshape(-1)
        assert torch.isfinite(a).all(), k
        d = (a - b).abs()
        tol = 1e-5 * (1 + b.abs()) if k in ("mean1", "rstd1") else 2 ** -6 * (0.05 + b.abs())
        frac = (d > tol).float().mean().item()
        rel = (d.norm() / (b.norm() + 1e-12)).item()
        print(f"{k:8s} frac>{'tol'} {frac:.2e} rel {rel:.2e} max {d.max().item():.3e}")
        assert frac <= 2e-3, (k, frac, d.max().item())
        assert rel <= 3e-3, (k, rel)
    if sp.slot >= 0:       # sample 0 had both branches dropped
        assert torch.equal(res[True]["out"][: M // B], x[: M // B])
    # ---- (b) directly against the oracle
    oo, _, _ = _oracle_block(m, eng, P, sp, x, B)
    a, b = res[True]["out"].cpu().reshape(-1), oo.detach().reshape(-1)
    d = (a - b).abs()
    rel = (d.norm() / b.norm()).item()
    print(f"fused block vs oracle: rel L2 {rel:.3e} max {d.max().item():.3e}")
    assert rel <= 2e-3 and d.max().item() <= 3e-2, (rel, d.max().item())


@pytest.mark.parametrize("stage,shifted,B", CASES)
def test_wide_block_backward(stage, shifted, B, dev_lib):      # (the fused backward reading h itself: development build)
    _backward(stage, shifted, B, False)


@pytest.mark.parametrize("stage,shifted,B", [(1, True, 2), (2, False, 2), (2, True, 16)])
def test_wide_block_backward_fp8_scores(stage, shifted, B, dev_lib):
    """engine.attn_fp8: the backward differentiates the function the forward ran (dS times the e4m3-rounded q, k)."""
    _backward(stage, shifted, B, True)


@pytest.mark.parametrize("stage,shifted,B", CASES)
def test_wide_block_pair_product_library(stage, shifted, B):
    """Arm (c) and the oracle comparison on the PRODUCT library (libtulip_hip.so, no dev_lib fixture): fused forward -> fused
    backward with the gelu'(h) hand-off is exactly what a training step launches, so the binary that is benched is the binary whose
    C = 192 / 384 backward meets the oracle per kernel (round-5 review, weak #1).  The development-only arm (fused backward reading
    h itself) stays in test_wide_block_backward."""
    from tulip_amd import _lib
    assert not _lib.dev_active()
    _backward(stage, shifted, B, False, dev_arm=False)


@pytest.mark.parametrize("stage,shifted,B", [(1, True, 2), (2, True, 16)])
def test_wide_block_pair_product_library_fp8_scores(stage, shifted, B):
    from tulip_amd import _lib
    assert not _lib.dev_active()
    _backward(stage, shifted, B, True, dev_arm=False)


def _backward(stage, shifted, B, fp8, dev_arm=True):
    m, eng, P, sp, M, x, xin = _setup(stage, shifted, B, seed=10 + stage, fp8=fp8)
    C, p = sp.C, sp.prefix
    saved = eng.overlap_wgrad
    eng.overlap_wgrad = False                                 # weight gradients and folds inline, on this stream
    out = torch.empty(M, C, device=DEV)
    eng.fuse_wide = False
    eng._block_fwd(P, sp, xin, out)
    dy = torch.randn(M, C, device=DEV)
    cast_buf = torch.zeros(M, C, device=DEV, dtype=torch.bfloat16)
    res = {}
    for fused in ((False, True) if dev_arm else (False,)):      # (True: the fused backward reading h itself -- development build)
        eng.fuse_wide_bwd = fused
        gflat = torch.zeros(eng.params.total, device=DEV)
        G = lambda name: gflat.data_ptr() + 4 * eng.params.offset[name]
        dx = dy.clone()
        cast_buf.zero_()
        eng._pending, eng._lagged_hook = [], None
        eng._block_bwd(P, sp, xin, dx, G, have_dyb=False, next_cast=(cast_buf, None, sp.H * sp.W))
        torch.cuda.synchronize()
        r = {"dx": dx.clone(), "dx_bf16": cast_buf.float().clone(), "dh": P[p + ".dh"].float().clone(),
             "dqkv": P[p + ".dqkv"].float().clone(), "dyb_a": P[p + ".dyb_a"].float().clone(),
             "dyb_m": P[p + ".dyb_m"].float().clone()}
        for n, q in m.named_parameters():
            if n.startswith(p + "."):
                o = eng.params.offset[n]
                r["g:" + n[len(p) + 1:]] = gflat[o:o + q.numel()].clone()
        res[fused] = r
    eng.fuse_wide = eng.fuse_wide_bwd = True
    # (c) the pair as the step runs it: fused forward -> fused backward, with gelu'(h) handed over in the fc1_pre buffer
    # (TULIP_BLOCK_FC1_GRAD, round 4) -- the saved tensors now come from the fused forward
    assert eng._hgrad_wide(sp, B) == eng.fc1_grad_wide
    eng._block_fwd(P, sp, xin, out)
    gflat = torch.zeros(eng.params.total, device=DEV)
    G = lambda name: gflat.data_ptr() + 4 * eng.params.offset[name]
    dx = dy.clone()
    cast_buf.zero_()
    eng._pending, eng._lagged_hook = [], None
    eng._block_bwd(P, sp, xin, dx, G, have_dyb=False, next_cast=(cast_buf, None, sp.H * sp.W))
    torch.cuda.synchronize()
    r = {"dx": dx.clone(), "dx_bf16": cast_buf.float().clone(), "dh": P[p + ".dh"].float().clone(),
         "dqkv": P[p + ".dqkv"].float().clone(), "dyb_a": P[p + ".dyb_a"].float().clone(), "dyb_m": P[p + ".dyb_m"].float().clone()}
    for n, q in m.named_parameters():
        if n.startswith(p + "."):
            o = eng.params.offset[n]
            r["g:" + n[len(p) + 1:]] = gflat[o:o + q.numel()].clone()
    res["pair"] = r
    eng.overlap_wgrad = saved
    for tag in ((True, "pair") if dev_arm else ("pair",)):
        for k, ref in res[False].items():
            a, b = res[tag][k].reshape(-1), ref.reshape(-1)
            assert torch.isfinite(a).all(), k
            rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
            print(f"{str(tag):5s} {k:44s} rel {rel:.3e}")
            assert b.norm().item() > 0, k
            # the unfused chain rounds d(norm input) to bf16 between its kernels, the fused one keeps it in fp32; the pair also
            # differs by the forward's isolated bf16 flips and the bf16 rounding of gelu'(h)
            lim = (2e-2 if "bias_table" in k else 6e-3) * (1.0 if tag is True else 1.35)
            assert rel <= lim, (tag, k, rel)
    res[True] = res["pair"]                                    # (b) below checks the pair against the oracle
    if sp.slot >= 0:       # sample 0: both branches dropped -> the block is the identity there
        assert torch.equal(res[True]["dx"][: M // B], dy[: M // B])
    # ---- (b) directly against the oracle's autograd of the block
    oo, xc, sd = _oracle_block(m, eng, P, sp, x, B, need_grad=True)
    (oo.reshape(M, C) * dy.cpu()).sum().backward()
    ra = ((res[True]["dx"].cpu() - xc.grad.reshape(M, C)).norm() / xc.grad.norm()).item()
    print(f"fused backward vs oracle autograd: dx rel L2 {ra:.3e}")
    assert ra <= 1e-2, ra
    for n, v in sd.items():
        if not v.is_floating_point():
            continue
        g = res[True]["g:" + n[len(p) + 1:]].cpu().reshape(v.shape)
        rel = ((g - v.grad).norm() / (v.grad.norm() + 1e-30)).item()
        print(f"  {n[len(p) + 1:]:40s} vs oracle rel {rel:.3e}")
        assert rel <= (1e-1 if "bias_table" in n else 1.5e-2), (n, rel)


@pytest.mark.parametrize("stage,B", [(1, 2), (2, 2), (1, 64), (2, 64)])
def test_l2_warm_up_changes_no_bit(stage, B):
    """The L2 warm-up at the head of the fused kernels (WeightWarm in csrc/swinw.hip: every workgroup of a one-wave grid,
    the first 256 workgroups of a larger one, batch 64 here) only touches weights: the forward's outputs and saved
    activations and the backward's data gradient and weight-gradient operands are bit-identical with it switched off."""
    from tulip_amd import ops
    m, eng, P, sp, M, x, xin = _setup(stage, True, B, seed=21)
    p = sp.prefix
    res = {}
    for warm in (0, 1):
        eng.no_warm = not warm           # TULIP_BLOCK_NO_WARM in the launch descriptor (per call: the library has no switches)
        out = torch.empty(M, sp.C, device=DEV)
        eng._block_fwd(P, sp, xin, out)
        dx = torch.randn(M, sp.C, device=DEV, generator=torch.Generator(DEV).manual_seed(5))
        saved, eng.overlap_wgrad = eng.overlap_wgrad, False
        gflat = torch.zeros(eng.params.total, device=DEV)
        eng._pending, eng._lagged_hook = [], None
        eng._block_bwd(P, sp, xin, dx, lambda name: gflat.data_ptr() + 4 * eng.params.offset[name], have_dyb=False)
        eng.overlap_wgrad = saved
        torch.cuda.synchronize()
        res[warm] = [out.clone(), dx.clone(), gflat.clone()] + [P[p + s].clone() for s in
                                                               (".xn1", ".qkv", ".o", ".xn2", ".h", ".g", ".dh", ".dqkv")]
    eng.no_warm = False
    for a, b in zip(res[0], res[1]):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def _setup96(shifted, B, seed, fp8=False):
    m, eng = _model(seed)
    eng.attn_fp8 = fp8
    P = eng.plan(B)
    sp = eng.enc_blocks[0][1 if shifted else 0]
    assert sp.shift == shifted and sp.C == 96 and eng._fusable96(sp) and eng.fuse_block96 and eng.fuse_block96_bwd
    M = B * sp.H * sp.W
    x = torch.randn(M, sp.C, device=DEV) * 1.5 + 0.2
    xin = P["enc0.in"]
    xin.copy_(x.view_as(xin))
    du = 0.5 + 0.5 * torch.rand(eng.n_drop_slots, B, device=DEV)
    du[:, 0] = 0.01                                           # sample 0: both branches of every block dropped
    eng.draw_drop_scales(P, True, du)
    return m, eng, P, sp, M, x, xin


@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("fp8", [False, True])
def test_c96_fused_block_directly_against_the_oracle(shifted, fp8):
    """swin96_fwd_kernel / swin96_bwd_kernel (csrc/swin96.hip, one launch each way for a stage-0 block) against the oracle's
    SwinTransformerBlock.forward (tulip.py:338-352) and its autograd under the same rounding model -- no unfused kernel in
    the loop (tests/test_ops_gpu.py compares the fused kernels with the launch sequences they replace)."""
    B = 2
    m, eng, P, sp, M, x, xin = _setup96(shifted, B, seed=31, fp8=fp8)
    C, p = sp.C, sp.prefix
    out = torch.full((M, C), float("nan"), device=DEV)
    eng._block_fwd(P, sp, xin, out)
    torch.cuda.synchronize()
    oo, xc, sd = _oracle_block(m, eng, P, sp, x, B, need_grad=True)
    a, b = out.cpu().reshape(-1), oo.detach().reshape(-1)
    d = (a - b).abs()
    rel = (d.norm() / b.norm()).item()
    print(f"fused C=96 block vs oracle: rel L2 {rel:.3e} max {d.max().item():.3e}")
    assert torch.isfinite(a).all() and rel <= 2e-3 and d.max().item() <= 3e-2, (rel, d.max().item())
    if sp.slot >= 0:
        assert torch.equal(out[: M // B], x[: M // B])
    # backward
    dy = torch.randn(M, C, device=DEV)
    saved, eng.overlap_wgrad = eng.overlap_wgrad, False       # weight gradients and folds inline, on this stream
    gflat = torch.zeros(eng.params.total, device=DEV)
    dx = dy.clone()
    eng._pending, eng._lagged_hook = [], None
    eng._block_bwd(P, sp, xin, dx, lambda name: gflat.data_ptr() + 4 * eng.params.offset[name], have_dyb=False)
    eng.overlap_wgrad = saved
    torch.cuda.synchronize()
    (oo.reshape(M, C) * dy.cpu()).sum().backward()
    ra = ((dx.cpu() - xc.grad.reshape(M, C)).norm() / xc.grad.norm()).item()
    print(f"fused C=96 backward vs oracle autograd: dx rel L2 {ra:.3e}")
    assert ra <= 1e-2, ra
    for n, v in sd.items():
        if not v.is_floating_point():
            continue
        o = eng.params.offset[n]
        g = gflat[o:o + v.numel()].cpu().reshape(v.shape)
        rel = ((g - v.grad).norm() / (v.grad.norm() + 1e-30)).item()
        print(f"  {n[len(p) + 1:]:40s} vs oracle rel {rel:.3e}")
        assert rel <= (1e-1 if "bias_table" in n else 1.5e-2), (n, rel)


def test_pack_multi():
    """tulip_pack_bf16_multi: fragment-major copies of weights and of their transposes (include/tulip_hip.h)."""
    from tulip_amd import ops

    def packed(w):                      # [N][K] -> [N/16][K/32][gq 4][t 16][8]
        N, K = w.shape
        return w.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)

    shapes = [(576, 192), (192, 192), (768, 192), (192, 768), (1152, 384), (64, 96)]
    srcs = [torch.randn(r, c, device=DEV).bfloat16() for r, c in shapes]
    ent, want = [], []
    for s_, (r, c) in zip(srcs, shapes):
        d0, d1 = (torch.zeros(r * c, device=DEV, dtype=torch.bfloat16) for _ in range(2))
        ent.append((s_, d0, r, c, 0)); want.append((d0, packed(s_)))
        if r % 32 == 0 and c % 16 == 0:
            ent.append((s_, d1, r, c, 1)); want.append((d1, packed(s_.t().contiguous())))
    items, n = ops.pack_items(ent)
    ops.pack_bf16_multi(items, n)
    torch.cuda.synchronize()
    for got, ref in want:
        assert torch.equal(got, ref)
