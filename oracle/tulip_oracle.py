"""CPU oracle for the TULIP Swin hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *restatement* of the algorithm in the reference's
``tulip/model/tulip.py`` as pure functions over a ``state_dict`` (numpy for the
integer/index ops, plain PyTorch fp32 eager ops for the float math).  It is the
checker for the HIP path and the ``cpu_baseline`` leg of ``bench.py``; nothing in
``tulip_amd/`` may import it (only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline do).

Parity pinning: the reference ships no tests/golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against the reference itself:
``tests/golden/make_golden.py`` imports ``/root/reference/tulip/model/tulip.py``
in the build container and (i) asserts this file reproduces its outputs and
parameter gradients, (ii) writes the fixtures under ``tests/golden/`` that
``tests/test_oracle_golden.py`` re-checks without the reference.

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class TulipConfig:
    """Constructor arguments of ``TULIP.__init__`` (tulip/model/tulip.py:531-535)."""
    img_size: Tuple[int, int] = (16, 1024)
    target_img_size: Tuple[int, int] = (64, 1024)
    patch_size: Tuple[int, int] = (1, 4)
    in_chans: int = 1
    embed_dim: int = 96
    window_size: Tuple[int, int] = (2, 8)
    depths: Tuple[int, ...] = (2, 2, 2, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    mlp_ratio: float = 4.0
    drop_path_rate: float = 0.1
    ln_eps: float = 1e-6
    pixel_shuffle: bool = True
    circular_padding: bool = True
    log_transform: bool = True
    patch_unmerging: bool = True

    @property
    def num_layers(self) -> int:
        return len(self.depths)

    @property
    def upscale_factor(self) -> int:
        # tulip.py:577
        t, i, p = self.target_img_size, self.img_size, self.patch_size
        return int(((t[0] * t[1]) / (i[0] * i[1])) ** 0.5) * 2 * int(((p[0] * p[1]) // 4) ** 0.5)

    @property
    def grid(self) -> Tuple[int, int]:
        # token grid after patch embedding (tulip.py:47)
        return self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]


def tulip_base_config(**kw) -> TulipConfig:  # tulip.py:739-746
    return TulipConfig(depths=(2, 2, 2, 2), embed_dim=96, num_heads=(3, 6, 12, 24), **kw)


def tulip_large_config(**kw) -> TulipConfig:  # tulip.py:748-755
    return TulipConfig(depths=(2,) * 5, embed_dim=96, num_heads=(3, 6, 12, 24, 48), **kw)


def tiny_config(**kw) -> TulipConfig:  # BASELINE.json configs[0]
    d = dict(img_size=(8, 256), target_img_size=(32, 256), depths=(2, 2), embed_dim=48, num_heads=(3, 6))
    d.update(kw)
    return TulipConfig(**d)


# --------------------------------------------------------------------------------------
# integer / index ops (bit-exact)
# --------------------------------------------------------------------------------------
def relative_position_index(wh: int, ww: int) -> np.ndarray:
    """(L,L) int64, value (dh + wh-1)*(2ww-1) + (dw + ww-1); tulip.py:228-240."""
    hh, wc = np.meshgrid(np.arange(wh), np.arange(ww), indexing="ij")
    h = hh.reshape(-1)
    w = wc.reshape(-1)
    dh = h[:, None] - h[None, :] + (wh - 1)
    dw = w[:, None] - w[None, :] + (ww - 1)
    return (dh * (2 * ww - 1) + dw).astype(np.int64)


def effective_window(H: int, window: Sequence[int], shift: bool) -> Tuple[Tuple[int, int], Tuple[int, int]]:
    """Window/shift actually used for a token grid of height H.

    tulip.py:284-287: when H < window_size[0] the block falls back to a (1, wh*ww) window and,
    for shifted blocks, a (0, wh*ww//2) shift.  tulip.py:219-222: regular shift = window//2.
    """
    wh, ww = int(window[0]), int(window[1])
    L = wh * ww
    if H < wh:
        return (1, L), ((0, L // 2) if shift else (0, 0))
    return (wh, ww), ((wh // 2, ww // 2) if shift else (0, 0))


def window_token_index(H: int, W: int, win: Tuple[int, int], shift: Tuple[int, int]) -> np.ndarray:
    """(nW, L) int64: flat natural token index h*W+w feeding slot l of window n.

    Composition of torch.roll by (-sh,-sw) (tulip.py:289-290) and the einops partition
    'B (Nh Mh) (Nw Mw) C -> (B Nh Nw) Mh Mw C' (tulip.py:248-252).  The inverse permutation
    (tulip.py:320-323) scatters results back to the same natural positions.
    """
    wh, ww = win
    sh, sw = shift
    nh, nw = H // wh, W // ww
    out = np.empty((nh * nw, wh * ww), dtype=np.int64)
    for wy in range(nh):
        for wx in range(nw):
            for i in range(wh):
                for j in range(ww):
                    hs, ws_ = wy * wh + i, wx * ww + j          # position in the rolled image
                    h, w = (hs + sh) % H, (ws_ + sw) % W        # rolled[hs] = x[(hs+sh) mod H]
                    out[wy * nw + wx, i * ww + j] = h * W + w
    return out


def shift_region_labels(H: int, W: int, win: Tuple[int, int], shift: Tuple[int, int]) -> np.ndarray:
    """(H,W) int64 region label map of ``create_mask`` (tulip.py:259-271), in rolled coordinates.

    Later slice assignments override earlier ones, and Python's ``-0 == 0`` makes the last slice
    ``[0:]`` cover everything when a shift component is 0 (backup window) -- reproduced literally.
    """
    wh, ww = win
    sh, sw = shift
    img = np.zeros((H, W), dtype=np.int64)
    h_slices = (slice(0, -wh), slice(-wh, -sh), slice(-sh, None))
    w_slices = (slice(0, -ww), slice(-ww, -sw), slice(-sw, None))
    cnt = 0
    for hs in h_slices:
        for ws_ in w_slices:
            img[hs, ws_] = cnt
            cnt += 1
    return img


def shift_attention_mask(H: int, W: int, win: Tuple[int, int], shift: Tuple[int, int]) -> np.ndarray:
    """(nW, L, L) float32 with 0 / -100 (tulip.py:273-280)."""
    wh, ww = win
    lab = shift_region_labels(H, W, win, shift)
    nh, nw = H // wh, W // ww
    lab = lab.reshape(nh, wh, nw, ww).transpose(0, 2, 1, 3).reshape(nh * nw, wh * ww)
    diff = lab[:, None, :] - lab[:, :, None]
    return np.where(diff != 0, np.float32(-100.0), np.float32(0.0)).astype(np.float32)


def patch_merge_gather_index(H: int, W: int) -> np.ndarray:
    """(H/2*W/2, 4) natural token index of the 4 concatenated sources, order
    [x(0::2,0::2), x(1::2,0::2), x(0::2,1::2), x(1::2,1::2)]  (tulip.py:92-99)."""
    out = np.empty(((H // 2) * (W // 2), 4), dtype=np.int64)
    for h in range(H // 2):
        for w in range(W // 2):
            r = h * (W // 2) + w
            out[r, 0] = (2 * h) * W + 2 * w
            out[r, 1] = (2 * h + 1) * W + 2 * w
            out[r, 2] = (2 * h) * W + 2 * w + 1
            out[r, 3] = (2 * h + 1) * W + 2 * w + 1
    return out


def pixel_shuffle_source_channel(c: int, i: int, j: int, r: int) -> int:
    """PixelShuffle(r): out[b,c,r*h+i,r*w+j] = in[b, c*r*r + i*r + j, h, w] (tulip.py:115,171)."""
    return c * r * r + i * r + j


def drop_path_rates(cfg: TulipConfig) -> Tuple[List[List[float]], List[List[float]]]:
    """Per-block stochastic-depth rates (tulip.py:409-410 encoder; :447,:452-453 decoder reuses
    the encoder stage slice selected by index = len(depths) - i - 2)."""
    dpr = [r.item() for r in torch.linspace(0, cfg.drop_path_rate, sum(cfg.depths))]
    enc = [dpr[sum(cfg.depths[:s]):sum(cfg.depths[:s + 1])] for s in range(cfg.num_layers)]
    dec = []
    for i in range(cfg.num_layers - 1):
        s = cfg.num_layers - i - 2
        dec.append(dpr[sum(cfg.depths[:s]):sum(cfg.depths[:s + 1])])
    return enc, dec


# --------------------------------------------------------------------------------------
# state_dict layout
# --------------------------------------------------------------------------------------
def state_dict_spec(cfg: TulipConfig) -> "Dict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered {key: (shape, kind)} in the reference's registration order.

    kind: 'linear_w' (trunc-normal .02), 'zeros', 'ones', 'conv_w'/'conv_b' (torch default conv
    init), 'bias_table' (trunc-normal .02), 'index' (int64 buffer).  Matches the 226-entry
    state_dict of tulip_base; registration order follows
    TULIP.__init__ (tulip.py:555-582): layers, layers_up, first_patch_expanding,
    skip_connection_layers, norm_up, patch_embed, decoder_pred, ps_head (or
    final_patch_expanding when pixel_shuffle=False; PatchExpanding instead of PatchUnmerging
    when patch_unmerging=False).
    """
    E, wh, ww = cfg.embed_dim, cfg.window_size[0], cfg.window_size[1]
    L = wh * ww
    nl = cfg.num_layers
    spec: Dict[str, Tuple[Tuple[int, ...], str]] = {}

    def block(prefix: str, C: int, nh: int):
        spec[f"{prefix}.norm1.weight"] = ((C,), "ones")
        spec[f"{prefix}.norm1.bias"] = ((C,), "zeros")
        spec[f"{prefix}.attn.relative_position_bias_table"] = (((2 * wh - 1) * (2 * ww - 1), nh), "bias_table")
        spec[f"{prefix}.attn.relative_position_index"] = ((L, L), "index")
        spec[f"{prefix}.attn.qkv.weight"] = ((3 * C, C), "linear_w")
        spec[f"{prefix}.attn.qkv.bias"] = ((3 * C,), "zeros")
        spec[f"{prefix}.attn.proj.weight"] = ((C, C), "linear_w")
        spec[f"{prefix}.attn.proj.bias"] = ((C,), "zeros")
        spec[f"{prefix}.norm2.weight"] = ((C,), "ones")
        spec[f"{prefix}.norm2.bias"] = ((C,), "zeros")
        Hd = int(C * cfg.mlp_ratio)
        spec[f"{prefix}.mlp.fc1.weight"] = ((Hd, C), "linear_w")
        spec[f"{prefix}.mlp.fc1.bias"] = ((Hd,), "zeros")
        spec[f"{prefix}.mlp.fc2.weight"] = ((C, Hd), "linear_w")
        spec[f"{prefix}.mlp.fc2.bias"] = ((C,), "zeros")

    def upsample(prefix: str, C: int):
        if cfg.patch_unmerging:       # PatchUnmerging, tulip.py:109-115
            spec[f"{prefix}.expand.weight"] = ((2 * C, C, 1, 1), "conv_w")
            spec[f"{prefix}.expand.bias"] = ((2 * C,), "conv_b")
        else:                         # PatchExpanding, tulip.py:126-132
            spec[f"{prefix}.expand.weight"] = ((2 * C, C), "linear_w")
            spec[f"{prefix}.norm.weight"] = ((C // 2,), "ones")
            spec[f"{prefix}.norm.bias"] = ((C // 2,), "zeros")

    for s in range(nl):
        C = E * 2 ** s
        for b in range(cfg.depths[s]):
            block(f"layers.{s}.blocks.{b}", C, cfg.num_heads[s])
        if s < nl - 1:
            spec[f"layers.{s}.downsample.norm.weight"] = ((4 * C,), "ones")
            spec[f"layers.{s}.downsample.norm.bias"] = ((4 * C,), "zeros")
            spec[f"layers.{s}.downsample.reduction.weight"] = ((2 * C, 4 * C), "linear_w")
    for i in range(nl - 1):
        s = nl - i - 2
        C = E * 2 ** s
        for b in range(cfg.depths[s]):
            block(f"layers_up.{i}.blocks.{b}", C, cfg.num_heads[s])
        if i < nl - 2:
            upsample(f"layers_up.{i}.upsample", C)
    upsample("first_patch_expanding", E * 2 ** (nl - 1))
    for i in range(nl - 1):
        C = E * 2 ** (nl - 2 - i)
        spec[f"skip_connection_layers.{i}.weight"] = ((C, 2 * C), "linear_w")
        spec[f"skip_connection_layers.{i}.bias"] = ((C,), "zeros")
    spec["norm_up.weight"] = ((E,), "ones")
    spec["norm_up.bias"] = ((E,), "zeros")
    kw = 8 if cfg.circular_padding else cfg.patch_size[1]
    spec["patch_embed.proj.weight"] = ((E, cfg.in_chans, cfg.patch_size[0], kw), "conv_w")
    spec["patch_embed.proj.bias"] = ((E,), "conv_b")
    spec["patch_embed.norm.weight"] = ((E,), "ones")
    spec["patch_embed.norm.bias"] = ((E,), "zeros")
    spec["decoder_pred.weight"] = ((cfg.in_chans, E, 1, 1), "conv_w")
    r2 = cfg.upscale_factor ** 2
    if cfg.pixel_shuffle:             # PixelShuffleHead, tulip.py:161-171
        spec["ps_head.conv_expand.0.weight"] = ((E * r2, E, 1, 1), "conv_w")
        spec["ps_head.conv_expand.0.bias"] = ((E * r2,), "conv_b")
    else:                             # FinalPatchExpanding, tulip.py:144-150
        spec["final_patch_expanding.expand.weight"] = ((E * r2, E), "linear_w")
        spec["final_patch_expanding.norm.weight"] = ((E,), "ones")
        spec["final_patch_expanding.norm.bias"] = ((E,), "zeros")
    return spec


def _key_seed(key: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in f"{seed}:{key}".encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def key_seeded_state_dict(cfg: TulipConfig, seed: int = 0, randomize_affine: bool = True) -> Dict[str, Tensor]:
    """Deterministic weights whose values depend only on (seed, key) -- independent of module
    construction order, so large models need no stored blob.

    With ``randomize_affine`` the biases / LayerNorm affine are perturbed too (the reference init
    sets them to 0/1, which would hide bias- and beta-path bugs in parity tests).
    """
    sd: Dict[str, Tensor] = {}
    wh, ww = cfg.window_size
    for key, (shape, kind) in state_dict_spec(cfg).items():
        if kind == "index":
            sd[key] = torch.from_numpy(relative_position_index(wh, ww))
            continue
        g = torch.Generator().manual_seed(_key_seed(key, seed))
        if kind in ("linear_w", "bias_table"):
            t = torch.randn(shape, generator=g).clamp_(-2, 2) * 0.02
        elif kind == "conv_w":
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "conv_b":
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        elif kind == "ones":
            t = torch.ones(shape)
            if randomize_affine:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif kind == "zeros":
            t = torch.zeros(shape)
            if randomize_affine:
                t = 0.02 * torch.randn(shape, generator=g)
        else:
            raise KeyError(kind)
        sd[key] = t.float()
    return sd


def synthetic_batch(cfg: TulipConfig, batch: int, seed: int = 1234) -> Tuple[Tensor, Tensor]:
    """Synthetic (lo, hi) range-image pair of SURVEY.md 8(d): r~U[0,1), 10% zeros, log1p,
    row-subsampled low-res input (datasets.py:68-70,117-125,285-294)."""
    g = torch.Generator().manual_seed(seed)
    Hh, Wh = cfg.target_img_size
    r = torch.rand(batch, cfg.in_chans, Hh, Wh, generator=g)
    r[torch.rand(batch, cfg.in_chans, Hh, Wh, generator=g) < 0.1] = 0
    hi = torch.log1p(r)
    lo = hi[:, :, 0::Hh // cfg.img_size[0], :].contiguous()
    return lo, hi


# --------------------------------------------------------------------------------------
# float ops
# --------------------------------------------------------------------------------------
class _Prec:
    """Rounding model.  ``lowp=False``: pure fp32 (the reference's fp32 forward).
    ``lowp=True``: round to bf16 exactly where the HIP path does -- every GEMM operand
    (activations and weights) and every bf16-stored intermediate -- while the residual stream,
    LayerNorm statistics, softmax and loss stay fp32 (same contract as CUDA autocast,
    SURVEY.md Appendix A, except that the residual stream is fp32 in every stage)."""

    def __init__(self, lowp: bool, attn_fp8: bool = False):
        self.lowp = lowp
        self.attn_fp8 = attn_fp8          # BASELINE configs[4]: attention scores from e4m3 q, k (engine.attn_fp8)

    def r(self, t: Tensor) -> Tensor:
        if not self.lowp:
            return t
        return _BF16Round.apply(t)


class _BF16Round(torch.autograd.Function):
    """bf16 round-to-nearest-even with a straight-through gradient that is itself rounded
    (the HIP backward stores the corresponding gradients in bf16)."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _FP8Round(torch.autograd.Function):
    """OCP e4m3 round-to-nearest-even (torch.float8_e4m3fn = the gfx950 fp8 format) with a straight-through
    gradient: the HIP backward multiplies dS with the ROUNDED q, k, i.e. differentiates the function the forward ran."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.float8_e4m3fn).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def linear(pr: _Prec, x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = pr.r(x) @ pr.r(w).t()
    return y if b is None else y + b


def patch_embed(pr: _Prec, sd, cfg: TulipConfig, x: Tensor) -> Tensor:
    """tulip.py:63-73.  Circular pad W by (2,2) then Conv2d(in_c->E,(p0,8),stride p) else a
    non-overlapping patch conv; 'B C H W -> B H W C'; LayerNorm(E)."""
    ph, pw = cfg.patch_size
    _, _, H, W = x.shape
    if H % ph != 0 or W % pw != 0:  # tulip.py:50-56 (argument order reproduced literally)
        x = F.pad(x, (0, ph - W % pw, 0, pw - H % ph, 0, 0))
    if cfg.circular_padding:
        x = F.pad(x, (2, 2, 0, 0), mode="circular")
    y = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=(ph, pw))
    y = y.permute(0, 2, 3, 1)
    return layer_norm(y, sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], cfg.ln_eps)


def window_attention(pr: _Prec, sd, prefix: str, cfg: TulipConfig, x: Tensor, nh: int, shift: bool) -> Tensor:
    """tulip.py:282-324 on the LayerNorm'ed tokens x:(B,H,W,C) -> (B,H,W,C)."""
    B, H, W, C = x.shape
    win, sft = effective_window(H, cfg.window_size, shift)
    L = win[0] * win[1]
    P = C // nh
    idx = torch.from_numpy(window_token_index(H, W, win, sft)).to(x.device)   # (nW, L)
    nW = idx.shape[0]
    tok = x.reshape(B, H * W, C)[:, idx.reshape(-1), :].reshape(B * nW, L, C)
    qkv = pr.r(linear(pr, tok, sd[f"{prefix}.attn.qkv.weight"], sd[f"{prefix}.attn.qkv.bias"]))
    qkv = qkv.reshape(B * nW, L, 3, nh, P).permute(2, 0, 3, 1, 4)        # (T, Bn, Nh, L, P)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if pr.attn_fp8:
        q, k = _FP8Round.apply(q), _FP8Round.apply(k)
    attn = (q @ k.transpose(-2, -1)) * (P ** -0.5)
    rpi = sd[f"{prefix}.attn.relative_position_index"].reshape(-1)
    bias = sd[f"{prefix}.attn.relative_position_bias_table"][rpi].reshape(L, L, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift:
        mask = torch.from_numpy(shift_attention_mask(H, W, win, sft)).to(x.device)  # (nW, L, L)
        attn = attn.reshape(B, nW, nh, L, L) + mask[None, :, None]
        attn = attn.reshape(B * nW, nh, L, L)
    attn = torch.softmax(attn, dim=-1)
    o = pr.r(attn) @ v                                                   # (Bn, Nh, L, P)
    o = pr.r(o.permute(0, 2, 1, 3).reshape(B * nW, L, C))
    o = linear(pr, o, sd[f"{prefix}.attn.proj.weight"], sd[f"{prefix}.attn.proj.bias"])
    out = torch.zeros(B, H * W, C, dtype=o.dtype, device=o.device)
    out[:, idx.reshape(-1), :] = o.reshape(B, nW * L, C)                 # inverse permutation
    return out.reshape(B, H, W, C)


def mlp(pr: _Prec, sd, prefix: str, x: Tensor) -> Tensor:
    """tulip.py:194-200; exact-erf GELU."""
    h = pr.r(linear(pr, x, sd[f"{prefix}.mlp.fc1.weight"], sd[f"{prefix}.mlp.fc1.bias"]))
    h = pr.r(F.gelu(h))
    return linear(pr, h, sd[f"{prefix}.mlp.fc2.weight"], sd[f"{prefix}.mlp.fc2.bias"])


def swin_block(pr: _Prec, sd, prefix: str, cfg: TulipConfig, x: Tensor, nh: int, shift: bool,
               keep: Optional[Tensor]) -> Tensor:
    """tulip.py:338-352.  ``keep``: per-sample DropPath multiplier floor(keep_prob+u)/keep_prob
    (tulip.py:21-30), or None for identity (eval / rate 0)."""
    y = window_attention(pr, sd, prefix, cfg, pr.r(layer_norm(x, sd[f"{prefix}.norm1.weight"],
                                                              sd[f"{prefix}.norm1.bias"], cfg.ln_eps)), nh, shift)
    if keep is not None:
        y = y * keep[0].view(-1, 1, 1, 1)
    x = x + y
    y = mlp(pr, sd, prefix, pr.r(layer_norm(x, sd[f"{prefix}.norm2.weight"], sd[f"{prefix}.norm2.bias"], cfg.ln_eps)))
    if keep is not None:
        y = y * keep[1].view(-1, 1, 1, 1)
    return x + y


def patch_merging(pr: _Prec, sd, prefix: str, cfg: TulipConfig, x: Tensor) -> Tensor:
    """tulip.py:101-106 (even H,W only: the reference's odd-size padding never triggers for the
    supported grids)."""
    B, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = pr.r(layer_norm(x, sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"], cfg.ln_eps))
    return linear(pr, x, sd[f"{prefix}.reduction.weight"], None)


def patch_unmerging(pr: _Prec, sd, prefix: str, x: Tensor) -> Tensor:
    """tulip.py:117-123: 1x1 conv C->2C (+bias), PixelShuffle(2), back to BHWC."""
    B, H, W, C = x.shape
    w = sd[f"{prefix}.expand.weight"].reshape(2 * C, C)
    z = linear(pr, x, w, sd[f"{prefix}.expand.bias"])                    # (B,H,W,2C)
    z = z.reshape(B, H, W, C // 2, 2, 2).permute(0, 1, 4, 2, 5, 3)       # (B,H,i,W,j,c)
    return z.reshape(B, 2 * H, 2 * W, C // 2)


def expand_rearrange(z: Tensor, P: int) -> Tensor:
    """'B H W (P1 P2 C) -> B (H P1) (W P2) C' with P1 = P2 = P (tulip.py:136, :154-156)."""
    B, H, W, PC = z.shape
    C = PC // (P * P)
    return z.reshape(B, H, W, P, P, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H * P, W * P, C)


def patch_expanding(pr: _Prec, sd, prefix: str, cfg: TulipConfig, x: Tensor) -> Tensor:
    """tulip.py:126-140 (patch_unmerging=False): Linear C->2C (no bias), 2x2 rearrange, LayerNorm(C/2)."""
    z = expand_rearrange(linear(pr, x, sd[f"{prefix}.expand.weight"], None), 2)
    return layer_norm(z, sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"], cfg.ln_eps)


def final_expanding_and_pred(pr: _Prec, sd, cfg: TulipConfig, x: Tensor) -> Tensor:
    """tulip.py:727-731 with pixel_shuffle=False: FinalPatchExpanding (Linear E->r^2 E no bias, r x r rearrange,
    LayerNorm(E); tulip.py:144-159), 'B H W C -> B C H W', conv1x1 E->in_chans (no bias)."""
    z = expand_rearrange(linear(pr, x, sd["final_patch_expanding.expand.weight"], None), cfg.upscale_factor)
    z = pr.r(layer_norm(z, sd["final_patch_expanding.norm.weight"], sd["final_patch_expanding.norm.bias"], cfg.ln_eps))
    wd = sd["decoder_pred.weight"].reshape(cfg.in_chans, cfg.embed_dim)
    return torch.einsum("bhwc,oc->bohw", z, wd)


def ps_head_and_pred(pr: _Prec, sd, cfg: TulipConfig, x: Tensor) -> Tensor:
    """tulip.py:720-731: conv1x1 E->E*r^2 (+bias), LeakyReLU(0.01), PixelShuffle(r),
    conv1x1 E->in_chans (no bias).  x:(B,H,W,E) is norm_up's output.  -> (B,in_chans,rH,rW)."""
    B, H, W, E = x.shape
    r = cfg.upscale_factor
    w = sd["ps_head.conv_expand.0.weight"].reshape(E * r * r, E)
    z = F.leaky_relu(linear(pr, x, w, sd["ps_head.conv_expand.0.bias"]), 0.01)   # (B,H,W,E*r*r)
    z = z.reshape(B, H, W, E, r, r)                                               # oc = c*r*r + i*r + j
    wd = sd["decoder_pred.weight"].reshape(cfg.in_chans, E)
    pred = torch.einsum("bhwcij,oc->bohiwj", z, wd)
    return pred.reshape(B, cfg.in_chans, r * H, r * W)


def forward_loss(cfg: TulipConfig, pred: Tensor, target: Tensor) -> Tuple[Tensor, Tensor]:
    """tulip.py:690-700."""
    loss = (pred - target).abs().mean()
    if cfg.log_transform:
        pixel = (torch.expm1(pred) - torch.expm1(target)).abs().mean()
    else:
        pixel = loss.clone()
    return loss, pixel


def drop_path_keep(rate: float, u: Tensor) -> Optional[Tensor]:
    """DropPath multiplier from uniform draws u:(B,) (tulip.py:25-29)."""
    if rate == 0.0:
        return None
    kp = 1.0 - rate
    return torch.floor(kp + u) / kp


def tulip_forward(sd: Dict[str, Tensor], cfg: TulipConfig, x: Tensor, target: Optional[Tensor],
                  lowp: bool = False, drop_u: Optional[Dict[str, Tensor]] = None,
                  taps: Optional[Dict[str, Tensor]] = None, attn_fp8: bool = False):
    """TULIP.forward (tulip.py:702-737), every flag combination of pixel_shuffle / patch_unmerging /
    circular_padding.

    ``drop_u``: {block prefix: (2,B) uniform draws} enables train-mode DropPath with explicit
    randomness (None = eval / identity).  ``taps`` collects per-stage activations.
    Returns (pred, loss, pixel_loss), or pred alone when target is None (mc_drop path).
    """
    pr = _Prec(lowp, attn_fp8)
    up = (lambda prefix, t: patch_unmerging(pr, sd, prefix, t)) if cfg.patch_unmerging else \
        (lambda prefix, t: patch_expanding(pr, sd, prefix, cfg, t))
    enc_rates, dec_rates = drop_path_rates(cfg)
    nl = cfg.num_layers

    def keep_for(prefix: str, rate: float):
        if drop_u is None or rate == 0.0:
            return None
        u = drop_u[prefix]
        return torch.stack([drop_path_keep(rate, u[0]), drop_path_keep(rate, u[1])])

    x = patch_embed(pr, sd, cfg, x)
    if taps is not None:
        taps["patch_embed"] = x
    saved = []
    for s in range(nl):
        saved.append(x)
        for b in range(cfg.depths[s]):
            p = f"layers.{s}.blocks.{b}"
            x = swin_block(pr, sd, p, cfg, x, cfg.num_heads[s], b % 2 == 1, keep_for(p, enc_rates[s][b]))
        if s < nl - 1:
            x = patch_merging(pr, sd, f"layers.{s}.downsample", cfg, x)
        if taps is not None:
            taps[f"layers.{s}"] = x
    x = up("first_patch_expanding", x)
    for i in range(nl - 1):
        s = nl - i - 2
        cat = torch.cat([x, saved[len(saved) - i - 2]], -1)
        x = linear(pr, cat, sd[f"skip_connection_layers.{i}.weight"], sd[f"skip_connection_layers.{i}.bias"])
        for b in range(cfg.depths[s]):
            p = f"layers_up.{i}.blocks.{b}"
            x = swin_block(pr, sd, p, cfg, x, cfg.num_heads[s], b % 2 == 1, keep_for(p, dec_rates[i][b]))
        if i < nl - 2:
            x = up(f"layers_up.{i}.upsample", x)
        if taps is not None:
            taps[f"layers_up.{i}"] = x
    x = pr.r(layer_norm(x, sd["norm_up.weight"], sd["norm_up.bias"], cfg.ln_eps))
    pred = ps_head_and_pred(pr, sd, cfg, x) if cfg.pixel_shuffle else final_expanding_and_pred(pr, sd, cfg, x)
    if target is None:
        return pred
    loss, pixel = forward_loss(cfg, pred, target)
    return pred, loss, pixel


def tulip_loss_and_grads(sd: Dict[str, Tensor], cfg: TulipConfig, x: Tensor, target: Tensor,
                         lowp: bool = False, drop_u=None, attn_fp8: bool = False):
    """Autograd of the oracle: (pred, loss, pixel_loss, {key: grad})."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    full = dict(sd)
    full.update(leaves)
    pred, loss, pixel = tulip_forward(full, cfg, x, target, lowp=lowp, drop_u=drop_u, attn_fp8=attn_fp8)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return pred.detach(), loss.detach(), pixel.detach(), grads


# --------------------------------------------------------------------------------------
# training-step semantics (engine_upsampling.py:66-100, lr_sched.py:9-21, main:282-283)
# --------------------------------------------------------------------------------------
def cosine_lr(step_epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """lr_sched.py:9-21 (per-iteration fractional epoch)."""
    if step_epoch < warmup_epochs:
        return lr * step_epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (step_epoch - warmup_epochs) / (epochs - warmup_epochs)))


def adamw_reference_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                         beta1: float = 0.9, beta2: float = 0.95, eps: float = 1e-8, wd: float = 0.01):
    """torch.optim.AdamW semantics (decoupled decay), used with betas (0.9,0.95) at main:283;
    weight decay applies to ndim>1 parameters only (timm param_groups, main:282)."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    mhat = m / (1 - beta1 ** step)
    vhat = v / (1 - beta2 ** step)
    p = p - lr * mhat / (vhat.sqrt() + eps)
    return p, m, v


def train_loop_reference(sd: Dict[str, Tensor], cfg: TulipConfig, batches, epochs_run: int, lr: float,
                         min_lr: float, warmup_epochs: float, epochs: float, accum_iter: int = 1,
                         betas=(0.9, 0.95), eps: float = 1e-8, wd: float = 0.01, lowp: bool = False):
    """engine_upsampling.py:46-124 on the oracle: per-window LR from `it/len + epoch` (:68-69), gradients of
    loss/accum_iter accumulated (:90), L2 norm of all gradients before the step (misc.py:303,317-329), AdamW with
    decay on ndim>1 parameters only (main:282-283), gradients cleared after the step (:96-97) and at the top of
    each epoch (:61).  Returns (loss per micro-step, lr per micro-step, grad norm per optimizer step)."""
    params = {k: v.clone() for k, v in sd.items() if v.is_floating_point()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    losses, lrs, norms, t, cur = [], [], [], 0, lr
    for epoch in range(epochs_run):
        acc = {k: torch.zeros_like(v) for k, v in params.items()}
        for it, (lo, hi) in enumerate(batches):
            if it % accum_iter == 0:
                cur = cosine_lr(it / len(batches) + epoch, lr, min_lr, warmup_epochs, epochs)
            full = dict(sd)
            full.update(params)
            _, loss, _, g = tulip_loss_and_grads(full, cfg, lo, hi, lowp=lowp)
            losses.append(loss.item())
            for k in acc:
                acc[k] += g[k] / accum_iter
            if (it + 1) % accum_iter == 0:
                norms.append(math.sqrt(sum(float((a.double() ** 2).sum()) for a in acc.values())))
                t += 1
                for k in params:
                    params[k], m[k], v2[k] = adamw_reference_step(params[k], acc[k], m[k], v2[k], t, cur, betas[0],
                                                                  betas[1], eps, wd if params[k].ndim > 1 else 0.0)
                    acc[k].zero_()
            lrs.append(cur)
    return losses, lrs, norms
