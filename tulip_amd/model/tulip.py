"""Drop-in replacement for the reference's ``model.tulip`` module (tulip/model/tulip.py).

Same public surface -- ``tulip_base`` / ``tulip_large`` / ``TULIP`` with identical constructor
arguments (tulip.py:531-535), ``forward(x, target, eval=False, mc_drop=False)`` (tulip.py:702) with
identical return values, and an identical ``state_dict`` (226 entries for tulip_base, same keys,
shapes, dtypes, registration order, and the same seeded initialisation because submodules are
created and initialised in the reference's order) -- but the compute is not PyTorch: every op of
the forward and backward pass is a hand-written gfx950 kernel from libtulip_hip.so, driven by
``tulip_amd.engine.TulipEngine``.  The sub-modules below only *hold parameters*; they have no
forward of their own.  There is no CPU / eager fallback: calling the model on a CPU tensor raises.
"""
from __future__ import annotations

import collections.abc
from functools import partial
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

__all__ = ["TULIP", "tulip_base", "tulip_large"]


class _ParamHolder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("tulip_amd sub-modules only hold parameters; call the TULIP module (HIP engine)")


class DropPath(_ParamHolder):
    """Stochastic depth marker (tulip.py:16-30); the per-sample multiplier is applied inside the
    residual GEMM epilogue by the engine."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob


class PatchEmbedding(_ParamHolder):  # tulip.py:33-48
    def __init__(self, img_size, patch_size, in_c, embed_dim, norm_layer, circular_padding):
        super().__init__()
        self.img_size, self.patch_size, self.circular_padding = tuple(img_size), tuple(patch_size), circular_padding
        kw = 8 if circular_padding else patch_size[1]
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=(patch_size[0], kw), stride=tuple(patch_size))
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]


class PatchMerging(_ParamHolder):  # tulip.py:76-81
    def __init__(self, dim, norm_layer):
        super().__init__()
        self.dim = dim
        self.norm = norm_layer(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)


class PatchUnmerging(_ParamHolder):  # tulip.py:109-115
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.expand = nn.Conv2d(dim, dim * 2, kernel_size=(1, 1))
        self.upsample = nn.PixelShuffle(2)


class PatchExpanding(_ParamHolder):  # tulip.py:126-132 (patch_unmerging=False)
    def __init__(self, dim, norm_layer):
        super().__init__()
        self.dim = dim
        self.expand = nn.Linear(dim, 2 * dim, bias=False)
        self.norm = norm_layer(dim // 2)


class FinalPatchExpanding(_ParamHolder):  # tulip.py:144-150 (pixel_shuffle=False)
    def __init__(self, dim, norm_layer, upscale_factor):
        super().__init__()
        self.dim, self.upscale_factor = dim, upscale_factor
        self.expand = nn.Linear(dim, (upscale_factor ** 2) * dim, bias=False)
        self.norm = norm_layer(dim)


class PixelShuffleHead(_ParamHolder):  # tulip.py:161-171
    def __init__(self, dim, upscale_factor):
        super().__init__()
        self.dim = dim
        self.conv_expand = nn.Sequential(nn.Conv2d(dim, dim * upscale_factor ** 2, kernel_size=(1, 1)),
                                         nn.LeakyReLU(inplace=True))
        self.upsample = nn.PixelShuffle(upscale_factor)


class Mlp(_ParamHolder):  # tulip.py:181-192
    def __init__(self, in_features, hidden_features, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop2 = nn.Dropout(drop)


def relative_position_index(wh: int, ww: int) -> torch.Tensor:
    """(L,L) int64: (dh + wh-1)*(2ww-1) + (dw + ww-1)   (tulip.py:228-239)."""
    hh, wc = np.meshgrid(np.arange(wh), np.arange(ww), indexing="ij")
    h, w = hh.reshape(-1), wc.reshape(-1)
    idx = (h[:, None] - h[None, :] + wh - 1) * (2 * ww - 1) + (w[:, None] - w[None, :] + ww - 1)
    return torch.from_numpy(idx.astype(np.int64))


class WindowAttention(_ParamHolder):  # tulip.py:203-246
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0, shift=False):
        super().__init__()
        ws = window_size if isinstance(window_size, collections.abc.Iterable) else (window_size, window_size)
        self.window_size = (int(ws[0]), int(ws[1]))
        self.num_heads, self.shift = num_heads, shift
        self.scale = (dim // num_heads) ** -0.5
        self.num_windows = self.window_size[0] * self.window_size[1]
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * self.window_size[0] - 1) * (2 * self.window_size[1] - 1), num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", relative_position_index(*self.window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)


class SwinTransformerBlock(_ParamHolder):  # tulip.py:326-336
    def __init__(self, dim, num_heads, window_size, shift, mlp_ratio, qkv_bias, drop, attn_drop, drop_path, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, qkv_bias, attn_drop, drop, shift)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)
        self.drop_path_rate = float(drop_path)
        self.shift = shift


def _stage_drop_rates(depths: Sequence[int], drop_path: float, index: int):
    dpr = [r.item() for r in torch.linspace(0, drop_path, sum(depths))]  # tulip.py:409-410
    return dpr[sum(depths[:index]):sum(depths[:index + 1])]


class BasicBlock(_ParamHolder):  # tulip.py:399-429
    def __init__(self, index, embed_dim, window_size, depths, num_heads, mlp_ratio, qkv_bias, drop_rate,
                 attn_drop_rate, drop_path, norm_layer, patch_merging):
        super().__init__()
        dim = embed_dim * 2 ** index
        rates = _stage_drop_rates(depths, drop_path, index)
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads[index], window_size, i % 2 == 1, mlp_ratio, qkv_bias, drop_rate,
                                 attn_drop_rate, rates[i], norm_layer) for i in range(depths[index])])
        self.downsample = PatchMerging(dim, norm_layer) if patch_merging else None


class BasicBlockUp(_ParamHolder):  # tulip.py:441-475
    def __init__(self, index, embed_dim, window_size, depths, num_heads, mlp_ratio, qkv_bias, drop_rate,
                 attn_drop_rate, drop_path, patch_expanding, norm_layer, patch_unmerging):
        super().__init__()
        index = len(depths) - index - 2
        dim = embed_dim * 2 ** index
        rates = _stage_drop_rates(depths, drop_path, index)
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads[index], window_size, i % 2 == 1, mlp_ratio, qkv_bias, drop_rate,
                                 attn_drop_rate, rates[i], norm_layer) for i in range(depths[index])])
        if patch_expanding:
            self.upsample = PatchUnmerging(dim) if patch_unmerging else PatchExpanding(dim, norm_layer)
        else:
            self.upsample = nn.Identity()


class TULIP(nn.Module):
    """tulip.py:530-737.  Constructor signature and defaults are the reference's."""

    def __init__(self, img_size=(32, 2048), target_img_size=(128, 2048), patch_size=(4, 4), in_chans: int = 1,
                 embed_dim: int = 96, window_size=4, depths: tuple = (2, 2, 6, 2), num_heads: tuple = (3, 6, 12, 24),
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.1, norm_layer=nn.LayerNorm, patch_norm: bool = True,
                 pixel_shuffle: bool = False, circular_padding: bool = False, swin_v2: bool = False,
                 log_transform: bool = False, patch_unmerging: bool = False):
        super().__init__()
        self.window_size = window_size
        self.depths, self.num_heads = tuple(depths), tuple(num_heads)
        self.num_layers = len(depths)
        self.embed_dim, self.mlp_ratio, self.qkv_bias = embed_dim, mlp_ratio, qkv_bias
        self.drop_rate, self.attn_drop_rate, self.drop_path = drop_rate, attn_drop_rate, drop_path_rate
        self.norm_layer = norm_layer
        self.img_size, self.target_img_size = tuple(img_size), tuple(target_img_size)
        self.patch_size, self.in_chans = tuple(patch_size), in_chans
        self.log_transform, self.patch_unmerging = log_transform, patch_unmerging
        self.pixel_shuffle, self.circular_padding = pixel_shuffle, circular_padding
        self.pos_drop = nn.Dropout(p=drop_rate)
        if swin_v2:
            # the reference's V2 branch reads self.patch_embed before it exists (tulip.py:602 vs :571)
            raise AttributeError("'TULIP' object has no attribute 'patch_embed'")
        if drop_rate != 0.0 or attn_drop_rate != 0.0:
            raise NotImplementedError("tulip_amd: element dropout (drop_rate/attn_drop_rate) is 0 in every reference "
                                      "configuration (tulip.py:741-743) and is not implemented in the HIP path")
        if not patch_norm:
            raise NotImplementedError("tulip_amd: patch_norm=False is not implemented (the patch-embedding kernel fuses the "
                                      "LayerNorm; every reference configuration keeps the default patch_norm=True)")
        if not isinstance(window_size, collections.abc.Iterable):
            window_size = (window_size, window_size)
        common = dict(embed_dim=embed_dim, window_size=window_size, depths=self.depths, num_heads=self.num_heads,
                      mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                      drop_path=drop_path_rate, norm_layer=norm_layer)
        # registration order below == the reference's (tulip.py:555-582) so that the same torch seed
        # yields the same initial weights and state_dict ordering
        self.layers = nn.ModuleList([BasicBlock(index=i, patch_merging=i < self.num_layers - 1, **common)
                                     for i in range(self.num_layers)])
        self.layers_up = nn.ModuleList([BasicBlockUp(index=i, patch_expanding=i < self.num_layers - 2,
                                                     patch_unmerging=patch_unmerging, **common)
                                        for i in range(self.num_layers - 1)])
        top = embed_dim * 2 ** (self.num_layers - 1)
        self.first_patch_expanding = PatchUnmerging(dim=top) if patch_unmerging else PatchExpanding(top, norm_layer)
        self.skip_connection_layers = nn.ModuleList([
            nn.Linear(2 * embed_dim * 2 ** (self.num_layers - 2 - i), embed_dim * 2 ** (self.num_layers - 2 - i))
            for i in range(self.num_layers - 1)])
        self.norm_up = norm_layer(embed_dim)
        self.patch_embed = PatchEmbedding(img_size, patch_size, in_chans, embed_dim,
                                          norm_layer if patch_norm else None, circular_padding)
        self.decoder_pred = nn.Conv2d(embed_dim, in_chans, kernel_size=(1, 1), bias=False)
        self.upscale_factor = int(((target_img_size[0] * target_img_size[1]) / (img_size[0] * img_size[1])) ** 0.5) \
            * 2 * int(((patch_size[0] * patch_size[1]) // 4) ** 0.5)           # tulip.py:577
        if pixel_shuffle:
            self.ps_head = PixelShuffleHead(embed_dim, self.upscale_factor)
        else:
            self.final_patch_expanding = FinalPatchExpanding(embed_dim, norm_layer, self.upscale_factor)
        self.apply(self.init_weights)
        self._engine = None
        self._ln_eps = float(self.norm_up.eps)
        # load_state_dict / misc.load_model copy into the fp32 views of the flat buffer: the bf16 GEMM operands must follow
        self.register_load_state_dict_post_hook(lambda module, _keys: module._mark_params_written())

    @staticmethod
    def init_weights(m):  # tulip.py:586-594
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ engine plumbing
    def engine(self):
        if self._engine is None:
            from ..engine import TulipEngine
            self._engine = TulipEngine(self)
        return self._engine

    def _mark_params_written(self):
        eng = getattr(self, "_engine", None)
        if eng is not None and eng.params is not None:
            eng.params.shadow_dirty = True

    def _apply(self, fn, *a, **k):
        # .to()/.cuda()/.float() re-create parameter storage: the engine re-flattens lazily
        out = super()._apply(fn, *a, **k)
        if getattr(self, "_engine", None) is not None:
            self._engine.invalidate()
        return out

    def forward(self, x, target, eval=False, mc_drop=False):
        """(pred, total_loss, pixel_loss), or pred alone when mc_drop (tulip.py:702-737)."""
        if not x.is_cuda:
            raise RuntimeError("tulip_amd.TULIP runs only on an AMD GPU (gfx950): the hot path is hand-written HIP "
                               "and has no CPU fallback.  Use oracle/tulip_oracle.py for CPU checks.")
        return self.engine().autograd_forward(x, target, mc_drop)


def tulip_base(**kwargs):  # tulip.py:739-746
    return TULIP(depths=(2, 2, 2, 2), embed_dim=96, num_heads=(3, 6, 12, 24), qkv_bias=True, mlp_ratio=4,
                 drop_path_rate=0.1, drop_rate=0, attn_drop_rate=0, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 **kwargs)


def tulip_large(**kwargs):  # tulip.py:748-755
    return TULIP(depths=(2, 2, 2, 2, 2), embed_dim=96, num_heads=(3, 6, 12, 24, 48), qkv_bias=True, mlp_ratio=4,
                 drop_path_rate=0.1, drop_rate=0, attn_drop_rate=0, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 **kwargs)
