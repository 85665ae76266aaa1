"""Tensor-level wrappers over the C ABI (tulip_amd/_lib.py).  Outputs are caller-allocated; every
call is asynchronous on torch's current HIP stream (so it is captured by torch.cuda.graph)."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BF16, EPI_F32, EPI_GELU_BWD, EPI_GELU_DUAL,  # noqa: F401
                   EPI_PIXSHUF2_F32, EPI_RESID_F32, EPI_SPLIT_F32, EPI_UNSHUF2_BF16, check)

BF16 = torch.bfloat16
F32 = torch.float32


def _p(t):
    """tensor -> device address; ints (precomputed addresses) and None pass through."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected contiguous {dtype} device tensor, got {t.dtype} "
                        f"{'cuda' if t.is_cuda else 'cpu'} contiguous={t.is_contiguous()}")


def gemm(A, B, M, N, K, *, lda, ldb, a_trans=False, b_trans=False, epi=EPI_BF16, bias=None, out=None, ldo=None,
         out2=None, ldo2=0, aux=None, ldaux=0, rowscale=None, rows_per_sample=1, accumulate=False, psH=0, psW=0,
         splits=1, workspace=None, workspace_bytes=0, touch=True, checked=False, mid=None, b_packed=False):
    """C[M,N] = opA . opB^T with a fused epilogue (see include/tulip_hip.h).  A/B/out may be views
    with an element offset (row strides passed as lda/ldb/ldo).  touch=False / checked=True: TULIP_GEMM_NO_TOUCH /
    TULIP_GEMM_CHECKED (measurement and bit-compare switches, per call)."""
    lib = _lib.load()
    accumulate = int(bool(accumulate)) | (0 if touch else _lib.GEMM_NO_TOUCH) | (_lib.GEMM_CHECKED if checked else 0)
    accumulate |= 0 if mid is None else (_lib.GEMM_MID if mid else _lib.GEMM_NO_MID)     # the 192 x 192 mid-size kernel: forced / never
    accumulate |= _lib.GEMM_B_PACKED if b_packed else 0      # B = the fragment-major copy of the [N][K] matrix: the small-K form
    rc = lib.tulip_gemm_bf16(_p(A), lda, int(a_trans), _p(B), ldb, int(b_trans), M, N, K, epi, _p(bias), _p(out),
                             ldo if ldo is not None else N, _p(out2), ldo2, _p(aux), ldaux, _p(rowscale),
                             rows_per_sample, int(accumulate), psH, psW, splits, _p(workspace), workspace_bytes,
                             _stream())
    check(rc, "tulip_gemm_bf16")


def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, C, eps, merge=False, B=0, H=0, W=0):
    rc = _lib.load().tulip_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, C, eps,
                                         int(merge), B, H, W, _stream())
    check(rc, "tulip_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, rows, C, merge=False, B=0, H=0, W=0, param_partials=None,
                  dx_bf16=None, cast_rowscale=None, cast_rows_per_sample=1):
    rc = _lib.load().tulip_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), rows, C,
                                         int(merge), B, H, W, _p(param_partials), _p(dx_bf16), _p(cast_rowscale),
                                         cast_rows_per_sample, _stream())
    check(rc, "tulip_layernorm_bwd")


def layernorm_bwd_splitk(slabs, nslab, x, mean, rstd, gamma, dres, dx, rows, C, merge=False, B=0, H=0, W=0,
                         param_partials=None, dx_bf16=None, cast_rowscale=None, cast_rows_per_sample=1):
    """layernorm_bwd on the raw split-K slabs of the data-gradient GEMM in front (tulip_layernorm_bwd_splitk)."""
    rc = _lib.load().tulip_layernorm_bwd_splitk(_p(slabs), nslab, _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx),
                                                rows, C, int(merge), B, H, W, _p(param_partials), _p(dx_bf16),
                                                _p(cast_rowscale), cast_rows_per_sample, _stream())
    check(rc, "tulip_layernorm_bwd_splitk")


def splitk_resid_ln_supported(N):
    return bool(_lib.load().tulip_splitk_resid_ln_supported(N))


def splitk_resid_ln(slabs, nslab, M, N, bias, aux, ldaux, rowscale, rows_per_sample, out, ldo, out_bf16, ldo2, gamma, beta,
                    ln_out, mean, rstd, eps):
    """Split-K fold + residual epilogue + the following LayerNorm in one launch (tulip_splitk_resid_ln)."""
    check(_lib.load().tulip_splitk_resid_ln(_p(slabs), nslab, M, N, _p(bias), _p(aux), ldaux, _p(rowscale), rows_per_sample,
                                            _p(out), ldo, _p(out_bf16), ldo2, _p(gamma), _p(beta), _p(ln_out), _p(mean),
                                            _p(rstd), eps, _stream()), "tulip_splitk_resid_ln")


def layernorm_bwd_partial_rows(rows, C):
    return _lib.load().tulip_layernorm_bwd_partial_rows(rows, C)


def reduce_rows2(part0, stride0, out0, n0, part1, stride1, out1, n1, nrows):
    check(_lib.load().tulip_reduce_rows2(_p(part0), stride0, _p(out0), n0, _p(part1), stride1, _p(out1), n1, nrows,
                                         _stream()), "tulip_reduce_rows2")


def reduce_region(part, stride, out, n, rows, overwrite=False, scatter_index=None, scatter_nh=0, scatter_len=0, adamw=False):
    """One tulip_reduce_region: out[i] (+)= sum_s part[s*stride + i]; see include/tulip_hip.h.  adamw: the fold takes the
    optimizer step of out[0..n) instead of storing the sum (reduce_rows_multi(..., adam=adamw_ref(...)))."""
    return _lib.ReduceRegion(_p(part), _p(out), stride, n, rows, int(overwrite), _p(scatter_index), scatter_nh,
                             scatter_len, int(adamw))


def wgrad_item(dY, ldy, X, ldx, Nw, Kw, Mtok, dW, db=None, splits=1, overwrite=False, adamw=False):
    return _lib.WgradItem(_p(dY), _p(X), _p(dW), _p(db), ldy, ldx, Nw, Kw, Mtok, splits, int(overwrite), int(adamw))


def adamw_ref(hyper, grad, param, exp_avg, exp_avg_sq, param_bf16, decay_mask64=None):
    """tulip_adamw_ref: the flat buffers tulip_wgrad_group_adamw / tulip_reduce_rows_multi_adamw step in (all laid out like
    `grad`); decay_mask64: tulip_adamw's mask (bit 0: weight decay for these 64 elements; None: everywhere)."""
    return _lib.AdamwRef(_p(hyper), _p(grad), _p(param), _p(exp_avg), _p(exp_avg_sq), _p(param_bf16), _p(decay_mask64))


def reduce_rows_multi(regions, adam=None):
    if len(regions) > _lib.REDUCE_REGIONS_MAX:
        raise ValueError("too many regions for one launch")
    arr = (_lib.ReduceRegion * max(len(regions), 1))(*regions)
    if adam is None:
        check(_lib.load().tulip_reduce_rows_multi(arr, len(regions), _stream()), "tulip_reduce_rows_multi")
    else:
        check(_lib.load().tulip_reduce_rows_multi_adamw(arr, len(regions), ctypes.byref(adam), _stream()),
              "tulip_reduce_rows_multi_adamw")


def wgrad_tiles(Nw, Kw, small_tiles=False):
    """Workgroup tiles per token split of a [Nw][Kw] weight gradient in the grouped launch (tulip_wgrad_tiles)."""
    return _lib.load().tulip_wgrad_tiles(Nw, Kw, _lib.WGRAD_SMALL_TILES if small_tiles else 0)


def wgrad_group_profiled(items, workspace, workspace_bytes, stamps, small_tiles=False):
    """tulip_wgrad_group_profiled: the grouped launch alone; stamps = int64 device tensor [workgroups, 4] (phase stamps)."""
    ia = (_lib.WgradItem * max(len(items), 1))(*items)
    check(_lib.load().tulip_wgrad_group_profiled(ia, len(items), _p(workspace), workspace_bytes,
                                                 _lib.WGRAD_SMALL_TILES if small_tiles else 0, _p(stamps), _stream()),
          "tulip_wgrad_group_profiled")


def wgrad_group(items, extra, workspace, workspace_bytes, fold=True, adam=None, small_tiles=False):
    """tulip_wgrad_group: grouped weight-gradient GEMM + one fold launch that also carries the `extra` regions.
    adam (ops.adamw_ref): items built with adamw=True take their optimizer step in the write-out (tulip_wgrad_group_adamw).
    small_tiles: TULIP_WGRAD_SMALL_TILES (the 64 x 96 tile everywhere; A/B measurements)."""
    ia = (_lib.WgradItem * max(len(items), 1))(*items)
    ea = (_lib.ReduceRegion * max(len(extra), 1))(*extra)
    fold = int(bool(fold)) | (_lib.WGRAD_SMALL_TILES if small_tiles else 0)
    if adam is None:
        check(_lib.load().tulip_wgrad_group(ia, len(items), ea, len(extra), _p(workspace), workspace_bytes, int(fold),
                                            _stream()), "tulip_wgrad_group")
    else:
        check(_lib.load().tulip_wgrad_group_adamw(ia, len(items), ea, len(extra), _p(workspace), workspace_bytes, int(fold),
                                                  ctypes.byref(adam), _stream()), "tulip_wgrad_group_adamw")


def wgrad_group_regions(items, workspace):
    """The fold regions of a grouped launch issued with fold=False (tulip_wgrad_group_regions)."""
    ia = (_lib.WgradItem * max(len(items), 1))(*items)
    out = (_lib.ReduceRegion * _lib.REDUCE_REGIONS_MAX)()
    n = _lib.load().tulip_wgrad_group_regions(ia, len(items), _p(workspace), out, _lib.REDUCE_REGIONS_MAX)
    check(min(n, 0), "tulip_wgrad_group_regions")
    return [out[i] for i in range(n)]


def layernorm_bwd_params(dy, x, mean, rstd, dgamma, dbeta, rows, C, merge=False, B=0, H=0, W=0):
    rc = _lib.load().tulip_layernorm_bwd_params(_p(dy), _p(x), _p(mean), _p(rstd), _p(dgamma), _p(dbeta), rows, C,
                                                int(merge), B, H, W, _stream())
    check(rc, "tulip_layernorm_bwd_params")


def patch_embed_fwd(img, w, b, gamma, beta, out, B, Cin, Hin, Win, E, p0, p1, kw, circular, eps, out_bf16=None,
                    ld_bf16=0, draw=None):
    """draw = (keep, scale, u_out, nslots, B, seed, counter): the step's DropPath draws (drop_path_scales) ride in this launch."""
    if draw is None:
        rc = _lib.load().tulip_patch_embed_fwd(_p(img), _p(w), _p(b), _p(gamma), _p(beta), _p(out), B, Cin, Hin, Win, E,
                                               p0, p1, kw, int(circular), eps, _p(out_bf16), ld_bf16, _stream())
        check(rc, "tulip_patch_embed_fwd")
        return
    keep, scale, u_out, nslots, Bd, seed, counter = draw
    dd = _lib.DropDraw(_p(keep), _p(scale), _p(u_out), nslots, Bd, int(seed) & (2 ** 64 - 1), _p(counter))
    rc = _lib.load().tulip_patch_embed_fwd_draw(_p(img), _p(w), _p(b), _p(gamma), _p(beta), _p(out), B, Cin, Hin, Win, E,
                                                p0, p1, kw, int(circular), eps, _p(out_bf16), ld_bf16, ctypes.byref(dd),
                                                _stream())
    check(rc, "tulip_patch_embed_fwd_draw")


def patch_embed_bwd(img, w, b, gamma, dout, dw, db, dgamma, dbeta, B, Cin, Hin, Win, E, p0, p1, kw, circular, eps,
                    partial_stride=0):
    rc = _lib.load().tulip_patch_embed_bwd(_p(img), _p(w), _p(b), _p(gamma), _p(dout), _p(dw), _p(db), _p(dgamma),
                                           _p(dbeta), B, Cin, Hin, Win, E, p0, p1, kw, int(circular), eps,
                                           partial_stride, _stream())
    check(rc, "tulip_patch_embed_bwd")


def patch_embed_bwd_blocks(ntok):
    return _lib.load().tulip_patch_embed_bwd_blocks(ntok)


def window_attn_bwd_partial_rows(B, H, W, nh, win):
    return _lib.load().tulip_window_attn_bwd_partial_rows(B, H, W, nh, win[0], win[1])


def window_attn_fwd(qkv, bias_table, rel_index, out, B, H, W, C, nh, win, shift, masked):
    rc = _lib.load().tulip_window_attn_fwd(_p(qkv), _p(bias_table), _p(rel_index), _p(out), B, H, W, C, nh, win[0],
                                           win[1], shift[0], shift[1], int(masked), _stream())
    check(rc, "tulip_window_attn_fwd")


def window_attn_bwd(qkv, dout, bias_table, rel_index, dqkv, dbias_dense, B, H, W, C, nh, win, shift, masked):
    rc = _lib.load().tulip_window_attn_bwd(_p(qkv), _p(dout), _p(bias_table), _p(rel_index), _p(dqkv),
                                           _p(dbias_dense), B, H, W, C, nh, win[0], win[1], shift[0], shift[1],
                                           int(masked), _stream())
    check(rc, "tulip_window_attn_bwd")


def cast_f32_bf16(x, y, rows, cols, rowscale=None, rows_per_sample=1):
    check(_lib.load().tulip_cast_f32_bf16(_p(x), _p(y), rows, cols, _p(rowscale), rows_per_sample, _stream()),
          "tulip_cast_f32_bf16")


def reduce_splits(slabs, out, n, splits):
    check(_lib.load().tulip_reduce_splits(_p(slabs), _p(out), n, splits, _stream()), "tulip_reduce_splits")


def gemm_packed_supported(M, N, K, splits=1) -> bool:
    return bool(_lib.load().tulip_gemm_packed_supported(M, N, K, splits))


def gemm_effective_splits(K, splits):
    return _lib.load().tulip_gemm_effective_splits(K, splits)


def cast_flat(x, y, n):
    check(_lib.load().tulip_cast_flat(_p(x), _p(y), n, _stream()), "tulip_cast_flat")


def cast_bf16_f32(x, y, n):
    check(_lib.load().tulip_cast_bf16_f32(_p(x), _p(y), n, _stream()), "tulip_cast_bf16_f32")


def tail_fwd(xn, We, be, wd, pred, B, H, W, E):
    check(_lib.load().tulip_tail_fwd(_p(xn), _p(We), _p(be), _p(wd), _p(pred), B, H, W, E, _stream()),
          "tulip_tail_fwd")


def tail_bwd(xn, We, be, wd, dpred, dz, dwd, B, H, W, E, target=None, gscale_dev=None, gscale=1.0):
    check(_lib.load().tulip_tail_bwd(_p(xn), _p(We), _p(be), _p(wd), _p(dpred), _p(dz), _p(dwd), B, H, W, E,
                                     _p(target), _p(gscale_dev), float(gscale), _stream()), "tulip_tail_bwd")


def tail_fused_bwd_supported(E):
    return bool(_lib.load().tulip_tail_fused_bwd_supported(int(E)))


def tail_bwd_dgrad(xn, We, be, wd, dpred, dxn, dwd, B, H, W, E, target=None, gscale_dev=None, gscale=1.0):
    """Head backward on the chain: dxn (bf16 [M][E]) and the decoder_pred partial rows; d(expand) is never written."""
    check(_lib.load().tulip_tail_bwd_dgrad(_p(xn), _p(We), _p(be), _p(wd), _p(dpred), _p(dxn), _p(dwd), B, H, W, E,
                                           _p(target), _p(gscale_dev), float(gscale), _stream()), "tulip_tail_bwd_dgrad")


def tail_bwd_dgrad_ln(xn, We, be, wd, dpred, dwd, B, H, W, E, x, mean, rstd, gamma, dx, ln_partials, dx_bf16=None,
                      cast_rowscale=None, cast_rows_per_sample=1, target=None, gscale_dev=None, gscale=1.0):
    """tail_bwd_dgrad with norm_up's backward in the epilogue: dx / dx_bf16 / [dgamma | dbeta] partial rows (one per 32 tokens)."""
    check(_lib.load().tulip_tail_bwd_dgrad_ln(_p(xn), _p(We), _p(be), _p(wd), _p(dpred), _p(dwd), B, H, W, E, _p(target),
                                              _p(gscale_dev), float(gscale), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx),
                                              _p(dx_bf16), _p(cast_rowscale), cast_rows_per_sample, _p(ln_partials),
                                              _stream()), "tulip_tail_bwd_dgrad_ln")


def tail_fwd_ln(x, gamma, beta, eps, xn, mean, rstd, We, be, wd, pred, B, H, W, E, target=None, loss_partials=None,
                log_transform=False):
    """norm_up + fused head (+ the L1 / pixel loss partial sums per 32 tokens) in one launch."""
    check(_lib.load().tulip_tail_fwd_ln(_p(x), _p(gamma), _p(beta), float(eps), _p(xn), _p(mean), _p(rstd), _p(We), _p(be),
                                        _p(wd), _p(pred), _p(target), _p(loss_partials), int(log_transform), B, H, W, E,
                                        _stream()), "tulip_tail_fwd_ln")


def l1_loss_final(partials, losses, nblocks, n, log_transform):
    check(_lib.load().tulip_l1_loss_final(_p(partials), _p(losses), nblocks, n, int(log_transform), _stream()),
          "tulip_l1_loss_final")


def tail_wgrad_splits(B, H, W, E):
    return _lib.load().tulip_tail_wgrad_splits(B, H, W, E)


def tail_wgrad(xn, We, be, wd, dpred, slabs_w, slabs_b, B, H, W, E, target=None, gscale_dev=None, gscale=1.0):
    """Expand-conv weight / bias gradient of the head as token-split slabs, d(expand) recomputed channel-sliced."""
    check(_lib.load().tulip_tail_wgrad(_p(xn), _p(We), _p(be), _p(wd), _p(dpred), _p(slabs_w), _p(slabs_b), B, H, W, E,
                                       _p(target), _p(gscale_dev), float(gscale), _stream()), "tulip_tail_wgrad")


def expand_norm_fwd(y, gamma, beta, mean, rstd, B, H, W, P, Cn, eps, out_bf16=None, ld=0, dotw=None, pred=None):
    """PatchExpanding / FinalPatchExpanding rearrange + LayerNorm (+ decoder_pred dot), see include/tulip_hip.h."""
    check(_lib.load().tulip_expand_norm_fwd(_p(y), _p(gamma), _p(beta), _p(out_bf16), ld, _p(dotw), _p(pred), _p(mean),
                                            _p(rstd), B, H, W, P, Cn, eps, _stream()), "tulip_expand_norm_fwd")


def expand_norm_bwd(y, mean, rstd, gamma, dy_nat, partials, B, H, W, P, Cn, dy_fine=None, ld=0, dpred=None, dotw=None,
                    beta=None):
    check(_lib.load().tulip_expand_norm_bwd(_p(dy_fine), ld, _p(dpred), _p(dotw), _p(y), _p(mean), _p(rstd), _p(gamma),
                                            _p(beta), _p(dy_nat), _p(partials), B, H, W, P, Cn, _stream()),
          "tulip_expand_norm_bwd")


def expand_norm_bwd_partial_rows(B, H, W, P):
    return _lib.load().tulip_expand_norm_bwd_partial_rows(B, H, W, P)


def l1_loss_fwd(pred, target, partials, losses, n, log_transform):
    check(_lib.load().tulip_l1_loss_fwd(_p(pred), _p(target), _p(partials), _p(losses), n, int(log_transform),
                                        _stream()), "tulip_l1_loss_fwd")


def l1_loss_bwd(pred, target, gscale_dev, gscale, dpred, n):
    check(_lib.load().tulip_l1_loss_bwd(_p(pred), _p(target), _p(gscale_dev), float(gscale), _p(dpred), n,
                                        _stream()), "tulip_l1_loss_bwd")


def adamw(p, g, m, v, p_bf16, n, hyper, decay_mask64=None, zero_grad=False):
    check(_lib.load().tulip_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(hyper), _p(decay_mask64),
                                  int(zero_grad), _stream()), "tulip_adamw")


def adamw_blocks(p, g, m, v, p_bf16, blocks, nblocks, hyper, decay_mask64=None, zero_grad=False):
    """tulip_adamw over the listed 64-float blocks only (blocks: int32 device tensor of block indices)."""
    check(_lib.load().tulip_adamw_blocks(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), _p(blocks), nblocks, _p(hyper),
                                         _p(decay_mask64), int(zero_grad), _stream()), "tulip_adamw_blocks")


def grad_norm(g, n, partials, out, scale_dev=None, scale=1.0):
    check(_lib.load().tulip_grad_norm(_p(g), n, _p(partials), _p(scale_dev), float(scale), _p(out), _stream()),
          "tulip_grad_norm")


def range_prep(raw, raw_dtype, batch_stride, row_stride, col_stride, base_offset, hi, lo, B, H, W, row_factor,
               col_factor, row_phase, col_phase, scale, gate, min_range, max_range, log_transform, roll_shift):
    check(_lib.load().tulip_range_prep(_p(raw), raw_dtype, batch_stride, row_stride, col_stride, base_offset, _p(hi),
                                       _p(lo), B, H, W, row_factor, col_factor, row_phase, col_phase, float(scale),
                                       int(gate), float(min_range), float(max_range), int(log_transform),
                                       int(roll_shift), _stream()), "tulip_range_prep")


# ---- evaluation post-processing / metrics (csrc/evalpost.hip)
def mc_aggregate(preds, passes, n, noise_threshold, out):
    check(_lib.load().tulip_mc_aggregate(_p(preds), passes, n, float(noise_threshold), _p(out), _stream()),
          "tulip_mc_aggregate")


def eval_postprocess(pred, hi, lo, pred_img, hi_img, partials, mae_out, H, W, h, w, log_transform, gate_min, gate_max,
                     keep_close):
    check(_lib.load().tulip_eval_postprocess(_p(pred), _p(hi), _p(lo), _p(pred_img), _p(hi_img), _p(partials),
                                             _p(mae_out), H, W, h, w, int(log_transform), float(gate_min),
                                             float(gate_max), float(keep_close), _stream()), "tulip_eval_postprocess")


def range_to_xyz(img, sin_h, cos_h, sin_v, cos_v, max_range, H, W, xyz):
    check(_lib.load().tulip_range_to_xyz(_p(img), _p(sin_h), _p(cos_h), _p(sin_v), _p(cos_v), float(max_range), H, W,
                                         _p(xyz), _stream()), "tulip_range_to_xyz")


def range_to_xyz_durlar(img, col_tables, row_tables, row_offset, max_range, origin_offset, z_offset, H, W, xyz):
    check(_lib.load().tulip_range_to_xyz_durlar(_p(img), _p(col_tables), _p(row_tables), _p(row_offset),
                                                float(max_range), float(origin_offset), float(z_offset), H, W,
                                                _p(xyz), _stream()), "tulip_range_to_xyz_durlar")


def voxel_metrics(pcd_pred, n_pred, pcd_gt, n_gt, is_f64, grid_size, bitmap_pred, bitmap_gt, bitmap_words, scratch,
                  out):
    check(_lib.load().tulip_voxel_metrics(_p(pcd_pred), n_pred, _p(pcd_gt), n_gt, int(is_f64), float(grid_size),
                                          _p(bitmap_pred), _p(bitmap_gt), bitmap_words, _p(scratch), _p(out),
                                          _stream()), "tulip_voxel_metrics")


def chamfer_sq(a, na, b, nb, is_f64, dist_a, dist_b, scratch, out):
    check(_lib.load().tulip_chamfer_sq(_p(a), na, _p(b), nb, int(is_f64), _p(dist_a), _p(dist_b), _p(scratch),
                                       _p(out), _stream()), "tulip_chamfer_sq")


def drop_path_scales(keep, scale, u_out, nslots, B, seed, counter):
    check(_lib.load().tulip_drop_path_scales(_p(keep), _p(scale), _p(u_out), nslots, B, int(seed) & (2 ** 64 - 1),
                                             _p(counter), _stream()), "tulip_drop_path_scales")


def kitti_range_map(points, n, rows, cols, ang_start_y, ang_res_y, ang_res_x, max_range, min_range, winner, out):
    check(_lib.load().tulip_kitti_range_map(_p(points), n, rows, cols, float(ang_start_y), float(ang_res_y),
                                            float(ang_res_x), float(max_range), float(min_range), _p(winner), _p(out),
                                            _stream()), "tulip_kitti_range_map")


def swin96_block_fwd(stamps=None, **kw):
    """tulip_swin96_block_fwd: keyword arguments are the fields of tulip_swin96_desc (tensors or addresses).
    stamps (int64 device tensor): the profiled twin."""
    d = _lib.Swin96Desc()
    for name, _t in _lib.Swin96Desc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked", "eps") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    if stamps is not None:
        check(_lib.load().tulip_swin96_block_fwd_profiled(ctypes.byref(d), _p(stamps), _stream()), "tulip_swin96_block_fwd_profiled")
        return
    check(_lib.load().tulip_swin96_block_fwd(ctypes.byref(d), _stream()), "tulip_swin96_block_fwd")


def _swin96_desc(kw):
    d = _lib.Swin96Desc()
    kw = dict(kw)
    for name, _t in _lib.Swin96Desc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked", "eps") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    return d


def swin96_pair_sync_bytes(B, H, W) -> int:
    return int(_lib.load().tulip_swin96_pair_sync_bytes(B, H, W))


def swin96_pair_fwd(first: dict, second: dict, sync):
    """tulip_swin96_pair_fwd: two consecutive C = 96 blocks in one launch; first / second hold the fields of tulip_swin96_desc,
    sync is the zero-initialised uint8 / int32 device tensor of swin96_pair_sync_bytes that these launches own."""
    d0, d1 = _swin96_desc(first), _swin96_desc(second)
    check(_lib.load().tulip_swin96_pair_fwd(ctypes.byref(d0), ctypes.byref(d1), _p(sync), sync.numel() * sync.element_size(),
                                            _stream()), "tulip_swin96_pair_fwd")


def swinw_supported(C, H, W) -> bool:
    return bool(_lib.load().tulip_swinw_supported(C, H, W))


def swinw_block_fwd(C, out_bf16=None, stamps=None, exchange=None, **kw):
    """tulip_swinw_block_fwd (C = 192 / 384): keyword arguments are the fields of tulip_swin96_desc.
    stamps (int64 device tensor): the profiled twin, per-wave shader-clock stamps at the phase boundaries.
    exchange (zeroed uint8 device tensor of swinw_split_bytes): tulip_swinw_block_fwd_split, two workgroups per window."""
    d = _lib.Swin96Desc()
    for name, _t in _lib.Swin96Desc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked", "eps") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    if exchange is not None:
        check(_lib.load().tulip_swinw_block_fwd_split(ctypes.byref(d), C, _p(out_bf16), _p(exchange),
                                                      exchange.numel() * exchange.element_size(), _p(stamps), _stream()),
              "tulip_swinw_block_fwd_split")
        return
    if stamps is not None:
        check(_lib.load().tulip_swinw_block_fwd_profiled(ctypes.byref(d), C, _p(out_bf16), _p(stamps), _stream()),
              "tulip_swinw_block_fwd_profiled")
        return
    check(_lib.load().tulip_swinw_block_fwd(ctypes.byref(d), C, _p(out_bf16), _stream()), "tulip_swinw_block_fwd")


def swinw_split_bytes(C, B, H, W) -> int:
    """tulip_swinw_split_bytes: size of the exchange buffer of the two-workgroups-per-window form, 0 where it does not exist."""
    return _lib.load().tulip_swinw_split_bytes(C, B, H, W)


def stamp_realtime(dst) -> None:
    """tulip_stamp_realtime: dst (int64 device tensor element) = the 100 MHz device clock at this point of the stream."""
    check(_lib.load().tulip_stamp_realtime(_p(dst), _stream()), "tulip_stamp_realtime")


def swinw_bwd_partial_rows(C, B, H, W) -> int:
    return _lib.load().tulip_swinw_bwd_partial_rows(C, B, H, W)


def swinw_block_bwd(C, exchange=None, **kw):
    """tulip_swinw_block_bwd: fields of tulip_swin96_bwd_desc; w_* are the TRANSPOSED bf16 weights.
    exchange: tulip_swinw_block_bwd_split (see swinw_block_fwd)."""
    d = _lib.Swin96BwdDesc()
    for name, _t in _lib.Swin96BwdDesc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    if exchange is not None:
        check(_lib.load().tulip_swinw_block_bwd_split(ctypes.byref(d), C, _p(exchange), exchange.numel() * exchange.element_size(),
                                                      _stream()), "tulip_swinw_block_bwd_split")
        return
    check(_lib.load().tulip_swinw_block_bwd(ctypes.byref(d), C, _stream()), "tulip_swinw_block_bwd")


def swind_supported(C, H, W, win) -> bool:
    return bool(_lib.load().tulip_swind_supported(C, H, W, int(win[0]), int(win[1])))


def swind_groups(C, B, H, W, win) -> int:
    """tulip_swind_groups: window groups of a deep-stage block launch = partial rows of its backward."""
    return _lib.load().tulip_swind_groups(C, B, H, W, int(win[0]), int(win[1]))


def swind_block_fwd(C, win, out_bf16=None, phases=15, stamps=None, **kw):
    """tulip_swind_block_fwd (C = 768 / 1536): keyword arguments are the fields of tulip_swin96_desc; w_* fragment-major copies."""
    d = _lib.Swin96Desc()
    for name, _t in _lib.Swin96Desc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked", "eps") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    check(_lib.load().tulip_swind_block_fwd(ctypes.byref(d), C, int(win[0]), int(win[1]), _p(out_bf16), phases, _p(stamps),
                                            _stream()), "tulip_swind_block_fwd")


def swind_block_bwd(C, win, d_norm_out, phases=15, stamps=None, **kw):
    """tulip_swind_block_bwd: fields of tulip_swin96_bwd_desc; w_* fragment-major copies of the TRANSPOSED weights; d_norm_out:
    fp32 [M][C] scratch that phases 2 / 8 fill for tulip_layernorm_bwd_splitk."""
    d = _lib.Swin96BwdDesc()
    for name, _t in _lib.Swin96BwdDesc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    check(_lib.load().tulip_swind_block_bwd(ctypes.byref(d), C, int(win[0]), int(win[1]), _p(d_norm_out), phases, _p(stamps),
                                            _stream()), "tulip_swind_block_bwd")


def pack_items(entries):
    """[(src address, dst address, rows, cols, transpose)] -> ctypes array for pack_bf16_multi (build once, launch often)."""
    return ((_lib.PackItem * max(len(entries), 1))(*[_lib.PackItem(_p(s), _p(d), r, c, int(t)) for s, d, r, c, t in entries]),
            len(entries))


def pack_bf16_multi(items, n):
    """Fragment-major bf16 copies (optionally of the transpose) of a list of matrices, see include/tulip_hip.h."""
    for k in range(0, n, _lib.PACK_MAX):
        cnt = min(_lib.PACK_MAX, n - k)
        check(_lib.load().tulip_pack_bf16_multi(ctypes.byref(items, k * ctypes.sizeof(_lib.PackItem)), cnt, _stream()),
              "tulip_pack_bf16_multi")


def swin96_bwd_partial_rows(B, H, W) -> int:
    return _lib.load().tulip_swin96_bwd_partial_rows(B, H, W)


def swin96_block_bwd(stamps=None, **kw):
    """tulip_swin96_block_bwd: keyword arguments are the fields of tulip_swin96_bwd_desc (tensors or addresses)."""
    d = _lib.Swin96BwdDesc()
    for name, _t in _lib.Swin96BwdDesc._fields_:
        v = kw.pop(name, None)
        setattr(d, name, _p(v) if name not in ("B", "H", "W", "shift_h", "shift_w", "masked") else v)
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    if stamps is not None:
        check(_lib.load().tulip_swin96_block_bwd_profiled(ctypes.byref(d), _p(stamps), _stream()), "tulip_swin96_block_bwd_profiled")
        return
    check(_lib.load().tulip_swin96_block_bwd(ctypes.byref(d), _stream()), "tulip_swin96_block_bwd")


# ---- the stage boundaries as one launch each (csrc/glue.hip): keyword arguments are the descriptor's fields
def _fill(cls, kw, ints):
    d = cls()
    for name, _t in cls._fields_:
        v = kw.pop(name, None)
        if name in ints:
            setattr(d, name, 0 if v is None else v)
        else:
            setattr(d, name, _p(v))
    if kw:
        raise TypeError(f"unknown fields {sorted(kw)}")
    return d


def merge_fwd_supported(Cin, B, H, W) -> bool:
    return bool(_lib.load().tulip_merge_fwd_supported(Cin, B, H, W))


def merge_fwd(**kw):
    """tulip_merge_fwd: PatchMerging.forward (2x2 gather + LayerNorm + reduction GEMM) in one launch."""
    d = _fill(_lib.MergeFwdDesc, kw, ("ld_bf16", "B", "H", "W", "Cin", "eps"))
    check(_lib.load().tulip_merge_fwd(ctypes.byref(d), _stream()), "tulip_merge_fwd")


def merge_bwd_supported(Cp, B, H, W) -> bool:
    return bool(_lib.load().tulip_merge_bwd_supported(Cp, B, H, W))


def merge_bwd_partial_rows(Cp, B, H, W) -> int:
    return _lib.load().tulip_merge_bwd_partial_rows(Cp, B, H, W)


def merge_bwd(**kw):
    """tulip_merge_bwd: [x_save half of the skip Linear's data gradient +] reduction data gradient + LayerNorm backward."""
    d = _fill(_lib.MergeBwdDesc, kw, ("cast_rows_per_sample", "B", "H", "W", "Cp"))
    check(_lib.load().tulip_merge_bwd(ctypes.byref(d), _stream()), "tulip_merge_bwd")


def unmerge_skip_supported(C, B, H, W) -> bool:
    return bool(_lib.load().tulip_unmerge_skip_supported(C, B, H, W))


def unmerge_skip_fwd(**kw):
    """tulip_unmerge_skip_fwd: PatchUnmerging.forward -> skip Linear(cat[...]) in one launch."""
    d = _fill(_lib.UnmergeSkipDesc, kw, ("B", "H", "W", "C"))
    check(_lib.load().tulip_unmerge_skip_fwd(ctypes.byref(d), _stream()), "tulip_unmerge_skip_fwd")


def skip_unmerge_bwd(**kw):
    """tulip_skip_unmerge_bwd: the skip Linear's data gradient (unmerged half) -> PatchUnmerging's data gradient."""
    d = _fill(_lib.SkipUnmergeBwdDesc, kw, ("cast_rows_per_sample", "B", "H", "W", "C"))
    check(_lib.load().tulip_skip_unmerge_bwd(ctypes.byref(d), _stream()), "tulip_skip_unmerge_bwd")
