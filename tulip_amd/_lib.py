"""ctypes binding of libtulip_hip.so (the C ABI declared in include/tulip_hip.h).

There is NO fallback: if the HIP library is missing or a symbol is absent this module raises, so
a GPU run can never silently execute anything but the hand-written gfx950 kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TULIP_HIP_LIB") or os.path.join(_PKG, "libtulip_hip.so")   # (TULIP_HIP_LIB: dev, A/B builds)

P, I, F, L, D = c_void_p, c_int, c_float, c_int64, c_double

class Swin96Desc(ctypes.Structure):
    """tulip_swin96_desc (include/tulip_hip.h)."""
    _fields_ = [(n, P) for n in ("x_in", "x1", "x_out", "xn1", "qkv", "attn_out", "xn2", "fc1_pre", "fc1_act", "mean1",
                                 "rstd1", "mean2", "rstd2", "w_qkv", "w_proj", "w_fc1", "w_fc2", "b_qkv", "b_proj",
                                 "b_fc1", "b_fc2", "norm1_weight", "norm1_bias", "norm2_weight", "norm2_bias",
                                 "bias_table", "rel_index", "drop_scale_attn", "drop_scale_mlp")] + \
               [(n, I) for n in ("B", "H", "W", "shift_h", "shift_w", "masked")] + [("eps", F)]


class Swin96BwdDesc(ctypes.Structure):
    """tulip_swin96_bwd_desc (include/tulip_hip.h)."""
    _fields_ = [(n, P) for n in ("dx", "x_in", "x1", "qkv", "fc1_pre", "mean1", "rstd1", "mean2", "rstd2", "w_qkv",
                                 "w_proj", "w_fc1", "w_fc2", "norm1_weight", "norm2_weight", "bias_table", "rel_index",
                                 "drop_scale_attn", "drop_scale_mlp", "d_out_mlp", "d_fc1_pre", "d_out_attn", "d_qkv",
                                 "dx_bf16", "dx_bf16_scale", "norm1_partials", "norm2_partials", "bias_partials")] + \
               [(n, I) for n in ("B", "H", "W", "shift_h", "shift_w", "masked")] + \
               [(n, P) for n in ("b_qkv", "b_fc1", "norm1_bias", "norm2_bias")]


class ReduceRegion(ctypes.Structure):
    """tulip_reduce_region (include/tulip_hip.h)."""
    _fields_ = [("partials", P), ("out", P), ("stride", L), ("n", L), ("rows", I), ("overwrite", I),
                ("scatter_index", P), ("scatter_nh", I), ("scatter_len", I), ("adamw", I)]


class WgradItem(ctypes.Structure):
    """tulip_wgrad_item (include/tulip_hip.h)."""
    _fields_ = [("dY", P), ("X", P), ("dW", P), ("db", P), ("ldy", I), ("ldx", I), ("Nw", I), ("Kw", I), ("Mtok", I),
                ("splits", I), ("overwrite", I), ("reserved_", I)]


class AdamwRef(ctypes.Structure):
    """tulip_adamw_ref (include/tulip_hip.h)."""
    _fields_ = [(n, P) for n in ("hyper", "grad", "param", "exp_avg", "exp_avg_sq", "param_bf16", "decay_mask64")]


class DropDraw(ctypes.Structure):
    """tulip_drop_draw (include/tulip_hip.h)."""
    _fields_ = [("keep", P), ("scale", P), ("u_out", P), ("nslots", I), ("B", I), ("seed", ctypes.c_uint64), ("counter", P)]


class PackItem(ctypes.Structure):
    """tulip_pack_item (include/tulip_hip.h)."""
    _fields_ = [("src", P), ("dst", P), ("rows", I), ("cols", I), ("transpose", I)]


class MergeFwdDesc(ctypes.Structure):      # tulip_merge_fwd_desc
    _fields_ = [("x", P), ("gamma", P), ("beta", P), ("w_packed", P), ("xm", P), ("mean", P), ("rstd", P), ("y", P),
                ("y_bf16", P), ("ld_bf16", I), ("B", I), ("H", I), ("W", I), ("Cin", I), ("eps", F)]


class MergeBwdDesc(ctypes.Structure):      # tulip_merge_bwd_desc
    _fields_ = [("dx_in", P), ("dy_skip", P), ("w_skip_t_packed", P), ("dyb", P), ("w_red_t_packed", P), ("x_prev", P),
                ("mean", P), ("rstd", P), ("gamma", P), ("dx_prev", P), ("param_partials", P), ("dx_bf16", P),
                ("cast_rowscale", P), ("cast_rows_per_sample", I), ("B", I), ("H", I), ("W", I), ("Cp", I)]


class UnmergeSkipDesc(ctypes.Structure):   # tulip_unmerge_skip_desc
    _fields_ = [("x_bf16", P), ("w_expand_packed", P), ("b_expand", P), ("cat", P), ("w_skip_packed", P), ("b_skip", P),
                ("out", P), ("B", I), ("H", I), ("W", I), ("C", I)]


class SkipUnmergeBwdDesc(ctypes.Structure):    # tulip_skip_unmerge_bwd_desc
    _fields_ = [("dy_skip", P), ("w_skip_t_packed", P), ("dz", P), ("w_expand_t_packed", P), ("dx", P), ("dx_bf16", P),
                ("cast_rowscale", P), ("cast_rows_per_sample", I), ("B", I), ("H", I), ("W", I), ("C", I)]


REDUCE_REGIONS_MAX, WGRAD_GROUP_MAX, PACK_MAX = 48, 16, 64
GEMM_NO_TOUCH, GEMM_CHECKED, GEMM_NO_MID, GEMM_MID, WGRAD_SMALL_TILES, BLOCK_NO_WARM = 0x100, 0x200, 0x400, 0x800, 0x100, 8     # per-call flag bits (tulip_hip.h)
GEMM_B_PACKED = 0x1000
ABI_VERSION = 6      # TULIP_ABI_VERSION of include/tulip_hip.h: the ctypes structs above mirror that layout

# name -> argtypes (must mirror include/tulip_hip.h; tests/test_cabi.py cross-checks against the header)
SIGNATURES = {
    "tulip_gemm_bf16": [P, I, I, P, I, I, I, I, I, I, P, P, I, P, I, P, I, P, I, I, I, I, I, P, L, P],
    "tulip_swin96_block_fwd": [P, P],
    "tulip_swin96_pair_sync_bytes": [I, I, I],
    "tulip_swin96_pair_fwd": [P, P, P, ctypes.c_size_t, P],
    "tulip_swin96_block_bwd": [P, P],
    "tulip_swin96_block_fwd_profiled": [P, P, P],
    "tulip_swin96_block_bwd_profiled": [P, P, P],
    "tulip_swin96_bwd_partial_rows": [I, I, I],
    "tulip_swinw_supported": [I, I, I],
    "tulip_swinw_block_fwd": [P, I, P, P],
    "tulip_swinw_block_fwd_profiled": [P, I, P, P, P],
    "tulip_swinw_split_bytes": [I, I, I, I],
    "tulip_swinw_block_fwd_split": [P, I, P, P, ctypes.c_size_t, P, P],
    "tulip_swinw_bwd_partial_rows": [I, I, I, I],
    "tulip_swinw_block_bwd": [P, I, P],
    "tulip_swinw_block_bwd_split": [P, I, P, ctypes.c_size_t, P],
    "tulip_swind_supported": [I, I, I, I, I],
    "tulip_swind_groups": [I, I, I, I, I, I],
    "tulip_swind_block_fwd": [P, I, I, I, P, I, P, P],
    "tulip_swind_block_bwd": [P, I, I, I, P, I, P, P],
    "tulip_pack_bf16_multi": [P, I, P],
    "tulip_layernorm_fwd": [P, P, P, P, P, P, I, I, F, I, I, I, I, P],
    "tulip_layernorm_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, I, P],
    "tulip_layernorm_bwd_splitk": [P, I, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, I, P],
    "tulip_splitk_resid_ln_supported": [I],
    "tulip_splitk_resid_ln": [P, I, I, I, P, P, I, P, I, P, I, P, I, P, P, P, P, P, F, P],
    "tulip_layernorm_bwd_partial_rows": [I, I],
    "tulip_layernorm_bwd_params": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "tulip_patch_embed_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, P, I, P],
    "tulip_patch_embed_fwd_draw": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, P, I, P, P],
    "tulip_patch_embed_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P],
    "tulip_patch_embed_bwd_blocks": [I],
    "tulip_window_attn_fwd": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "tulip_window_attn_bwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "tulip_window_attn_bwd_partial_rows": [I, I, I, I, I, I],
    "tulip_cast_f32_bf16": [P, P, I, I, P, I, P],
    "tulip_reduce_splits": [P, P, L, I, P],
    "tulip_reduce_rows2": [P, L, P, L, P, L, P, L, I, P],
    "tulip_reduce_rows_multi": [P, I, P],
    "tulip_reduce_rows_multi_adamw": [P, I, P, P],
    "tulip_wgrad_group": [P, I, P, I, P, L, I, P],
    "tulip_wgrad_tiles": [I, I, I],
    "tulip_wgrad_group_regions": [P, I, P, P, I],
    "tulip_wgrad_group_adamw": [P, I, P, I, P, L, I, P, P],
    "tulip_wgrad_group_profiled": [P, I, P, L, I, P, P],
    "tulip_gemm_effective_splits": [I, I],
    "tulip_gemm_packed_supported": [I, I, I, I],
    "tulip_cast_flat": [P, P, L, P],
    "tulip_cast_bf16_f32": [P, P, L, P],
    "tulip_tail_fwd": [P, P, P, P, P, I, I, I, I, P],
    "tulip_tail_bwd": [P, P, P, P, P, P, P, I, I, I, I, P, P, F, P],
    "tulip_tail_fused_bwd_supported": [I],
    "tulip_tail_bwd_dgrad": [P, P, P, P, P, P, P, I, I, I, I, P, P, F, P],
    "tulip_tail_bwd_dgrad_ln": [P, P, P, P, P, P, I, I, I, I, P, P, F, P, P, P, P, P, P, P, I, P, P],
    "tulip_tail_fwd_ln": [P, P, P, F, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "tulip_l1_loss_final": [P, P, I, L, I, P],
    "tulip_tail_wgrad_splits": [I, I, I, I],
    "tulip_tail_wgrad": [P, P, P, P, P, P, P, I, I, I, I, P, P, F, P],
    "tulip_expand_norm_fwd": [P, P, P, P, I, P, P, P, P, I, I, I, I, I, F, P],
    "tulip_expand_norm_bwd": [P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "tulip_expand_norm_bwd_partial_rows": [I, I, I, I],
    "tulip_l1_loss_fwd": [P, P, P, P, L, I, P],
    "tulip_l1_loss_bwd": [P, P, P, F, P, L, P],
    "tulip_adamw": [P, P, P, P, P, L, P, P, I, P],
    "tulip_adamw_blocks": [P, P, P, P, P, P, I, P, P, I, P],
    "tulip_drop_path_scales": [P, P, P, I, I, ctypes.c_uint64, P, P],
    "tulip_grad_norm": [P, L, P, P, F, P, P],
    "tulip_kitti_range_map": [P, L, I, I, F, F, F, F, F, P, P, P],
    "tulip_range_prep": [P, I, L, L, L, L, P, P, I, I, I, I, I, I, I, F, I, F, F, I, I, P],
    "tulip_mc_aggregate": [P, I, L, F, P, P],
    "tulip_eval_postprocess": [P, P, P, P, P, P, P, I, I, I, I, I, F, F, F, P],
    "tulip_range_to_xyz": [P, P, P, P, P, F, I, I, P, P],
    "tulip_range_to_xyz_durlar": [P, P, P, P, F, F, D, I, I, P, P],
    "tulip_voxel_metrics": [P, L, P, L, I, D, P, P, L, P, P, P],
    "tulip_chamfer_sq": [P, L, P, L, I, P, P, P, P, P],
    "tulip_merge_fwd_supported": [I, I, I, I],
    "tulip_merge_fwd": [P, P],
    "tulip_merge_bwd_supported": [I, I, I, I],
    "tulip_merge_bwd_partial_rows": [I, I, I, I],
    "tulip_merge_bwd": [P, P],
    "tulip_unmerge_skip_supported": [I, I, I, I],
    "tulip_unmerge_skip_fwd": [P, P],
    "tulip_skip_unmerge_bwd": [P, P],
    "tulip_stamp_realtime": [P, P],
    "tulip_abi_version": [],
    "tulip_build_arch": [],
    "tulip_dev_variants": [],
}

(EPI_BF16, EPI_GELU_DUAL, EPI_GELU_BWD, EPI_F32, EPI_RESID_F32, EPI_PIXSHUF2_F32, _EPI_RETIRED_6,
 EPI_SPLIT_F32, EPI_UNSHUF2_BF16) = range(9)

_lib = None


class TulipHipError(RuntimeError):
    pass


DEV_LIB_PATH = os.path.join(_PKG, "libtulip_hip_dev.so")    # -DTULIP_DEV_VARIANTS=1: + the forms no step launches (tulip_hip.h)
_libs = {}
# (the engine's environment switches that select a development-only kernel form select the development build with it)
_dev_depth = [1 if (os.environ.get("TULIP_HIP_DEV", "0") == "1" or os.environ.get("TULIP_SWIN96_RECOMPUTE", "0") != "0" or
                    os.environ.get("TULIP_SWINW_SPLIT_BWD", "0") != "0" or os.environ.get("TULIP_FC1_GRAD_WIDE", "1") == "0" or
                    os.environ.get("TULIP_FUSE_WIDE", "1") == "0") else 0]


def _open(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise TulipHipError(
            f"{path} not found: build it with `python -m tulip_amd.csrc.build` "
            "(tulip_amd has no CPU or PyTorch fallback for the hot path)")
    import torch  # noqa: F401  (loads libamdhip64 first)
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TulipHipError(f"symbol {name} missing from {path}") from e
        fn.argtypes = argtypes
        fn.restype = c_char_p if name == "tulip_build_arch" else c_int
    if lib.tulip_abi_version() != ABI_VERSION:
        raise TulipHipError(f"{path}: ABI version mismatch; rebuild")
    return lib


def load() -> ctypes.CDLL:
    """dlopen the in-tree library (import torch first so it binds to torch's HIP runtime): libtulip_hip.so, or -- inside
    `dev_library()` / under TULIP_HIP_DEV=1 -- libtulip_hip_dev.so, the same symbols plus the kernel forms no step launches."""
    global _lib
    path = DEV_LIB_PATH if _dev_depth[0] and not os.environ.get("TULIP_HIP_LIB") else LIB_PATH
    lib = _libs.get(path)
    if lib is None:
        lib = _libs[path] = _open(path)
    _lib = lib
    return lib


class dev_library:
    """with _lib.dev_library(): ...  -- the launches inside go to the development build (profiled twins, the recomputing C = 96
    backward, the backward's split form, the h-saving wide forms).  Launches captured into a graph keep their library."""
    def __enter__(self):
        _dev_depth[0] += 1
        return load()

    def __exit__(self, *exc):
        _dev_depth[0] -= 1
        return False


def dev_active() -> bool:
    """Would a launch issued now go to the development build?"""
    return bool(_dev_depth[0]) and not os.environ.get("TULIP_HIP_LIB")


def check(rc: int, what: str) -> None:
    if rc != 0:
        if rc == -1:
            raise TulipHipError(f"{what}: unsupported argument combination (TULIP_ERR_ARG)")
        if rc == -3:
            raise TulipHipError(f"{what}: this form exists only in the development build (TULIP_ERR_NOT_BUILT): "
                                "`with tulip_amd._lib.dev_library():` or TULIP_HIP_DEV=1")
        raise TulipHipError(f"{what}: HIP launch failed (hipError_t {-rc - 1000})")
