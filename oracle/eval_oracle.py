"""CPU restatement of the reference's evaluation post-processing and metrics.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tests/golden/make_golden.py; the product path
(tulip_amd/evaluation.py + csrc/evalpost.hip) never imports it.

Pinning.  tests/golden/g10_eval.npz is produced by running the REFERENCE's own `evaluate` and `MCdrop`
(engine_upsampling.py:126-355, :361-608) and its util/evaluation.py functions in the build container on
seeded synthetic images with a fixed stand-in model (make_golden.py: golden_eval), and this file is asserted
against it there and in tests/test_eval_cpu.py.  One piece is NOT pinned: the Chamfer distance.  The
reference calls the third-party CUDA extension `chamfer_distance` (ChamferDistance()(a, b) -> dist1, dist2,
idx1, idx2; evaluation.py:4,125-135), which is neither vendored in /root/reference nor installed here and
has no pinned version in the reference's requirements.  Its published algorithm (the widely used
chrdiller/pyTorchChamferDistance kernel) returns, for every point, the SQUARED Euclidean distance to its
nearest neighbour in the other cloud, computed in float32; `chamfer_sq` restates that and the reference's
reduction mean(dist1)+mean(dist2).  Chamfer parity is therefore "unpinned" (the fixture's chamfer values come
from this restatement, plugged into the reference's loop in place of the extension).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

# ---------------------------------------------------------------------------------------------------
# gates and per-dataset constants
# ---------------------------------------------------------------------------------------------------
MAX_RANGE = {"kitti": 80, "carla": 80, "durlar": 120}     # img_to_pcd_* arguments, engine:222-223,232-233,250-251


def pred_gate(dataset: str, mc_drop: bool) -> Tuple[float, float]:
    """Range gate applied to the prediction only (engine:183-190; MCdrop :436-443 uses 0 for KITTI)."""
    if dataset == "durlar":
        return 0.3 / 120, 1
    if dataset == "kitti" and mc_drop:
        return 0, 1
    return 2 / 80, 1


def mc_aggregate(preds: torch.Tensor, noise_threshold: float) -> torch.Tensor:
    """engine:421-426: mean over the passes, unbiased std, prediction zeroed where std > threshold*mean."""
    pred = torch.mean(preds, dim=0, keepdim=True)
    std = torch.std(preds, dim=0, keepdim=True)
    pred = pred.clone()
    pred[std > noise_threshold * pred] = 0
    return pred


def postprocess(pred: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, dataset: str, log_transform: bool,
                mc_drop: bool = False, keep_close_scan: bool = False):
    """engine:176-251 for ONE image (B=1 tensors (1,1,H,W) / (1,1,h,w)):
    expm1 of all three when log_transform (:178-181); prediction gate (:183-190); mae = mean|pred-hi| BEFORE the
    low-res rows are restored (:192-193); mae_low_res over rows 0::f (:224-228; CARLA only when the widths agree,
    :208-216); rows 0::f of the prediction overwritten with the input (:217,230,245); keep_close_scan zeroes
    both images above 0.25 (durlar in `evaluate` :247-249, kitti in `MCdrop` :487-489).
    Returns (mae, mae_low_res, pred_img (H,W) float32, hi_img (H,W) float32)."""
    if log_transform:
        pred, hi, lo = torch.expm1(pred), torch.expm1(hi), torch.expm1(lo)
    g0, g1 = pred_gate(dataset, mc_drop)
    pred = torch.where((pred >= g0) & (pred <= g1), pred, 0)
    mae = (pred - hi).abs().mean().item()
    H, h = hi.shape[2], lo.shape[2]
    p = pred[0, 0].numpy().copy()
    t = hi[0, 0].numpy().copy()
    l = lo[0, 0].numpy()
    rows = range(0, H, H // h)
    if dataset == "carla" and lo.shape[3] != hi.shape[3]:
        mae_low = 0
    else:
        mae_low = np.abs(p[rows, :] - l).mean()
        p[rows, :] = l
    if keep_close_scan and ((dataset == "durlar" and not mc_drop) or (dataset == "kitti" and mc_drop)):
        p[p > 0.25] = 0
        t[t > 0.25] = 0
    return mae, float(mae_low), p, t


# ---------------------------------------------------------------------------------------------------
# range image -> point cloud (util/evaluation.py)
# ---------------------------------------------------------------------------------------------------
def kitti_tables(rows: int = 64, cols: int = 1024):
    """Per-row / per-column trigonometry of img_to_pcd_kitti (evaluation.py:52-87), float32 like numpy
    computes it there: vertical = float32(row*26.8/(rows-1)) - 24.8, horizontal = -float32(col+1-cols/2)*360/cols+90,
    both /180*pi.  Returns (sin_h, cos_h) over columns and (sin_v, cos_v) over rows."""
    ang_res_y, ang_res_x = 26.8 / (rows - 1), 360 / cols
    v = np.float32(np.arange(rows, dtype=np.float64) * ang_res_y) - 24.8
    h = -np.float32(np.arange(cols, dtype=np.float64) + 1 - (cols / 2)) * ang_res_x + 90.0
    v = v / 180.0 * np.pi
    h = h / 180.0 * np.pi
    assert v.dtype == np.float32 and h.dtype == np.float32
    return np.sin(h), np.cos(h), np.sin(v), np.cos(v)


def carla_tables(rows: int, cols: int):
    """img_to_pcd_carla (evaluation.py:90-116): elevation linspace(-15,15,rows), azimuth linspace(-180,180,cols,
    endpoint=False), cast to float32 before deg2rad."""
    v = np.deg2rad(np.linspace(start=-15, stop=15, num=rows).astype(np.float32))
    h = np.deg2rad(np.linspace(start=-180, stop=180, num=cols, endpoint=False).astype(np.float32))
    assert v.dtype == np.float32 and h.dtype == np.float32
    return np.sin(h), np.cos(h), np.sin(v), np.cos(v)


def spherical_pcd(img: np.ndarray, tables, maximum_range) -> np.ndarray:
    """x = sin(h)cos(v)r, y = cos(h)cos(v)r, z = sin(v)r with r = img*maximum_range, all float32, points in
    row-major pixel order (evaluation.py:76-85, :108-114)."""
    sh, ch, sv, cv = tables
    r = img.astype(np.float32) * maximum_range
    x = sh[None, :] * cv[:, None] * r
    y = ch[None, :] * cv[:, None] * r
    z = np.broadcast_to(sv[:, None], r.shape) * r
    return np.stack((x, y, z), axis=-1).reshape(-1, 3)


# DurLAR (Ouster OS1-128) beam geometry tables, evaluation.py:7-19
DURLAR_OFFSET_LUT = np.tile(np.array([48, 32, 16, 0]), 32)
DURLAR_ORIGIN_OFFSET = 0.015806
DURLAR_Z_OFFSET = 0.03618
DURLAR_ANGLE_OFF = math.pi * 4.2285 / 180.


def durlar_elevation_lut(ref_lut: Optional[np.ndarray] = None) -> np.ndarray:
    """The 128 beam elevations (degrees) are sensor calibration DATA (evaluation.py:11): loaded from the golden
    fixture by the tests, handed in here."""
    if ref_lut is None:
        raise ValueError("pass the elevation table (tests/golden/durlar_beam_tables.npz)")
    return np.asarray(ref_lut, dtype=np.float64)


def durlar_pcd(img: np.ndarray, elevation_lut: np.ndarray, maximum_range=120) -> np.ndarray:
    """img_to_pcd_durlar (evaluation.py:21-50): pixel (u=col, v=row) has encoder angle 2pi - u*2pi/cols and beam
    elevation lut[v]; r-origin_offset is taken in float32 (float32 image * int, minus a Python float), everything
    after in float64; the point lands at index v*cols + (u + cols - offset_lut[v]) % cols."""
    rows, cols = img.shape[:2]
    u = np.arange(cols)
    enc = 2.0 * math.pi - (((cols + u) % cols) * (math.pi * 2.0 / cols))
    el = math.pi * elevation_lut / 180.
    rng = (img.astype(np.float32) * maximum_range) - DURLAR_ORIGIN_OFFSET          # float32 (rows, cols)
    assert rng.dtype == np.float32
    x = rng * np.cos(enc + DURLAR_ANGLE_OFF)[None, :] * np.cos(el)[:, None] + (DURLAR_ORIGIN_OFFSET * np.cos(enc))[None, :]
    y = rng * np.sin(enc + DURLAR_ANGLE_OFF)[None, :] * np.cos(el)[:, None] + (DURLAR_ORIGIN_OFFSET * np.sin(enc))[None, :]
    z = rng * np.sin(el)[:, None]
    pts = np.stack((-x, -y, z + DURLAR_Z_OFFSET), axis=-1)                           # (rows, cols, 3) float64
    out = np.zeros((rows * cols, 3))
    v = np.arange(rows)
    dst = v[:, None] * cols + (u[None, :] + cols - DURLAR_OFFSET_LUT[:rows][:, None]) % cols
    out[dst.reshape(-1)] = pts.reshape(-1, 3)
    return out


# ---------------------------------------------------------------------------------------------------
# metrics
# ---------------------------------------------------------------------------------------------------
def chamfer_sq(points1: np.ndarray, points2: np.ndarray, chunk: int = 2048, dtype=np.float32) -> float:
    """chamfer_distance (evaluation.py:125-135) with the extension restated: squared nearest-neighbour
    distances in `dtype`, direct (a-b)^2 form, mean(dist1) + mean(dist2).  UNPINNED (see module header)."""
    a = torch.from_numpy(np.ascontiguousarray(points1)).to(torch.float32 if dtype == np.float32 else torch.float64)
    b = torch.from_numpy(np.ascontiguousarray(points2)).to(a.dtype)

    def nn_sq(p, q):
        out = torch.empty(p.shape[0], dtype=p.dtype)
        for s in range(0, p.shape[0], chunk):
            d = p[s:s + chunk, None, :] - q[None, :, :]
            out[s:s + chunk] = (d * d).sum(-1).min(dim=1).values
        return out

    return (nn_sq(a, b).double().mean() + nn_sq(b, a).double().mean()).item()


def voxel_metrics(pcd_pred: np.ndarray, pcd_gt: np.ndarray, grid_size: float):
    """engine:254-271 + evaluation.py:148-175 without materialising the dense boolean grids: a voxel is the integer
    triple ((p - min_coord)/grid_size).astype(int) with min_coord over BOTH clouds, in the clouds' own dtype
    (float32 for KITTI/CARLA, float64 for DurLAR); IoU = |P&G|/|P|G|, precision = |P&G|/|P|, recall = |P&G|/|G|,
    f1 = 2pr/(p+r)."""
    both = np.vstack((pcd_pred, pcd_gt))
    mn, mx = np.min(both, axis=0), np.max(both, axis=0)
    dims = ((mx - mn) / grid_size).astype(int) + 1

    def keys(p):
        idx = ((p - mn) / grid_size).astype(int)
        return np.unique((idx[:, 0].astype(np.int64) * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2])

    kp, kg = keys(pcd_pred), keys(pcd_gt)
    inter = np.intersect1d(kp, kg, assume_unique=True).size
    union = kp.size + kg.size - inter
    iou = inter / union
    precision = inter / kp.size
    recall = inter / kg.size
    f1 = 2 * (precision * recall) / (precision + recall)
    return iou, precision, recall, f1, dims


def synthetic_eval_case(dataset: str, H: int, W: int, h: int, w: int, seed: int, log_transform: bool):
    """Seeded (pred, hi, lo) in the model's value space: a smooth scene, a prediction = target + noise with
    negatives, values beyond the gates and exact hits on the gate thresholds."""
    g = torch.Generator().manual_seed(seed)
    jj = torch.arange(W)[None, :] / W
    ii = torch.arange(H)[:, None] / H
    scene = 0.08 + 0.5 * (0.5 + 0.5 * torch.sin(6.28 * (3 * jj + ii))) * (0.3 + 0.7 * ii)
    scene = scene + 0.02 * torch.rand(H, W, generator=g)
    scene = torch.where(torch.rand(H, W, generator=g) < 0.1, torch.zeros(()), scene)       # no-return pixels
    pred = scene + 0.01 * torch.randn(H, W, generator=g)
    u = torch.rand(H, W, generator=g)
    pred = torch.where(u < 0.03, -pred, pred)
    pred = torch.where((u >= 0.03) & (u < 0.05), pred + 1.0, pred)
    pred = torch.where((u >= 0.05) & (u < 0.10), pred * 0.05, pred)
    hi = scene[None, None].float()
    pred = pred[None, None].float()
    lo = hi[:, :, 0::H // h, 0::max(1, W // w)].contiguous()
    if log_transform:
        hi, lo, pred = torch.log1p(hi), torch.log1p(lo), torch.log1p(pred.clamp(min=-0.5))
    return pred.contiguous(), hi.contiguous(), lo.contiguous()
