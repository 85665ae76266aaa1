"""The step after the hot path: the reference's evaluation loops
(engine_upsampling.py `evaluate` :126-355 and `MCdrop` :361-608) with every per-image computation on the GPU.

The reference pulls each prediction to the host and runs numpy there (gate, MAE, row restore, range image ->
point cloud, two dense boolean voxel grids of ~1 GB each, a CUDA Chamfer extension in between).  Here the
prediction never leaves HBM: csrc/evalpost.hip does the post-processing, the projection, a bitmap voxel set with
O(points) work, and a brute-force nearest-neighbour Chamfer; per-image results land in a device table that is
read back once at the end of the loop.  `evaluate` / `MCdrop` keep the reference's signatures, its results
files and its quirks (MCdrop's KITTI gate starts at 0, `keep_close_scan` applies to DurLAR in `evaluate` and to
KITTI in `MCdrop`, MCdrop's results file only carries mae and chamfer_dist).  Plot/tensorboard/.ply side
outputs are not produced.  No CPU fallback: the HIP library is required.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import numpy as np
import torch

from . import ops

# img_to_pcd_* maximum_range arguments (engine:222-223,232-233,250-251)
MAX_RANGE = {"kitti": 80, "carla": 80, "durlar": 120}
# largest normalised range a cloud can hold, for sizing the voxel bitmaps: KITTI inputs are not gated
# (datasets.py:285-286; ranges up to 120 m = 1.5), CARLA/DurLAR inputs and every prediction are gated to <= 1
_VALUE_BOUND = {"kitti": 1.6, "carla": 1.05, "durlar": 1.05}

# Ouster OS1-128 beam calibration used for DurLAR (sensor data, util/evaluation.py:7-19)
DURLAR_ELEVATION_DEG = (
    21.42, 21.12, 20.81, 20.5, 20.2, 19.9, 19.58, 19.26, 18.95, 18.65, 18.33, 18.02, 17.68, 17.37, 17.05, 16.73,
    16.4, 16.08, 15.76, 15.43, 15.1, 14.77, 14.45, 14.11, 13.78, 13.45, 13.13, 12.79, 12.44, 12.12, 11.77, 11.45,
    11.1, 10.77, 10.43, 10.1, 9.74, 9.4, 9.06, 8.72, 8.36, 8.02, 7.68, 7.34, 6.98, 6.63, 6.29, 5.95, 5.6, 5.25, 4.9,
    4.55, 4.19, 3.85, 3.49, 3.15, 2.79, 2.44, 2.1, 1.75, 1.38, 1.03, 0.68, 0.33, -0.03, -0.38, -0.73, -1.07, -1.45,
    -1.8, -2.14, -2.49, -2.85, -3.19, -3.54, -3.88, -4.26, -4.6, -4.95, -5.29, -5.66, -6.01, -6.34, -6.69, -7.05,
    -7.39, -7.73, -8.08, -8.44, -8.78, -9.12, -9.45, -9.82, -10.16, -10.5, -10.82, -11.19, -11.52, -11.85, -12.18,
    -12.54, -12.87, -13.2, -13.52, -13.88, -14.21, -14.53, -14.85, -15.2, -15.53, -15.84, -16.16, -16.5, -16.83,
    -17.14, -17.45, -17.8, -18.11, -18.42, -18.72, -19.06, -19.37, -19.68, -19.97, -20.31, -20.61, -20.92, -21.22)
DURLAR_PIXEL_OFFSET = (48, 32, 16, 0) * 32
DURLAR_ORIGIN_OFFSET = 0.015806
DURLAR_Z_OFFSET = 0.03618
DURLAR_ANGLE_OFF = math.pi * 4.2285 / 180.

RESULT_COLUMNS = ("mae", "mae_low_res", "chamfer_dist", "iou", "precision", "recall", "f1", "status")


def pred_gate(dataset_select: str, mc_drop: bool):
    """engine:183-190 (`evaluate`) / :436-443 (`MCdrop`: KITTI keeps 0 <= p <= 1)."""
    if dataset_select == "durlar":
        return 0.3 / 120, 1.0
    if dataset_select == "kitti" and mc_drop:
        return 0.0, 1.0
    if dataset_select in ("kitti", "carla"):
        return 2 / 80, 1.0
    raise NotImplementedError(f"Cannot find the dataset: {dataset_select}")       # engine:252-253


class RangeEvaluator:
    """Per-image metrics of one dataset configuration.  `__call__(pred, lo, hi)` takes (1,1,H,W)/(1,1,h,w) device
    tensors in the model's value space and returns a device float64 tensor ordered as RESULT_COLUMNS; nothing
    synchronises with the host."""

    def __init__(self, dataset_select: str, img_size_low_res, img_size_high_res, log_transform: bool,
                 grid_size: float = 0.1, keep_close_scan: bool = False, mc_drop: bool = False, device="cuda"):
        self.ds, self.mc = dataset_select, bool(mc_drop)
        self.gate = pred_gate(dataset_select, self.mc)
        self.h, self.w = (int(v) for v in img_size_low_res)
        self.H, self.W = (int(v) for v in img_size_high_res)
        self.log_transform, self.grid_size = bool(log_transform), float(grid_size)
        # keep_close_scan acts on DurLAR in evaluate (:247-249) and on KITTI in MCdrop (:487-489)
        self.keep_close = 0.25 if keep_close_scan and ((dataset_select == "durlar" and not self.mc) or
                                                       (dataset_select == "kitti" and self.mc)) else 0.0
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("tulip_amd.evaluation runs on the GPU only (HIP kernels, no CPU fallback)")
        H, W, dev = self.H, self.W, self.device
        R = MAX_RANGE[dataset_select]
        if dataset_select == "kitti":
            if (H, W) != (64, 1024):          # img_to_pcd_kitti hard-codes the sensor grid (evaluation.py:53-57)
                raise ValueError(f"cannot reshape array of size {H * W} into shape ({64 * 1024},)")
            self.f64 = False
            ang_res_y, ang_res_x = 26.8 / (H - 1), 360 / W
            v = np.float32(np.arange(H, dtype=np.float64) * ang_res_y) - 24.8           # :69
            hz = -np.float32(np.arange(W, dtype=np.float64) + 1 - (W / 2)) * ang_res_x + 90.0   # :70
            v, hz = v / 180.0 * np.pi, hz / 180.0 * np.pi                              # :72-73
            z_lo, z_hi = math.sin(math.radians(-24.8)), math.sin(math.radians(2.0))
        elif dataset_select == "carla":
            self.f64 = False
            v = np.deg2rad(np.linspace(start=-15, stop=15, num=H).astype(np.float32))           # :94,:105-106
            hz = np.deg2rad(np.linspace(start=-180, stop=180, num=W, endpoint=False).astype(np.float32))
            z_lo, z_hi = math.sin(math.radians(-15)), math.sin(math.radians(15))
        else:
            if H > len(DURLAR_ELEVATION_DEG):
                raise IndexError(f"index {H - 1} is out of bounds for axis 0 with size {len(DURLAR_ELEVATION_DEG)}")
            self.f64 = True
            u = np.arange(W)
            enc = 2.0 * math.pi - (((W + u) % W) * (math.pi * 2.0 / W))                 # :28-30
            el = math.pi * np.array(DURLAR_ELEVATION_DEG[:H], dtype=np.float64) / 180.  # :32
            colt = np.concatenate([np.cos(enc + DURLAR_ANGLE_OFF), np.sin(enc + DURLAR_ANGLE_OFF),
                                   DURLAR_ORIGIN_OFFSET * np.cos(enc), DURLAR_ORIGIN_OFFSET * np.sin(enc)])
            rowt = np.concatenate([np.cos(el), np.sin(el)])
            self.colt = torch.from_numpy(colt).to(dev)
            self.rowt = torch.from_numpy(rowt).to(dev)
            self.row_offset = torch.tensor(DURLAR_PIXEL_OFFSET[:H], dtype=torch.int32, device=dev)
            z_lo, z_hi = math.sin(el.min()), math.sin(el.max())
        if not self.f64:
            assert v.dtype == np.float32 and hz.dtype == np.float32
            self.tables = [torch.from_numpy(np.ascontiguousarray(t)).to(dev)
                           for t in (np.sin(hz), np.cos(hz), np.sin(v), np.cos(v))]
        # voxel bitmaps: an upper bound of the grid from the sensor geometry (the reference sizes it from the data)
        Rb = R * _VALUE_BOUND[dataset_select] + 0.1
        dims = [int(2 * Rb / self.grid_size) + 2] * 2 + [int(Rb * (max(z_hi, 0) - min(z_lo, 0)) / self.grid_size) + 3]
        self.bitmap_words = (dims[0] * dims[1] * dims[2] + 31) // 32
        self.bm_pred = torch.zeros(self.bitmap_words, dtype=torch.int32, device=dev)
        self.bm_gt = torch.zeros(self.bitmap_words, dtype=torch.int32, device=dev)
        n = H * W
        pt = torch.float64 if self.f64 else torch.float32
        self.pcd_pred = torch.empty(n, 3, dtype=pt, device=dev)
        self.pcd_gt = torch.empty(n, 3, dtype=pt, device=dev)
        self.pred_img = torch.empty(H, W, dtype=torch.float32, device=dev)
        self.hi_img = torch.empty(H, W, dtype=torch.float32, device=dev)
        self.dist_a = torch.empty(n, dtype=torch.float32, device=dev)
        self.dist_b = torch.empty(n, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(6 * 1024 + 16, dtype=torch.float64, device=dev)
        self.mae = torch.zeros(2, dtype=torch.float32, device=dev)
        self.vox = torch.zeros(8, dtype=torch.float64, device=dev)
        self.cd = torch.zeros(1, dtype=torch.float64, device=dev)

    def point_clouds(self, pred, lo, hi):
        """post-processing + projection only: fills self.pred_img/hi_img/pcd_pred/pcd_gt/mae"""
        H, W = self.H, self.W
        for t, shp in ((pred, (H, W)), (hi, (H, W)), (lo, (self.h, self.w))):
            if t.dtype != torch.float32 or not t.is_cuda or t.numel() != shp[0] * shp[1]:
                raise ValueError(f"expected a float32 device tensor with {shp[0]}x{shp[1]} pixels (one image)")
        pred, lo, hi = pred.contiguous(), lo.contiguous(), hi.contiguous()
        ops.eval_postprocess(pred, hi, lo, self.pred_img, self.hi_img, self.scratch, self.mae, H, W, self.h, self.w,
                             self.log_transform, self.gate[0], self.gate[1], self.keep_close)
        R = MAX_RANGE[self.ds]
        for img, pcd in ((self.pred_img, self.pcd_pred), (self.hi_img, self.pcd_gt)):
            if self.f64:
                ops.range_to_xyz_durlar(img, self.colt, self.rowt, self.row_offset, R, DURLAR_ORIGIN_OFFSET,
                                        DURLAR_Z_OFFSET, H, W, pcd)
            else:
                ops.range_to_xyz(img, *self.tables, R, H, W, pcd)

    def __call__(self, pred: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
        self.point_clouds(pred, lo, hi)
        n = self.H * self.W
        ops.voxel_metrics(self.pcd_pred, n, self.pcd_gt, n, self.f64, self.grid_size, self.bm_pred, self.bm_gt,
                          self.bitmap_words, self.scratch, self.vox)
        ops.chamfer_sq(self.pcd_gt, n, self.pcd_pred, n, self.f64, self.dist_a, self.dist_b, self.scratch, self.cd)
        return torch.cat([self.mae.double(), self.cd, self.vox[:4], self.vox[7:8]])


def mc_aggregate(preds: torch.Tensor, noise_threshold: float) -> torch.Tensor:
    """(passes,1,H,W) -> (1,1,H,W): engine:421-426."""
    if preds.dtype != torch.float32 or not preds.is_cuda:
        raise ValueError("preds must be a float32 device tensor")
    preds = preds.contiguous()
    out = torch.empty((1,) + tuple(preds.shape[1:]), dtype=torch.float32, device=preds.device)
    ops.mc_aggregate(preds, preds.shape[0], out.numel(), noise_threshold, out)
    return out


def _samples(batch):
    lo, hi = batch
    if isinstance(lo, dict):
        lo, hi = lo["sample"], hi["sample"]
    return lo, hi


def _finish(table, args, log_writer, file_name: str, keys):
    """results file (engine:330-334 / :590-594) + the averages the loops log (engine:340-347)."""
    res = torch.stack(table).cpu().numpy() if table else np.zeros((0, len(RESULT_COLUMNS)))
    col = {k: res[:, i] for i, k in enumerate(RESULT_COLUMNS)}
    evaluation_metrics = {k: (col[k].tolist() if k in keys else [])
                          for k in ("mae", "chamfer_dist", "iou", "precision", "recall", "f1")}
    out_dir = getattr(args, "output_dir", None)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, file_name)
        with open(path, "w") as file:
            json.dump(evaluation_metrics, file)
        print(f"Dictionary saved to {path}")
    n = max(len(table), 1)
    avg = {"loss": float(col["mae"].sum() / n), "iou": float(col["iou"].sum() / n),
           "cd": float(col["chamfer_dist"].sum() / n), "f1": float(col["f1"].sum() / n),
           "precision": float(col["precision"].sum() / n), "recall": float(col["recall"].sum() / n),
           "mae_low_res": float(col["mae_low_res"].sum() / n)}
    if log_writer is not None:
        for k in ("iou", "cd", "loss", "f1", "precision", "recall"):
            log_writer.add_scalar(f"Metrics/test_average_{k}", avg[k], 0)
    avg["per_image"] = evaluation_metrics
    return avg


@torch.no_grad()
def evaluate(data_loader, model, device, log_writer=None, args=None):
    """engine_upsampling.py:126-355.  Returns the averages it logs plus the per-image lists it saves."""
    ev = RangeEvaluator(args.dataset_select, args.img_size_low_res, args.img_size_high_res, args.log_transform,
                        args.grid_size, getattr(args, "keep_close_scan", False), False, device)
    model.eval()
    table = []
    for batch in data_loader:
        lo, hi = _samples(batch)
        lo = lo.to(device, non_blocking=True).float()
        hi = hi.to(device, non_blocking=True).float()
        pred, _, _ = model(lo, hi, eval=True)
        for b in range(pred.shape[0]):            # the reference runs with batch size 1 (:160-161)
            table.append(ev(pred[b:b + 1], lo[b:b + 1], hi[b:b + 1]))
    return _finish(table, args, log_writer, "results.txt", ("mae", "chamfer_dist", "iou", "precision", "recall", "f1"))


@torch.no_grad()
def MCdrop(data_loader, model, device, log_writer=None, args=None):
    """engine_upsampling.py:361-608: `num_mcdropout_iterations` forwards of the same input in tiles of 8, mean /
    std / threshold, then the evaluate pipeline with MCdrop's own gate; results_mcdrop.txt holds mae and
    chamfer_dist only (:526-534).  Every reference configuration has dropout p=0, so the passes are identical
    unless the model says otherwise."""
    iteration, iteration_batch = args.num_mcdropout_iterations, 8
    assert iteration > iteration_batch                                                # :369
    ev = RangeEvaluator(args.dataset_select, args.img_size_low_res, args.img_size_high_res, args.log_transform,
                        args.grid_size, getattr(args, "keep_close_scan", False), True, device)
    model.eval()
    table = []
    for batch in data_loader:
        lo, hi = _samples(batch)
        lo = lo.to(device, non_blocking=True).float()
        hi = hi.to(device, non_blocking=True).float()
        preds = torch.empty((iteration,) + tuple(hi.shape[1:]), dtype=torch.float32, device=device)
        for i in range(int(np.ceil(iteration / iteration_batch))):
            nb = iteration_batch if (iteration - i * iteration_batch) > iteration_batch else (iteration - i * iteration_batch)
            preds[i * iteration_batch:i * iteration_batch + nb] = model(lo.tile(nb, 1, 1, 1), hi, mc_drop=True)
        pred = mc_aggregate(preds, args.noise_threshold)
        table.append(ev(pred, lo, hi))
    return _finish(table, args, log_writer, "results_mcdrop.txt", ("mae", "chamfer_dist"))
