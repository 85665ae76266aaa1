"""CPU restatement of the reference's range-image readers and input transforms.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tests/golden/make_golden.py; the product path
(tulip_amd/data.py + csrc/prep.hip) never imports it.

Pinned against the reference: tests/golden/g8_transforms.npz holds outputs of the reference's own
classes (util/datasets.py ScaleTensor / FilterInvalidPixels / DownsampleTensor(/Width) / LogTransform /
RandomRollRangeMap composed as build_{kitti,durlar,carla}_upsampling_dataset compose them) and of its
npy_loader / rimg_loader on files written by make_golden.py; tests/test_data_cpu.py replays them.
torchvision's ToTensor is absent from the image: for a float32 (H, W) array it is `from_numpy(a)[None]`
(no value scaling for float input), which is what both the generator and this file use.
"""
from __future__ import annotations

import io
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class PrepSpec:
    """One dataset's transform chain (datasets.py:244-340)."""
    scale: float                    # ScaleTensor factor
    gate: Optional[Tuple[float, float]]   # FilterInvalidPixels(min, max) on the scaled value, or None
    name: str = ""


# datasets.py:249-250 (durlar), :285-286 (kitti: no gate), :321-322 (carla)
DATASETS = {
    "durlar": PrepSpec(1 / 120, (0.3 / 120, 1), "durlar"),
    "kitti": PrepSpec(1 / 80, None, "kitti"),
    "carla": PrepSpec(1 / 80, (2 / 80, 1), "carla"),
}


def npy_range(buf: bytes) -> np.ndarray:
    """npy_loader (datasets.py:175-179): (H, W, 2) [range m, intensity] -> channel 0 as float32."""
    a = np.load(io.BytesIO(buf))
    return a[..., 0].astype(np.float32)


def rimg_range(buf: bytes) -> np.ndarray:
    """rimg_loader (datasets.py:181-193): two native uints (s0, s1), then float16 payload stored as
    (s1, s0); transposed to (s0, s1) and flipped along both axes."""
    size = np.frombuffer(buf, dtype=np.uint, count=2)
    pay = np.frombuffer(buf, dtype=np.float16, offset=2 * np.dtype(np.uint).itemsize)
    img = pay.reshape(int(size[1]), int(size[0])).transpose()
    return np.flip(img).astype(np.float32)


def range_prep(raw: torch.Tensor, spec: PrepSpec, img_size_low_res, img_size_high_res, log_transform: bool,
               roll_shift: Optional[int] = None, row_phase: int = 0, col_phase: int = 0):
    """raw (B, H, W) float32 metres -> (low_res (B,1,h,w), high_res (B,1,H,W)) as the loaders hand them to the
    model: x*scale (datasets.py:138-142) -> keep min<=x<=max else 0 (:144-151) -> low-res rows phase::f
    (:117-125) and columns phase::fw when the widths differ (:127-135, :289-290) -> log1p (:68-70) ->
    torch.roll by the shared shift along W (:96-107)."""
    H, W = img_size_high_res
    h, w = img_size_low_res
    assert raw.shape[-2:] == (H, W)
    x = raw.float()[:, None] * spec.scale
    if spec.gate is not None:
        x = torch.where((x >= spec.gate[0]) & (x <= spec.gate[1]), x, 0)
    f = H // h
    lo = x[:, :, range(row_phase, H + row_phase, f), :]
    if W // w > 1:
        lo = lo[:, :, :, range(col_phase, W + col_phase, W // w)]
    hi = x
    if log_transform:
        lo, hi = torch.log1p(lo), torch.log1p(hi)
    if roll_shift is not None:
        lo, hi = torch.roll(lo, roll_shift, -1), torch.roll(hi, roll_shift, -1)
    return lo.contiguous(), hi.contiguous()


def synthetic_raw(B: int, H: int, W: int, seed: int, max_m: float = 130.0) -> torch.Tensor:
    """Seeded raw range images in metres with the cases the gates care about: zeros (no return), values
    below the minimum range, values above the maximum, and values sitting exactly on the thresholds."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, H, W, generator=g) * max_m
    u = torch.rand(B, H, W, generator=g)
    x = torch.where(u < 0.15, torch.zeros(()), x)
    x = torch.where((u >= 0.15) & (u < 0.25), x * 0.02, x)         # 0 .. 2.6 m: around both minimum ranges
    edge = torch.tensor([0.3, 2.0, 80.0, 120.0, 0.29999998, 1.9999999, 80.00001, 120.00001])
    idx = torch.randint(0, H * W, (B, 64), generator=g)
    flat = x.reshape(B, -1)
    for b in range(B):
        flat[b, idx[b]] = edge[torch.arange(64) % edge.numel()]
    return flat.reshape(B, H, W).contiguous()


# ---------------------------------------------------------------------------------------------------
# KITTI point cloud -> range image (the producer of the .npy files): kitti_utils/sample_kitti_dataset.py:24-66,139-160
# ---------------------------------------------------------------------------------------------------
KITTI_PROJECTION = dict(image_rows=64, image_cols=1024, ang_start_y=24.8, ang_res_y=26.8 / 63, ang_res_x=360 / 1024,
                        max_range=120, min_range=0)


def kitti_pixel_of_points(points: np.ndarray, image_rows, image_cols, ang_start_y, ang_res_y, ang_res_x, **_):
    """(rowId, colId, range) of every point, in numpy float32 arithmetic like create_range_map :27-44:
    row = rint((atan2(z, sqrt(x^2+y^2))*180/pi + ang_start_y) / ang_res_y);
    col = -trunc((atan2(x, y)*180/pi - 90) / ang_res_x) + cols/2, minus cols where >= cols."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    assert x.dtype == np.float32
    vert = np.arctan2(z, np.sqrt(x * x + y * y)) * 180.0 / np.pi
    row = np.int_(np.round(( vert + ang_start_y) / ang_res_y))
    hor = np.arctan2(x, y) * 180.0 / np.pi
    col = -np.int_((hor - 90.0) / ang_res_x) + image_cols / 2
    col[col >= image_cols] -= image_cols
    col = col.astype(np.int64)
    rng = np.sqrt(x * x + y * y + z * z)
    return row, col, rng


def kitti_range_map(points: np.ndarray, image_rows=64, image_cols=1024, ang_start_y=24.8, ang_res_y=26.8 / 63,
                    ang_res_x=360 / 1024, max_range=120, min_range=0) -> np.ndarray:
    """create_range_map (:24-66): (N,4) [x,y,z,intensity] float32 -> (rows, cols, 2) [range, intensity] float32.
    Ranges outside [min_range, max_range] become 0 (:41-42) -- the intensities of those points are KEPT, because the
    reference tests the already zeroed range (:45-46); points outside the image are dropped (:49); when several points
    hit a pixel the LAST one in point order stays (fancy-index assignment :56-57)."""
    row, col, rng = kitti_pixel_of_points(points, image_rows, image_cols, ang_start_y, ang_res_y, ang_res_x)
    rng = rng.copy()
    rng[rng > max_range] = 0
    rng[rng < min_range] = 0
    inten = points[:, 3]
    ok = (row >= 0) & (row < image_rows) & (col >= 0) & (col < image_cols)
    out = np.zeros((image_rows, image_cols, 2), dtype=np.float32)
    # last point in order wins: iterate winners explicitly instead of relying on assignment order
    pix = row[ok] * image_cols + col[ok]
    idx = np.nonzero(ok)[0]
    winner = np.full(image_rows * image_cols, -1, dtype=np.int64)
    np.maximum.at(winner, pix, idx)
    hit = winner >= 0
    flat = out.reshape(-1, 2)
    flat[hit, 0] = rng[winner[hit]]
    flat[hit, 1] = inten[winner[hit]]
    return out


def synthetic_kitti_scan(n: int, seed: int) -> np.ndarray:
    """Seeded Velodyne-like scan: 64 beams between -24.8 and +2 degrees with jitter, full azimuth, ranges 1..130 m
    (some beyond max_range), a few points above/below the vertical field of view, duplicates on the same pixel."""
    g = np.random.default_rng(seed)
    el = np.deg2rad(g.choice(np.linspace(-24.8, 2.0, 64), n) + g.normal(0, 0.08, n))
    el[: n // 50] = np.deg2rad(g.uniform(-30, 6, n // 50))
    az = g.uniform(-np.pi, np.pi, n)
    r = g.uniform(1.0, 130.0, n)
    pts = np.stack([r * np.cos(el) * np.sin(az), r * np.cos(el) * np.cos(az), r * np.sin(el), g.uniform(0, 1, n)], -1)
    return pts.astype(np.float32)
