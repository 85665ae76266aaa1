"""The step before the hot path: the reference's range-image files and input transforms
(util/datasets.py) with the arithmetic moved onto the GPU.

The reference decodes a file on a CPU worker (npy_loader/rimg_loader, datasets.py:175-193), runs a chain
of per-image torch ops there (ToTensor -> ScaleTensor -> FilterInvalidPixels -> DownsampleTensor(/Width) ->
LogTransform -> RandomRollRangeMap, :244-340) once for the low-res and once for the high-res copy, and
collates.  Here the host only reads file payloads into one pinned staging buffer; ONE kernel
(csrc/prep.hip, tulip_range_prep) reads the raw payload in place (interleaved .npy channel 0, transposed
+ flipped float16 .rimg) and writes the model's (low_res, high_res) batch.  No CPU fallback: without the
HIP library `RangePrep.__call__` raises.
"""
from __future__ import annotations

import os
import struct
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

NPY_EXTENSIONS = (".npy", ".rimg", ".bin")        # datasets.py:39

# dataset_select -> (ScaleTensor factor, FilterInvalidPixels (min, max) or None): datasets.py:249-250,285-286,321-322
DATASET_PREP = {
    "durlar": (1 / 120, (0.3 / 120, 1.0)),
    "kitti": (1 / 80, None),
    "carla": (1 / 80, (2 / 80, 1.0)),
}


class RangePrep:
    """Callable with the semantics of the reference's (transform_low_res, transform_high_res) pair for one
    dataset_select; constructor arguments mirror the `args` fields build_*_upsampling_dataset read."""

    def __init__(self, dataset_select: str, img_size_low_res: Sequence[int], img_size_high_res: Sequence[int],
                 log_transform: bool = False, roll_shift: Optional[int] = None, row_phase: int = 0,
                 col_phase: int = 0):
        if dataset_select not in DATASET_PREP:
            raise KeyError(dataset_select)          # generate_dataset: dataset_list[args.dataset_select] (datasets.py:49-51)
        self.scale, self.gate = DATASET_PREP[dataset_select]
        self.h, self.w = (int(v) for v in img_size_low_res)
        self.H, self.W = (int(v) for v in img_size_high_res)
        self.f = self.H // self.h                   # downsample_factor = output_size[0] // input_size[0]
        self.fw = max(1, self.W // self.w)          # DownsampleTensorWidth only when the ratio is > 1 (:289-290)
        if self.h * self.f != self.H or self.w * self.fw != self.W:
            raise ValueError("high-res size must be an integer multiple of the low-res size")
        self.log_transform = bool(log_transform)
        # RandomRollRangeMap draws ONE shift when the dataset is built (:100-104) and keeps it for the whole run
        self.roll_shift = 0 if roll_shift is None else int(roll_shift)
        self.row_phase, self.col_phase = row_phase, col_phase

    # -- raw layouts ---------------------------------------------------------------------------------
    def _run(self, raw: torch.Tensor, dtype_code: int, sb: int, si: int, sj: int, base: int, B: int,
             want_hi: bool, want_lo: bool):
        if not raw.is_cuda:
            raise RuntimeError("tulip_amd.data.RangePrep runs on the GPU only (HIP kernel, no CPU fallback)")
        dev = raw.device
        hi = torch.empty(B, 1, self.H, self.W, dtype=torch.float32, device=dev) if want_hi else None
        lo = torch.empty(B, 1, self.h, self.w, dtype=torch.float32, device=dev) if want_lo else None
        gmin, gmax = self.gate if self.gate is not None else (0.0, 0.0)
        ops.range_prep(raw, dtype_code, sb, si, sj, base, hi, lo, B, self.H, self.W, self.f, self.fw, self.row_phase,
                       self.col_phase, self.scale, self.gate is not None, gmin, gmax, self.log_transform,
                       self.roll_shift)
        return lo, hi

    def __call__(self, raw: torch.Tensor, want_hi: bool = True, want_lo: bool = True):
        """raw: (B,H,W) float32 metres, or the (B,H,W,2) [range, intensity] payload of B .npy files
        (channel 0 is read in place).  Returns (low_res (B,1,h,w), high_res (B,1,H,W))."""
        if raw.dtype != torch.float32 or not raw.is_contiguous():
            raise TypeError("raw must be a contiguous float32 tensor")
        if raw.dim() == 4:
            B, H, W, C = raw.shape
            sj = C
        elif raw.dim() == 3:
            B, H, W = raw.shape
            sj = 1
        else:
            raise ValueError("raw must be (B,H,W) or (B,H,W,C)")
        if (H, W) != (self.H, self.W):
            raise ValueError(f"raw image is {H}x{W}, img_size_high_res is {self.H}x{self.W}")
        return self._run(raw, 0, H * W * sj, W * sj, sj, 0, B, want_hi, want_lo)

    def from_rimg(self, payload: torch.Tensor, want_hi: bool = True, want_lo: bool = True):
        """payload: (B, s1, s0) float16, the bytes of B .rimg files after their (s0, s1) header.  The
        reference reshapes to (s1, s0), transposes and flips both axes (datasets.py:181-193): pixel (i,j)
        is payload[s1-1-j, s0-1-i], read in place through negative strides."""
        if payload.dtype != torch.float16 or payload.dim() != 3 or not payload.is_contiguous():
            raise TypeError("payload must be a contiguous (B, s1, s0) float16 tensor")
        B, s1, s0 = payload.shape
        if (s0, s1) != (self.H, self.W):
            raise ValueError(f".rimg header says {s0}x{s1}, img_size_high_res is {self.H}x{self.W}")
        return self._run(payload, 1, s0 * s1, -1, -s0, s0 * s1 - 1, B, want_hi, want_lo)


# -- file payloads (host I/O only: no arithmetic on the pixels) ------------------------------------------
def read_npy_payload(path: str) -> np.ndarray:
    """The array npy_loader opens (datasets.py:175-179), WITHOUT its channel-0 copy: (H, W, 2) float32."""
    a = np.load(path)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    return a


def read_rimg_payload(path: str) -> Tuple[int, int, np.ndarray]:
    """(s0, s1, payload (s1, s0) float16) of a CARLA .rimg file (datasets.py:181-188)."""
    usz = np.dtype(np.uint).itemsize
    with open(path, "rb") as f:
        s0, s1 = struct.unpack("=" + ("Q" if usz == 8 else "I") * 2, f.read(2 * usz))
        pay = np.fromfile(f, dtype=np.float16)
    return s0, s1, pay.reshape(s1, s0)


def list_range_files(root: str) -> List[str]:
    """RangeMapFolder(class_dir=False) sample order (datasets.py:196-222): every file with a range-map
    extension under root, directories and file names walked in sorted order."""
    out = []
    for d, dirs, files in sorted(os.walk(root, followlinks=True)):
        for fn in sorted(files):
            if fn.lower().endswith(NPY_EXTENSIONS):
                out.append(os.path.join(d, fn))
    return out


class DeviceRangeLoader:
    """Batches of (low_res, high_res) device tensors from a directory of range images: the reference's
    PairDataset(RangeMapFolder(low), RangeMapFolder(high)) + DataLoader collation (datasets.py:153-162,
    244-304; main_lidar_upsampling.py:188-214) for the case its scripts use, where both roots are the same
    directory.  Payloads are staged in one pinned buffer per batch and transformed by one kernel launch."""

    def __init__(self, root: str, prep: RangePrep, batch_size: int, device="cuda", shuffle: bool = False,
                 drop_last: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1):
        self.files = list_range_files(root)
        self.prep, self.B, self.device = prep, batch_size, torch.device(device)
        self.shuffle, self.drop_last, self.seed, self.epoch = shuffle, drop_last, seed, 0
        self.rank, self.world = rank, world_size
        self._stages, self._events, self._turn = [None, None], [None, None], 0

    def set_epoch(self, epoch: int):                # DistributedSampler.set_epoch (main:311)
        self.epoch = epoch

    def _order(self) -> List[int]:
        n = len(self.files)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        if self.world > 1:                          # DistributedSampler: pad to a multiple, then stride by rank
            total = -(-n // self.world) * self.world
            idx = (idx + idx[: total - n])[self.rank:total:self.world]
        return idx

    def __len__(self) -> int:
        n = len(self._order())
        return n // self.B if self.drop_last else -(-n // self.B)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        idx = self._order()
        for s in range(0, len(idx), self.B):
            chunk = idx[s:s + self.B]
            if len(chunk) < self.B and self.drop_last:
                return
            paths = [self.files[i] for i in chunk]
            if paths[0].lower().endswith(".rimg"):
                pays = [read_rimg_payload(p)[2] for p in paths]
                yield self.prep.from_rimg(self._upload(pays, torch.float16))
            else:
                pays = [read_npy_payload(p) for p in paths]
                yield self.prep(self._upload(pays, torch.float32))

    def _upload(self, pays, dtype) -> torch.Tensor:
        """Stack payloads in one of two pinned staging buffers and start the H2D copy; a buffer is rewritten
        only after the copy that last read it has completed."""
        k, n, shape = self._turn, len(pays), tuple(pays[0].shape)
        self._turn ^= 1
        st = self._stages[k]
        if st is None or st.dtype != dtype or tuple(st.shape[1:]) != shape or st.shape[0] < n:
            st = self._stages[k] = torch.empty((max(n, self.B),) + shape, dtype=dtype).pin_memory()
            self._events[k] = torch.cuda.Event()
        else:
            self._events[k].synchronize()
        for b, a in enumerate(pays):
            st[b].copy_(torch.from_numpy(a))
        dev = st[:n].to(self.device, non_blocking=True)
        self._events[k].record(torch.cuda.current_stream(self.device))
        return dev


# -- the producer of the KITTI .npy files ------------------------------------------------------------------
class KittiRangeProjector:
    """kitti_utils/sample_kitti_dataset.py create_range_map (:24-66) with the reference's parameters (:139-145) as a
    device kernel: a (N,4) float32 [x,y,z,intensity] Velodyne scan -> the (64,1024,2) [range m, intensity] array the
    reference stores as .npy and `RangePrep` reads in place."""

    def __init__(self, image_rows: int = 64, image_cols: int = 1024, ang_start_y: float = 24.8,
                 ang_res_y: float = 26.8 / 63, ang_res_x: float = 360 / 1024, max_range: float = 120,
                 min_range: float = 0, device="cuda"):
        self.rows, self.cols = image_rows, image_cols
        self.params = (ang_start_y, ang_res_y, ang_res_x, max_range, min_range)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("KittiRangeProjector runs on the GPU only (HIP kernel, no CPU fallback)")
        self._winner = torch.full((image_rows * image_cols,), -1, dtype=torch.int32, device=self.device)

    def __call__(self, points: torch.Tensor) -> torch.Tensor:
        if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 4 or not points.is_cuda \
                or not points.is_contiguous():
            raise TypeError("points must be a contiguous (N,4) float32 device tensor")
        out = torch.empty(self.rows, self.cols, 2, dtype=torch.float32, device=points.device)
        ops.kitti_range_map(points, points.shape[0], self.rows, self.cols, *self.params, self._winner, out)
        return out
