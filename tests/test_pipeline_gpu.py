"""End to end on the GPU box, the way main_lidar_upsampling.py strings the pieces together (:188-316): range-image
files on disk -> DeviceRangeLoader (one transform kernel per batch) -> train_one_epoch (fused step) -> evaluate()
(device post-processing + metrics) -> results.txt.  CARLA preset because its projection accepts any image size."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import tulip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_files_to_training_to_metrics(tmp_path):
    from tests.test_model_gpu import build
    from tulip_amd import data as D, evaluation as EV
    from tulip_amd.trainer import Trainer, train_one_epoch
    cfg = O.tiny_config(drop_path_rate=0.0)                    # 8x256 -> 32x256
    H, W = cfg.target_img_size
    h, w = cfg.img_size
    rng = np.random.default_rng(0)
    root = tmp_path / "data" / "train"
    root.mkdir(parents=True)
    jj, ii = np.meshgrid(np.arange(W) / W, np.arange(H) / H)
    for i in range(8):                                         # smooth scenes, 5..60 m, some no-return pixels
        r = 5 + 50 * (0.5 + 0.5 * np.sin(6.28 * (2 * jj + ii + 0.1 * i))) * (0.4 + 0.6 * ii)
        r[rng.random((H, W)) < 0.05] = 0
        np.save(root / f"{i:08d}.npy", np.stack([r, rng.random((H, W))], -1).astype(np.float32))
    prep = D.RangePrep("carla", (h, w), (H, W), log_transform=True)
    loader = D.DeviceRangeLoader(str(root), prep, 4, shuffle=True, seed=1)
    assert len(loader) == 2
    lo, hi = next(iter(loader))
    assert lo.shape == (4, 1, h, w) and hi.shape == (4, 1, H, W) and lo.is_cuda
    assert torch.equal(lo, hi[:, :, 0::H // h, :])             # low-res rows are rows 0::f of the target

    sd = O.key_seeded_state_dict(cfg, seed=0)
    m = build(cfg, sd, train=True)
    tr = Trainer(m, 4, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.01)
    args = SimpleNamespace(lr=2e-3, min_lr=1e-5, warmup_epochs=1, epochs=12)
    hist = []
    for epoch in range(12):
        loader.set_epoch(epoch)
        hist.append(train_one_epoch(tr, loader, epoch, args)["loss"])
    assert all(np.isfinite(hist)) and hist[-1] < 0.6 * hist[1], hist     # it learns (epoch 0 runs at lr ~ 0)

    eargs = SimpleNamespace(img_size_low_res=(h, w), img_size_high_res=(H, W), grid_size=0.1, log_transform=True,
                            dataset_select="carla", output_dir=str(tmp_path / "out"), keep_close_scan=False)
    val = D.DeviceRangeLoader(str(root), prep, 1, shuffle=False, drop_last=False)
    avg = EV.evaluate(val, m, DEV, None, eargs)
    res = json.load(open(tmp_path / "out" / "results.txt"))
    assert len(res["mae"]) == 8 and all(np.isfinite(res[k]).all() for k in res)
    assert 0 < avg["iou"] <= 1 and 0 < avg["precision"] <= 1 and avg["cd"] > 0
    assert not m.training                                        # evaluate() leaves the model in eval mode
