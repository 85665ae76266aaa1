#!/usr/bin/env python3
"""Measurements for the 8(f) rows around the hot path (run on the GPU box):
  prep     : tulip_range_prep, KITTI B=8 from the interleaved .npy payload -> GB/s against the HBM roofline
  infer    : eval forward, tulip_base KITTI B=8, HIP-graph replay -> images/s (the MCdrop tile)
  evaluate : per-image post-processing + projection + voxel metrics + Chamfer -> ms/image, Chamfer pair rate
             against the fp32 vector-ALU rate, and the oracle (numpy/torch CPU) timed on the same image
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    from tulip_amd import data as D, evaluation as EV
    from tulip_amd.infer import GraphedForward
    from tulip_amd.model.tulip import tulip_base
    from oracle import data_oracle as DO, eval_oracle as EO, tulip_oracle as O
    dev = "cuda"
    out = {}
    # ---- prep
    B, H, W = 8, 64, 1024
    raw = DO.synthetic_raw(B, H, W, seed=1)
    payload = torch.stack([raw, torch.rand_like(raw)], -1).contiguous().to(dev)
    prep = D.RangePrep("kitti", (16, 1024), (64, 1024), True)
    ms = timed(lambda: prep(payload), 200)
    algo = B * H * W * 4 * (1 + 1 + 0.25)            # channel-0 read + hi write + lo write
    touched = B * H * W * 4 * (2 + 1 + 0.25)         # the interleaved intensity shares the cache lines
    t0 = time.perf_counter()
    for _ in range(5):
        DO.range_prep(raw, DO.DATASETS["kitti"], (16, 1024), (64, 1024), True)
    cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    out["prep"] = {"workload": "kitti B=8 64x1024 from (H,W,2) payload", "ms": ms, "algorithmic_GBps": algo / ms / 1e6,
                   "touched_GBps": touched / ms / 1e6, "hbm_peak_GBps": 8000, "oracle_cpu_ms": cpu_ms,
                   "note": "2.6 MB per launch: launch-latency bound, not bandwidth bound"}
    big = torch.rand(64, 128, 2048, device=dev) * 100
    prep_big = D.RangePrep("durlar", (32, 2048), (128, 2048), True)
    ms = timed(lambda: prep_big(big), 50)
    out["prep_large"] = {"workload": "durlar B=64 128x2048", "ms": ms,
                         "algorithmic_GBps": 64 * 128 * 2048 * 4 * 2.25 / ms / 1e6}
    # ---- inference
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=(2, 8),
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(dev).eval()
    cfg = O.TulipConfig()
    for Bi in (8, 64):
        lo, hi = O.synthetic_batch(cfg, Bi, seed=2)
        gf = GraphedForward(m, Bi)
        gf(lo.to(dev))
        ms = timed(lambda: gf(), 50)
        out[f"infer_B{Bi}"] = {"workload": f"tulip_base KITTI 16x1024->64x1024 eval forward B={Bi}, graph replay",
                               "ms": ms, "images_per_s": Bi / ms * 1e3}
    # ---- evaluate
    for ds, HW, hw in (("kitti", (64, 1024), (16, 1024)), ("durlar", (128, 2048), (32, 2048))):
        pred, hi, lo = EO.synthetic_eval_case(ds, *HW, *hw, seed=3, log_transform=True)
        ev = EV.RangeEvaluator(ds, hw, HW, True, 0.1, False, False, dev)
        p, l, h_ = pred.to(dev), lo.to(dev), hi.to(dev)
        ms_all = timed(lambda: ev(p, l, h_), 10, warm=2)
        ms_pc = timed(lambda: ev.point_clouds(p, l, h_), 50)
        n = HW[0] * HW[1]
        from tulip_amd import ops
        ms_cd = timed(lambda: ops.chamfer_sq(ev.pcd_gt, n, ev.pcd_pred, n, ev.f64, ev.dist_a, ev.dist_b, ev.scratch,
                                             ev.cd), 10, warm=2)
        ms_vox = timed(lambda: ops.voxel_metrics(ev.pcd_pred, n, ev.pcd_gt, n, ev.f64, 0.1, ev.bm_pred, ev.bm_gt,
                                                 ev.bitmap_words, ev.scratch, ev.vox), 50)
        pairs = 2.0 * n * n
        rec = {"workload": f"{ds} {HW[0]}x{HW[1]} one image", "ms_total": ms_all, "ms_post_and_projection": ms_pc,
               "ms_voxel_metrics": ms_vox, "ms_chamfer": ms_cd, "chamfer_Gpairs_per_s": pairs / ms_cd / 1e6,
               # 8 flops per pair (3 sub, 3 mul/fma, 2 add folded into fma, 1 min counted as 1)
               "chamfer_valu_TFLOPs": pairs * 8 / ms_cd / 1e9, "valu_fp32_peak_TFLOPs": 157.3}
        if ds == "kitti":
            t0 = time.perf_counter()
            mae, mae_low, p_img, t_img = EO.postprocess(pred, hi, lo, ds, True)
            pp, pg = (EO.spherical_pcd(im, EO.kitti_tables(), 80) for im in (p_img, t_img))
            t1 = time.perf_counter()
            EO.voxel_metrics(pp, pg, 0.1)
            t2 = time.perf_counter()
            EO.chamfer_sq(pg, pp)
            t3 = time.perf_counter()
            rec["oracle_cpu_ms"] = {"post_and_projection": (t1 - t0) * 1e3, "voxel_metrics_sparse": (t2 - t1) * 1e3,
                                    "chamfer_bruteforce": (t3 - t2) * 1e3, "threads": torch.get_num_threads()}
        out[f"evaluate_{ds}"] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
