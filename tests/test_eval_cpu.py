"""SURVEY 8(f)-2 / 8(f)-4 on CPU: oracle/eval_oracle.py against g10_eval.npz, the outputs of the REFERENCE's
evaluate()/MCdrop() and util/evaluation.py run in the build container (tests/golden/make_golden.py:golden_eval)."""
import os

import numpy as np
import pytest
import torch

from oracle import eval_oracle as EO


@pytest.fixture(scope="module")
def g10(golden_dir):
    return np.load(os.path.join(golden_dir, "g10_eval.npz"))


def mc_stack(pred, n=12):
    out = []
    for c in range(n):
        g = torch.Generator().manual_seed(9000 + c)
        out.append(pred[0] + 0.01 * torch.randn(pred.shape[1:], generator=g) *
                   (torch.rand(pred.shape[1:], generator=g) < 0.3))
    return torch.stack(out)


def run_case(g10, name, k):
    ds = str(g10["case_dataset"][g10["cases"].tolist().index(name)])
    ci, H, W, h, w, log_t, mc, keep, n_img = (int(v) for v in g10[f"{name}_meta"])
    pred, hi, lo = EO.synthetic_eval_case(ds, H, W, h, w, seed=500 + 10 * ci + k, log_transform=bool(log_t))
    thr = 0.0005 if ds == "durlar" else 0.03
    if mc:
        pred = EO.mc_aggregate(mc_stack(pred), thr)
    mae, mae_low, p_img, t_img = EO.postprocess(pred, hi, lo, ds, bool(log_t), mc_drop=bool(mc), keep_close_scan=bool(keep))
    if ds == "kitti":
        pcd = [EO.spherical_pcd(im, EO.kitti_tables(), 80) for im in (p_img, t_img)]
    elif ds == "carla":
        pcd = [EO.spherical_pcd(im, EO.carla_tables(H, W), 80) for im in (p_img, t_img)]
    else:
        pcd = [EO.durlar_pcd(im, g10["durlar_elevation_lut"], 120) for im in (p_img, t_img)]
    return ds, mae, mae_low, pcd


@pytest.mark.parametrize("name", ["kitti", "carla", "carla_w", "durlar", "kitti_mc", "durlar_mc"])
def test_oracle_eval_chain_vs_reference(g10, name):
    n_img = int(g10[f"{name}_meta"][-1])
    for k in range(n_img):
        ds, mae, mae_low, (pp, pg) = run_case(g10, name, k)
        assert abs(mae - g10[f"{name}_mae"][k]) <= 1e-7
        assert abs(mae_low - float(g10[f"{name}_{k}_mae_low"])) <= 1e-7
        assert np.array_equal(pp[::97], g10[f"{name}_{k}_pcd_pred_sample"])
        assert np.array_equal(pg[::97], g10[f"{name}_{k}_pcd_gt_sample"])
        assert pp.dtype == (np.float64 if ds == "durlar" else np.float32)
        iou, prec, rec, f1, dims = EO.voxel_metrics(pp, pg, 0.1)
        assert np.array_equal(dims, g10[f"{name}_{k}_dims"])
        assert np.array_equal(np.array([iou, prec, rec, f1]), g10[f"{name}_{k}_voxel"])
        if g10[f"{name}_iou"].size:                      # MCdrop never appends these lists (engine:526-534)
            assert (iou, prec, rec, f1) == tuple(g10[f"{name}_{m}"][k] for m in ("iou", "precision", "recall", "f1"))
        else:
            assert name.endswith("_mc")


def test_chamfer_restatement_small(g10):
    """UNPINNED metric (extension absent): the float32 restatement against float64 brute force and the fixture."""
    _, _, _, (pp, pg) = run_case(g10, "carla", 0)
    cd32 = EO.chamfer_sq(pg, pp)
    cd64 = EO.chamfer_sq(pg, pp, dtype=np.float64)
    assert abs(cd32 - g10["carla_chamfer_dist"][0]) <= 1e-6 * cd32
    assert abs(cd32 - cd64) <= 1e-5 * cd64


def test_mc_aggregate_matches_torch_semantics():
    g = torch.Generator().manual_seed(0)
    preds = torch.rand(12, 1, 8, 16, generator=g)
    preds[:, :, :2] = 0.5                                   # identical passes: std 0, kept unless mean < 0
    preds[:, :, 2] = -0.1                                   # 0 > thr*negative: removed
    out = EO.mc_aggregate(preds, 0.03)
    assert torch.equal(out[0, 0, :2], torch.full((2, 16), 0.5))
    assert torch.equal(out[0, 0, 2], torch.zeros(16))
    ref_std = preds.double().std(dim=0, unbiased=True)
    keep = ~(ref_std > 0.03 * preds.double().mean(0))
    assert torch.equal((out[0] != 0) | (preds.mean(0) == 0), keep | (preds.mean(0) == 0))
