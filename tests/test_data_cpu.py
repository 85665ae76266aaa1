"""SURVEY 8(f)-3 on CPU: the data oracle against the reference-generated fixture g8 (bit exact), and the host
side of tulip_amd.data (file order, payload readers, error behaviour).  No GPU compute."""
import os

import numpy as np
import pytest
import torch

from oracle import data_oracle as DO
from tulip_amd import data as D


@pytest.fixture(scope="module")
def g8(golden_dir):
    return np.load(os.path.join(golden_dir, "g8_transforms.npz"))


def case_args(z, name):
    seed, h, w, H, W, log_t, shift = (int(v) for v in z[f"{name}_meta"])
    return seed, (h, w), (H, W), bool(log_t), (None if shift < 0 else shift)


def test_oracle_transforms_match_reference_bit_exact(g8):
    for name in g8["cases"].tolist():
        seed, lo_size, hi_size, log_t, shift = case_args(g8, name)
        raw = DO.synthetic_raw(2, hi_size[0], hi_size[1], seed=seed)
        lo, hi = DO.range_prep(raw, DO.DATASETS[name.split("_")[0]], lo_size, hi_size, log_t, shift)
        assert np.array_equal(lo.numpy(), g8[f"{name}_lo"]), name
        assert np.array_equal(hi.numpy(), g8[f"{name}_hi"]), name
        assert lo.shape == (2, 1) + lo_size and hi.shape == (2, 1) + hi_size


def test_gate_cases_are_exercised(g8):
    """the fixture holds pixels the gates zero and pixels sitting on both thresholds"""
    hi = g8["durlar_lin_hi"]
    seed = int(g8["durlar_lin_meta"][0])
    raw = DO.synthetic_raw(2, 64, 64, seed=seed).numpy()[:, None]
    rolled = np.roll(raw, int(g8["durlar_lin_meta"][6]), axis=-1)
    assert ((rolled > 120.0) & (hi == 0)).any() and ((rolled > 0) & (rolled < 0.3) & (hi == 0)).any()
    assert (hi == np.float32(1.0)).any() and (hi == np.float32(0.3) * np.float32(1 / 120)).any()


def test_oracle_loaders_match_reference(g8):
    assert np.array_equal(DO.npy_range(g8["npy_bytes"].tobytes()), g8["npy_expected"])
    r = DO.rimg_range(g8["rimg_bytes"].tobytes())
    assert r.dtype == np.float32 and np.array_equal(r, g8["rimg_expected"])


def test_payload_readers(tmp_path, g8):
    p = tmp_path / "a.npy"
    p.write_bytes(g8["npy_bytes"].tobytes())
    a = D.read_npy_payload(str(p))
    assert a.shape == (16, 48, 2) and np.array_equal(a[..., 0], g8["npy_expected"])
    p = tmp_path / "a.rimg"
    p.write_bytes(g8["rimg_bytes"].tobytes())
    s0, s1, pay = D.read_rimg_payload(str(p))
    assert (s0, s1) == g8["rimg_expected"].shape and pay.shape == (s1, s0)
    # the in-place addressing the kernel uses: pixel (i,j) = payload[s1-1-j, s0-1-i]
    i, j = np.meshgrid(np.arange(s0), np.arange(s1), indexing="ij")
    assert np.array_equal(pay[s1 - 1 - j, s0 - 1 - i].astype(np.float32), g8["rimg_expected"])


def test_file_order_matches_dataset_folder(tmp_path):
    for rel in ["b/2.npy", "b/10.npy", "a/z.npy", "a/A.NPY", "c.rimg", "a/skip.txt", "a/sub/q.bin"]:
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"")
    got = [os.path.relpath(f, tmp_path) for f in D.list_range_files(str(tmp_path))]
    assert got == ["c.rimg", "a/A.NPY", "a/z.npy", "a/sub/q.bin", "b/10.npy", "b/2.npy"]


def test_loader_sharding_and_len(tmp_path):
    for i in range(11):
        (tmp_path / f"{i:03d}.npy").write_bytes(b"")
    prep = D.RangePrep("kitti", (16, 64), (64, 64), True)
    ld = [D.DeviceRangeLoader(str(tmp_path), prep, 2, shuffle=True, seed=3, rank=r, world_size=2) for r in (0, 1)]
    o0, o1 = ld[0]._order(), ld[1]._order()
    assert len(o0) == len(o1) == 6 and set(o0) | set(o1) == set(range(11))
    assert len(ld[0]) == 3
    ld[0].set_epoch(1)
    assert ld[0]._order() != o0


def test_prep_argument_errors():
    with pytest.raises(KeyError):
        D.RangePrep("nuscenes", (16, 64), (64, 64))
    with pytest.raises(ValueError):
        D.RangePrep("kitti", (16, 64), (60, 64))
    prep = D.RangePrep("kitti", (16, 64), (64, 64), True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        prep(torch.zeros(1, 64, 64))
    with pytest.raises(ValueError):
        prep(torch.zeros(1, 32, 64))


def test_kitti_projection_oracle_vs_reference(golden_dir):
    """producer of the .npy files: oracle restatement of create_range_map against the reference's output (g11)"""
    z = np.load(os.path.join(golden_dir, "g11_kitti_projection.npz"))
    pts = DO.synthetic_kitti_scan(int(z["n"]), int(z["seed"]))
    img = DO.kitti_range_map(pts)
    assert img.shape == (64, 1024, 2) and img.dtype == np.float32
    assert np.array_equal(img[:, ::3, :], z["every_third_column"])
    assert np.array_equal(np.array([(img[..., 0] > 0).sum(), (img[..., 1] > 0).sum()]), z["nonzero"])
    np.testing.assert_allclose(img.astype(np.float64).sum(axis=(0, 1)), z["sums"], rtol=1e-12)
    # the reference's quirk: ranges beyond max_range are zeroed, their intensities are not
    assert (img[..., 1] > 0).sum() > (img[..., 0] > 0).sum()
