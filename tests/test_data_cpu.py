"""Input pipeline: a tiny StereoShapeNet-shaped tree written with Pillow / scipy.io and read
back through data.StereoShapeNet."""
import os

import numpy as np
import pytest
import torch


def _make_tree(root, n_models=3, views=(0, 1), size=137):
    from PIL import Image
    from scipy.io import savemat
    rng = np.random.default_rng(0)
    vols = {}
    for m in range(n_models):
        tax, model = "02691156", f"model{m:02d}"
        rdir = os.path.join(root, "ShapeNetStereoRendering", tax, model)
        os.makedirs(rdir)
        os.makedirs(os.path.join(root, "ShapeNetVox32", tax), exist_ok=True)
        vol = (rng.random((32, 32, 32)) > 0.7).astype(np.uint8)
        vols[model] = vol
        savemat(os.path.join(root, "ShapeNetVox32", tax, model + ".mat"), {"Volume": vol})
        for v in views:
            for side in "lr":
                rgba = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
                rgba[: size // 2, :, 3] = 0                       # transparent top half -> white background
                Image.fromarray(rgba, "RGBA").save(os.path.join(rdir, "render_%02d_%s.png" % (v, side)))
    return vols


def test_dataset_reads_the_readme_layout(s3r, tmp_path):
    vols = _make_tree(str(tmp_path))
    ds = s3r.data.StereoShapeNet(str(tmp_path))
    assert len(ds) == 3 * 2
    left, right, vol = ds[0]
    assert left.shape == (3, 224, 224) and right.shape == (3, 224, 224) and vol.shape == (32, 32, 32)
    assert left.dtype == torch.float32 and 0.0 <= left.min() and left.max() <= 1.0
    assert torch.all(left[:, :100, :] == 1.0)                      # transparent pixels composited over white
    assert not torch.equal(left, right)
    assert torch.equal(vol, torch.from_numpy(vols["model00"].astype(np.float32)))
    got = list(s3r.data.batches(ds, 4))
    assert [b[0].shape[0] for b in got] == [4, 2] and got[0][2].shape == (4, 32, 32, 32)
    only = s3r.data.StereoShapeNet(str(tmp_path), views=[1])
    assert len(only) == 3 and all(v == 1 for _, _, v in only.items)


def test_dataset_with_exr_disparity(s3r, tmp_path):
    """with_disparity=True: the disp_%02d_{l,r}.exr files of README.md:75-76 through exr.py; views without both files
    are not listed."""
    _make_tree(str(tmp_path), n_models=2, views=(0, 1))
    rng = np.random.default_rng(5)
    want = {}
    for m in range(2):
        rdir = os.path.join(str(tmp_path), "ShapeNetStereoRendering", "02691156", f"model{m:02d}")
        for side in "lr":                                          # view 0 only
            d = (rng.random((224, 224), dtype=np.float32) * 60)
            d[:30] = np.inf
            want[(m, side)] = d
            s3r.exr.write_exr(os.path.join(rdir, "disp_00_%s.exr" % side), {"R": d, "G": d, "B": d}, "ZIP", half=False)
    ds = s3r.data.StereoShapeNet(str(tmp_path), with_disparity=True)
    assert len(ds) == 2 and all(v == 0 for _, _, v in ds.items)
    left, right, vol, dl, dr = ds[1]
    assert dl.shape == (224, 224) and dl.dtype == torch.float32
    assert np.array_equal(dl.numpy(), want[(1, "l")]) and np.array_equal(dr.numpy(), want[(1, "r")])
    batch = next(iter(s3r.data.batches(ds, 2)))
    assert len(batch) == 5 and batch[3].shape == (2, 224, 224)
    assert len(s3r.data.StereoShapeNet(str(tmp_path))) == 4       # without disparity every view is listed


def test_dataset_errors(s3r, tmp_path):
    with pytest.raises(FileNotFoundError, match="README.md:73-77"):
        s3r.data.StereoShapeNet(str(tmp_path / "nope"))
