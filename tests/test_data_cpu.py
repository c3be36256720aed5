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
    assert left.dtype == torch.uint8                               # renders stay 8-bit up to the first kernel
    assert torch.all(left[:, :100, :] == 255)                      # transparent pixels composited over white
    assert not torch.equal(left, right)
    # render_dtype="float32" is the same decode scaled by 1/255 on the host, one division per sample
    fl = s3r.data.StereoShapeNet(str(tmp_path), render_dtype="float32")[0][0]
    assert fl.dtype == torch.float32 and 0.0 <= fl.min() and fl.max() <= 1.0 and torch.all(fl[:, :100, :] == 1.0)
    assert torch.equal(fl, left.float() / 255.0) and torch.equal(fl, s3r.data.renders_to_float(left))
    with pytest.raises(ValueError):
        s3r.data.StereoShapeNet(str(tmp_path), render_dtype="float16")
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


def test_composite_over_white_is_the_rounded_8bit_blend(s3r):
    rng = np.random.default_rng(3)
    rgba = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    rgba[0, 0] = (10, 20, 30, 0)
    rgba[0, 1] = (10, 20, 30, 255)
    got = s3r.data.composite_over_white_u8(rgba)
    assert got.dtype == np.uint8 and tuple(got[0, 0]) == (255, 255, 255) and tuple(got[0, 1]) == (10, 20, 30)
    c, a = rgba[..., :3].astype(np.float64), rgba[..., 3:4].astype(np.float64)
    exact = (c * a + 255.0 * (255.0 - a)) / 255.0
    assert np.array_equal(got, np.floor(exact + 0.5).astype(np.uint8))


def test_u8_scale_formula_is_the_correctly_rounded_quotient():
    """The stem kernels scale an 8-bit sample u by 1/255 as q = u * r, q' = fma(fma(-q, 255, u), r, q) with r =
    fp32(1/255) (csrc/s3r_kernels.h render_f32).  For every u in 0..255 that is the correctly rounded u / 255 — what
    numpy / torch compute on the host — checked here in exact rational arithmetic (the GPU tests then compare the two
    entries bit for bit)."""
    from fractions import Fraction

    def r32(fr):                                   # nearest float32 to an exact rational, ties to even
        f = np.float32(float(fr))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        return min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1))

    def fma(a, b, c):
        return r32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))

    r = np.float32(1.0) / np.float32(255.0)
    assert r == r32(Fraction(1, 255))
    plain_mul_wrong = 0
    for u in range(256):
        want = np.float32(u) / np.float32(255.0)
        assert want == r32(Fraction(u, 255))
        q = r32(Fraction(u) * Fraction(float(r)))
        plain_mul_wrong += int(q != want)
        assert fma(fma(-q, np.float32(255.0), np.float32(u)), r, q) == want, u
    assert plain_mul_wrong > 0                     # (a bare multiply by 1/255 is NOT enough: 126 of 256 values differ)
