"""Input pipeline of the path: StereoShapeNet files -> (left, right, volume) batches.

File layout as the reference's config documents it (/root/reference/README.md:73-77):
    LEFT_RENDERING_PATH   <root>/ShapeNetStereoRendering/<taxonomy>/<model>/render_%02d_l.png
    RIGHT_RENDERING_PATH  <root>/ShapeNetStereoRendering/<taxonomy>/<model>/render_%02d_r.png
    LEFT/RIGHT_DISP_PATH  .../disp_%02d_{l,r}.exr          (ground-truth disparity: read with this package's own
                                                            minimal OpenEXR decoder, exr.py — pyexr/OpenEXR are
                                                            absent from this image — when with_disparity=True)
    VOLUME_PATH           <root>/ShapeNetVox32/<taxonomy>/<model>.mat

The reference's own transforms (crop / background / normalisation constants) are on unmounted branches
(SURVEY.md §0).  What is EVIDENCED by the mount: the five path templates above and the file types.  What is
INVENTED by this build (each a constant of this file, to be overwritten from the reference the day it is mounted —
tools/resurvey.py lists the reference's own): compositing RGBA renders over a WHITE background IN 8 BITS (rounded
integer blend, as an image library composites); scaling to [0,1] by 1/255 with NO mean / std normalisation; BILINEAR
resize to 224x224 when a render has another size (no crop); `volume > 0`
as the occupancy rule for the .mat grids; taking the first of Z / V / Y / disparity / R as the EXR disparity channel
(exr.disparity_channel) with "finite and >= 0" as its validity rule; block-mean pooling of the 224x224 maps to the
28x28 read-out resolution (downsample_disparity).  Decoding is host work; batches are handed to the GPU through
graph.PrefetchingLoader so the copy overlaps the forward.

Renders stay 8-BIT all the way to the first kernel (render_dtype="uint8", the default): the PNGs are 8-bit, a uint8 batch
is a quarter of the bytes across PCIe, and the stem scales by 1/255 as it reads (`s3r_encoder_forward_u8`) with the same
single rounding as the host conversion — render_dtype="float32" yields exactly `uint8 / 255` computed here, and both
give bit-identical predictions.
"""
from __future__ import annotations

import os
import re
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

IMG = 224
RENDER_DIR, VOLUME_DIR = "ShapeNetStereoRendering", "ShapeNetVox32"
_VIEW = re.compile(r"render_(\d+)_l\.png$")


def composite_over_white_u8(rgba: np.ndarray) -> np.ndarray:
    """(H,W,4) uint8 RGBA -> (H,W,3) uint8 over a white background: round((c * a + 255 * (255 - a)) / 255) in integers."""
    c = rgba[..., :3].astype(np.uint32)
    a = rgba[..., 3:4].astype(np.uint32)
    return ((c * a + 255 * (255 - a) + 127) // 255).astype(np.uint8)


def renders_to_float(u8) :
    """The host form of the stem's 1/255 scaling: float32(u) / float32(255), one correctly rounded division per sample
    (numpy array or torch tensor, host or device).  `model(renders_to_float(x))` == `model(x)` bit for bit for uint8 x.
    The divisor is a TENSOR on the input's device: dividing a device tensor by a Python scalar makes PyTorch multiply by
    fp32(1/255) instead, which is not the correctly rounded quotient for 126 of the 256 codes."""
    if isinstance(u8, torch.Tensor):
        return torch.div(u8.to(torch.float32), torch.tensor(255.0, dtype=torch.float32, device=u8.device))
    return u8.astype(np.float32) / np.float32(255.0)


def _load_png(path: str) -> np.ndarray:
    """One render as (3,224,224) uint8, composited over white."""
    try:
        from PIL import Image
    except ImportError as e:                              # pragma: no cover
        raise RuntimeError("decoding PNG renders needs Pillow") from e
    with Image.open(path) as im:
        im = im.convert("RGBA")
        if im.size != (IMG, IMG):
            im = im.resize((IMG, IMG), Image.BILINEAR)
        rgba = np.asarray(im, dtype=np.uint8)
    return composite_over_white_u8(rgba).transpose(2, 0, 1).copy()          # white background, CHW


def _load_volume(path: str) -> np.ndarray:
    from scipy.io import loadmat
    m = loadmat(path)
    for k, v in m.items():
        if not k.startswith("__") and isinstance(v, np.ndarray) and v.ndim == 3:
            if v.shape != (32, 32, 32):
                raise RuntimeError(f"{path}: volume {k} has shape {v.shape}, expected 32^3")
            return (v > 0).astype(np.float32)
    raise RuntimeError(f"{path}: no 3D array found")


def _load_disparity(path: str) -> np.ndarray:
    from . import exr
    d = exr.disparity_channel(exr.read_exr(path))
    if d.shape != (IMG, IMG):
        raise RuntimeError(f"{path}: disparity map is {d.shape}, expected {(IMG, IMG)} (the renders' resolution)")
    return np.ascontiguousarray(d, dtype=np.float32)


def downsample_disparity(d: torch.Tensor, size: int = 28) -> torch.Tensor:
    """(N,H,W) ground-truth disparity in render pixels -> (N,size,size): the mean over the VALID pixels (finite, >= 0)
    of each H/size x W/size block, inf where a block has none — the resolution `Stereo2Voxel.disparity` predicts at
    (values stay in render pixels)."""
    n, h, w = d.shape
    if h % size or w % size:
        raise ValueError(f"{h}x{w} maps do not tile into {size}x{size} blocks")
    blocks = d.reshape(n, size, h // size, size, w // size).permute(0, 1, 3, 2, 4).reshape(n, size, size, -1)
    valid = torch.isfinite(blocks) & (blocks >= 0)
    cnt = valid.sum(-1)
    mean = torch.where(valid, blocks, torch.zeros_like(blocks)).sum(-1) / cnt.clamp(min=1)
    return torch.where(cnt > 0, mean, torch.full_like(mean, float("inf")))


class StereoShapeNet(torch.utils.data.Dataset):
    """One item per (taxonomy, model, view): left, right (3,224,224) uint8 (render_dtype="uint8", default: the modules
    take them as they are) or float32 in [0,1] (render_dtype="float32": uint8 / 255), volume (32,32,32) {0,1};
    with_disparity=True appends the left / right ground-truth disparity maps (224,224) float32 (render pixels; the
    files' own invalid markers — inf / negative — are kept) and lists only the views that have both EXR files."""

    def __init__(self, root: str, taxonomies: Optional[Sequence[str]] = None, views: Optional[Sequence[int]] = None,
                 with_disparity: bool = False, render_dtype: str = "uint8"):
        if render_dtype not in ("uint8", "float32"):
            raise ValueError("render_dtype must be 'uint8' or 'float32'")
        self.with_disparity, self.render_dtype = with_disparity, render_dtype
        rdir = os.path.join(root, RENDER_DIR)
        if not os.path.isdir(rdir):
            raise FileNotFoundError(f"{rdir} not found (expected the layout of README.md:73-77 under {root})")
        self.root, self.items = root, []
        for tax in sorted(os.listdir(rdir)):
            if taxonomies is not None and tax not in taxonomies:
                continue
            tdir = os.path.join(rdir, tax)
            if not os.path.isdir(tdir):
                continue
            for model in sorted(os.listdir(tdir)):
                mdir = os.path.join(tdir, model)
                vol = os.path.join(root, VOLUME_DIR, tax, model + ".mat")
                if not os.path.isdir(mdir) or not os.path.exists(vol):
                    continue
                for f in sorted(os.listdir(mdir)):
                    m = _VIEW.match(f)
                    if m and (views is None or int(m.group(1)) in views) and \
                            os.path.exists(os.path.join(mdir, f.replace("_l.png", "_r.png"))):
                        v = int(m.group(1))
                        if with_disparity and not all(os.path.exists(os.path.join(mdir, "disp_%02d_%s.exr" % (v, sd)))
                                                      for sd in "lr"):
                            continue
                        self.items.append((tax, model, v))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        tax, model, view = self.items[i]
        mdir = os.path.join(self.root, RENDER_DIR, tax, model)
        left = _load_png(os.path.join(mdir, "render_%02d_l.png" % view))
        right = _load_png(os.path.join(mdir, "render_%02d_r.png" % view))
        if self.render_dtype == "float32":
            left, right = renders_to_float(left), renders_to_float(right)
        vol = _load_volume(os.path.join(self.root, VOLUME_DIR, tax, model + ".mat"))
        if self.with_disparity:
            dl = _load_disparity(os.path.join(mdir, "disp_%02d_l.exr" % view))
            dr = _load_disparity(os.path.join(mdir, "disp_%02d_r.exr" % view))
            return (torch.from_numpy(left), torch.from_numpy(right), torch.from_numpy(vol), torch.from_numpy(dl),
                    torch.from_numpy(dr))
        return torch.from_numpy(left), torch.from_numpy(right), torch.from_numpy(vol)


def batches(ds: StereoShapeNet, batch: int, indices: Optional[Sequence[int]] = None, workers: int = 0
            ) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """Sequential (left, right, volume[, disp_left, disp_right]) host batches over `indices` (default: the whole set)."""
    sub = ds if indices is None else torch.utils.data.Subset(ds, list(indices))
    yield from torch.utils.data.DataLoader(sub, batch_size=batch, shuffle=False, num_workers=workers, pin_memory=False)
