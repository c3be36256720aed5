"""ORACLE — test infrastructure, NOT product code.  **Parity unpinned.**

CPU fp32 PyTorch restatement of this build's own `arch_spec` (the reference's layer table is
not in the mount: /root/reference/README.md:5 says the source lives on the unmounted
Stereo2Voxel / Stereo2Point branches).  There are no reference golden
vectors, known-answer tests or fixtures to pin it against, so every parity claim made with
this file is "HIP path == this build's own restatement", never "== the reference".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (stereo-3d-reconstruction_amd/) never does; it fails loudly without its HIP
library instead.

What each function restates (reference locations are the most specific the mount supports):
  OracleEncoder      shared 2D conv tower over left/right renders   (README.md:73-74)
  cost_volume        bidirectional shift-and-diff disparity volume  (README.md:75-76)
  OracleDecoder      3D conv hourglass -> 32^3 occupancy + sigmoid  (README.md:77;    SURVEY §8a row 3)
  eval-mode BN+ReLU  per-channel affine after each conv             
  OraclePointHead    latent -> (B,2048,3) cloud                     (README.md:36;    SURVEY §8a row 5)
  chamfer_distance   the op extensions/chamfer_dist provides        (README.md:64-65)
"""
from __future__ import annotations

import importlib
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
spec = importlib.import_module("stereo-3d-reconstruction_amd.arch_spec")


class _Block(nn.Module):
    """conv (+ eval BatchNorm) (+ activation), built from one arch_spec.Layer."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer
        if layer.op == "conv2d":
            self.conv = nn.Conv2d(layer.cin, layer.cout, layer.k, layer.s, layer.p, dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm2d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "conv3d":
            self.conv = nn.Conv3d(layer.cin, layer.cout, layer.k, layer.s, layer.p, dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm3d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "deconv3d":
            self.conv = nn.ConvTranspose3d(layer.cin, layer.cout, layer.k, layer.s, layer.p, output_padding=layer.opad,
                                           dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm3d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "deconv2d":
            self.conv = nn.ConvTranspose2d(layer.cin, layer.cout, layer.k, layer.s, layer.p, output_padding=layer.opad,
                                           dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm2d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "linear":
            self.conv = nn.Linear(layer.cin, layer.cout, bias=True)
            self.bn = None
        else:
            raise ValueError(layer.op)

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.layer.act == "relu":
            x = F.relu(x)
        elif self.layer.act == "sigmoid":
            x = torch.sigmoid(x)
        elif self.layer.act == "leaky_relu":
            x = F.leaky_relu(x, spec.act_param(self.layer))
        elif self.layer.act == "elu":
            x = F.elu(x, spec.act_param(self.layer))
        elif self.layer.act == "tanh":
            x = torch.tanh(x)
        elif self.layer.act != "none":
            raise ValueError(self.layer.act)
        return x


class _Chain(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.names = [l.name for l in layers]
        for l in layers:
            self.add_module(l.name, _Block(l))

    def forward(self, x, upto=None):
        for n in self.names:
            x = getattr(self, n)(x)
            if n == upto:
                break
        return x


class OracleEncoder(_Chain):
    """(N,3,224,224) -> (N,32,28,28); the same weights serve the left and the right view."""

    def __init__(self):
        super().__init__(spec.ENCODER)


def cost_volume(feat_l: torch.Tensor, feat_r: torch.Tensor, max_disp: int = spec.MAX_DISP) -> torch.Tensor:
    """Bidirectional shift-and-diff volume, (B,C,H,W) x2 -> (B,2C,D,H,W).

    channels [0,C):   left-referenced   L[b,c,h,w] - R[b,c,h,w-d]   (0 where w-d <  0)
    channels [C,2C):  right-referenced  R[b,c,h,w] - L[b,c,h,w+d]   (0 where w+d >= W)
    """
    B, C, H, W = feat_l.shape
    vol = feat_l.new_zeros(B, 2 * C, max_disp, H, W)
    for d in range(max_disp):            # the stock-PyTorch formulation: a Python loop over D
        if d >= W:
            break
        vol[:, :C, d, :, d:] = feat_l[:, :, :, d:] - feat_r[:, :, :, : W - d]
        vol[:, C:, d, :, : W - d] = feat_r[:, :, :, : W - d] - feat_l[:, :, :, d:]
    return vol


class OracleDecoder(_Chain):
    """(B,64,28,28,28) -> (B,32,32,32) occupancy probabilities."""

    def __init__(self):
        super().__init__(spec.DECODER)

    def forward(self, vol, upto=None):
        x = super().forward(vol, upto)
        if upto is None or upto == "d4":
            x = x.squeeze(1)
        return x

    def latent(self, vol):
        return super().forward(vol, "v6")


class OraclePointHead(_Chain):
    """(B,512,4,4,4) latent -> (B,2048,3)."""

    def __init__(self):
        super().__init__(spec.POINT_HEAD)

    def forward(self, latent):
        x = super().forward(latent.flatten(1))
        return x.view(-1, spec.N_POINTS, 3)


class OracleStereo2Voxel(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = OracleEncoder()
        self.decoder = OracleDecoder()

    def forward(self, left, right):
        B = left.shape[0]
        feats = self.encoder(torch.cat([left, right], 0))
        vol = cost_volume(feats[:B], feats[B:])
        return self.decoder(vol)


class OracleStereo2Point(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = OracleEncoder()
        self.decoder = _Chain(spec.DECODER_DOWN)
        self.point_head = OraclePointHead()

    def forward(self, left, right):
        B = left.shape[0]
        feats = self.encoder(torch.cat([left, right], 0))
        vol = cost_volume(feats[:B], feats[B:])
        return self.point_head(self.decoder(vol))


def chamfer_distance(p: torch.Tensor, q: torch.Tensor):
    """Squared-L2 nearest-neighbour distances both ways.

    p (B,N,3), q (B,M,3) -> dist1 (B,N) = min_j |p_i-q_j|^2, dist2 (B,M) = min_i |p_i-q_j|^2,
    idx1 (B,N) int32, idx2 (B,M) int32 (first minimum).  Computed in the difference form
    (dx^2+dy^2+dz^2), not the |p|^2+|q|^2-2pq expansion, so fp32 results are well conditioned.
    """
    d = (p[:, :, None, :] - q[:, None, :, :])
    d = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    dist1, idx1 = d.min(dim=2)
    dist2, idx2 = d.min(dim=1)
    return dist1, dist2, idx1.to(torch.int32), idx2.to(torch.int32)


def chamfer_loss(p, q):
    """mean(dist1) + mean(dist2) — the scalar a ChamferDistance module reduces to."""
    d1, d2, _, _ = chamfer_distance(p, q)
    return d1.mean() + d2.mean()


def disparity_wta(feat_l: torch.Tensor, feat_r: torch.Tensor, max_disp: int = spec.MAX_DISP):
    """Winner-take-all disparity from the cost volume's own costs (README.md:75-76 documents the
    ground truth this is scored against).  (B,C,H,W) x2 -> disp_l, disp_r (B,H,W) fp32, integer-valued:
        disp_l[b,h,w] = first argmin_{d <= min(max_disp-1, w)}     sum_c |L[b,c,h,w] - R[b,c,h,w-d]|
        disp_r[b,h,w] = first argmin_{d <= min(max_disp-1, W-1-w)} sum_c |R[b,c,h,w] - L[b,c,h,w+d]|
    The channel sum is accumulated c = 0..C-1 one channel at a time in fp32 (not torch.sum, whose order is
    unspecified), so near-ties resolve the same way in every implementation of this definition."""
    B, C, H, W = feat_l.shape
    inf = torch.full((B, H, W), float("inf"))
    best_l, best_r = inf.clone(), inf.clone()
    arg_l, arg_r = torch.zeros(B, H, W), torch.zeros(B, H, W)
    for d in range(min(max_disp, W)):
        cl = torch.zeros(B, H, W - d)
        cr = torch.zeros(B, H, W - d)
        for c in range(C):
            diff = (feat_l[:, c, :, d:] - feat_r[:, c, :, : W - d]).abs()      # |L(w) - R(w-d)| == |R(w') - L(w'+d)|
            cl = cl + diff
            cr = cr + diff
        full_l, full_r = inf.clone(), inf.clone()
        full_l[:, :, d:] = cl
        full_r[:, :, : W - d] = cr
        upd_l, upd_r = full_l < best_l, full_r < best_r
        best_l, best_r = torch.where(upd_l, full_l, best_l), torch.where(upd_r, full_r, best_r)
        arg_l, arg_r = torch.where(upd_l, torch.full_like(arg_l, d), arg_l), torch.where(upd_r, torch.full_like(arg_r, d), arg_r)
    return arg_l, arg_r


def disparity_epe(pred: torch.Tensor, gt: torch.Tensor):
    """Per-sample end-point error over the valid ground-truth pixels (finite, >= 0) and their count."""
    valid = torch.isfinite(gt) & (gt >= 0)
    err = torch.where(valid, (pred - torch.where(valid, gt, torch.zeros_like(gt))).abs(), torch.zeros_like(gt)).double()
    n = valid.flatten(1).sum(1)
    s = err.flatten(1).sum(1)
    return torch.where(n > 0, s / n.clamp(min=1), torch.zeros_like(s)).float(), n.to(torch.int32)


def voxel_iou(pred: torch.Tensor, gt: torch.Tensor, th: float = 0.5) -> torch.Tensor:
    """Per-sample IoU of thresholded occupancy grids: |pred>th & gt>th| / |pred>th | gt>th|."""
    a, b = pred > th, gt > th
    inter = (a & b).flatten(1).sum(1).float()
    union = (a | b).flatten(1).sum(1).float()
    return torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
