"""The bulk + remainder launch cut (plan_tail_cut) must never change a result: any tile shape produces the
same bits, so a layer computed in two launches with different tiles equals the single-launch result."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import s3r
outs = []
# every tile shape the dual launch is built for (0, 1, 4, 7), 16-byte gathers (28-wide rows) and dword gathers (14-wide
# rows, 3D), workgroup counts just over one or two 256-workgroup rounds
for op, n, tile, batches in (("conv2d", 28, 0, (42, 43, 50)), ("conv2d", 28, 1, (42, 85)), ("conv2d", 28, 4, (84, 90)),
                             ("conv2d", 28, 7, (22, 43)), ("conv3d", 14, 0, (12, 13)), ("conv3d", 14, 7, (7, 13))):
    L = s3r.arch_spec.Layer("t", op, 64, 128, 3, 1, 1)
    ch = s3r.modules._HipChain([L], n)
    s3r.seed_module(ch, 3)
    ch.to("cuda:0")
    ch.tile_override["t"] = tile
    nd = s3r.arch_spec.ndim(L)
    for B in batches:
        x = torch.randn((B, 64) + (n,) * nd, generator=torch.Generator().manual_seed(B)).cuda()
        outs.append(ch._run(x).cpu())
torch.save(outs, sys.argv[1])
"""


def test_tail_cut_is_bitwise_neutral(tmp_path):
    """"cut": the default — bulk and re-tiled remainder in ONE launch (conv_glds_dual_kernel); "two": the same cut as two
    launches (S3R_NO_DUAL); "nocut": one launch of the layer's own tile.  All three give the same bits."""
    res = {}
    for tag, env in (("cut", {}), ("two", {"S3R_NO_DUAL": "1"}), ("nocut", {"S3R_NO_TAIL_CUT": "1"})):
        out = tmp_path / f"{tag}.pt"
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, "-c", _CHILD % ROOT, str(out)], check=True, env=e)
        res[tag] = torch.load(out)
    assert len(res["cut"]) == 13
    for a, b, c in zip(res["cut"], res["two"], res["nocut"]):
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c)


_RING_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import s3r
outs = []
# 64x64 tiles on launches of a few workgroups (the six-stage LDS ring's territory): K loops shorter than, equal to and
# longer than the ring (1, 5, 6 and 144 K tiles), stride 1 (16-byte gathers) and stride 2 (dword gathers), a transposed
# layer, and a split-K layer
for name, op, cin, cout, k, s, p, n, B in (("a", "conv2d", 16, 64, 1, 1, 0, 8, 2), ("b", "conv2d", 80, 64, 1, 1, 0, 8, 3),
                                           ("c", "conv2d", 96, 96, 1, 1, 0, 8, 1), ("d", "conv2d", 256, 256, 3, 1, 1, 28, 1),
                                           ("e", "conv2d", 64, 64, 3, 2, 1, 28, 2), ("f", "deconv3d", 32, 64, 4, 2, 1, 4, 1),
                                           ("g", "conv3d", 128, 64, 3, 1, 1, 7, 1)):
    L = s3r.arch_spec.Layer(name, op, cin, cout, k, s, p)
    ch = s3r.modules._HipChain([L], n)
    s3r.seed_module(ch, 3)
    ch.to("cuda:0")
    ch.tile_override[name] = 3
    if name == "g":
        ch.ksplit_override[name] = 4
    nd = s3r.arch_spec.ndim(L)
    x = torch.randn((B, cin) + (n,) * nd, generator=torch.Generator().manual_seed(7)).cuda()
    outs.append(ch._run(x).cpu())
torch.save(outs, sys.argv[1])
"""


def test_deep_lds_ring_is_bitwise_neutral(tmp_path):
    """The six-stage ring (sparse launches of the 64 x 64 tile) walks K in the same order as the two-stage loop: same
    bits, for K loops shorter and longer than the ring."""
    res = {}
    for tag, env in (("ring", {}), ("plain", {"S3R_DEEP_RING": "0"})):
        out = tmp_path / f"{tag}.pt"
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, "-c", _RING_CHILD % ROOT, str(out)], check=True, env=e)
        res[tag] = torch.load(out)
    assert len(res["ring"]) == 7
    for a, b in zip(res["ring"], res["plain"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
