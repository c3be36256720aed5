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
L = s3r.arch_spec.Layer("t", "conv2d", 64, 128, 3, 1, 1)
ch = s3r.modules._HipChain([L], 28)
s3r.seed_module(ch, 3)
ch.to("cuda:0")
ch.tile_override["t"] = 0                      # 128x128 tiles: B*784/128 workgroups
outs = []
for B in (42, 43, 50):                          # 258, 264, 307 workgroups: just over one 256-workgroup round
    x = torch.randn(B, 64, 28, 28, generator=torch.Generator().manual_seed(B)).cuda()
    outs.append(ch._run(x).cpu())
torch.save(outs, sys.argv[1])
"""


def test_tail_cut_is_bitwise_neutral(tmp_path):
    res = {}
    for tag, env in (("cut", {}), ("nocut", {"S3R_NO_TAIL_CUT": "1"})):
        out = tmp_path / f"{tag}.pt"
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, "-c", _CHILD % ROOT, str(out)], check=True, env=e)
        res[tag] = torch.load(out)
    for a, b in zip(res["cut"], res["nocut"]):
        assert torch.equal(a, b)
