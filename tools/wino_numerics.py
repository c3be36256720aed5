#!/usr/bin/env python3
"""Error of every convolution algorithm form (direct, one-axis Winograd, two-axis Winograd) against an fp64 convolution, per layer
and input distribution (randn, relu, 1000 + randn, heavy-tailed; plain and zero-sum kernels): largest error over the problem's
scale (max sum |w||x|), over the output's maximum, and relative L2.  Run on the GPU box: `python tools/wino_numerics.py`."""
import sys, torch
sys.path.insert(0, "/root/repo")
import s3r
import torch.nn.functional as F
dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
g = torch.Generator().manual_seed(123)
cases = {}
for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP)):
    for l, n_in, _ in spec.trace(layers, n0):
        cases[l.name] = (l, n_in)
for name in ("e4", "e7", "v1", "v3", "v5", "v6", "d2"):
    l, n_in = cases[name]
    B = 1 if name == "v1" else 2
    shape = (B, l.cin) + (n_in,) * spec.ndim(l)
    inputs = {"randn": torch.randn(shape, generator=g), "relu": torch.randn(shape, generator=g).relu(), "offset": 1000.0 + torch.randn(shape, generator=g),
              "heavy": torch.randn(shape, generator=g).abs() * torch.exp(2.0 * torch.randn(shape, generator=g))}
    for zero_sum in (False, True):
        ch = s3r.modules._HipChain([l], n_in, precision="fp32")
        s3r.seed_module(ch, 5)
        blk = getattr(ch, l.name)
        with torch.no_grad():
            if zero_sum:
                w = blk.conv.weight
                w -= w.mean(dim=tuple(range(2, w.dim())), keepdim=True)
            blk.bn.weight.fill_(1.0); blk.bn.bias.zero_(); blk.bn.running_mean.zero_(); blk.bn.running_var.fill_(1.0 - spec.BN_EPS)
            blk.conv.bias.zero_()
        w64 = blk.conv.weight.detach().double()
        ch.to(dev)
        for kind, x in inputs.items():
            if l.op == "deconv3d":
                want = F.conv_transpose3d(x.double(), w64, None, 2, 1); mag = F.conv_transpose3d(x.double().abs(), w64.abs(), None, 2, 1)
            else:
                f = F.conv3d if l.op == "conv3d" else F.conv2d
                want = f(x.double(), w64, None, 1, l.p); mag = f(x.double().abs(), w64.abs(), None, 1, l.p)
            want = want.clamp_min(0.0)
            scale, oscale = float(mag.max()), float(want.abs().max())
            out = []
            forms = [(L.ALGO_DIRECT, -1, "direct")]
            if name != "v6": forms.append((L.ALGO_WINOGRAD, 0, "1axis"))
            if l.op == "conv3d": forms.append((L.ALGO_WINOGRAD, 3, "2axis"))
            for algo, tile, tag in forms:
                ch.algo_override[l.name] = algo
                if tile >= 0: ch.tile_override[l.name] = tile
                else: ch.tile_override.pop(l.name, None)
                got = ch._run(x.to(dev)).cpu().double()
                e = (got - want).abs()
                out.append(f"{tag}: max/scale {float(e.max())/scale:.1e} max/omax {float(e.max())/oscale:.1e} relL2 {float(e.norm()/want.norm()):.1e}")
            print(name, kind, "zs" if zero_sum else "  ", " | ".join(out), flush=True)
