#!/usr/bin/env python3
"""Three-axis transposed Winograd form (algo = WINOGRAD, tile = 6) against the oracle block and against the two-axis form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r
from oracle import s2v_oracle as oracle
spec, L = s3r.arch_spec, s3r._lib
Layer = spec.Layer
dev = "cuda:0"
dec = {l.name: l for l in spec.DECODER}
cases = [([Layer("ta", "deconv3d", 32, 64, 4, 2, 1)], 8, 1), ([Layer("tb", "deconv3d", 64, 72, 4, 2, 1)], 8, 3),
         ([Layer("tc", "deconv3d", 32, 24, 4, 2, 1)], 16, 2), ([dec["d2"]], 8, 2), ([dec["d3"]], 16, 2),
         ([dec["d3"], dec["d4"]], 16, 3), ([dec["d3"], dec["d4"]], 16, 32)]
for layers, n_in, B in cases:
    ch = s3r.modules._HipChain(layers, n_in, precision="fp32")
    s3r.seed_module(ch, 7)
    blocks = [oracle._Block(l).eval() for l in layers]
    for l, blk in zip(layers, blocks):
        blk.load_state_dict(getattr(ch, l.name).state_dict())
    ch.to(dev)
    x = torch.randn((B, layers[0].cin) + (n_in,) * 3, generator=torch.Generator().manual_seed(3)).relu_()
    with torch.no_grad():
        want = x
        for blk in blocks:
            want = blk(want)
    two = ch._run(x.to(dev)).cpu()
    ch.algo_override[layers[0].name], ch.tile_override[layers[0].name] = L.ALGO_WINOGRAD, 6
    three = ch._run(x.to(dev)).cpu()
    again = ch._run(x.to(dev)).cpu()
    one = ch._run(x[B - 1:].to(dev)).cpu()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(f"{'+'.join(l.name for l in layers):8s} n={n_in} B={B}: 3-axis vs oracle {rel(three, want):.3e}  2-axis vs oracle {rel(two, want):.3e}  "
          f"deterministic {torch.equal(three, again)}  batch-invariant {torch.equal(one[0], three[B - 1])}", flush=True)
