#!/usr/bin/env python3
"""Transposed-convolution forms on arbitrary (cin, cout, edge, batch): two-axis (library's launch form) vs three-axis, ms per call
(the layer's own profiler record: difference pass + class kernel [+ finish]) and the aux share.
  python tools/deconv_bench.py 256,128,8,32 128,64,16,32 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r
L, spec = s3r._lib, s3r.arch_spec
dev = "cuda:0"
for arg in sys.argv[1:]:
    cin, cout, n, B = (int(v) for v in arg.split(","))
    layer = spec.Layer("t", "deconv3d", cin, cout, 4, 2, 1)
    ch = s3r.modules._HipChain([layer], n)
    s3r.seed_module(ch, 1)
    ch.to(dev)
    x = torch.randn(B, cin, n, n, n, device=dev).relu_()
    line = f"{cin:4d}->{cout:4d} edge {n:2d} B {B:3d}:"
    for name, tile in (("two-axis", -1), ("three-axis", 6)):
        ch.algo_override["t"], ch.tile_override["t"] = L.ALGO_WINOGRAD, tile
        ms, aux = [], []
        try:
            for it in range(8):
                s3r.profile_detail(1)
                s3r.profile_enable(16)
                ch._run(x)
                rec = s3r.profile_read(16)
                s3r.profile_enable(0)
                if it >= 2:
                    ms.append(sum(r["ms"] for r in rec if r["family"] == "conv_mfma"))
                    aux.append(sum(r["ms"] for r in rec if r["family"] == "aux"))
        except s3r.S3RError as e:
            line += f"  {name}: n/a"
            continue
        ms.sort(); aux.sort()
        flops = 2.0 * B * cin * n ** 3 * cout * 64 * (27 / 64 if tile == 6 else 36 / 64)
        m, a = ms[len(ms) // 2], aux[len(aux) // 2]
        line += f"  {name}: {m:.4f} ms (aux {a:.4f}, kernel at {flops / (m - a) / 1e9 / 157.3:.3f} of the pipe)"
    print(line, flush=True)
