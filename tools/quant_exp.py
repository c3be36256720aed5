#!/usr/bin/env python3
"""Experiment: throughput of one layer vs batch size (is the loss on mid-size layers workgroup-count
quantisation over the 256 CUs?).  python tools/quant_exp.py --layer e7 --tile 0 --batches 20:50"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="e7")
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--batches", default="20:50")
a = ap.parse_args()
spec = s3r.arch_spec
dev = torch.device("cuda:0")
case = None
for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP)):
    for l, n_in, _ in spec.trace(layers, n0):
        if l.name == a.layer:
            case = (l, n_in)
l, n_in = case
ch = s3r.modules._HipChain([l], n_in)
s3r.seed_module(ch, 1)
ch.to(dev)
ch.tile_override[l.name] = a.tile
lo, hi = (int(v) for v in a.batches.split(":"))
tiles = {0: (128, 128), 1: (64, 256), 2: (32, 256), 3: (64, 64), 6: (128, 64), 7: (64, 128)}
bm, bn = tiles[a.tile]
for B in range(lo, hi + 1):
    x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), device=dev).relu_()
    flops = 2.0 * spec.layer_macs(l, n_in) * B
    ms = []
    for r in range(6):
        s3r.profile_enable(8)
        ch._run(x)
        rec = s3r.profile_read(8)
        s3r.profile_enable(0)
        if r:
            ms.append([q for q in rec if q["family"] == "conv_mfma"][0]["ms"])
    ms.sort()
    med = ms[len(ms) // 2]
    out = spec.out_size(l, n_in)
    npos = B * out ** spec.ndim(l)
    wgs = -(-npos // bn) * -(-l.cout // bm)
    print(f"B={B:3d} wgs={wgs:6d} wgs/CU={wgs / 256:6.2f}  {med:7.4f} ms  {flops / med / 1e9:6.1f} TF")
