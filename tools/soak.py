#!/usr/bin/env python3
"""Soak: the same batches through the forward many times, eager and back to back, every output compared bitwise with the
first one for that batch (a race in a counted-vmcnt pipeline or a ring shows up as a rare mismatch).
    python tools/soak.py [--iters 300]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda:0")
bad = 0
for precision, variant, batches, u8 in (("bf16", "voxel", (256,), False), ("bf16", "voxel", (64, 7), False),
                                       ("fp32", "voxel", (32, 3, 1), False), ("fp32", "point", (8,), False),
                                       ("bf16", "point", (16,), False), ("bf16", "voxel", (256, 130), True),
                                       ("fp32", "voxel", (32, 5), True)):
    model = (s3r.Stereo2Voxel if variant == "voxel" else s3r.Stereo2Point)(precision)
    s3r.seed_module(model, 5)
    model.to(dev)
    data = {b: tuple(t.to(dev) for t in s3r.synthetic_pairs(b, seed=100 + b)) for b in batches}
    if u8:                                             # 8-bit renders: the stems scale by 1/255 as they read
        data = {b: tuple((t * 255).round().to(torch.uint8) for t in v) for b, v in data.items()}
    ref = {b: model(*data[b]).clone() for b in batches}
    torch.cuda.synchronize()
    n_bad = 0
    for i in range(a.iters):
        b = batches[i % len(batches)]
        out = model(*data[b])
        if not torch.equal(out, ref[b]):
            n_bad += 1
    torch.cuda.synchronize()
    print(f"{precision} {variant} batches {batches}{' (8-bit renders)' if u8 else ''}: {a.iters} forwards, {n_bad} mismatches", flush=True)
    bad += n_bad
sys.exit(1 if bad else 0)
