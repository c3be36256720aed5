#!/usr/bin/env python3
"""Micro-benchmark of the Stereo2Point-only kernels at BASELINE.json configs[3] size (B = 32): the three point-head
linear layers (weight streaming: GB/s of weights) and the Chamfer kernel (pairs/s, "TFLOP/s" at 8 flops per pair),
HIP-event timed by the library's profiler, median of --rounds.

    python tools/point_bench.py [--batch 32] [--rounds 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import s3r  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--points", type=int, default=2048)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = s3r.arch_spec
    B = args.batch
    head = s3r.PointHead()
    s3r.seed_module(head, 0)
    head.to(dev)
    x = torch.randn(B, spec.LATENT_C, 4, 4, 4, device=dev)
    g = torch.Generator().manual_seed(0)
    p = torch.rand(B, args.points, 3, generator=g).to(dev)
    q = torch.rand(B, args.points, 3, generator=g).to(dev)
    big = torch.empty(64 << 20, device=dev)                   # 256 MB: flushed through the caches between rounds
    res = {}
    for r in range(args.rounds + 2):
        big.add_(1.0)                                         # evict weights / clouds from L2 and the Infinity Cache
        s3r.profile_enable(64)
        head(x)
        s3r.chamfer_distance(p, q)
        rec = s3r.profile_read(64)
        s3r.profile_enable(0)
        if r < 2:
            continue
        for e in rec:
            key = (e["family"], e["tag"])
            res.setdefault(key, []).append((e["ms"], e["bytes"], e["flops"]))
    names = {300 + i: l.name for i, l in enumerate(spec.POINT_HEAD)}
    for (fam, tag), v in sorted(res.items()):
        ms = sorted(t for t, _, _ in v)[len(v) // 2]
        by, fl = v[0][1], v[0][2]
        label = names.get(tag, fam)
        if fam == "linear":
            print(f"{label:8s} {ms * 1e3:8.1f} us   {by / ms / 1e6:8.1f} GB/s algorithmic (weights + activations, incl. the split-K finish)")
        elif fam == "chamfer":
            print(f"chamfer  {ms * 1e3:8.1f} us   {fl / ms / 1e9:8.2f} TFLOP/s at 8 flops per pair ({2.0 * B * args.points ** 2 / ms / 1e9:.2f} T pairs/s)")


if __name__ == "__main__":
    main()
