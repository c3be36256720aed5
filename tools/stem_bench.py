#!/usr/bin/env python3
"""The bf16 stem (e1: fp32 NCHW renders -> bf16 NHWC features) alone at BASELINE configs[2] size, for profiling:
python tools/stem_bench.py [--batch 256] [--rounds 20]   (prints the library profiler's per-launch time)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--rounds", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
spec = s3r.arch_spec
ch = s3r.modules._HipChain([spec.ENCODER[0]], spec.IMG_HW, precision="bf16")
s3r.seed_module(ch, 0)
ch.to(dev)
x = torch.rand(2 * a.batch, 3, 224, 224, device=dev)
ms = []
for r in range(a.rounds + 2):
    s3r.profile_enable(8)
    ch._run(x)
    rec = s3r.profile_read(8)
    s3r.profile_enable(0)
    if r >= 2:
        ms.append(sum(e["ms"] for e in rec))
ms.sort()
by = 2 * a.batch * (3 * 224 * 224 * 4 + 112 * 112 * 32 * 2)
print(f"stem bf16 B={a.batch}: median {ms[len(ms) // 2] * 1e3:.1f} us  min {ms[0] * 1e3:.1f} us  {by / ms[len(ms) // 2] / 1e9:.2f} TB/s algorithmic")
