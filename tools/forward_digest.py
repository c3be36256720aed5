#!/usr/bin/env python3
"""sha256 of the forward's output bits at a few batches: run under two builds (S3R_LIB=...) to show that a kernel change kept
every bit.    python tools/forward_digest.py [--bf16] [--point]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r
prec = "bf16" if "--bf16" in sys.argv else "fp32"
model = (s3r.Stereo2Point if "--point" in sys.argv else s3r.Stereo2Voxel)(prec)
s3r.seed_module(model, 0)
model.to("cuda:0")
for B in (1, 2, 5, 32):
    left, right = s3r.synthetic_pairs(B, seed=3)
    y = model(left.cuda(), right.cuda())
    torch.cuda.synchronize()
    print(B, hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
