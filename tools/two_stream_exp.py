#!/usr/bin/env python3
"""Experiment: does running two half-batches on two HIP streams (tails of one kernel overlapping the body of
another) beat one full-batch stream?  python tools/two_stream_exp.py [--batch 32]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--splits", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, S = a.batch, a.splits
models = [s3r.Stereo2Voxel() for _ in range(S + 1)]
s3r.seed_module(models[0], 0)
for m in models[1:]:
    m.load_state_dict(models[0].state_dict())
for m in models:
    m.to(dev)
left, right = s3r.synthetic_pairs(B, seed=1000)
left, right = left.to(dev), right.to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
h = B // S


def one():
    return models[S](left, right)


def multi():
    cur = torch.cuda.current_stream(dev)
    outs = []
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(models[i](left[i * h:(i + 1) * h], right[i * h:(i + 1) * h]))
    for st in streams:
        cur.wait_stream(st)
    return torch.cat(outs, 0)


for fn in (one, multi):
    for _ in range(3):
        y = fn()
torch.cuda.synchronize()
assert torch.equal(one(), multi())
for name, fn in (("one stream", one), (f"{S} streams", multi), ("one stream", one), (f"{S} streams", multi)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{name:12s}: {dt * 1e3:7.3f} ms/step  {B / dt:8.1f} pairs/s")
