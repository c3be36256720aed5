#!/usr/bin/env python3
"""Experiment: do two INDEPENDENT batches in flight on two HIP streams (tails of one kernel overlapping the body
of another) beat running them back to back?  python tools/two_stream_exp.py [--batch 32] [--half]
--half: split ONE batch over the two streams instead (each kernel half-size)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--half", action="store_true")
ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
a = ap.parse_args()
dev = torch.device("cuda:0")
B, S = a.batch, a.streams
per = B // S if a.half else B
models = [s3r.Stereo2Voxel(a.precision) for _ in range(S)]
s3r.seed_module(models[0], 0)
for m in models[1:]:
    m.load_state_dict(models[0].state_dict())
for m in models:
    m.to(dev)
data = [tuple(t.to(dev) for t in s3r.synthetic_pairs(per, seed=1000 + i)) for i in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
graphs = [s3r.GraphedForward(m, per, dev) for m in models]
for g, (l, r) in zip(graphs, data):
    g.left.copy_(l); g.right.copy_(r)


def sequential():
    for g in graphs:
        g()


def concurrent():
    cur = torch.cuda.current_stream(dev)
    for g, st in zip(graphs, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            g()
    for st in streams:
        cur.wait_stream(st)


for fn in (sequential, concurrent):
    for _ in range(3):
        fn()
torch.cuda.synchronize()
pairs = per * S
for name, fn in (("back to back", sequential), ("concurrent", concurrent), ("back to back", sequential), ("concurrent", concurrent)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{name:13s}: {dt * 1e3:7.3f} ms per {pairs} pairs  {pairs / dt:8.1f} pairs/s")
