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
ap.add_argument("--stagger-ms", type=float, default=0.0, help="free-running streams, the second started this much later (no joins between steps)")
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


def free_running(steps):
    """Each stream replays its own graph back to back; stream i starts i * stagger later (host busy-wait), nothing joins them
    until the end: a phase offset between two identical kernel sequences persists while both queues stay full."""
    cur = torch.cuda.current_stream(dev)
    for st in streams:
        st.wait_stream(cur)
    for i, (g, st) in enumerate(zip(graphs, streams)):
        if i:
            t1 = time.perf_counter() + a.stagger_ms * 1e-3
            while time.perf_counter() < t1:
                pass
        with torch.cuda.stream(st):
            g()
    for _ in range(steps - 1):
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g()
    for st in streams:
        cur.wait_stream(st)


if a.stagger_ms > 0:
    free_running(3)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        free_running(a.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"free-running, stagger {a.stagger_ms} ms: {dt * 1e3:7.3f} ms per {per * S} pairs  {per * S / dt:8.1f} pairs/s")

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
