#!/usr/bin/env python3
"""Experiment: do the HBM-bound front layers of the bf16 encoder (e1-e3 / e1-e5) run faster when the batch goes
through them in chunks whose activations fit the 256 MB Infinity Cache (each layer then reads what the previous one
just wrote from the cache)?   python tools/mall_chunk_exp.py [--images 512] [--upto e3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=512)
ap.add_argument("--upto", default="e3")
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
x = torch.rand(a.images, 3, 224, 224, device=dev)
for parts in (1, 2, 4, 8):
    encs = [s3r.Encoder(a.precision) for _ in range(parts)]          # one arena per chunk size is enough, but keep it simple
    s3r.seed_module(encs[0], 0)
    for e in encs[1:]:
        e.load_state_dict(encs[0].state_dict())
    for e in encs:
        e.to(dev)
    per = a.images // parts
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(parts)]
    def run():
        for e, xc in zip(encs, xs):
            e._run(xc, a.upto)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    print(f"{a.precision} e1..{a.upto} on {a.images} images in {parts} part(s): {min(ts) * 1e3:.3f} ms")
