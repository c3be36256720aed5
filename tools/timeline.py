#!/usr/bin/env python3
"""Per-workgroup timeline of one conv layer (diagnostic build with -DS3R_ABLATE, run with S3R_ABL=7):
prologue / main loop / epilogue durations and how many workgroups of a CU are inside their MFMA loop at a time.
  S3R_ABL=7 python tools/timeline.py --layer e2 [--tile 1]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="e2")
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--wino", action="store_true", help="the layer's Winograd class kernel (wino_body stamps) instead of the direct kernel")
ap.add_argument("--form", type=int, default=-1, help="--wino: launch form code (layer_bench --algo 2 --tiles)")
a = ap.parse_args()
spec = s3r.arch_spec
dev = torch.device("cuda:0")
for layers, n0, mult in ((spec.ENCODER, spec.IMG_HW, 2), (spec.DECODER, spec.MAX_DISP, 1)):
    for l, n_in, _ in spec.trace(layers, n0):
        if l.name == a.layer:
            case = (l, n_in, mult * a.batch)
l, n_in, B = case
ch = s3r.modules._HipChain([l], n_in)
s3r.seed_module(ch, 1)
ch.to(dev)
if a.tile >= 0:
    ch.tile_override[l.name] = a.tile
if a.wino:
    ch.algo_override[l.name] = s3r._lib.ALGO_WINOGRAD
    ch.tile_override[l.name] = a.form
x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), device=dev).relu_()
for _ in range(3):
    ch._run(x)
torch.cuda.synchronize()
lib = s3r.load_library()
fn = lib.s3r_debug_read_timeline_wino if a.wino else lib.s3r_debug_read_timeline
if a.wino:                                   # stamps of the last run only (the grid differs between the launch forms)
    lib.s3r_debug_clear_timeline_wino()
    ch._run(x)
    torch.cuda.synchronize()
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
N = 65536
buf = np.zeros((N, 6), dtype=np.uint64)
assert fn(buf.ctypes.data, N) == N
t = buf[buf[:, 4] > 0].astype(np.int64)
t0 = t[:, 1].min()
cu, st, ls, le, en = t[:, 0], (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0, (t[:, 4] - t0) / 100.0
iss = (t[:, 5] - t0) / 100.0
print(f"{l.name}: {len(t)} workgroups stamped (first launch of the layer), {len(np.unique(cu))} CUs, kernel span {en.max():.1f} us")
print(f"  prologue  med {np.median(ls - st):6.2f} us   p90 {np.percentile(ls - st, 90):6.2f}")
print(f"  main loop med {np.median(le - ls):6.2f} us   p90 {np.percentile(le - ls, 90):6.2f}")
print(f"  epilogue  issue med {np.median(iss - le):6.2f} us p90 {np.percentile(iss - le, 90):6.2f};  stores landed med "
      f"{np.median(en - le):6.2f} us p90 {np.percentile(en - le, 90):6.2f}")
# per CU: fraction of the kernel span with k workgroups inside their main loop
span = en.max()
grid = np.linspace(0, span, 2000)
hist = np.zeros(12)
cover = []
for c in np.unique(cu):
    m = cu == c
    inloop = ((grid[:, None] >= ls[m][None, :]) & (grid[:, None] < le[m][None, :])).sum(1)
    resident = ((grid[:, None] >= st[m][None, :]) & (grid[:, None] < en[m][None, :])).sum(1)
    for k in range(12):
        hist[k] += (inloop == k).sum()
    cover.append(((inloop > 0).mean(), resident.mean(), m.sum()))
hist /= hist.sum()
print("  share of (CU, time) with k workgroups in their MFMA loop: " + " ".join(f"{k}:{h:.3f}" for k, h in enumerate(hist) if h > 0.0005))
cover = np.array(cover)
print(f"  per CU: loop-covered time {cover[:, 0].mean():.3f} (min {cover[:, 0].min():.3f}), mean resident workgroups {cover[:, 1].mean():.2f}, "
      f"workgroups per CU {cover[:, 2].min():.0f}..{cover[:, 2].max():.0f}")
last_end = np.array([en[cu == c].max() for c in np.unique(cu)])
print(f"  CU finish times: min {last_end.min():.1f} med {np.median(last_end):.1f} max {last_end.max():.1f} us")
