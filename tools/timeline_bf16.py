#!/usr/bin/env python3
"""Per-workgroup timeline of one bf16 layer, plane kernel or (r06) per-tap kernel (diagnostic build: `make -C stereo-3d-reconstruction_amd/csrc abl`):
  S3R_LIB=tools/alt/abl.so S3R_ABL=7 python tools/timeline_bf16.py --layer e2 [--tile 22] [--batch 256]
phases per workgroup: tables (position decode, LDS tables), image (first image DMA issued -> landed), loop, epilogue
(issue), stores (landed); and how many workgroups of a CU are inside their MFMA loop at a time."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="e2")
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
spec = s3r.arch_spec
dev = torch.device("cuda:0")
for layers, n0, mult in ((spec.ENCODER, spec.IMG_HW, 2), (spec.DECODER, spec.MAX_DISP, 1)):
    for l, n_in, _ in spec.trace(layers, n0):
        if l.name == a.layer:
            case = (l, n_in, mult * a.batch)
l, n_in, B = case
ch = s3r.modules._HipChain([l], n_in, precision="bf16")
s3r.seed_module(ch, 1)
ch.to(dev)
if a.tile >= 0:
    ch.tile_override[l.name] = a.tile
x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), device=dev).to(torch.bfloat16)
x = x.permute(0, *range(2, x.dim()), 1).contiguous()
for _ in range(3):
    ch._run(x)
torch.cuda.synchronize()
lib = s3r.load_library()
fn = lib.s3r_debug_read_timeline_h
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
N = 65536
buf = np.zeros((N, 8), dtype=np.uint64)
assert fn(buf.ctypes.data, N) == N
t = buf[buf[:, 6] > 0].astype(np.int64)
t0 = t[:, 1].min()
cu = t[:, 0]
st, tb, im, le, ei, en = [(t[:, i] - t0) / 100.0 for i in range(1, 7)]
print(f"{l.name}: {len(t)} workgroups stamped, {len(np.unique(cu))} CUs, kernel span {en.max():.1f} us")
def stat(name, v):
    print(f"  {name:28s} med {np.median(v):7.2f} us   mean {v.mean():7.2f}   p90 {np.percentile(v, 90):7.2f}")
stat("tables (decode, LDS tables)", tb - st)
stat("first image DMA -> landed", im - tb)
stat("K loop", le - im)
stat("epilogue (issue)", ei - le)
stat("stores landed", en - ei)
stat("workgroup lifetime", en - st)
if (t[:, 7] > 0).any():      # per-tap kernel: ticks wave 0 spent in the loop's `s_waitcnt vmcnt(0)` — the NEXT K tile's operands not landed yet
    wt = t[:, 7] / 100.0
    stat("  of the K loop: waiting on DMA", wt)
    print(f"  share of the K loop spent waiting for the next K tile's operands: {float((wt / np.maximum(le - im, 1e-9)).mean()):.3f}")
span = en.max()
grid = np.linspace(0, span, 2000)
hist = np.zeros(12)
res = []
for c in np.unique(cu):
    m = cu == c
    inloop = ((grid[:, None] >= im[m][None, :]) & (grid[:, None] < le[m][None, :])).sum(1)
    resident = ((grid[:, None] >= st[m][None, :]) & (grid[:, None] < en[m][None, :])).sum(1)
    for k in range(12):
        hist[k] += (inloop == k).sum()
    res.append(resident.mean())
hist /= hist.sum()
print("  share of (CU, time) with k workgroups in their K loop: " + " ".join(f"{k}:{h:.3f}" for k, h in enumerate(hist) if h > 0.0005))
print(f"  mean resident workgroups per CU: {np.mean(res):.2f}")
# gap between one workgroup's end and the next one's entry on the same CU slot (dispatch latency)
gaps = []
for c in np.unique(cu):
    m = cu == c
    e = np.sort(en[m]); s_ = np.sort(st[m])
    k = min(len(e), len(s_)) - 3
    if k > 0:
        gaps.extend((s_[3:3 + k] - e[:k]).tolist())
if gaps:
    print(f"  next entry - an earlier exit on the same CU (3 slots): med {np.median(gaps):.2f} us")
