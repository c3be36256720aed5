#!/usr/bin/env python3
"""Phases of one steady-state pass of the bf16 stem kernel per workgroup (diagnostic build: `make -C
stereo-3d-reconstruction_amd/csrc abl`):  S3R_LIB=tools/alt/abl.so S3R_ABL=32 python tools/timeline_stem.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import s3r

dev = torch.device("cuda:0")
spec = s3r.arch_spec
ch = s3r.modules._HipChain([spec.ENCODER[0]], spec.IMG_HW, precision="bf16")
s3r.seed_module(ch, 0)
ch.to(dev)
x = torch.rand(512, 3, 224, 224, device=dev)
for _ in range(3):
    ch._run(x)
torch.cuda.synchronize()
fn = s3r.load_library().s3r_debug_read_stem_timeline
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((4096, 16), dtype=np.uint64)
assert fn(buf.ctypes.data, 4096) == 4096
t = buf[buf[:, 6] > 0].astype(np.int64)
print(f"{len(t)} workgroups stamped (pass 5 of each); times in us (100 MHz counter)")
seq = [(0, "pass start"), (1, "next rows issued"), (2, "this pass's rows landed"), (3, "barrier passed"), (7, "28 operands read"),
       (8, "tile 1 MFMAs done"), (9, "tile 1 in LDS slab"), (4, "tile 1 stores issued"), (12, "tile 2 MFMAs done"),
       (13, "tile 2 in LDS slab"), (5, "tile 2 stores issued"), (6, "closing barrier passed")]
prev = None
for idx, name in seq:
    if prev is not None:
        d = (t[:, idx] - t[:, prev]) / 100.0
        print(f"  -> {name:28s} med {np.median(d):6.2f}  mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}")
    prev = idx
d = (t[:, 6] - t[:, 0]) / 100.0
print(f"  whole pass                      med {np.median(d):6.2f}  mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}")
