#!/usr/bin/env python3
"""Where the eval loop's time goes (runner.py --test on a host-resident list): page-locking the list, H2D rate by dtype,
the forward, the metric.  python tools/eval_breakdown.py [--precision bf16] [--batch 256] [--samples 3072]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--samples", type=int, default=3072)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = s3r.Stereo2Voxel(a.precision)
s3r.seed_module(model, 0)
model.to(dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rd in ("uint8", "float32"):
    left, right, gt = s3r.evaluate.synthetic_eval_set(a.samples, 0, rd)
    B = a.batch
    model(left[:B].to(dev), right[:B].to(dev))
    rt = torch.cuda.cudart()
    t0 = sync()
    for t in (left, right, gt):
        assert int(rt.cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)) == 0
    t1 = sync()
    mb = sum(t.numel() * t.element_size() for t in (left, right, gt)) / 1e6
    print(f"[{rd}] hostRegister {mb:.0f} MB: {1e3 * (t1 - t0):.1f} ms", flush=True)
    for name, t in (("left", left), ("gt", gt)):
        sl = t[:B]
        d = sl.to(dev, non_blocking=True)
        t0 = sync()
        for _ in range(5):
            d = sl.to(dev, non_blocking=True)
        t1 = sync()
        print(f"[{rd}] H2D {name} batch {sl.numel() * sl.element_size() / 1e6:.1f} MB: {1e3 * (t1 - t0) / 5:.2f} ms "
              f"= {5 * sl.numel() * sl.element_size() / (t1 - t0) / 1e9:.1f} GB/s  is_pinned={sl.is_pinned()}", flush=True)
    l, r, g = left[:B].to(dev), right[:B].to(dev), gt[:B].to(dev)
    t0 = sync()
    for _ in range(5):
        pred = model(l, r)
    t1 = sync()
    for _ in range(5):
        for th in s3r.evaluate.THRESHOLDS:
            s3r.voxel_iou(pred, g, th)
    t2 = sync()
    print(f"[{rd}] forward {1e3 * (t1 - t0) / 5:.2f} ms/batch, 4 x IoU {1e3 * (t2 - t1) / 5:.2f} ms/batch", flush=True)
    t0 = sync()
    for t in (left, right, gt):
        rt.cudaHostUnregister(t.data_ptr())
    t1 = sync()
    print(f"[{rd}] hostUnregister: {1e3 * (t1 - t0):.1f} ms", flush=True)
    for n in (a.samples // 4, a.samples):
        t0 = sync()
        res = s3r.evaluate.test_net(model, left[:n], right[:n], gt[:n], batch=B, device=dev)
        t1 = sync()
        print(f"[{rd}] test_net {n} samples: {1e3 * (t1 - t0):.1f} ms = {n / (t1 - t0):.0f} pairs/s", flush=True)
