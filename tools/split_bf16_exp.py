#!/usr/bin/env python3
"""Bounded experiment: an fp32-accuracy convolution on the bf16 matrix cores by operand splitting.

x = x1 + x2 + x3 and w = w1 + w2 + w3 exactly (three bf16 pieces hold fp32's 24 significant bits); the six products x1w1, x1w2, x2w1,
x1w3, x2w2, x3w1 carry everything down to 2^-24 relative ("bf16x3, 6-product"); three products (x1w1, x1w2, x2w1) carry 2^-16.  A
bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the form IS an ordinary bf16 convolution over 6 (3) times the
input channels: [x1 x1 x2 x1 x2 x3] against [w1 w2 w1 w3 w2 w1].  That is how it is measured here — the library's own bf16 kernels on
a 6x / 3x-channel layer, nothing new built — against the exact-fp32 MFMA kernel on v2 (Conv3d 64 -> 128, k3 s2 p1, 28^3 -> 14^3) at
B = 32.  Operand splitting itself (one VALU pass over the activations, 1.5x / 1x their bytes) is NOT charged: an upper bound on the gain.

    python tools/split_bf16_exp.py            # GPU timing + CPU error emulation (exact products, fp64 accumulation)

Kill rule: < 1.25x on v2 alone, or error > 2x the fp32 kernel's.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import s3r  # noqa: E402

spec = s3r.arch_spec


def split3(t):
    t1 = t.to(torch.bfloat16).float()
    t2 = (t - t1).to(torch.bfloat16).float()
    t3 = (t - t1 - t2).to(torch.bfloat16).float()
    return t1, t2, t3


def error_emulation():
    torch.manual_seed(0)
    x = torch.randn(1, 64, 12, 12, 12).relu()
    w = torch.randn(128, 64, 3, 3, 3) / (27 * 64) ** .5
    want = torch.nn.functional.conv3d(x.double(), w.double(), stride=2, padding=1)
    rel = lambda a: float((a - want).norm() / want.norm())
    x1, x2, x3 = split3(x)
    w1, w2, w3 = split3(w)
    assert torch.equal(x1 + x2 + x3, x) and torch.equal(w1 + w2 + w3, w)
    conv = lambda a, b: torch.nn.functional.conv3d(a.double(), b.double(), stride=2, padding=1)      # exact products, fp64 sums
    six = conv(x1, w1) + conv(x1, w2) + conv(x2, w1) + conv(x1, w3) + conv(x2, w2) + conv(x3, w1)
    three = conv(x1, w1) + conv(x1, w2) + conv(x2, w1)
    fp32 = torch.nn.functional.conv3d(x, w, stride=2, padding=1).double()
    print(f"error vs fp64 (truncation only, sums in fp64):  6-product {rel(six):.2e}   3-product {rel(three):.2e}   "
          f"plain bf16 {rel(conv(x1, w1)):.2e}   torch fp32 conv {rel(fp32):.2e}  (the fp32 MFMA kernel: 3e-7 .. 1e-6, tools/wino_numerics.py)")


def gpu_timing(B=32, rounds=7):
    dev = "cuda:0"
    v2 = [l for l in spec.DECODER if l.name == "v2"][0]

    def time_chain(layer, prec, x):
        ch = s3r.modules._HipChain([layer], 28, precision=prec)
        s3r.seed_module(ch, 1)
        ch.to(dev)
        ms = []
        for it in range(rounds + 2):
            s3r.profile_enable(8)
            ch._run(x)
            rec = [r for r in s3r.profile_read(8) if r["family"] == "conv_mfma"]
            s3r.profile_enable(0)
            if it >= 2:
                ms.append(sum(r["ms"] for r in rec))
        ms.sort()
        return ms[len(ms) // 2]

    x = torch.randn(B, 64, 28, 28, 28, device=dev).relu_()
    t32 = time_chain(v2, "fp32", x)
    flops = 2.0 * spec.layer_macs(v2, 28) * B
    print(f"v2 exact-fp32 MFMA kernel, B = {B}: {t32:.4f} ms  ({flops / t32 / 1e9:.1f} TFLOP/s)")
    for mult, name in ((1, "plain bf16"), (3, "3-product split"), (6, "6-product split")):
        layer = spec.Layer("v2", "conv3d", 64 * mult, 128, 3, 2, 1)
        xb = torch.randn(B, 28, 28, 28, 64 * mult, device=dev).relu_().to(torch.bfloat16)      # channels-last physical
        t = time_chain(layer, "bf16", xb.permute(0, 4, 1, 2, 3))
        print(f"v2 as a bf16 convolution over {64 * mult:3d} channels ({name}): {t:.4f} ms  = {flops / t / 1e9:.1f} fp32-equivalent TFLOP/s, "
              f"{t32 / t:.2f}x the fp32 kernel  (bf16 MFMA rate {mult * flops / t / 1e9:.0f} TFLOP/s)")


if __name__ == "__main__":
    error_emulation()
    if torch.cuda.is_available():
        gpu_timing()
