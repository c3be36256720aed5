"""Known-answer vectors for ONE Conv2d and ONE ConvTranspose3d layer, for the compiled-C consumer of include/s3r.h
(tests/c_abi/consumer_gpu.c, tests/test_c_abi_gpu.py).  Produced by the ORACLE's layer block (torch CPU fp32: conv + eval
BatchNorm + ReLU) — "parity unpinned": /root/reference holds no code to generate vectors from (README.md:5).

    python tests/golden/make_conv_kat.py      # rewrites tests/golden/conv_kat.npz

Inputs and parameters are stored (they are small), so the C program needs nothing but raw arrays.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import s3r  # noqa: E402
from oracle import s2v_oracle as O  # noqa: E402

CASES = {
    # name: (arch_spec.Layer, batch, edge)
    "c2d": (s3r.arch_spec.Layer("c2d", "conv2d", 32, 48, 3, 1, 1), 2, 8),
    "dc3": (s3r.arch_spec.Layer("dc3", "deconv3d", 32, 32, 4, 2, 1), 1, 4),
}


def main():
    out = {}
    for name, (layer, batch, n) in CASES.items():
        g = torch.Generator().manual_seed(11)
        blk = O._Block(layer).eval()
        with torch.no_grad():
            for p in blk.conv.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            blk.bn.weight.copy_(torch.rand(layer.cout, generator=g) + 0.5)
            blk.bn.bias.copy_(torch.randn(layer.cout, generator=g) * 0.1)
            blk.bn.running_mean.copy_(torch.randn(layer.cout, generator=g) * 0.1)
            blk.bn.running_var.copy_(torch.rand(layer.cout, generator=g) + 0.5)
            nd = s3r.arch_spec.ndim(layer)
            x = torch.randn((batch, layer.cin) + (n,) * nd, generator=g)
            y = blk(x)
        out.update({f"{name}_x": x.numpy(), f"{name}_w": blk.conv.weight.detach().numpy(), f"{name}_b": blk.conv.bias.detach().numpy(),
                    f"{name}_gamma": blk.bn.weight.detach().numpy(), f"{name}_beta": blk.bn.bias.detach().numpy(),
                    f"{name}_mean": blk.bn.running_mean.numpy(), f"{name}_var": blk.bn.running_var.numpy(), f"{name}_y": y.numpy()})
    np.savez_compressed(os.path.join(HERE, "conv_kat.npz"), **out)
    print("wrote conv_kat.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
