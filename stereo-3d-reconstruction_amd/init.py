"""Seeded random initialisation shared by the HIP modules and the tests' oracle.

BASELINE.json configs[0] runs on random-init weights (no checkpoint can be fetched here), so the
init must (a) be reproducible from a seed on any host and (b) keep activations O(1) through 18
layers so that parity tests exercise real signal instead of a sigmoid(0) plateau.  It works on a
`state_dict` by key, so any module with the arch_spec key layout (the HIP modules, the oracle)
ends up with bit-identical parameters for the same seed.
"""
from __future__ import annotations

import math

import torch


def seeded_state_dict(module: torch.nn.Module, seed: int = 0, randomize_bn: bool = True):
    """Return a new state_dict for `module` (He-normal convs, small biases, non-trivial BN stats)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for key, ref in module.state_dict().items():
        shape, leaf = tuple(ref.shape), key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=ref.dtype)
        elif _is_batchnorm(module, key):          # (by module TYPE, not by name: any key layout, tools/resurvey.py)
            if not randomize_bn:
                t = torch.ones(shape) if leaf in ("weight", "running_var") else torch.zeros(shape)
            elif leaf == "weight":
                t = 0.75 + 0.5 * torch.rand(shape, generator=g)
            elif leaf == "running_var":
                t = 0.5 + torch.rand(shape, generator=g)
            else:                                   # bias, running_mean
                t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "weight":
            w = torch.randn(shape, generator=g)
            if "deconv" in key or _is_transposed(module, key):
                # ConvTranspose weight is (Cin, Cout, k..): each output sums Cin*k^d/s^d taps
                fan_in = shape[0] * math.prod(shape[2:]) / 8.0
            else:
                fan_in = math.prod(shape[1:])
            t = w * math.sqrt(2.0 / fan_in)
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            raise KeyError(f"unexpected state_dict key {key}")
        out[key] = t.to(ref.dtype)
    return out


def _owner(module, key):
    sub = module
    for part in key.split(".")[:-1]:
        sub = getattr(sub, part)
    return sub


def _is_batchnorm(module, key):
    return isinstance(_owner(module, key), torch.nn.modules.batchnorm._BatchNorm)


def _is_transposed(module, key):
    sub = module
    for part in key.split(".")[:-1]:
        sub = getattr(sub, part)
    return isinstance(sub, (torch.nn.ConvTranspose3d, torch.nn.ConvTranspose2d))


def seed_module(module: torch.nn.Module, seed: int = 0, randomize_bn: bool = True):
    sd = seeded_state_dict(module, seed, randomize_bn)
    module.load_state_dict({k: v.to(module.state_dict()[k].device) for k, v in sd.items()})
    return module


def synthetic_pairs(batch: int, seed: int = 0, device="cpu"):
    """left,right = torch.rand(B,3,224,224) with generator seeds (seed, seed+1)."""
    gl = torch.Generator(device="cpu").manual_seed(seed)
    gr = torch.Generator(device="cpu").manual_seed(seed + 1)
    left = torch.rand(batch, 3, 224, 224, generator=gl)
    right = torch.rand(batch, 3, 224, 224, generator=gr)
    return left.to(device), right.to(device)
