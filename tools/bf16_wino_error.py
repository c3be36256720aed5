#!/usr/bin/env python3
"""What a bf16 Winograd form would cost in accuracy, measured on the CPU in fp64 / fp32 arithmetic: a 3 x 3
stride-1 pad-1 convolution computed (a) directly with operands rounded to bf16 and fp32 accumulation — what conv_bf16*_kernel
does —, (b) as F(4,3) along H with the TRANSFORMED operands rounded to bf16 (the transform itself in fp32, rounded once: the
best a bf16-MFMA Winograd kernel can do), (c) as F(4,3) x F(4,3), (d) as F(2,3) along H; and a 2 x 2 valid correlation — one
parity class of a transposed k4 s2 convolution, two of its three axes — directly and as F(2,2) x F(2,2).  Errors are relative L2
against the fp64 convolution of the UNROUNDED operands; the bf16 path's bar is tests/test_bf16_gpu.py's.
    python tools/bf16_wino_error.py"""
import torch

torch.manual_seed(0)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def bf(t):
    return t.float().bfloat16().double()


def wino(x, w, mats, axes):
    """x (B,C,H,W) zero-padded by 1; Winograd along `axes` (subset of (2, 3)); products on bf16-rounded transformed operands."""
    BT, G, AT = mats
    m, n = AT.shape[0], BT.shape[0]
    B, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    K = w.shape[0]
    if axes == (2,):
        g = H // m
        y = torch.zeros(B, K, H, W, dtype=torch.float64)
        U = bf(torch.einsum("ak,ockw->ocaw", G, w))                               # (K,C,n,3)
        for q in range(g):
            rows = xp[:, :, m * q:m * q + n, :]                                   # (B,C,n,W+2)
            V = bf(torch.einsum("ai,bciw->bcaw", BT, rows).float())               # fp32 transform, rounded once
            M = torch.zeros(B, K, n, W, dtype=torch.float64)
            for tw in range(3):
                M += torch.einsum("ocat,bcaw->boaw", U[:, :, :, tw:tw + 1], V[:, :, :, tw:tw + W])
            y[:, :, m * q:m * q + m, :] = torch.einsum("ua,boaw->bouw", AT, M.float().double())
        return y
    g = H // m
    y = torch.zeros(B, K, H, W, dtype=torch.float64)
    U = bf(torch.einsum("ak,bl,ockl->ocab", G, G, w))
    for q in range(g):
        for s in range(g):
            win = xp[:, :, m * q:m * q + n, m * s:m * s + n]
            V = bf(torch.einsum("ai,bj,bcij->bcab".replace("bcij->bcab", "xcij->xcab"), BT, BT, win).float())
            M = torch.einsum("ocab,xcab->xoab", U, V)
            y[:, :, m * q:m * q + m, m * s:m * s + m] = torch.einsum("ua,vb,xoab->xouv", AT, AT, M.float().double())
    return y


def rel(a, b):
    return float((a - b).norm() / b.norm())


for name, C, K, n in (("e4-like 64->128, 16^2", 64, 128, 16), ("e7-like 256->256, 12^2", 256, 256, 12)):
    for data in ("randn", "relu(randn) (a post-ReLU activation)"):
        x = torch.randn(2, C, n, n, dtype=torch.float64)
        if data != "randn":
            x = x.clamp_min(0)
        w = torch.randn(K, C, 3, 3, dtype=torch.float64) / (3 * C ** .5)
        want = torch.nn.functional.conv2d(x, w, padding=1)
        direct = torch.nn.functional.conv2d(bf(x), bf(w), padding=1)
        res = {"direct bf16": rel(direct, want),
               "F(2,3) along H": rel(wino(x, w, (BT2, G2, AT2), (2,)), want),
               "F(4,3) along H": rel(wino(x, w, (BT4, G4, AT4), (2,)), want),
               "F(4,3) x F(4,3)": rel(wino(x, w, (BT4, G4, AT4), (2, 3)), want)}
        base = res["direct bf16"]
        print(f"{name}, {data}: " + "; ".join(f"{k} {v:.2e} ({v / base:.1f}x)" for k, v in res.items()))


# ---- the transposed layers: one parity class is a 2 x 2 [x 2] correlation; F(2,2) along two of its axes
BT22 = torch.tensor([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=torch.float64)
G22 = torch.tensor([[1, 0], [1, 1], [0, 1]], dtype=torch.float64)
AT22 = torch.tensor([[1, 1, 0], [0, 1, 1]], dtype=torch.float64)
for name, C, K, n in (("d3-like 128->64, 16^2", 128, 64, 16), ("d2-like 256->128, 8^2", 256, 128, 8)):
    for data in ("randn", "relu(randn)"):
        x = torch.randn(2, C, n + 1, n + 1, dtype=torch.float64)
        if data != "randn":
            x = x.clamp_min(0)
        w = torch.randn(K, C, 2, 2, dtype=torch.float64) / (2 * C ** .5)
        want = torch.nn.functional.conv2d(x, w)
        direct = torch.nn.functional.conv2d(bf(x), bf(w))
        xb = bf(x)                                                    # the activation tensor IS bf16; differences formed in fp32
        U = bf(torch.einsum("ak,bl,ockl->ocab", G22, G22, w))
        y = torch.zeros_like(want)
        for q in range(n // 2):
            for s in range(n // 2):
                V = bf(torch.einsum("ai,bj,xcij->xcab", BT22, BT22, xb[:, :, 2 * q:2 * q + 3, 2 * s:2 * s + 3]).float())
                M = torch.einsum("ocab,xcab->xoab", U, V)
                y[:, :, 2 * q:2 * q + 2, 2 * s:2 * s + 2] = torch.einsum("ua,vb,xoab->xouv", AT22, AT22, M.float().double())
        print(f"{name}, {data}: direct bf16 {rel(direct, want):.2e}; F(2,2) x F(2,2) {rel(y, want):.2e} "
              f"({rel(y, want) / rel(direct, want):.1f}x)")
