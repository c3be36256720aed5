"""Parameter-general layers (ABI 8, csrc/s3r_general.hip): any fp32 Conv2d / Conv3d / ConvTranspose2d / ConvTranspose3d — kernel size,
stride, padding, dilation, output padding, channel counts that are not multiples of 16, LeakyReLU / ELU / Tanh — runs through the
direct kernel (behind a staging pass where the tuned paths do not reach) and must agree with the oracle's block, i.e. with
torch.nn on the CPU, at the path's fp32 tolerance.  The shapes are NOT this build's network: they are what a reference layer table
may hold the day tools/resurvey.py prints it."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _check(s3r, oracle, layers, n_in, B, seed=0):
    dev, spec = "cuda:0", s3r.arch_spec
    ch = s3r.modules._HipChain(layers, n_in, precision="fp32")
    s3r.seed_module(ch, 11 + seed)
    blocks = [oracle._Block(l).eval() for l in layers]
    for l, blk in zip(layers, blocks):
        blk.load_state_dict(getattr(ch, l.name).state_dict())
    ch.to(dev)
    x = torch.randn((B, layers[0].cin) + (n_in,) * spec.ndim(layers[0]), generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        want = x
        for blk in blocks:
            want = blk(want)
    got = ch._run(x.to(dev))
    assert tuple(got.shape) == tuple(want.shape), (layers, n_in, got.shape, want.shape)
    assert torch.equal(got, ch._run(x.to(dev))), (layers, "determinism")
    assert torch.equal(ch._run(x[B - 1:].to(dev))[0], got[B - 1]), (layers, "batch invariance")
    rel = _rel(got.cpu(), want)
    assert rel < 1e-5, (layers, n_in, rel)


def test_named_shapes(s3r, oracle):
    L = s3r.arch_spec.Layer
    cases = [
        ([L("a", "conv2d", 3, 8, 7, 2, 3)], 33, 2),                                   # a ResNet-style stem: k7 s2 p3, Cin = 3
        ([L("a", "conv2d", 16, 24, 5, 1, 2)], 12, 3),                                 # k5
        ([L("a", "conv2d", 32, 32, 3, 1, 2, True, "relu", 2)], 16, 2),                 # dilation 2 (same size)
        ([L("a", "conv3d", 16, 16, 3, 1, 3, True, "none", 3)], 9, 1),                  # dilation 3 in 3D
        ([L("a", "conv3d", 5, 7, 1, 1, 0)], 6, 2),                                    # 1 x 1 x 1, odd channels
        ([L("a", "conv2d", 20, 40, 3, 2, 1, True, "leaky_relu")], 15, 2),              # Cin % 16 != 0, LeakyReLU (default slope)
        ([L("a", "conv2d", 16, 16, 3, 1, 1, False, "leaky_relu", 1, 0, 0.2)], 8, 1),   # slope 0.2, no BN
        ([L("a", "conv3d", 32, 32, 3, 1, 1, True, "elu")], 8, 2),                      # ELU on a layer that otherwise has Winograd forms
        ([L("a", "conv2d", 32, 16, 3, 1, 1, True, "tanh")], 8, 2),
        ([L("a", "deconv2d", 16, 8, 4, 2, 1)], 7, 2),                                 # the usual 2D upsampler
        ([L("a", "deconv2d", 12, 20, 3, 2, 1, True, "relu", 1, 1)], 6, 3),             # k3 s2 p1 output_padding 1
        ([L("a", "deconv2d", 8, 8, 2, 2, 0)], 5, 1),                                  # k2 s2 p0
        ([L("a", "deconv3d", 16, 16, 3, 1, 1)], 6, 2),                                # stride 1 (a flipped convolution)
        ([L("a", "deconv3d", 8, 12, 4, 2, 1)], 5, 2),                                 # the network's shape but Cin % 16 != 0
        ([L("a", "deconv3d", 16, 8, 5, 3, 2, True, "tanh", 1, 2)], 4, 1),              # k5 s3 p2 output_padding 2
        ([L("a", "deconv2d", 16, 16, 3, 2, 2, True, "none", 2, 1)], 6, 2),             # dilation 2, p2, output_padding 1
        ([L("a", "conv2d", 3, 16, 3, 1, 1), L("b", "conv2d", 16, 10, 3, 2, 1, True, "elu"), L("c", "deconv2d", 10, 6, 4, 2, 1)], 16, 2),
        ([L("a", "conv3d", 4, 16, 3, 1, 1), L("b", "conv3d", 16, 16, 3, 1, 1), L("c", "deconv3d", 16, 4, 2, 2, 0, False, "tanh")], 8, 2),
    ]
    for i, (layers, n_in, B) in enumerate(cases):
        _check(s3r, oracle, layers, n_in, B, seed=i)


def _exec_ratio(s3r, layer, n_in, B):
    """FLOPs the layer's kernels execute on the matrix cores / its algorithmic FLOPs (the library's own profiler record)"""
    ch = s3r.modules._HipChain([layer], n_in, precision="fp32")
    s3r.seed_module(ch, 3)
    ch.to("cuda:0")
    x = torch.randn((B, layer.cin) + (n_in,) * s3r.arch_spec.ndim(layer), device="cuda:0")
    s3r.profile_enable(16)
    ch._run(x)
    rec = [r for r in s3r.profile_read(16) if r["family"] == "conv_mfma"]
    s3r.profile_enable(0)
    return sum(r["exec_flops"] for r in rec) / sum(r["flops"] for r in rec), rec


def test_transposed_layers_run_as_residue_classes(s3r, oracle):
    """r06 (VERDICT r05 #5c): ANY ConvTranspose with dilation 1 is stride^ndim residue classes, each a stride-1 convolution launch over
    the halo-padded input — not a zero-stuffed tensor convolved with the whole kernel (stride^ndim times the multiplications).  Values
    against torch.nn for kernels smaller / equal / larger than the stride, strides 1-4, output padding, classes without a tap; and the
    executed multiplications stay within 5 % of the algorithmic count (channel counts that are multiples of 16)."""
    L = s3r.arch_spec.Layer
    cases = [
        ([L("a", "deconv2d", 64, 32, 4, 2, 1)], 16, 2),                                # the usual upsampler, cin % 16 == 0
        ([L("a", "deconv2d", 32, 32, 2, 2, 0)], 9, 2),                                 # k = s: one tap per class
        ([L("a", "deconv2d", 16, 16, 2, 3, 0)], 5, 2),                                 # k < s: one class per axis has NO tap (bias only)
        ([L("a", "deconv2d", 16, 16, 1, 2, 0, True, "relu", 1, 1)], 6, 1),              # k1 s2 + output padding
        ([L("a", "deconv2d", 16, 24, 5, 2, 2, True, "relu", 1, 1)], 7, 3),              # k5 s2 p2 op1 (3 + 2 taps)
        ([L("a", "deconv2d", 32, 16, 7, 4, 3, True, "none", 1, 3)], 5, 1),              # k7 s4 p3 op3
        ([L("a", "deconv3d", 32, 32, 2, 2, 0)], 6, 2),                                 # 3D k2 s2
        ([L("a", "deconv3d", 16, 16, 3, 2, 1, True, "relu", 1, 1)], 5, 2),              # 3D k3 s2 p1 op1
        ([L("a", "deconv3d", 16, 8, 4, 3, 0)], 4, 1),                                  # 3D k4 s3
        ([L("a", "deconv3d", 48, 32, 3, 1, 1)], 7, 2),                                 # stride 1: a flipped convolution, one class
        ([L("a", "deconv2d", 20, 12, 4, 2, 1, False, "elu")], 11, 2),                  # odd channels + ELU + no BN
        # k == stride, pad 0: ONE GEMM over (cout, tap) rows with a depth-to-space store (no class launches)
        ([L("a", "deconv2d", 32, 24, 3, 3, 0)], 7, 2),
        ([L("a", "deconv2d", 16, 40, 4, 4, 0, True, "sigmoid")], 5, 3),
        ([L("a", "deconv3d", 24, 10, 2, 2, 0, True, "leaky_relu")], 5, 2),              # + odd channels (staged for the channel padding only)
        ([L("a", "deconv3d", 16, 72, 3, 3, 0, False, "none")], 3, 1),                   # 27 taps x 72 couts = 1944 GEMM rows
    ]
    for i, (layers, n_in, B) in enumerate(cases):
        _check(s3r, oracle, layers, n_in, B, seed=80 + i)
    for layer, n_in, B in ((L("a", "deconv2d", 64, 32, 4, 2, 1), 32, 4), (L("a", "deconv2d", 32, 32, 2, 2, 0), 28, 4),
                           (L("a", "deconv3d", 32, 32, 2, 2, 0), 14, 2), (L("a", "deconv3d", 16, 16, 3, 2, 1, True, "relu", 1, 1), 14, 2),
                           (L("a", "deconv2d", 48, 48, 3, 2, 1, True, "relu", 1, 1), 28, 4)):
        ratio, rec = _exec_ratio(s3r, layer, n_in, B)
        assert ratio <= 1.05, (layer, n_in, ratio)
    # channel counts that are not multiples of 16 pay their zero rows: 24 -> 32 is 4/3, no more
    ratio, _ = _exec_ratio(s3r, L("a", "deconv2d", 24, 24, 4, 2, 1), 28, 4)
    assert ratio <= 1.05 * 32 / 24
    # dilation > 1 keeps the zero-stuffed form (documented in include/s3r.h): stride^ndim
    ratio, _ = _exec_ratio(s3r, L("a", "deconv2d", 16, 16, 3, 2, 2, True, "none", 2, 1), 12, 2)
    assert 3.0 < ratio < 5.0


def test_residue_class_layers_read_their_producers_halo_in_place(s3r, oracle):
    """Behind another layer of a chain a residue-class ConvTranspose with cin % 16 == 0 is NOT staged: the chain gives its input
    the halo the classes read (want_halo) and the class kernels read it in place — same values, no staging pass, no scratch."""
    import ctypes as C
    L = s3r.arch_spec.Layer
    lib = s3r.load_library()
    chains = [
        ([L("a", "conv2d", 16, 32, 3, 1, 1), L("b", "deconv2d", 32, 16, 4, 2, 1), L("c", "deconv2d", 16, 8, 2, 2, 0, True, "none")], 12, 2),
        ([L("a", "conv3d", 16, 32, 3, 1, 1), L("b", "deconv3d", 32, 16, 2, 2, 0), L("c", "deconv3d", 16, 16, 3, 2, 1, True, "tanh", 1, 1)], 6, 2),
        ([L("a", "deconv2d", 32, 32, 5, 3, 1), L("b", "deconv2d", 32, 16, 3, 1, 1)], 5, 1),
        # producers that run Winograd forms (finish kernels that build padded planes) writing the WIDER halos such consumers read
        ([L("a", "conv2d", 32, 32, 3, 1, 1), L("b", "deconv2d", 32, 16, 7, 2, 0)], 8, 2),             # two-axis 2D producer, halo 3
        ([L("a", "conv3d", 32, 32, 3, 1, 1), L("b", "deconv3d", 32, 16, 7, 2, 0, True, "tanh")], 8, 1), # two-axis 3D producer, halo 3
        ([L("a", "conv2d", 32, 64, 3, 1, 1), L("b", "deconv2d", 64, 16, 4, 2, 1)], 40, 2),            # one-axis F(4,3) producer (edge 40), halo 1
        ([L("a", "deconv3d", 32, 32, 4, 2, 1), L("b", "deconv2d", 32, 16, 4, 2, 1)][:1] + [L("c", "deconv3d", 32, 16, 5, 2, 2, True, "relu", 1, 1)], 4, 2),  # tuned transposed producer, halo 1
    ]
    for i, (layers, n_in, B) in enumerate(chains):
        _check(s3r, oracle, layers, n_in, B, seed=90 + i)
    d = s3r._lib.make_desc(L("b", "deconv2d", 32, 16, 4, 2, 1), 2, 12, in_halo=1)
    assert lib.s3r_conv_scratch_elems(C.byref(d)) == 0                       # halo 1 is what k4 s2 p1 reads: in place
    d0 = s3r._lib.make_desc(L("b", "deconv2d", 32, 16, 4, 2, 1), 2, 12, in_halo=0)
    assert lib.s3r_conv_scratch_elems(C.byref(d0)) == -(-(2 * 32 * 14 * 14) // 256) * 256      # unpadded input: the staged copy
    # the profile of the chain's second and third layer: the class launch only, no staging pass
    ch = s3r.modules._HipChain(chains[0][0], 12, precision="fp32")
    s3r.seed_module(ch, 1)
    ch.to("cuda:0")
    s3r.profile_enable(32)
    ch._run(torch.randn(2, 16, 12, 12, device="cuda:0"))
    rec = [r for r in s3r.profile_read(32) if r["family"] == "conv_mfma"]
    s3r.profile_enable(0)
    assert [r["launches"] for r in rec][1:] == [1, 1], rec       # (b: four residue classes in ONE launch over a class table; c: k == stride,
                                                                 #  one depth-to-space GEMM) — and no staging pass in front of either


def test_rgb_first_layers_are_unfolded_and_leaky_relu_is_fused(s3r, oracle):
    """r06, what a DispNet-style front end holds: Conv2d 3 -> 64 k7 s2 p3 + LeakyReLU(0.1).  cin <= 8 layers are staged as im2col — a 1 x 1
    GEMM over cin k^nd rows (147 -> 160) instead of 16 channel rows per tap of which 13 are zero (5.3 x the multiplications) — and
    LeakyReLU with a slope in [0, 1] is max(t, slope t) in the direct kernel's epilogue (and its split-K finish), not a pass of its own."""
    L = s3r.arch_spec.Layer
    cases = [
        ([L("a", "conv2d", 3, 64, 7, 2, 3, True, "leaky_relu", 1, 0, 0.1)], 64, 2),
        ([L("a", "conv2d", 6, 32, 5, 2, 2, False, "leaky_relu", 1, 0, 0.1)], 40, 3),
        ([L("a", "conv3d", 1, 16, 3, 1, 1)], 12, 2),
        ([L("a", "conv3d", 2, 24, 4, 2, 1, True, "elu")], 10, 2),
        ([L("a", "conv2d", 3, 16, 3, 1, 2, True, "relu", 2)], 17, 2),                                  # dilation 2 through the unfolding
        ([L("a", "conv2d", 32, 48, 3, 2, 1, True, "leaky_relu", 1, 0, 0.2)], 20, 2),                  # fused on a plain direct layer
        ([L("a", "conv3d", 128, 32, 3, 1, 1, True, "leaky_relu", 1, 0, 1.0)], 6, 2),                  # slope 1: identity; deep K: split-K finish
        ([L("a", "conv2d", 32, 32, 3, 1, 1, True, "leaky_relu", 1, 0, 0.0)], 8, 2),                   # slope 0: ReLU
        ([L("a", "conv2d", 16, 16, 3, 1, 1, True, "leaky_relu", 1, 0, 1.5)], 8, 2),                   # slope > 1 is min(t, slope t): the pass
        ([L("a", "deconv2d", 32, 16, 4, 2, 1, True, "leaky_relu", 1, 0, 0.1)], 9, 2),                 # residue classes + fused LeakyReLU
        ([L("a", "deconv2d", 16, 16, 2, 2, 0, True, "leaky_relu", 1, 0, 0.1)], 9, 2),                 # depth-to-space store + fused LeakyReLU
    ]
    for i, (layers, n_in, B) in enumerate(cases):
        _check(s3r, oracle, layers, n_in, B, seed=120 + i)
    ratio, rec = _exec_ratio(s3r, L("a", "conv2d", 3, 64, 7, 2, 3, True, "leaky_relu", 1, 0, 0.1), 224, 8)
    assert ratio <= 160 / 147 + 1e-6, ratio
    assert sum(r["launches"] for r in rec) == 2, rec                   # the unfolding pass + ONE GEMM launch: no activation pass
    with s3r.debug_overrides(ksplit={"a": 2}):
        _check(s3r, oracle, [L("a", "conv3d", 64, 32, 3, 2, 1, True, "leaky_relu", 1, 0, 0.3)], 9, 2, seed=140)      # split-K finish applies it
    # a batch whose unfolded copy passes 1 GiB goes through in sub-batches (here 16 + 4 samples of 64 MB each): same values, and a
    # sample's bits do not depend on the pass it went through (_check compares the last sample alone with it inside the batch)
    _check(s3r, oracle, [L("a", "conv2d", 8, 16, 7, 1, 3, True, "leaky_relu", 1, 0, 0.1)], 200, 20, seed=141)


def test_linear_layers_take_every_activation(s3r, oracle):
    """a linear layer with LeakyReLU / ELU / Tanh ran with NO activation (the LINEAR branch of geometry() returned before
    the range check and the linear epilogue knows none / ReLU / sigmoid only).  They are a pass behind the layer now."""
    L = s3r.arch_spec.Layer
    for i, (act, par) in enumerate((("leaky_relu", None), ("leaky_relu", 0.3), ("elu", 0.7), ("tanh", None), ("relu", None), ("sigmoid", None),
                                    ("none", None))):
        _check(s3r, oracle, [L("a", "linear", 96, 40, 1, 1, 0, False, act, 1, 0, par)], 1, 3, seed=40 + i)
    _check(s3r, oracle, [L("a", "linear", 64, 64, 1, 1, 0, False, "elu"), L("b", "linear", 64, 10, 1, 1, 0, False, "tanh")], 1, 2, seed=50)
    # the flat entry has no parameter argument: it refuses what it cannot express instead of dropping it
    import ctypes as C
    lib = s3r.load_library()
    x = torch.zeros(2, 16, device="cuda:0"); w = torch.zeros(8, 16, device="cuda:0"); y = torch.zeros(2, 8, device="cuda:0")
    sc = torch.zeros(max(1, lib.s3r_linear_scratch_elems(2, 16, 8)), device="cuda:0")
    rc = lib.s3r_linear_forward(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 2, 16, 8, 3, sc.data_ptr(), sc.numel(), None)
    assert rc == -1 and b"LeakyReLU" in lib.s3r_last_error()


def test_staged_layers_build_halos_of_any_width(s3r, oracle):
    """the staged copy builds its own halo, so its effective padding is not bound by the caller-halo limit of 8"""
    L = s3r.arch_spec.Layer
    _check(s3r, oracle, [L("a", "deconv2d", 16, 8, 7, 1, 0, True, "relu", 2)], 6, 2, seed=60)          # pe = 2 * 6 - 0 = 12
    _check(s3r, oracle, [L("a", "conv2d", 10, 12, 7, 1, 9, True, "none", 3)], 9, 1, seed=61)           # cin % 16 != 0, pad 9
    _check(s3r, oracle, [L("a", "deconv3d", 8, 8, 5, 2, 0, False, "tanh", 3)], 3, 1, seed=62)          # pe = 12 in 3D


def test_split_k_is_honoured_on_general_layers_that_are_not_staged(s3r, oracle):
    """a caller-forced split-K on a cin % 16 == 0 layer with LeakyReLU / ELU / Tanh was silently ignored by the forward
    while the scratch query planned its slabs; now the layer runs the normal direct path (split as asked) + the activation pass"""
    import ctypes as C
    L = s3r.arch_spec.Layer
    layer = L("a", "conv3d", 64, 32, 3, 2, 1, True, "elu")
    lib = s3r.load_library()
    d1 = s3r._lib.make_desc(layer, 2, 9, in_halo=1)
    d4 = s3r._lib.make_desc(layer, 2, 9, in_halo=1, ksplit=4)
    assert lib.s3r_conv_scratch_elems(C.byref(d4)) > lib.s3r_conv_scratch_elems(C.byref(d1)) >= 0
    with s3r.debug_overrides(ksplit={"a": 4}):
        _check(s3r, oracle, [layer], 9, 2, seed=70)
    with s3r.debug_overrides(ksplit={"a": 1}):
        _check(s3r, oracle, [layer], 9, 2, seed=70)


def test_random_sweep(s3r, oracle):
    """A seeded sweep over the parameter space (what a property-based run would draw; kept deterministic so that a failure names
    its case)."""
    L = s3r.arch_spec.Layer
    rng = random.Random(2025)
    done = 0
    while done < 48:
        nd = rng.choice((2, 3))
        tr = rng.random() < 0.4
        k = rng.choice((1, 2, 3, 4, 5, 7) if nd == 2 else (1, 2, 3, 4, 5))
        s = rng.choice((1, 2, 3) if tr else (1, 2))
        dil = rng.choice((1, 1, 1, 2))
        cin = rng.choice((1, 3, 8, 16, 20, 32, 48))
        cout = rng.choice((2, 7, 16, 33, 64, 70))
        act = rng.choice(("none", "relu", "sigmoid", "leaky_relu", "elu", "tanh"))
        if tr:
            p = rng.randrange(0, dil * (k - 1) + 1)
            op = rng.randrange(0, max(s, dil))
            layer = L("g", "deconv%dd" % nd, cin, cout, k, s, p, rng.random() < 0.7, act, dil, op)
        else:
            p = rng.randrange(0, min(4, dil * (k - 1)) + 1)
            layer = L("g", "conv%dd" % nd, cin, cout, k, s, p, rng.random() < 0.7, act, dil)
        n_in = rng.randrange(3, 13 if nd == 3 else 24)
        n_out = s3r.arch_spec.out_size(layer, n_in)
        if n_out < 1 or n_out > (40 if nd == 3 else 96) or (cout == 1 and k == 1):
            continue
        _check(s3r, oracle, [layer], n_in, rng.choice((1, 2, 3)), seed=done)
        done += 1
