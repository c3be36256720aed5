"""bf16 MFMA path (BASELINE.json configs[2]): channels-last bf16 activations, fp32 accumulation.

The oracle stays fp32.  Per-layer tests feed it the SAME bf16-rounded inputs and weights the HIP kernel
sees, so what is compared is accumulation order + the final rounding to bf16 (one bf16 ulp = 2^-8
relative).  The end-to-end test reports how far 18 layers of bf16 activations drift from the fp32 oracle;
north_star's criterion for this path is voxel IoU, not 1e-4.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-30)).item()


def _bf(x):
    return x.to(torch.bfloat16).float()


def _layer_cases(spec):
    cases = []
    for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP)):
        for l, n_in, _ in spec.trace(layers, n0):
            cases.append((l, n_in))
    return cases


@pytest.fixture(params=[32, 16], ids=["mfma32x32x16", "mfma16x16x32"])
def mfma_shape(request, monkeypatch):
    """Both matrix instructions the kernels are built for (S3R_BF16_MFMA; the library's default is whichever is the
    faster by wall on random data, s3r_conv_bf16.hip): every per-layer / per-tile check runs under each."""
    monkeypatch.setenv("S3R_BF16_MFMA", str(request.param))
    return request.param


@pytest.mark.parametrize("idx", range(18))
def test_each_layer_bf16_vs_oracle(s3r, oracle, idx, mfma_shape):
    spec = s3r.arch_spec
    layer, n_in = _layer_cases(spec)[idx]
    B = 2
    ch = s3r.modules._HipChain([layer], n_in, precision="bf16")
    s3r.seed_module(ch, 7)
    blk = oracle._Block(layer).eval()
    sd = getattr(ch, layer.name).state_dict()
    stem, head = layer.name == "e1", layer.name == "d4"
    if not stem and not head:                      # the MFMA layers see bf16 weights
        sd = dict(sd)
        sd["conv.weight"] = _bf(sd["conv.weight"])
    blk.load_state_dict(sd)
    g = torch.Generator().manual_seed(idx)
    x = torch.randn((B, layer.cin) + (n_in,) * spec.ndim(layer), generator=g)
    if not stem:
        x = _bf(x)
    with torch.no_grad():
        want = blk(x)
    ch.to(DEV)
    if stem:
        got = ch._run(x.to(DEV))
    else:
        xin = x.to(DEV).to(torch.bfloat16)
        xin = xin.permute(0, *range(2, xin.dim()), 1).contiguous()          # physical channels-last
        got = ch._run(xin)
    assert got.shape == want.shape
    if head:
        assert got.dtype == torch.float32
        assert rel_l2(got.cpu(), want) < 1e-5
        return
    assert got.dtype == torch.bfloat16
    # bf16 output: each element is the fp32 result rounded to 8 significant bits
    assert rel_l2(got.cpu(), want) < 3e-3, (layer.name, rel_l2(got.cpu(), want))
    err = (got.cpu().float() - want).abs()
    assert (err <= 2.0 ** -7 * want.abs() + 1e-3 * want.abs().max()).all(), layer.name


# 1, 2, 4: per-tap gather (64-channel K tiles where Cin allows, +16 = 32-channel), 3: its 128 x 128-cout tile;
# 5, 6 / 21, 22: plane-reuse gather, 23: its 256 x 128-cout tile;
# 9, 10: row-reuse gather (x128 / x256 positions)
@pytest.mark.parametrize("tm", [1, 2, 3, 4, 5, 6, 9, 10, 17, 18, 19, 21, 22, 23])
@pytest.mark.parametrize("kind", ["conv3d_s1", "conv3d_s2", "deconv", "conv2d_s2", "conv3d_k4_valid_ks2", "cout32",
                                  "conv2d_s1_w28", "conv3d_s1_w14", "conv3d_s1_c64", "deconv_c64", "deconv_c128_w8",
                                  "conv3d_k4_valid_c128_ks2", "conv2d_s1_c128_w9", "cout36_narrow_stores", "cout100_s2", "c128_cout128_ks2"])
def test_bf16_tiles_and_split_k(s3r, oracle, tm, kind, mfma_shape):
    Layer = s3r.arch_spec.Layer
    layer, n_in, B, ks = {
        "conv3d_s1": (Layer("t", "conv3d", 32, 96, 3, 1, 1), 7, 3, 0),
        "conv3d_s2": (Layer("t", "conv3d", 64, 160, 3, 2, 1), 9, 2, 0),
        "deconv": (Layer("t", "deconv3d", 32, 48, 4, 2, 1), 5, 3, 0),
        "conv2d_s2": (Layer("t", "conv2d", 64, 64, 3, 2, 1), 13, 5, 0),
        "conv3d_k4_valid_ks2": (Layer("t", "conv3d", 64, 40, 4, 1, 0), 7, 2, 2),
        "cout32": (Layer("t", "conv2d", 256, 32, 1, 1, 0), 9, 3, 4),
        "conv2d_s1_w28": (Layer("t", "conv2d", 64, 64, 3, 1, 1), 28, 3, 0),
        "conv3d_s1_w14": (Layer("t", "conv3d", 32, 64, 3, 1, 1), 14, 2, 0),
        "conv3d_s1_c64": (Layer("t", "conv3d", 64, 96, 3, 1, 1), 7, 3, 0),
        "deconv_c64": (Layer("t", "deconv3d", 64, 48, 4, 2, 1), 5, 3, 0),
        "deconv_c128_w8": (Layer("t", "deconv3d", 128, 64, 4, 2, 1), 8, 2, 0),
        "conv3d_k4_valid_c128_ks2": (Layer("t", "conv3d", 128, 40, 4, 1, 0), 7, 2, 2),
        "conv2d_s1_c128_w9": (Layer("t", "conv2d", 128, 64, 3, 1, 1), 9, 21, 0),
        "cout36_narrow_stores": (Layer("t", "conv2d", 64, 36, 3, 1, 1), 9, 3, 0),    # Cout % 8 != 0: 2-byte stores
        "cout100_s2": (Layer("t", "conv3d", 32, 100, 3, 2, 1), 9, 2, 0),             # Cout % 8 != 0, two cout tiles
        "c128_cout128_ks2": (Layer("t", "conv3d", 128, 128, 3, 1, 1), 6, 2, 2),       # split-K under the 128-cout tiles
    }[kind]
    ch = s3r.modules._HipChain([layer], n_in, precision="bf16")
    s3r.seed_module(ch, 7)
    ch.tile_override["t"] = tm
    if ks:
        ch.ksplit_override["t"] = ks
    blk = oracle._Block(layer).eval()
    sd = dict(ch.t.state_dict())
    sd["conv.weight"] = _bf(sd["conv.weight"])
    blk.load_state_dict(sd)
    x = _bf(torch.randn((B, layer.cin) + (n_in,) * s3r.arch_spec.ndim(layer), generator=torch.Generator().manual_seed(3)))
    with torch.no_grad():
        want = blk(x)
    xin = x.to(DEV).to(torch.bfloat16)
    xin = xin.permute(0, *range(2, xin.dim()), 1).contiguous()
    kc = 64 if tm in (5, 6) else 32            # plane-reuse gather: 5, 6 = 64-channel K tiles, 21, 22 = 32-channel
    plane_ok = (layer.s == 1 or layer.op == "deconv3d") and layer.cin % kc == 0 and (ks == 0 or (layer.cin // kc) % ks == 0)
    if tm in (6, 22, 23) and kind in ("conv3d_k4_valid_c128_ks2", "conv3d_k4_valid_ks2"):
        plane_ok = False                       # 16-position planes: a 256-position tile spans 17 of them (+ halos)
    if tm == 6 and kind == "c128_cout128_ks2":
        plane_ok = False                       # 36-position planes: 506 image rows of 128 B, above the 480-row cap
    wide_ok = (-(-layer.cout // 64)) % 2 == 0          # 128-cout workgroup tiles need an even number of 64-cout tiles
    if (tm == 10 and kind in ("conv3d_s2", "cout100_s2")) or (tm in (5, 6, 21, 22, 23) and not plane_ok) or \
            (tm in (3, 19, 23) and not wide_ok):
        # 5-wide rows at stride 2: the 256-position reuse image exceeds its LDS budget; the plane-reuse gather
        # needs stride 1 and whole 64-channel chunks per split
        with pytest.raises(s3r.S3RError):
            ch.to(DEV)._run(xin)
        return
    got = ch.to(DEV)._run(xin)
    assert rel_l2(got.cpu(), want) < 3e-3
    assert torch.equal(ch._run(xin), got)


def test_cost_volume_bf16_bit_exact(s3r, oracle):
    g = torch.Generator().manual_seed(2)
    fl, fr = _bf(torch.randn(3, 32, 28, 28, generator=g)), _bf(torch.randn(3, 32, 28, 28, generator=g))
    want = oracle.cost_volume(fl, fr).to(torch.bfloat16)          # fp32 difference of bf16 values, rounded once
    cv = s3r.CostVolume(precision="bf16")
    a = fl.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = fr.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = cv(a, b)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert torch.equal(got.cpu(), want)
    vp = cv.forward_padded(a, b)                                    # physical (B,30,30,30,64), zero halo
    assert vp.shape == (3, 30, 30, 30, 64)
    assert torch.equal(vp[:, 1:-1, 1:-1, 1:-1, :].permute(0, 4, 1, 2, 3).cpu(), want)
    border = vp.clone()
    border[:, 1:-1, 1:-1, 1:-1, :] = 0
    assert border.abs().max().item() == 0


def test_stereo2voxel_bf16_vs_fp32_oracle(s3r, oracle, mfma_shape):
    hip = s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(hip, 0)
    ref = oracle.OracleStereo2Voxel().eval()
    ref.load_state_dict(hip.state_dict())               # same fp32 state_dict on both sides
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(4, seed=0)
    with torch.no_grad():
        want = ref(left, right)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    assert got.shape == (4, 32, 32, 32) and got.dtype == torch.float32
    r, m = rel_l2(got, want), (got - want).abs().max().item()
    iou = oracle.voxel_iou(got, want).min().item()
    print(f"bf16 path vs fp32 oracle: rel_l2={r:.3e} max|d|={m:.3e} IoU@0.5(min over samples)={iou:.5f}")
    assert r < 2e-2 and m < 5e-2
    assert iou > 0.98
    # batch-invariance and determinism hold on this path too
    one = hip(left[2:3].to(DEV), right[2:3].to(DEV)).cpu()
    assert torch.equal(one[0], got[2])


def test_bf16_mfma_layout_exact_integers(s3r, mfma_shape):
    """A 1x1 conv on small-integer data is exact in bf16 x bf16 -> fp32: a swapped lane / register / cout-permutation
    map of either matrix instruction shows up as a mismatch (asymmetric weights, asymmetric input; 96 couts = one
    full and one half 64-cout tile; K = 64 = two 32-channel chunks)."""
    import torch.nn.functional as F
    L = s3r.arch_spec.Layer("t", "conv2d", 64, 96, 1, 1, 0, bn=False, act="none")
    ch = s3r.modules._HipChain([L], 9, precision="bf16")
    w = (torch.arange(96 * 64, dtype=torch.float32).reshape(96, 64, 1, 1) % 7) - 3
    w[5, 3] = 11
    w[70, 40] = -9
    ch.t.conv.weight.data.copy_(w)
    ch.t.conv.bias.data.copy_(torch.arange(96, dtype=torch.float32) % 5)
    x = (torch.arange(3 * 64 * 81, dtype=torch.float32).reshape(3, 64, 9, 9) % 5) - 2
    x[1, 2, 3, 4] = 9
    want = F.conv2d(x, w, ch.t.conv.bias.data)               # |values| < 2^8 * small: exact in bf16 after rounding? keep small
    assert float(want.abs().max()) < 256 and torch.equal(want.to(torch.bfloat16).float(), want)
    ch.to(DEV)
    xin = x.to(DEV).to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    for tm in (1, 2, 17, 21, 22):
        ch.tile_override["t"] = tm
        try:
            got = ch._run(xin)
        except s3r.S3RError:
            continue                                         # (plane codes refuse 1x1 shapes they cannot tile)
        assert torch.equal(got.float().cpu(), want), tm


def test_bf16_stem_many_images_against_the_fp32_stem(s3r):
    """The bf16 stem (fp32 MFMA over rows staged by LDS-DMA, persistent workgroups) on enough images that every
    workgroup makes several passes over its two row slabs (80 images = 2240 passes on <= 512 resident workgroups),
    checked against the fp32 path's stem — a different kernel (VALU, NCHW) — rounded to bf16: what may differ is the
    summation order (1e-6) and hence, rarely, the last bf16 bit.  The image border (top row, left column) must be exact
    zeros' worth: compare those rows / columns separately."""
    spec = s3r.arch_spec
    g = torch.Generator().manual_seed(11)
    x = torch.rand(80, 3, spec.IMG_HW, spec.IMG_HW, generator=g).to(DEV)
    hb = s3r.modules._HipChain([spec.ENCODER[0]], spec.IMG_HW, precision="bf16")
    hf = s3r.modules._HipChain([spec.ENCODER[0]], spec.IMG_HW, precision="fp32")
    s3r.seed_module(hb, 5)
    hf.load_state_dict(hb.state_dict())
    hb.to(DEV), hf.to(DEV)
    got = hb._run(x).float()                                   # logical (80,32,112,112)
    want32 = hf._run(x)
    want = want32.to(torch.bfloat16).float()
    assert got.shape == want.shape
    ulp = 2.0 ** -7 * want32.abs() + 1e-6                      # one bf16 step at the value's magnitude
    diff = (got - want).abs()
    assert (diff <= ulp).all()
    assert (diff > 0).float().mean().item() < 0.02             # the last bit differs on a per-cent of the elements at most
    for sl in (got[:, :, 0, :] - want[:, :, 0, :], got[:, :, :, 0] - want[:, :, :, 0]):
        assert (sl.abs() <= ulp.max()).all()
    again = hb._run(x).float()
    assert torch.equal(again, got)                             # deterministic across launches


def test_bf16_odd_empty_and_chunked_batches(s3r):
    """Ragged sizes on the bf16 path: an odd batch equals its samples run one by one (bitwise), an empty batch is an
    empty result, and the disparity read-out works from the bf16 encoder's features."""
    hip = s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(hip, 3)
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(5, seed=11)
    got = hip(left.to(DEV), right.to(DEV))
    for i in range(5):
        assert torch.equal(hip(left[i:i + 1].to(DEV), right[i:i + 1].to(DEV))[0], got[i])
    assert hip(left[:0].to(DEV), right[:0].to(DEV)).shape == (0, 32, 32, 32)
    dl, dr = hip.disparity(left.to(DEV), right.to(DEV))
    assert dl.shape == (5, 28, 28) and dr.shape == (5, 28, 28)
    assert float(dl.min()) >= 0 and float(dl.max()) <= 8 * 27 and bool((dl % 8 == 0).all())


@pytest.mark.parametrize("n", [2, 7, 16, 33, 64, 100])
def test_bf16_results_do_not_depend_on_the_batch_size(s3r, n, mfma_shape):
    """The library switches tiles with the batch (plane-reuse <-> row-reuse / per-tap, 128 x 128 <-> 128 x 64) but never
    the K summation order of a layer: sample 0 is bitwise the same alone and inside any batch."""
    hip = s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(hip, 6)
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(n, seed=17)
    alone = hip(left[:1].to(DEV), right[:1].to(DEV))[0].clone()
    got = hip(left.to(DEV), right.to(DEV))
    assert torch.equal(got[0], alone)
    assert torch.equal(hip(left[n - 1:].to(DEV), right[n - 1:].to(DEV))[0], got[n - 1])


def test_bf16_batch_above_the_chunk_limit(s3r):
    """258 pairs run as chunks of 256 + 2 (32-bit buffer offsets bound one launch): every sample still equals the
    sample run alone."""
    hip = s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(hip, 1)
    hip.to(DEV)
    n = s3r.modules.MAX_CHUNK + 2
    left, right = s3r.synthetic_pairs(n, seed=13)
    got = hip(left.to(DEV), right.to(DEV))
    assert got.shape == (n, 32, 32, 32)
    for i in (0, 255, 256, n - 1):
        assert torch.equal(hip(left[i:i + 1].to(DEV), right[i:i + 1].to(DEV))[0], got[i])


def test_bf16_hip_graph_replay_matches_eager(s3r):
    hip = s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(hip, 0)
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(3, seed=31)
    left, right = left.to(DEV), right.to(DEV)
    want = hip(left, right).clone()
    g = s3r.GraphedForward(hip, 3, DEV)
    assert torch.equal(g(left, right), want)
    l2, r2 = s3r.synthetic_pairs(3, seed=32)
    want2 = hip(l2.to(DEV), r2.to(DEV)).clone()
    assert torch.equal(g(l2.to(DEV), r2.to(DEV)), want2)


def test_bf16_autotune_keeps_the_result_within_bf16_tolerance(s3r):
    """Measured per-layer (tile, split-K) choices change summation orders, nothing else."""
    hip = s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(hip, 2)
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(4, seed=5)
    left, right = left.to(DEV), right.to(DEV)
    before = hip(left, right).clone()
    chosen = hip.autotune(left, right, rounds=1)
    assert set(chosen) >= {"e2", "v1", "d3"} and all(v["ms"] > 0 for v in chosen.values())
    after = hip(left, right)
    assert rel_l2(after.cpu(), before.cpu()) < 5e-3
    assert torch.equal(hip(left, right), after)


def test_stereo2point_bf16_vs_fp32_oracle(s3r, oracle):
    """Stereo2Point with the convolutional part on the bf16 path (the point head stays fp32): same state_dict as the
    fp32 module, point cloud within bf16 tolerance of the fp32 oracle, batch-invariant."""
    hip = s3r.Stereo2Point(precision="bf16")
    s3r.seed_module(hip, 1)
    ref = oracle.OracleStereo2Point().eval()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(3, seed=2)
    with torch.no_grad():
        want = ref(left, right)
    got = hip(left.to(DEV), right.to(DEV))
    assert got.shape == (3, 2048, 3) and got.dtype == torch.float32
    assert rel_l2(got.cpu(), want) < 2e-2
    assert torch.equal(hip(left[1:2].to(DEV), right[1:2].to(DEV))[0], got[1])
    assert hip(left[:0].to(DEV), right[:0].to(DEV)).shape == (0, 2048, 3)
    assert s3r.Stereo2Point().state_dict().keys() == hip.state_dict().keys()
    dl, dr = hip.disparity(left.to(DEV), right.to(DEV))                  # the read-out is shared with Stereo2Voxel
    assert dl.shape == (3, 28, 28) and dr.shape == (3, 28, 28)


def test_bf16_and_fp32_modules_share_a_state_dict(s3r):
    a, b = s3r.Stereo2Voxel(), s3r.Stereo2Voxel(precision="bf16")
    assert a.state_dict().keys() == b.state_dict().keys()
    assert all(v.dtype in (torch.float32, torch.int64) for v in b.state_dict().values())


def test_bf16_eval_metric_within_1e3_of_fp32_path(s3r):
    """north_star's criterion for this path, literally: the evaluation metric (mean voxel IoU against ground truth,
    per threshold) computed with the bf16 path differs from the fp32 path's by < 1e-3 (same weights, same data)."""
    a, b = s3r.Stereo2Voxel(), s3r.Stereo2Voxel(precision="bf16")
    s3r.seed_module(a, 0)
    b.load_state_dict(a.state_dict())
    a.to(DEV), b.to(DEV)
    left, right, gt = s3r.evaluate.synthetic_eval_set(16, 3)
    ra = s3r.evaluate.test_net(a, left, right, gt, batch=8, device=DEV)
    rb = s3r.evaluate.test_net(b, left, right, gt, batch=8, device=DEV)
    for t, x, y in zip(ra["thresholds"], ra["mean_iou"], rb["mean_iou"]):
        print(f"threshold {t}: mean IoU fp32 path {x:.5f}  bf16 path {y:.5f}  |diff| {abs(x - y):.2e}")
        assert abs(x - y) < 1e-3


def test_rows_kernel_equals_the_plane_kernel_bitwise(s3r, oracle, mfma_shape):
    """e2 at large image counts runs the row-persistent kernel (tile code 40: seven waves slide down a strip of rows,
    weights resident in LDS, input rows in a ten-slot ring).  Its K order is the plane kernel's, so the two must agree
    BIT FOR BIT (which is what lets the library choose between them by batch size); 1, 2 and 4 strips per image."""
    spec = s3r.arch_spec
    L = spec.ENCODER[1]
    ch = s3r.modules._HipChain([L], 112, precision="bf16")
    s3r.seed_module(ch, 3)
    ch.to(DEV)
    g = torch.Generator().manual_seed(5)
    for n_img in (128, 256, 512):                                   # 4, 2, 1 strips per image
        x = torch.randn((n_img, 112, 112, 32), generator=g).to(torch.bfloat16).to(DEV)   # physical channels-last
        x = x.permute(0, 3, 1, 2)                                                         # logical (N,32,112,112) view
        ch.tile_override["e2"] = 22
        want = ch._run(s3r.modules._check_input_cl(x, "x", (32, 112, 112)))
        ch.tile_override["e2"] = 40
        got = ch._run(s3r.modules._check_input_cl(x, "x", (32, 112, 112)))
        assert torch.equal(got, want), n_img
        del ch.tile_override["e2"]
        assert torch.equal(ch._run(s3r.modules._check_input_cl(x, "x", (32, 112, 112))), want)   # the library's own pick
    # against the oracle on a few images (bf16-rounded operands, fp32 accumulation, one bf16 rounding at the end)
    blk = oracle._Block(L).eval()
    sd = {k: (_bf(v) if k.endswith("conv.weight") else v) for k, v in ch.e2.state_dict().items()}
    blk.load_state_dict(sd)
    xs = x[:2].float().cpu()
    with torch.no_grad():
        ref = blk(xs)
    ch.tile_override["e2"] = 40
    got = ch._run(s3r.modules._check_input_cl(x, "x", (32, 112, 112)))[:2].float().cpu()
    assert rel_l2(got, ref) < 6e-3
    # a batch too small for it is refused under a forced code (the library then picks the plane kernel itself)
    with pytest.raises(s3r.S3RError):
        ch._run(s3r.modules._check_input_cl(x[:8], "x", (32, 112, 112)))
