"""Parity tests proper: the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

The oracle is this build's own PyTorch-CPU restatement of arch_spec — PARITY UNPINNED with respect to
the reference, whose model code is not in the mount.  Tolerance: BASELINE.json's
north_star states 1e-4 relative for fp32 outputs and IoU within 1e-3; the tests below hold the HIP
path to tighter bounds where fp32 allows (written next to each assert).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel_l2(a, b):
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


@pytest.fixture(scope="module")
def models(s3r, oracle):
    hip = s3r.Stereo2Voxel()
    s3r.seed_module(hip, 0)
    ref = oracle.OracleStereo2Voxel().eval()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV)
    return hip, ref


def _single(s3r, layer, in_size):
    ch = s3r.modules._HipChain([layer], in_size)
    s3r.seed_module(ch, 7)
    return ch


def _oracle_block(oracle, layer, sd):
    blk = oracle._Block(layer).eval()
    blk.load_state_dict(sd)
    return blk


# ------------------------------------------------------------------ MFMA operand/accumulator maps
def test_mfma_layout_exact_integers(s3r):
    """A 1x1 conv with small-integer data is exact in fp32: any swapped lane/register map shows up
    as a mismatch (asymmetric weights, asymmetric input)."""
    L = s3r.arch_spec.Layer("t", "conv2d", 16, 32, 1, 1, 0, bn=False, act="none")
    ch = _single(s3r, L, 8)
    w = torch.arange(32 * 16, dtype=torch.float32).reshape(32, 16, 1, 1) % 7 - 3
    w[5, 3] = 11
    ch.t.conv.weight.data.copy_(w)
    ch.t.conv.bias.data.zero_()
    x = (torch.arange(2 * 16 * 64, dtype=torch.float32).reshape(2, 16, 8, 8) % 5) - 2
    x[1, 2, 3, 4] = 9
    want = F.conv2d(x, w)
    got = ch.to(DEV)._run(x.to(DEV)).cpu()
    assert torch.equal(got, want)


# ------------------------------------------------------------------ every layer of the arch, alone
def _layer_cases(spec):
    cases = []
    for stage, layers, n0 in (("enc", spec.ENCODER, spec.IMG_HW), ("dec", spec.DECODER, spec.MAX_DISP)):
        for l, n_in, _ in spec.trace(layers, n0):
            cases.append((l, n_in))
    return cases


@pytest.mark.parametrize("idx", range(18))
def test_each_layer_vs_oracle(s3r, oracle, idx):
    spec = s3r.arch_spec
    layer, n_in = _layer_cases(spec)[idx]
    B = 2
    ch = _single(s3r, layer, n_in)
    blk = _oracle_block(oracle, layer, getattr(ch, layer.name).state_dict())
    g = torch.Generator().manual_seed(idx)
    x = torch.randn((B, layer.cin) + (n_in,) * spec.ndim(layer), generator=g)
    with torch.no_grad():
        want = blk(x)
    got = ch.to(DEV)._run(x.to(DEV)).cpu()
    assert got.shape == want.shape
    # fp32 FMA chain vs MKL-DNN blocked summation: differences are pure rounding order and grow with the
    # reduction depth K (v6: K = 256*4^3 = 16384 sequential fp32 FMAs; measured 2.4e-6)
    assert rel_l2(got, want) < 5e-6, (layer.name, rel_l2(got, want))
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("tile", range(8))
@pytest.mark.parametrize("kind", ["conv3d_s1", "conv3d_s1_w8", "conv3d_s1_w8_dword", "conv3d_s2", "deconv",
                                  "deconv_w4", "conv2d_s2", "conv2d_s1_w12", "conv3d_k4_valid"])
def test_every_tile_configuration(s3r, oracle, tile, kind):
    """All eight MFMA tile shapes x both gather widths (dword / 16-byte LDS-DMA) on ragged problem sizes
    (N not a tile multiple, Cout not a BM multiple, rows that are / are not a multiple of 4)."""
    Layer = s3r.arch_spec.Layer
    layer, n_in, B, vec = {
        "conv3d_s1": (Layer("t", "conv3d", 32, 96, 3, 1, 1), 7, 3, 0),             # Nw=7: dword gather
        "conv3d_s1_w8": (Layer("t", "conv3d", 32, 96, 3, 1, 1), 8, 3, 0),          # Nw=8: 16-byte gather
        "conv3d_s1_w8_dword": (Layer("t", "conv3d", 32, 96, 3, 1, 1), 8, 3, 1),    # same, dword forced
        "conv3d_s2": (Layer("t", "conv3d", 16, 160, 3, 2, 1), 9, 2, 0),
        "deconv": (Layer("t", "deconv3d", 32, 48, 4, 2, 1), 5, 3, 0),
        "deconv_w4": (Layer("t", "deconv3d", 32, 48, 4, 2, 1), 4, 3, 0),           # input rows of 4: 16-byte
        "conv2d_s2": (Layer("t", "conv2d", 48, 64, 3, 2, 1), 13, 5, 0),
        "conv2d_s1_w12": (Layer("t", "conv2d", 48, 64, 3, 1, 1), 12, 5, 0),
        "conv3d_k4_valid": (Layer("t", "conv3d", 16, 40, 4, 1, 0), 7, 2, 0),       # v6-like: k4, no padding
    }[kind]
    ch = _single(s3r, layer, n_in)
    ch.tile_override["t"] = tile + 16 * vec
    blk = _oracle_block(oracle, layer, ch.t.state_dict())
    x = torch.randn((B, layer.cin) + (n_in,) * s3r.arch_spec.ndim(layer), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = blk(x)
    out_w = want.shape[-1] if layer.op != "deconv3d" else n_in
    dword = vec == 1 or (layer.op != "deconv3d" and layer.s != 1) or out_w % 4 != 0
    if tile == 5 and dword:      # the 64 x 512 tile is wider than one dword-gather row piece allows
        with pytest.raises(s3r.S3RError):
            ch.to(DEV)._run(x.to(DEV))
        return
    got = ch.to(DEV)._run(x.to(DEV)).cpu()
    assert rel_l2(got, want) < 2e-6


@pytest.mark.parametrize("ksplit", [1, 2, 4])
@pytest.mark.parametrize("kind", ["conv3d_k4_valid", "deconv", "conv3d_s2"])
def test_split_k(s3r, oracle, kind, ksplit):
    """Split-K (partial slabs + the deterministic finish kernel) against the oracle, on ragged sizes."""
    Layer = s3r.arch_spec.Layer
    layer, n_in, B = {
        "conv3d_k4_valid": (Layer("t", "conv3d", 64, 40, 4, 1, 0), 7, 3),
        "deconv": (Layer("t", "deconv3d", 64, 48, 4, 2, 1, act="sigmoid"), 5, 3),
        "conv3d_s2": (Layer("t", "conv3d", 128, 160, 3, 2, 1, bn=False, act="none"), 9, 2),
    }[kind]
    ch = _single(s3r, layer, n_in)
    ch.ksplit_override["t"] = ksplit
    blk = _oracle_block(oracle, layer, ch.t.state_dict())
    x = torch.randn((B, layer.cin) + (n_in,) * 3, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = blk(x)
    ch.to(DEV)
    got = ch._run(x.to(DEV))
    assert rel_l2(got.cpu(), want) < 2e-6
    assert torch.equal(ch._run(x.to(DEV)), got)          # deterministic: slabs are summed in a fixed order
    for tile in (0, 3, 7):
        ch.tile_override["t"] = tile
        assert torch.equal(ch._run(x.to(DEV)), got)      # the tile shape never changes the bits
    ch.ksplit_override["t"] = 3                           # does not divide cin/16
    with pytest.raises(s3r.S3RError):
        ch._run(x.to(DEV))


def test_conv_linearity_full_size(s3r):
    """Size-independent property at BASELINE size (B=32): conv(2x) == 2*conv(x) bit-exactly when the
    epilogue is the identity (scaling by 2 is exact in fp32)."""
    L = s3r.arch_spec.Layer("t", "conv3d", 64, 64, 3, 1, 1, bn=False, act="none")
    ch = _single(s3r, L, 28)
    ch.t.conv.bias.data.zero_()
    ch.to(DEV)
    x = torch.randn(32, 64, 28, 28, 28, device=DEV)
    y1 = ch._run(x)
    y2 = ch._run(2 * x)
    assert torch.equal(y2, 2 * y1)


# ------------------------------------------------------------------ cost volume: bit exact
@pytest.mark.parametrize("shape", [(2, 32, 28, 28, 28), (1, 5, 7, 6, 10), (3, 4, 12, 5, 8), (2, 3, 9, 4, 7)])
def test_cost_volume_bit_exact(s3r, oracle, shape):
    B, Cc, D, H, W = shape
    g = torch.Generator().manual_seed(11)
    fl, fr = torch.randn(B, Cc, H, W, generator=g), torch.randn(B, Cc, H, W, generator=g)
    want = oracle.cost_volume(fl, fr, D)
    got = s3r.CostVolume(D)(fl.to(DEV), fr.to(DEV)).cpu()
    assert torch.equal(got, want)


# ------------------------------------------------------------------ whole network
def test_stereo2voxel_vs_oracle_b2(s3r, oracle, models):
    hip, ref = models
    left, right = s3r.synthetic_pairs(2, seed=0)
    with torch.no_grad():
        want = ref(left, right)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    assert got.shape == (2, 32, 32, 32)
    assert rel_l2(got, want) < 1e-5                    # north_star bound: 1e-4 relative
    assert (got - want).abs().max().item() < 1e-4
    assert oracle.voxel_iou(got, want).min().item() > 1 - 1e-3   # north_star: IoU within 1e-3


def test_stereo2voxel_vs_golden(s3r, models, golden_dir):
    hip, _ = models
    z = np.load(f"{golden_dir}/s2v_b2_seed0.npz")
    left, right = s3r.synthetic_pairs(2, seed=0)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    want = torch.from_numpy(z["occupancy"])
    assert rel_l2(got, want) < 1e-5
    feats = hip.encoder(torch.cat([left, right]).to(DEV)).cpu()
    assert rel_l2(feats, torch.from_numpy(z["features"])) < 1e-5


def test_encoder_pair_entry_equals_the_concatenated_batch(s3r, models):
    """s3r_encoder_forward(left, right): the tower reads the two render tensors in place (no torch.cat inside the
    forward): bit for bit the features of the concatenated batch, also when the two tensors are far apart in memory."""
    hip, _ = models
    left, right = s3r.synthetic_pairs(3, seed=41)
    left, right = left.to(DEV), right.to(DEV)
    spacer = torch.empty(1 << 20, device=DEV)                # (keeps the two allocations from being neighbours)
    right2 = right.clone()
    want = hip.encoder(torch.cat([left, right]))
    assert torch.equal(hip.encoder.forward_pair(left, right2), want)
    assert hip.encoder.forward_pair(left[:0], right[:0]).shape == (0, 32, 28, 28)
    with pytest.raises(RuntimeError):
        hip.encoder.forward_pair(left, right[:2])
    bf = s3r.Stereo2Voxel(precision="bf16")
    bf.load_state_dict(hip.state_dict())
    bf.to(DEV)
    assert torch.equal(bf.encoder.forward_pair(left, right2), bf.encoder(torch.cat([left, right])))
    del spacer


def test_stage_by_stage_vs_oracle(s3r, oracle, models):
    hip, ref = models
    left, right = s3r.synthetic_pairs(2, seed=5)
    with torch.no_grad():
        f_ref = ref.encoder(torch.cat([left, right]))
        v_ref = oracle.cost_volume(f_ref[:2], f_ref[2:])
    f = hip.encoder(torch.cat([left, right]).to(DEV))
    assert rel_l2(f.cpu(), f_ref) < 5e-6
    v = hip.cost_volume(f[:2], f[2:])
    assert rel_l2(v.cpu(), v_ref) < 1e-5
    for name in hip.decoder.names[:-1]:
        with torch.no_grad():
            d_ref = ref.decoder(v_ref, upto=name)
        d = hip.decoder(v_ref.to(DEV), upto=name).cpu()
        assert rel_l2(d, d_ref) < 5e-6, name


def test_padded_handoff_equals_plain_tensors_and_keeps_halo_zero(s3r, oracle, models):
    """Stereo2Voxel.forward hands the cost volume to the decoder as a halo-padded resident buffer; the
    module-level path (plain tensors, the library pads) must give the same bits, the halo must stay zero,
    and the interior must be the oracle's volume exactly."""
    hip, ref = models
    left, right = s3r.synthetic_pairs(2, seed=11)
    left, right = left.to(DEV), right.to(DEV)
    fused = hip(left, right)
    f = hip.encoder(torch.cat([left, right]))
    plain = hip.decoder(hip.cost_volume(f[:2], f[2:]))
    assert torch.equal(fused, plain)
    vp = hip.cost_volume.forward_padded(f[:2], f[2:])
    assert vp.shape == (2, 64, 30, 30, 30)
    inner = vp[:, :, 1:-1, 1:-1, 1:-1]
    assert torch.equal(inner, hip.cost_volume(f[:2], f[2:]))
    assert vp.abs().sum().item() == inner.abs().sum().item() or torch.equal(
        vp.sum(), inner.sum())                                      # nothing outside the interior
    border = vp.clone()
    border[:, :, 1:-1, 1:-1, 1:-1] = 0
    assert border.abs().max().item() == 0.0
    with torch.no_grad():
        fr = ref.encoder(torch.cat([left, right]).cpu())
    assert torch.equal(inner.cpu(), oracle.cost_volume(f[:2].cpu(), f[2:].cpu()))
    assert rel_l2(f.cpu(), fr) < 5e-6


def test_workspace_relayout_between_batch_sizes(s3r, models):
    """The activation arena is laid out per batch size; alternating batch sizes must re-zero it
    (stale interiors of one layout would sit in another layout's halos)."""
    hip, _ = models
    left, right = s3r.synthetic_pairs(5, seed=21)
    left, right = left.to(DEV), right.to(DEV)
    want5 = hip(left, right).clone()
    want2 = hip(left[:2], right[:2]).clone()
    assert torch.equal(want2, want5[:2])
    for _ in range(2):
        assert torch.equal(hip(left, right), want5)
        assert torch.equal(hip(left[:2], right[:2]), want2)
        assert torch.equal(hip(left[3:4], right[3:4])[0], want5[3])


def test_batch32_matches_per_sample_bitwise(s3r, models):
    """BASELINE size (B=32).  Every output voxel's K-order is fixed by the kernel, so a sample's result
    cannot depend on where it sits in the batch: batch-32 outputs equal batch-1/2 outputs bit for bit."""
    hip, _ = models
    left, right = s3r.synthetic_pairs(32, seed=3)
    left, right = left.to(DEV), right.to(DEV)
    full = hip(left, right)
    assert full.shape == (32, 32, 32, 32)
    assert torch.isfinite(full).all() and full.min() >= 0 and full.max() <= 1
    for s in (0, 13, 31):
        one = hip(left[s:s + 1], right[s:s + 1])
        assert torch.equal(one[0], full[s])
    again = hip(left, right)
    assert torch.equal(again, full)            # deterministic (no atomics on the voxel path)


@pytest.mark.parametrize("n", [2, 5, 11, 21, 33, 47, 64, 100])
def test_results_do_not_depend_on_the_batch_size(s3r, models, n):
    """The library changes tiles, cuts layers into a bulk and a re-tiled remainder (one launch or two) and re-plans both
    with the workgroup count — i.e. with the batch — but never a layer's K order or split-K factor: the first and the last
    sample are bitwise the same alone and inside any batch (workgroup counts just under / over whole 256-workgroup rounds
    included: the batch sizes walk the remainder logic through cut / no-cut decisions on different layers)."""
    hip, _ = models
    left, right = s3r.synthetic_pairs(n, seed=23)
    left, right = left.to(DEV), right.to(DEV)
    got = hip(left, right).clone()
    assert torch.equal(hip(left[:1], right[:1])[0], got[0])
    assert torch.equal(hip(left[n - 1:], right[n - 1:])[0], got[n - 1])
    assert torch.equal(hip(left, right), got)


def test_batch32_sample_vs_oracle(s3r, models):
    hip, ref = models
    left, right = s3r.synthetic_pairs(32, seed=3)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    with torch.no_grad():
        want = ref(left[20:22], right[20:22])
    assert rel_l2(got[20:22], want) < 1e-5


def test_odd_and_empty_batches(s3r, models):
    hip, ref = models
    left, right = s3r.synthetic_pairs(3, seed=9)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    with torch.no_grad():
        want = ref(left, right)
    assert rel_l2(got, want) < 1e-5
    empty = hip(left[:0].to(DEV), right[:0].to(DEV))
    assert empty.shape == (0, 32, 32, 32)


def test_state_dict_roundtrip_changes_output(s3r, models):
    """load_state_dict must invalidate the packed-weight cache."""
    hip, _ = models
    left, right = s3r.synthetic_pairs(1, seed=0)
    left, right = left.to(DEV), right.to(DEV)
    y0 = hip(left, right).clone()
    sd0 = {k: v.clone() for k, v in hip.state_dict().items()}
    other = s3r.Stereo2Voxel()
    s3r.seed_module(other, 123)
    hip.load_state_dict(other.state_dict())
    y1 = hip(left, right).clone()
    assert not torch.equal(y0, y1)
    hip.load_state_dict(sd0)
    assert torch.equal(hip(left, right), y0)


# ------------------------------------------------------------------ Stereo2Point + Chamfer + IoU
def test_stereo2point_vs_oracle(s3r, oracle):
    hip = s3r.Stereo2Point()
    s3r.seed_module(hip, 1)
    ref = oracle.OracleStereo2Point().eval()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(2, seed=2)
    with torch.no_grad():
        want = ref(left, right)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    assert got.shape == (2, 2048, 3)
    assert rel_l2(got, want) < 1e-5
    # ragged sizes: an empty batch, and a sample of an odd batch equals the sample alone (bitwise)
    assert hip(left[:0].to(DEV), right[:0].to(DEV)).shape == (0, 2048, 3)
    l3, r3 = s3r.synthetic_pairs(3, seed=6)
    got3 = hip(l3.to(DEV), r3.to(DEV))
    assert torch.equal(hip(l3[2:3].to(DEV), r3[2:3].to(DEV))[0], got3[2])


def test_stereo2point_vs_golden_points(s3r, golden_dir):
    """The committed Stereo2Point fixture (tests/golden/make_golden.py: oracle, seed 1, pairs seed 2) on the GPU."""
    z = np.load(f"{golden_dir}/s2p_chamfer.npz")
    hip = s3r.Stereo2Point()
    s3r.seed_module(hip, 1)
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(2, seed=2)
    got = hip(left.to(DEV), right.to(DEV)).cpu()
    assert rel_l2(got, torch.from_numpy(z["points"])) < 1e-5


def test_stereo2point_and_chamfer_at_baseline_size(s3r, oracle):
    """BASELINE.json configs[3] at its stated size: Stereo2Point forward at B = 32 + Chamfer distance between
    (32, 2048, 3) clouds.  Two samples against the oracle; the rest through size-independent properties (a
    sample's cloud does not depend on its batch; Chamfer is symmetric under swapping the clouds, zero against
    itself, and its indices reproduce its distances)."""
    hip = s3r.Stereo2Point()
    s3r.seed_module(hip, 4)
    ref = oracle.OracleStereo2Point().eval()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV)
    B = 32
    left, right = s3r.synthetic_pairs(B, seed=61)
    pts = hip(left.to(DEV), right.to(DEV))
    assert pts.shape == (B, 2048, 3)
    with torch.no_grad():
        want = ref(left[[0, 31]], right[[0, 31]])
    assert rel_l2(pts[[0, 31]].cpu(), want) < 1e-5
    assert torch.equal(hip(left[17:18].to(DEV), right[17:18].to(DEV))[0], pts[17])
    gt = (torch.rand(B, 2048, 3, generator=torch.Generator().manual_seed(9)) - 0.5).to(DEV)
    d1, d2, i1, i2 = s3r.chamfer_distance(pts, gt)
    for k in (3, 30):                                         # two samples against the oracle, bit for bit
        w1, w2, wi1, wi2 = oracle.chamfer_distance(pts[k:k + 1].cpu(), gt[k:k + 1].cpu())
        assert torch.equal(d1[k].cpu(), w1[0]) and torch.equal(d2[k].cpu(), w2[0])
        assert torch.equal(i1[k].cpu().long(), wi1[0].long()) and torch.equal(i2[k].cpu().long(), wi2[0].long())
    e1, e2, j1, j2 = s3r.chamfer_distance(gt, pts)            # swapping the clouds swaps the outputs
    assert torch.equal(e1, d2) and torch.equal(e2, d1) and torch.equal(j1, i2) and torch.equal(j2, i1)
    z1, z2, _, _ = s3r.chamfer_distance(pts, pts)
    assert float(z1.abs().max()) == 0.0 and float(z2.abs().max()) == 0.0
    nn = torch.gather(gt, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3))        # the indices reproduce the distances
    assert torch.allclose(((pts - nn) ** 2).sum(-1), d1, rtol=1e-5, atol=1e-7)
    assert torch.equal(s3r.chamfer_distance(pts, gt)[0], d1)                   # deterministic


@pytest.mark.parametrize("shape", [(32, 32768, 1024), (5, 1024, 6144), (33, 96, 40), (3, 50, 7), (70, 4096, 100)])
def test_linear_layer(s3r, oracle, shape):
    """Point-head linear kernel (fp32-MFMA weight streaming, deterministic split-K) incl. ragged batch /
    cout and the Cin % 32 != 0 fallback."""
    B, cin, cout = shape
    L = s3r.arch_spec.Layer("t", "linear", cin, cout, 1, 1, 0, bn=False, act="relu")
    ch = _single(s3r, L, 1)
    blk = _oracle_block(oracle, L, ch.t.state_dict())
    x = torch.randn(B, cin, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = blk(x)
    ch.to(DEV)
    got = ch._run(x.to(DEV))
    assert rel_l2(got.cpu(), want) < 2e-6
    assert torch.equal(ch._run(x.to(DEV)), got)


@pytest.mark.parametrize("shape", [(2, 2048, 2048), (3, 100, 1500), (1, 1, 1), (2, 1025, 7)])
def test_chamfer_exact(s3r, oracle, shape):
    B, N, M = shape
    g = torch.Generator().manual_seed(4)
    p, q = torch.rand(B, N, 3, generator=g), torch.rand(B, M, 3, generator=g)
    w1, w2, wi1, wi2 = oracle.chamfer_distance(p, q)
    d1, d2, i1, i2 = s3r.chamfer_distance(p.to(DEV), q.to(DEV))
    assert torch.equal(d1.cpu(), w1) and torch.equal(d2.cpu(), w2)      # same fp32 operation order: bit exact
    assert torch.equal(i1.cpu(), wi1) and torch.equal(i2.cpu(), wi2)
    loss = s3r.ChamferDistance()(p.to(DEV), q.to(DEV)).item()
    assert abs(loss - oracle.chamfer_loss(p, q).item()) < 1e-6


def test_chamfer_collisions_first_minimum(s3r, oracle):
    p = torch.tensor([[[0.0, 0, 0], [1, 1, 1]]])
    q = torch.tensor([[[1.0, 1, 1], [0, 0, 0], [0, 0, 0], [1, 1, 1]]])
    d1, d2, i1, i2 = s3r.chamfer_distance(p.to(DEV), q.to(DEV))
    assert i1.cpu().tolist() == [[1, 0]] and i2.cpu().tolist() == [[1, 0, 0, 1]]
    assert d1.abs().max().item() == 0 and d2.abs().max().item() == 0


def test_voxel_iou_exact(s3r, oracle):
    g = torch.Generator().manual_seed(8)
    a, b = torch.rand(5, 32, 32, 32, generator=g), torch.rand(5, 32, 32, 32, generator=g)
    b[4] = 0                      # empty ground truth
    a[4] = 0                      # and empty prediction -> IoU defined as 1
    got = s3r.voxel_iou(a.to(DEV), b.to(DEV)).cpu()
    assert torch.equal(got, oracle.voxel_iou(a, b))


@pytest.mark.parametrize("shape", [(3, 32, 28, 28, 28), (2, 5, 7, 13, 40), (1, 64, 9, 57, 16), (4, 3, 1, 1, 4)])
def test_disparity_wta_bit_exact(s3r, oracle, shape):
    """Winner-take-all disparity read-out vs the oracle: integer results, bit-exact, including
    max_disp > width, a single-pixel map, exact ties (first minimum) and a planted shift."""
    B, Cc, H, W, D = shape
    g = torch.Generator().manual_seed(sum(shape))
    fl, fr = torch.randn(B, Cc, H, W, generator=g), torch.randn(B, Cc, H, W, generator=g)
    if W > 6:
        fr[0, :, :, : W - 5] = fl[0, :, :, 5:]            # sample 0: a true disparity of 5
    fl[-1, :, :, : W // 2] = 1.0                          # last sample: constant patches -> ties
    fr[-1, :, :, : W // 2] = 1.0
    want_l, want_r = oracle.disparity_wta(fl, fr, D)
    got_l, got_r = s3r.disparity_wta(fl.to(DEV), fr.to(DEV), D)
    assert got_l.shape == (B, H, W) and got_l.dtype == torch.float32
    assert torch.equal(got_l.cpu(), want_l) and torch.equal(got_r.cpu(), want_r)
    if W > 6 and D > 5 and B > 1:                        # (B == 1: sample 0 also carries the constant patch)
        assert (got_l[0, :, 5:] == 5).all()


def test_disparity_epe_exact(s3r, oracle):
    g = torch.Generator().manual_seed(21)
    pred = torch.randint(0, 28, (6, 28, 28), generator=g).float() * 8
    gt = torch.rand(6, 28, 28, generator=g) * 200
    gt[0, :5] = float("inf")                              # background as EXR marks it
    gt[1, 3] = -1.0
    gt[2, 0, 0] = float("nan")
    gt[5] = float("inf")                                  # no valid pixel at all
    epe, n = s3r.disparity_epe(pred.to(DEV), gt.to(DEV))
    want_e, want_n = oracle.disparity_epe(pred, gt)
    assert torch.equal(n.cpu(), want_n) and n.dtype == torch.int32
    assert (epe.cpu() - want_e).abs().max().item() <= 1e-6 * want_e.abs().max().item()    # fp64 sums, fp32 result
    assert epe[5].item() == 0.0


def test_model_disparity_and_eval_driver(s3r, oracle, models):
    """Stereo2Voxel.disparity = read-out of the model's own encoder features (x8 to render pixels); the eval driver
    pools EPE over valid pixels exactly as the oracle does."""
    hip, _ = models
    left, right, _ = s3r.evaluate.synthetic_eval_set(5, 4)
    dl, dr = hip.disparity(left.to(DEV), right.to(DEV))
    feats = hip.encoder(torch.cat([left, right], 0).to(DEV)).cpu()
    wl, wr = oracle.disparity_wta(feats[:5], feats[5:], s3r.arch_spec.MAX_DISP)
    assert torch.equal(dl.cpu(), wl * 8) and torch.equal(dr.cpu(), wr * 8)
    g = torch.Generator().manual_seed(9)
    gl, gr = torch.rand(5, 28, 28, generator=g) * 220, torch.rand(5, 28, 28, generator=g) * 220
    gl[0, :10] = float("inf")
    res = s3r.evaluate.test_disparity(hip, left, right, gl, gr, batch=2, device=DEV)
    el, nl = oracle.disparity_epe(wl * 8, gl)
    er, nr = oracle.disparity_epe(wr * 8, gr)
    assert res["valid_left"] == int(nl.sum()) and res["valid_right"] == int(nr.sum())
    assert abs(res["epe_left"] - float((el.double() * nl).sum() / nl.sum())) < 1e-4
    assert abs(res["epe_right"] - float((er.double() * nr).sum() / nr.sum())) < 1e-4


def test_point_eval_driver_matches_oracle_chamfer(s3r, oracle):
    hip = s3r.Stereo2Point()
    s3r.seed_module(hip, 5)
    ref = oracle.OracleStereo2Point().eval()
    ref.load_state_dict(hip.state_dict())
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(5, seed=8)
    gt = torch.rand(5, 1500, 3, generator=torch.Generator().manual_seed(4)) - 0.5
    res = s3r.evaluate.test_point_net(hip, left, right, gt, batch=2, device=DEV)
    with torch.no_grad():
        pred = ref(left, right)
    d1, d2, _, _ = oracle.chamfer_distance(pred, gt)
    want = d1.mean(1) + d2.mean(1)
    assert res["samples"] == 5 and res["per_sample"].shape == (5,)
    assert (res["per_sample"] - want).abs().max().item() < 1e-4 * want.abs().max().item()
    assert abs(res["mean_chamfer"] - want.mean().item()) < 1e-4 * want.mean().item()


def test_eval_driver_matches_oracle_iou(s3r, oracle, models):
    """evaluate.test_net (device-side IoU per threshold) against the oracle's forward + IoU."""
    hip, ref = models
    left, right, gt = s3r.evaluate.synthetic_eval_set(5, 1)
    res = s3r.evaluate.test_net(hip, left, right, gt, batch=2, device=DEV)
    assert res["samples"] == 5 and res["per_sample"].shape == (5, 4)
    with torch.no_grad():
        pred = ref(left, right)
    for j, t in enumerate(res["thresholds"]):
        a, b = pred > t, gt > 0.5
        want = (a & b).flatten(1).sum(1).float() / (a | b).flatten(1).sum(1).float().clamp(min=1)
        assert (res["per_sample"][:, j] - want).abs().max().item() < 1e-3      # north_star: IoU within 1e-3
        assert abs(res["mean_iou"][j] - want.mean().item()) < 1e-3


# ------------------------------------------------------------------ error behaviour of the boundary
def test_errors(s3r, models):
    hip, _ = models
    with pytest.raises(RuntimeError):
        hip(torch.zeros(1, 3, 224, 224), torch.zeros(1, 3, 224, 224))                 # CPU tensors
    with pytest.raises(RuntimeError):
        hip(torch.zeros(1, 3, 224, 224, device=DEV).half(), torch.zeros(1, 3, 224, 224, device=DEV).half())
    with pytest.raises(RuntimeError):
        hip(torch.zeros(1, 3, 200, 224, device=DEV), torch.zeros(1, 3, 200, 224, device=DEV))
    with pytest.raises(RuntimeError):
        hip(torch.zeros(2, 3, 224, 224, device=DEV), torch.zeros(1, 3, 224, 224, device=DEV))
    with pytest.raises(RuntimeError):
        hip.train()
    with pytest.raises(RuntimeError):
        s3r.chamfer_distance(torch.zeros(1, 0, 3, device=DEV), torch.zeros(1, 4, 3, device=DEV))
    # disparity read-out / end-point error: shape, device and size contracts
    f = torch.zeros(1, 8, 4, 6, device=DEV)
    with pytest.raises(RuntimeError):
        s3r.disparity_wta(f, torch.zeros(1, 8, 4, 7, device=DEV))                     # shapes differ
    with pytest.raises(RuntimeError):
        s3r.disparity_wta(f.cpu(), f.cpu())                                           # no CPU fallback
    with pytest.raises(s3r.S3RError):
        s3r.disparity_wta(f, f, max_disp=0)
    with pytest.raises(s3r.S3RError):                                                 # a row pair must fit 64 KiB of LDS
        s3r.disparity_wta(torch.zeros(1, 64, 2, 200, device=DEV), torch.zeros(1, 64, 2, 200, device=DEV))
    with pytest.raises(RuntimeError):
        s3r.disparity_epe(torch.zeros(2, 4, 4, device=DEV), torch.zeros(2, 4, 5, device=DEV))
    assert s3r.disparity_epe(torch.zeros(0, 4, 4, device=DEV), torch.zeros(0, 4, 4, device=DEV))[0].shape == (0,)
    dl0, dr0 = s3r.disparity_wta(f[:0], f[:0])
    assert dl0.shape == (0, 4, 6) and dr0.shape == (0, 4, 6)


def test_two_host_threads_two_streams(s3r):
    """INTEGRATION.md's threading contract: the library is re-entrant — two host threads, each with its own module and
    HIP stream, run forwards concurrently and get the bits a serial run gets; error strings are per thread."""
    import threading
    mods, wants, ins = [], [], []
    for k in range(2):
        m = s3r.Stereo2Voxel("bf16" if k else "fp32")
        s3r.seed_module(m, 10 + k)
        m.to(DEV)
        l, r = s3r.synthetic_pairs(3, seed=40 + k)
        l, r = l.to(DEV), r.to(DEV)
        mods.append(m), ins.append((l, r)), wants.append(m(l, r).clone())
    torch.cuda.synchronize()
    got, errs = [None, None], [None, None]

    def work(k):
        try:
            st = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st):
                for _ in range(5):
                    y = mods[k](*ins[k])
                if k == 0:                                      # a deliberate error on this thread only
                    try:
                        s3r.disparity_wta(torch.zeros(1, 64, 2, 200, device=DEV), torch.zeros(1, 64, 2, 200, device=DEV))
                    except s3r.S3RError as e:
                        errs[k] = str(e)
            st.synchronize()
            got[k] = y.clone()
        except Exception as e:                                  # pragma: no cover
            errs[k] = f"unexpected {type(e).__name__}: {e}"

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert errs[1] is None and errs[0] is not None and "64 KiB" in errs[0]
    assert torch.equal(got[0], wants[0]) and torch.equal(got[1], wants[1])


def test_non_default_stream(s3r, models):
    hip, _ = models
    left, right = s3r.synthetic_pairs(1, seed=0)
    left, right = left.to(DEV), right.to(DEV)
    want = hip(left, right)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(st):
        got = hip(left, right)
    st.synchronize()
    assert torch.equal(got, want)


def test_hip_graph_replay_matches_eager(s3r):
    hip = s3r.Stereo2Voxel()          # own instance: capture pins the module's buffers to one batch shape
    s3r.seed_module(hip, 0)
    hip.to(DEV)
    left, right = s3r.synthetic_pairs(3, seed=31)
    left, right = left.to(DEV), right.to(DEV)
    want = hip(left, right).clone()
    g = s3r.GraphedForward(hip, 3, DEV)
    got = g(left, right).clone()
    assert torch.equal(got, want)
    l2, r2 = s3r.synthetic_pairs(3, seed=32)
    want2 = hip(l2.to(DEV), r2.to(DEV)).clone()
    assert torch.equal(g(l2.to(DEV), r2.to(DEV)), want2)     # replay with new data in the static inputs
    assert torch.equal(hip(left, right), want)                # eager calls of the captured shape still work
    with pytest.raises(RuntimeError):
        g(left[:2], right[:2])
    # the graph holds the module's resident buffers: other shapes / new weights on the same module must raise
    with pytest.raises(RuntimeError, match="HIP graph"):
        hip(left[:2], right[:2])
    other = s3r.Stereo2Voxel()
    s3r.seed_module(other, 9)
    with pytest.raises(RuntimeError, match="HIP graph"):
        hip.load_state_dict(other.state_dict())
        hip(left, right)


def test_prefetching_loader_order_and_values(s3r, models):
    hip, _ = models
    batches = [s3r.synthetic_pairs(2, seed=40 + i) for i in range(3)]
    outs = [hip(l, r).clone() for l, r in s3r.PrefetchingLoader(batches, DEV)]
    for (l, r), o in zip(batches, outs):
        assert torch.equal(o, hip(l.to(DEV), r.to(DEV)))


def test_dataset_eval_matches_tensor_eval(s3r, models, tmp_path):
    """evaluate.test_dataset (PNG/MAT decode -> PrefetchingLoader -> HIP forward -> device IoU) equals
    evaluate.test_net on the same decoded tensors."""
    from tests.test_data_cpu import _make_tree
    hip, _ = models
    _make_tree(str(tmp_path), n_models=3, views=(0,), size=224)
    ds = s3r.data.StereoShapeNet(str(tmp_path))
    a = s3r.evaluate.test_dataset(hip, ds, batch=2, device=DEV)
    items = [ds[i] for i in range(len(ds))]
    left, right, gt = (torch.stack([it[k] for it in items]) for k in range(3))
    b = s3r.evaluate.test_net(hip, left, right, gt, batch=2, device=DEV)
    assert a["samples"] == 3 and torch.equal(a["per_sample"], b["per_sample"])
    assert list(a["per_taxonomy"]) == ["02691156"] and a["per_taxonomy"]["02691156"]["samples"] == 3
    assert a["per_taxonomy"]["02691156"]["mean_iou"] == pytest.approx(a["mean_iou"])


def test_dataset_eval_vs_oracle_on_an_independent_decode(s3r, oracle, models, tmp_path):
    """The same chain against the ORACLE, with the files decoded here and not by data.py: Pillow -> float RGBA composited
    over white in float64, rounded to the 8-bit code, / 255; scipy.io for the volume; the oracle's CPU forward and a
    plain-torch IoU.  (test_dataset_eval_matches_tensor_eval above is HIP against HIP.)"""
    import os
    import numpy as np
    from PIL import Image
    from scipy.io import loadmat
    from tests.test_data_cpu import _make_tree
    hip, ref = models
    _make_tree(str(tmp_path), n_models=3, views=(0,), size=224)
    got = s3r.evaluate.test_dataset(hip, s3r.data.StereoShapeNet(str(tmp_path)), batch=2, device=DEV)
    lefts, rights, vols = [], [], []
    for m in range(3):
        rdir = os.path.join(str(tmp_path), "ShapeNetStereoRendering", "02691156", f"model{m:02d}")
        for side, dst in (("l", lefts), ("r", rights)):
            rgba = np.asarray(Image.open(os.path.join(rdir, f"render_00_{side}.png")).convert("RGBA"), dtype=np.float64)
            al = rgba[..., 3:4] / 255.0
            rgb = np.rint(rgba[..., :3] * al + 255.0 * (1.0 - al))              # over white, to the nearest 8-bit code
            dst.append(torch.from_numpy((rgb / 255.0).astype(np.float32).transpose(2, 0, 1).copy()))
        vols.append(torch.from_numpy(loadmat(os.path.join(str(tmp_path), "ShapeNetVox32", "02691156",
                                                          f"model{m:02d}.mat"))["Volume"].astype(np.float32)))
    left, right, gt = torch.stack(lefts), torch.stack(rights), torch.stack(vols)
    with torch.no_grad():
        pred = ref(left, right)
    for j, t in enumerate(got["thresholds"]):
        p_, g_ = pred > t, gt > 0.5
        want = (p_ & g_).flatten(1).sum(1).float() / (p_ | g_).flatten(1).sum(1).float().clamp(min=1)
        assert (got["per_sample"][:, j].cpu() - want).abs().max().item() < 1e-3          # north_star: IoU within 1e-3
        assert abs(got["mean_iou"][j] - want.mean().item()) < 1e-3


def test_dataset_eval_with_exr_disparity(s3r, models, tmp_path):
    """The whole next-row chain: PNG / MAT / EXR decode -> prefetch -> forward + IoU, disparity read-out + end-point error
    against the EXR ground truth (block-averaged to the read-out's resolution) — equal to the tensor-level drivers."""
    import os
    import numpy as np
    from tests.test_data_cpu import _make_tree
    hip, _ = models
    _make_tree(str(tmp_path), n_models=3, views=(0,), size=224)
    rng = np.random.default_rng(3)
    for m in range(3):
        rdir = os.path.join(str(tmp_path), "ShapeNetStereoRendering", "02691156", f"model{m:02d}")
        for side in "lr":
            d = (rng.random((224, 224), dtype=np.float32) * 200)
            d[:, :40] = np.inf
            s3r.exr.write_exr(os.path.join(rdir, "disp_00_%s.exr" % side), {"Z": d}, "ZIP", half=(m == 1))
    ds = s3r.data.StereoShapeNet(str(tmp_path), with_disparity=True)
    a = s3r.evaluate.test_dataset(hip, ds, batch=2, device=DEV)
    items = [ds[i] for i in range(len(ds))]
    left, right, gt, dl, dr = (torch.stack([it[k] for it in items]) for k in range(5))
    b = s3r.evaluate.test_net(hip, left, right, gt, batch=2, device=DEV)
    c = s3r.evaluate.test_disparity(hip, left, right, s3r.data.downsample_disparity(dl), s3r.data.downsample_disparity(dr),
                                    batch=2, device=DEV)
    assert torch.equal(a["per_sample"], b["per_sample"])
    # (the block means are formed on the device there and on the host here: fp32 summation order differs)
    assert abs(a["epe_left"] - c["epe_left"]) < 1e-5 * c["epe_left"] and abs(a["epe_right"] - c["epe_right"]) < 1e-5 * c["epe_right"]
    assert a["epe_left"] > 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("launcher,world", [("self", 2), ("torchrun", 2), ("self", 8)])
def test_bench_ranks_share_one_gpu(tmp_path, launcher, world):
    """bench.py's N>1 path (sharded batch, all-gather collation, barrier, max-over-ranks timing, rank-0 JSON) run
    with `world` ranks sharing cuda:0 over gloo — the RCCL run itself needs a multi-GPU node — at two ranks and at the
    real world size of BASELINE configs[4] (eight processes, eight arenas, eight host launch threads on one device).
    "self": plain `python bench.py --gpus N ...` as the driver invokes it (bench starts its own ranks as a child
    process); "torchrun": under an external launcher, as the contract's N>1 command line does."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "4", "--backend",
            "gloo", "--same-device", "--no-cpu-baseline"]
    if launcher == "self":
        cmd = [sys.executable] + args
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args
        env = dict(os.environ)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=580, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["config"]["global_batch"] == 4 * world and d["scaling"] == "weak"
    assert d["n_ranks_seen"] == world and d["collective_backend"] == "gloo"
    assert d["value"] > 0 and "cpu_baseline" not in d and "secondary" not in d


def test_bench_line_carries_roofline_border_excluded_and_secondaries(tmp_path):
    """The default N=1 line: roofline with frac and frac_border_excluded, no PMC constants unless the committed
    counter file matches the sources that ran, and the configs[2] / configs[3] measurements under `secondary`."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("S3R_")}      # the DEFAULT line: no kernel-policy overrides
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=500, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    # `frac` is what the matrix cores EXECUTE over their peak (<= 1: the pipe's utilisation); `frac_credited` charges the same
    # time with the direct form's FLOPs (with most convolutions on the Winograd kernels it may pass 1)
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["frac"] <= r["frac_credited"] < 1.6
    assert 0 < r["frac_credited_border_excluded"] < r["frac_credited"]
    assert 0.4 < r["executed_over_algorithmic_mfma_flops"] <= 1.0
    assert abs(r["frac"] - r["frac_credited"] * r["executed_over_algorithmic_mfma_flops"]) < 2e-3
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert set(r["what_ran"]) == {"e2", "e3", "e4", "e5", "e6", "e7", "e8", "v1", "v2", "v3", "v4", "v5", "v6", "d1", "d2", "d3"}
    assert r["what_ran"]["e3"] == "direct" and r["what_ran"]["v1"].startswith("winograd")
    # the eager pass the kernel times come from: sum of ALL kernels of a step <= the step itself
    assert r["kernel_ms_per_step"] <= r["all_kernels_ms_per_step"] <= r["eager_step_ms"] * 1.001
    if r["traffic"] is not None:                              # only ever from a counter file hashed to THESE sources
        import bench
        assert r["pmc_source"]["csrc_sha256"] == bench.csrc_sha256()
    sec = d["secondary"]
    assert sec["bf16_b256"]["dtype"] == "bf16" and sec["bf16_b256"]["value"] > d["value"]
    assert sec["bf16_b256"]["roofline"]["peak"] == 2500.0
    assert sec["point_b32"]["value"] > 0 and "chamfer" in sec["point_b32"]["kernels"] and "linear" in sec["point_b32"]["kernels"]
    # three independent batches in flight on three streams: the same work per step, never the headline; not slower than 0.9 x it
    assert sec["in_flight3_b32"]["streams"] == 3 and sec["in_flight3_b32"]["value"] > 0.9 * d["value"]


def test_maximum_sizes_chunking_and_limits(s3r, models):
    """Above MAX_CHUNK pairs the modules split the batch (every tensor of one C-ABI call must stay below 2^31
    elements / 4 GiB); a single chain call over the limit is refused, not silently wrapped."""
    hip, _ = models
    B = s3r.modules.MAX_CHUNK + 2
    g = torch.Generator().manual_seed(5)
    base_l, base_r = torch.rand(6, 3, 224, 224, generator=g).to(DEV), torch.rand(6, 3, 224, 224, generator=g).to(DEV)
    reps = -(-B // 6)
    left, right = base_l.repeat(reps, 1, 1, 1)[:B].contiguous(), base_r.repeat(reps, 1, 1, 1)[:B].contiguous()
    out = hip(left, right)
    assert out.shape == (B, 32, 32, 32)
    small = hip(base_l, base_r)
    for i in (0, 5, 255, 256, B - 1):                      # both sides of the chunk boundary
        assert torch.equal(out[i], small[i % 6])
    del out, left, right
    torch.cuda.empty_cache()
    with pytest.raises(s3r.S3RError, match="split the batch"):
        hip.encoder._run(torch.zeros(1500, 3, 224, 224, device=DEV))      # e2's 1500x64x114x114 output > 4 GiB
