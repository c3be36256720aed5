"""Winograd-along-H path of the fp32 3 x 3 [x 3] stride-1 convolutions (F(4,3)) and the transposed convolutions (F(2,2) inside
the parity classes): csrc/s3r_conv_wino.hip.  S3R_WINO (read once, at load): unset / 1 = the library's policy (every layer
that has the form), 0 = never; each setting runs in its own child process.  It computes the same convolution with 1/2 (9/16)
of the multiplications in a different summation order, so the bar against the direct kernels is the oracle at the path's
fp32 tolerance (north_star: 1e-4 relative; measured here ~1e-6), not bit-equality; what must stay bitwise are the properties
that do not depend on the algorithm — determinism, batch invariance — and the equality of the Winograd kernel's launch forms."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_CHILD = r'''
import sys, json, torch
sys.path.insert(0, %(root)r)
import s3r
from oracle import s2v_oracle as oracle
dev = torch.device("cuda:0")
spec = s3r.arch_spec
out = {"layers": {}, "ok": True}

def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())

# ---- every eligible layer alone, two batch sizes, against the oracle's block
cases = []
for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP)):
    for l, n_in, _ in spec.trace(layers, n0):
        if l.op in ("conv2d", "conv3d") and l.k == 3 and l.s == 1 and l.p == 1 and l.cin %% 16 == 0:
            cases.append((l, n_in))
        if l.op == "conv3d" and l.k == 4 and l.s == 1 and l.p == 0:       # v6: the two-axis F(2,4) x F(2,4) form
            cases.append((l, n_in))
        if l.op == "deconv3d":                                   # F(2,2) along H inside the parity classes
            cases.append((l, n_in))
for l, n_in in cases:
    ch = s3r.modules._HipChain([l], n_in, precision="fp32")
    s3r.seed_module(ch, 7)
    blk = oracle._Block(l).eval()
    blk.load_state_dict(getattr(ch, l.name).state_dict())
    ch.to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((3, l.cin) + (n_in,) * spec.ndim(l), generator=g)
    with torch.no_grad():
        want = blk(x)
    got = ch._run(x.to(dev))
    again = ch._run(x.to(dev))
    one = ch._run(x[1:2].to(dev))
    out["layers"][l.name] = {"rel": rel_l2(got.cpu(), want), "deterministic": bool(torch.equal(got, again)),
                             "batch_invariant": bool(torch.equal(one[0], got[1])),
                             "launches": None}
# ---- d3 with the occupancy head fused into its epilogue (the chain's own plan)
dl = {l.name: (l, n_in) for l, n_in, _ in spec.trace(spec.DECODER, spec.MAX_DISP)}
ch = s3r.modules._HipChain([dl["d3"][0], dl["d4"][0]], dl["d3"][1], precision="fp32")
s3r.seed_module(ch, 9)
blocks = [oracle._Block(dl[k][0]).eval() for k in ("d3", "d4")]
for k, blk in zip(("d3", "d4"), blocks):
    blk.load_state_dict(getattr(ch, k).state_dict())
ch.to(dev)
x = torch.randn((3, 128, 16, 16, 16), generator=torch.Generator().manual_seed(4))
with torch.no_grad():
    want = blocks[1](blocks[0](x))
got = ch._run(x.to(dev))
out["head_rel"] = rel_l2(got.cpu(), want)
out["head_max"] = float((got.cpu() - want).abs().max())
out["head_batch_invariant"] = bool(torch.equal(ch._run(x[2:3].to(dev))[0], got[2]))
# ---- the whole forward against the oracle, and batch invariance of the whole forward
m = s3r.Stereo2Voxel(); s3r.seed_module(m, 0); m.to(dev)
ref = oracle.OracleStereo2Voxel(); ref.load_state_dict(m.state_dict()); ref.eval()
l, r = s3r.synthetic_pairs(3, seed=71)
with torch.no_grad():
    want = ref(l, r)
got = m(l.to(dev), r.to(dev)).clone()
out["model_rel"] = rel_l2(got.cpu(), want)
out["model_max"] = float((got.cpu() - want).abs().max())
out["model_batch_invariant"] = bool(torch.equal(m(l[2:3].to(dev), r[2:3].to(dev))[0], got[2]))
torch.save(got.cpu(), sys.argv[1])
print("RESULT " + json.dumps(out))
'''


def _run_child(tmp_path, flag):
    path = str(tmp_path / f"wino_{flag}.pt")
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}, path], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=dict(os.environ, S3R_WINO=flag))
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:]), torch.load(path)


def test_winograd_path_vs_oracle_and_invariants(tmp_path):
    res, out1 = _run_child(tmp_path, "1")                       # the library's policy: every eligible layer on the Winograd kernel
    assert set(res["layers"]) == {"e2", "e4", "e6", "e7", "v1", "v3", "v5", "v6", "d1", "d2", "d3"}
    assert res["head_rel"] < 1e-5 and res["head_max"] < 1e-5 and res["head_batch_invariant"]
    for name, r in res["layers"].items():
        assert r["rel"] < 1e-5, (name, r)                       # north_star: 1e-4 relative
        assert r["deterministic"] and r["batch_invariant"], (name, r)
    assert res["model_rel"] < 1e-5 and res["model_max"] < 1e-4 and res["model_batch_invariant"]
    res0, out0 = _run_child(tmp_path, "0")                      # direct kernels only
    assert res0["model_rel"] < 1e-5 and res0["model_batch_invariant"]
    # the switch does something: same convolution, other summation — other bits within the same bar
    assert not torch.equal(out0, out1)
    assert float((out0.double() - out1.double()).norm() / out0.double().norm()) < 1e-5


def test_cost_volume_writes_the_transformed_planes_bitwise(s3r, monkeypatch):
    """The cost volume written directly in the layout its consumer's Winograd kernel reads — the 36 two-axis plane sets
    (S3R_LAYOUT_WINO_DH: what the default policy's v1 takes) or the six one-axis plane sets (S3R_LAYOUT_WINO_H) — equals the
    input transform of the padded volume (to fp32 rounding against a float64 evaluation; bitwise through the decoder, which
    runs its own transform kernel in front of the same class kernel when handed the padded volume); both models take the
    hand-off by themselves; a batch too large for one transformed call keeps the plain hand-off."""
    if os.environ.get("S3R_WINO") not in (None, "1"):
        pytest.skip("the library's own policy is under test (S3R_WINO is read once, at load)")
    dev, L = "cuda:0", s3r._lib
    m = s3r.Stereo2Voxel()
    s3r.seed_module(m, 0)
    m.to(dev)
    l, r = s3r.synthetic_pairs(3, seed=5)
    feats = m.encoder.forward_pair(l.to(dev), r.to(dev))
    vol = m.cost_volume.forward_padded(feats[:3], feats[3:]).clone()                # (3, 64, 30, 30, 30), zero halo
    BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], dtype=torch.float64, device=dev)
    rows = torch.stack([vol[:, :, :, k:k + 25:4, :].double() for k in range(6)])    # padded rows 4q + k, q = 0 .. 6
    want1 = torch.einsum("ck,kbxdqw->cbxdqw", BT, rows)                             # F(4,3)'s input transform along H
    # ---- the default policy: v1 on the two-axis kernel
    assert m.decoder.wino_input_layout(3) == L.LAYOUT_WINO_DH
    planes2 = m.cost_volume.forward_wino2(feats[:3], feats[3:]).clone()             # (36, 3, 64, 7, 7, 30)
    deps = torch.stack([want1[:, :, :, k:k + 25:4] for k in range(6)])              # (depth k, row class, b, c, sd, q, w)
    want2 = torch.einsum("ak,kcbxsqw->acbxsqw", BT, deps).reshape(36, 3, 64, 7, 7, 30)
    assert planes2.shape == want2.shape
    assert (planes2.double() - want2).abs().max().item() < 1e-5 * max(1.0, want2.abs().max().item())
    a = m.decoder.forward_padded(vol)
    b = m.decoder.forward_padded(planes2, in_layout=L.LAYOUT_WINO_DH)
    assert torch.equal(a, b)
    assert torch.equal(m(l.to(dev), r.to(dev)), a)                                   # the model's own forward takes it
    assert m.decoder.wino_input_layout(300) == L.LAYOUT_PLAIN                        # 300 x 13.5 MB of planes: two calls
    with pytest.raises(RuntimeError):
        m.decoder.forward_padded(planes2[:, :, :, :, :, :29], in_layout=L.LAYOUT_WINO_DH)
    # ---- the one-axis kernel (forced per layer) and its layout
    m.decoder.algo_override["v1"], m.decoder.tile_override["v1"] = L.ALGO_WINOGRAD, 0
    assert m.decoder.wino_input_layout(3) == L.LAYOUT_WINO_H
    planes = m.cost_volume.forward_wino(feats[:3], feats[3:]).clone()               # (6, 3, 64, 30, 7, 30)
    assert planes.shape == want1.shape
    assert (planes.double() - want1).abs().max().item() < 2e-6 * max(1.0, want1.abs().max().item())
    assert not planes[:, :, :, 0].any() and not planes[:, :, :, -1].any()           # the depth-halo planes stay zero
    a1 = m.decoder.forward_padded(vol)
    m.decoder.tile_override.pop("v1")                       # (a transformed input takes no form override)
    m.decoder.algo_override["v1"] = L.ALGO_AUTO
    b1 = m.decoder.forward_padded(planes, in_layout=L.LAYOUT_WINO_H)
    assert torch.equal(a1, b1) and not torch.equal(a1, a)
    m.decoder.algo_override.pop("v1")
    p = s3r.Stereo2Point()
    s3r.seed_module(p, 1)
    p.to(dev)
    fp = p.encoder.forward_pair(l.to(dev), r.to(dev))
    lat_a = p.decoder.forward_padded(p.cost_volume.forward_padded(fp[:3], fp[3:]))
    vol_b, layout = p.cost_volume.forward_for(p.decoder, fp[:3], fp[3:])
    assert layout == L.LAYOUT_WINO_DH
    assert torch.equal(lat_a, p.decoder.forward_padded(vol_b, in_layout=layout))


def _wino_layers(spec):
    out = []
    for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP)):
        for l, n_in, _ in spec.trace(layers, n0):
            if (l.op in ("conv2d", "conv3d") and l.k == 3 and l.s == 1 and l.p == 1 and l.cin % 32 == 0) or l.op == "deconv3d":
                out.append((l, n_in))
    return out


def _positions(spec, l, n_in, batch):
    """GEMM positions of the Winograd launch (groups of R output rows) and its serial workgroup count."""
    if l.op == "deconv3d":
        n = batch * (n_in // 2) * (n_in // 2) * n_in               # (depth pair, row pair, column)
        return n, -(-l.cout // 64) * -(-n // 64) * 8
    n = batch * (n_in if l.op == "conv3d" else 1) * -(-n_in // 4) * n_in
    return n, -(-l.cout // 64) * -(-n // 64)


def test_every_launch_form_gives_the_serial_forms_bits(s3r):
    """The class-parallel form (one workgroup per (tile, class), class sums through slabs, `wino_finish_kernel`) and the dual
    form (bulk serial + remainder class-parallel in one launch) run the same MFMA sequence per class and the same output
    transform as the serial form: bit for bit, on every layer that has a Winograd kernel, at a batch that makes the dual
    form's cut fall inside the layer (W > 256 serial workgroups) and at a small one."""
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    for l, n_in in _wino_layers(spec):
        batches = [1]
        b = 1
        while _positions(spec, l, n_in, b)[1] <= 264 and b < 64:
            b += 1
        batches.append(b)
        for B in batches:
            ch = s3r.modules._HipChain([l], n_in, precision="fp32")
            s3r.seed_module(ch, 11)
            ch.to(dev)
            ch.algo_override[l.name] = L.ALGO_WINOGRAD
            x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), generator=torch.Generator().manual_seed(B)).to(dev)
            outs = {}
            for form in (0, 1, 2):
                ch.tile_override[l.name] = form
                s3r.profile_enable(8)
                outs[form] = ch._run(x).clone()
                rec = [r for r in s3r.profile_read(8) if r["family"] == "conv_mfma"]
                s3r.profile_enable(0)
                assert len(rec) == 1 and rec[0]["ran"].startswith("winograd"), rec
                if form == 1:
                    assert rec[0]["ran"] == "winograd-class-parallel", (l.name, form, rec)
                if form == 2 and _positions(spec, l, n_in, B)[1] > 256:
                    assert rec[0]["ran"] == "winograd-dual", (l.name, B, form, rec)
            for form, y in outs.items():
                assert torch.equal(y, outs[0]), (l.name, B, form, float((y - outs[0]).abs().max()))
            ch.tile_override.pop(l.name)                           # ... and so does whatever form the library plans itself
            s3r.profile_enable(8)                                  # (unless it plans the two-axis kernel: another algorithm)
            auto = ch._run(x)
            rec = [r for r in s3r.profile_read(8) if r["family"] == "conv_mfma"]
            s3r.profile_enable(0)
            if rec[0]["ran"] != "winograd-2axis":
                assert torch.equal(auto, outs[0]), (l.name, B, "auto form", rec[0]["ran"])
    # d3 with the occupancy head fused: serial epilogue vs the finish kernel's cout walk
    dl = {l.name: (l, n_in) for l, n_in, _ in spec.trace(spec.DECODER, spec.MAX_DISP)}
    ch = s3r.modules._HipChain([dl["d3"][0], dl["d4"][0]], dl["d3"][1], precision="fp32")
    s3r.seed_module(ch, 9)
    ch.to(dev)
    ch.algo_override["d3"] = L.ALGO_WINOGRAD
    for B in (1, 5):                                               # (5 samples: 320 serial workgroups, the dual form cuts inside)
        x = torch.randn((B, 128, 16, 16, 16), generator=torch.Generator().manual_seed(4)).to(dev)
        outs = {}
        for form in (0, 1, 2):
            ch.tile_override["d3"] = form
            outs[form] = ch._run(x).clone()
        assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0]), B


_FORMS_CHILD = r'''
import sys, torch
sys.path.insert(0, %(root)r)
import s3r
dev = torch.device("cuda:0")
m = s3r.Stereo2Voxel(); s3r.seed_module(m, 0); m.to(dev)
outs = []
for B in (1, 2, 4, 8, 32):
    l, r = s3r.synthetic_pairs(B, seed=90 + B)
    outs.append(m(l.to(dev), r.to(dev)).clone().cpu())
torch.save(outs, sys.argv[1])
'''


def test_whole_forward_is_form_invariant_at_every_batch(tmp_path):
    """B = 1, 2, 4, 8, 32: the forward under the library's own launch plan (class-parallel on sparse grids, dual where a
    layer's last round is mostly empty; the transposed layers' depth differences materialised or formed in the kernel by
    input size; the two-axis layers class-parallel or semi-fused) equals the forward with every one-axis Winograd layer forced
    to the serial form, with the depth differences forced either way, and with the two-axis layers forced to either of their
    forms, with the stem writing its plain activation for a transform kernel instead of e2's planes itself, and with e6 doing
    the same instead of writing e7's — bitwise."""
    res = {}
    for flag, env_set in (("auto", {}), ("serial", {"S3R_WINO_FORM": "0"}), ("mat0", {"S3R_DWINO_MAT": "0"}),
                          ("mat1", {"S3R_DWINO_MAT": "1"}), ("cp2", {"S3R_WINO2_FORM": "0"}), ("semi2", {"S3R_WINO2_FORM": "1"}),
                          ("stem0", {"S3R_STEM_WINO": "0"}), ("handoff0", {"S3R_WINO_HANDOFF": "0"})):
        path = str(tmp_path / f"forms_{flag}.pt")
        env = dict(os.environ)
        env.pop("S3R_WINO_FORM", None)
        env.pop("S3R_DWINO_MAT", None)
        env.pop("S3R_WINO2_FORM", None)
        env.pop("S3R_STEM_WINO", None)
        env.pop("S3R_WINO_HANDOFF", None)
        env.update(env_set)
        r = subprocess.run([sys.executable, "-c", _FORMS_CHILD % {"root": ROOT}, path], capture_output=True, text=True,
                           timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = torch.load(path)
    for flag in ("serial", "mat0", "mat1", "cp2", "semi2", "stem0", "handoff0"):
        for a, b in zip(res["auto"], res[flag]):
            assert torch.equal(a, b), (flag, a.shape)


def test_winograd_on_offset_and_heavy_tailed_inputs(s3r):
    """Every algorithm form against an fp64 convolution, beside the direct kernel's own error on the same data (the table
    `tools/wino_numerics.py` prints): randn, a large DC offset (1000 + randn: the input transforms' rows sum to zero, so the
    offset cancels BEFORE the multiplications), heavy tails (|randn| exp(2 randn)) and zero-sum kernels.  Measured: direct
    rel-L2 3e-7 .. 1e-6; one-axis Winograd 0.7 .. 2.3e-6; two-axis 2.5 .. 4.7e-6 (largest element error 1.7e-5 of the output's
    maximum).  Bars: rel-L2 < 1e-5 for every form; largest error < 5e-5 of the problem's own scale (max sum of |w| |x|:
    zero-sum kernels on an offset cancel 1000-sized terms, so errors relative to the OUTPUT are ill-conditioned for any
    fp32 summation, the direct one included) and — where the output is not a cancellation — < 5e-5 of the output's maximum
    (north_star: 1e-4); the one-axis form within 8x, the two-axis form within 16x of the direct kernel's rel-L2."""
    import torch.nn.functional as F
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    g = torch.Generator().manual_seed(123)
    cases = {l.name: (l, n) for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP))
             for l, n, _ in spec.trace(layers, n0)}
    for name in ("e4", "e7", "v3", "v5", "v6", "d2"):
        l, n_in = cases[name]
        shape = (2, l.cin) + (n_in,) * spec.ndim(l)
        inputs = {"randn": torch.randn(shape, generator=g), "offset": 1000.0 + torch.randn(shape, generator=g),
                  "heavy": torch.randn(shape, generator=g).abs() * torch.exp(2.0 * torch.randn(shape, generator=g))}
        forms = [("direct", L.ALGO_DIRECT, -1)]
        if name != "v6":
            forms.append(("one-axis", L.ALGO_WINOGRAD, 0))
        if l.op != "deconv3d":
            forms.append(("two-axis", L.ALGO_WINOGRAD, 3))
        for zero_sum in (False, True):
            ch = s3r.modules._HipChain([l], n_in, precision="fp32")
            s3r.seed_module(ch, 5)
            blk = getattr(ch, l.name)
            with torch.no_grad():
                if zero_sum:                                       # every kernel sums to zero over its taps
                    w = blk.conv.weight
                    w -= w.mean(dim=tuple(range(2, w.dim())), keepdim=True)
                blk.bn.weight.fill_(1.0); blk.bn.bias.zero_(); blk.bn.running_mean.zero_(); blk.bn.running_var.fill_(1.0 - spec.BN_EPS)
                blk.conv.bias.zero_()
            w64 = blk.conv.weight.detach().double()
            ch.to(dev)
            for kind, x in inputs.items():
                if l.op == "deconv3d":
                    want = F.conv_transpose3d(x.double(), w64, None, 2, 1)
                    mag = F.conv_transpose3d(x.double().abs(), w64.abs(), None, 2, 1)
                else:
                    f = F.conv3d if l.op == "conv3d" else F.conv2d
                    want, mag = f(x.double(), w64, None, 1, l.p), f(x.double().abs(), w64.abs(), None, 1, l.p)
                want = want.clamp_min(0.0)                          # the layer's ReLU
                scale, out_scale = float(mag.max()), float(want.abs().max())
                rel = {}
                for tag, algo, tile in forms:
                    ch.algo_override[l.name] = algo
                    if tile >= 0:
                        ch.tile_override[l.name] = tile
                    else:
                        ch.tile_override.pop(l.name, None)
                    err = (ch._run(x.to(dev)).cpu().double() - want).abs()
                    rel[tag] = float(err.norm() / want.norm())
                    where = (name, kind, zero_sum, tag, float(err.max()), scale, out_scale, rel)
                    assert float(err.max()) < 5e-5 * scale, where
                    if not (zero_sum and kind == "offset"):         # (the cancellation case: see above)
                        assert rel[tag] < 1e-5 and float(err.max()) < 5e-5 * out_scale, where
                if "one-axis" in rel:
                    assert rel["one-axis"] < 8 * rel["direct"] + 1e-7, (name, kind, zero_sum, rel)
                if "two-axis" in rel:
                    assert rel["two-axis"] < 16 * rel["direct"] + 1e-7, (name, kind, zero_sum, rel)


def test_two_axis_form_vs_oracle_and_invariants(s3r, oracle):
    """The two-axis form (tile = 3 under algo = WINOGRAD; what AUTO resolves every stride-1 layer over an edge <= 28 to — over
    D and H in 3D, over H and W in 2D, where the positions of a sub-batch lie flat): against the oracle at 1e-5, deterministic,
    batch-invariant, the same bits from both launch forms, and — where the layer has both algorithms — within rounding of the
    one-axis kernel but not its bits.  e4 (edge 56) is forced onto it too: the form is general, only its slabs make it a loss
    there.  e7 at 3 samples: 147 positions, a ragged last GEMM tile."""
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    dl = {l.name: (l, n_in) for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP))
          for l, n_in, _ in spec.trace(layers, n0)}
    for name, B in (("v5", 5), ("v6", 5), ("v3", 2), ("v1", 1), ("e7", 3), ("e6", 1), ("e4", 2)):
        l, n_in = dl[name]
        ch = s3r.modules._HipChain([l], n_in, precision="fp32")
        s3r.seed_module(ch, 21)
        blk = oracle._Block(l).eval()
        blk.load_state_dict(getattr(ch, l.name).state_dict())
        ch.to(dev)
        x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), generator=torch.Generator().manual_seed(8))
        with torch.no_grad():
            want = blk(x)
        ch.algo_override[name], ch.tile_override[name] = L.ALGO_WINOGRAD, 3
        s3r.profile_enable(8)
        got = ch._run(x.to(dev)).clone()
        rec = [r for r in s3r.profile_read(8) if r["family"] == "conv_mfma"]
        s3r.profile_enable(0)
        assert rec[0]["ran"] == "winograd-2axis" and rec[0]["exec_flops"] < 0.5 * rec[0]["flops"], rec
        rel = float((got.cpu().double() - want.double()).norm() / want.double().norm())
        assert rel < 1e-5, (name, rel)
        assert torch.equal(ch._run(x.to(dev)), got)                                   # deterministic
        assert torch.equal(ch._run(x[B - 1:].to(dev))[0], got[B - 1])                 # batch-invariant
        for form in (4, 5):                                                           # class-parallel / semi-fused launch forms
            if name == "v6" and form == 5:
                continue                                                              # (F(2,4): class-parallel only)
            ch.tile_override[name] = form
            assert torch.equal(ch._run(x.to(dev)), got), (name, form)
        ch.tile_override[name] = 3
        if name != "v6":
            ch.tile_override[name] = 0
            one = ch._run(x.to(dev))
            assert not torch.equal(one, got)
            assert float((one.double() - got.double()).norm() / got.double().norm()) < 1e-5


def test_shapes_that_are_not_the_networks(s3r, oracle):
    """The descriptors are general, the network's shapes are few: every algorithm and launch form on layer shapes the network does
    not have — edges that are not multiples of the group size (ragged last groups along one or both axes), couts that fill
    neither a 64-row tile nor a 128-row weight pad, position counts that end inside a GEMM tile — against the oracle block at
    1e-5, all forms of one algorithm bitwise equal, AUTO equal to one of the forced algorithms."""
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    Layer = spec.Layer
    cases = [  # (layer, edge, batch, algorithm / form codes it must support under algo = WINOGRAD)
        (Layer("t2a", "conv2d", 32, 48), 6, 3, (0, 1, 2, 3, 4, 5)),
        (Layer("t2b", "conv2d", 64, 96), 9, 2, (0, 1, 2, 3, 4, 5)),
        (Layer("t2c", "conv2d", 32, 130), 30, 1, (0, 1, 2, 3, 4, 5)),
        (Layer("t2d", "conv2d", 96, 64), 7, 5, (0, 1, 2, 3, 4, 5)),
        (Layer("t2e", "conv2d", 32, 64), 28, 7, (0, 1, 2, 3, 4, 5)),
        (Layer("t3d", "conv3d", 96, 32), 4, 2, (0, 1, 2, 3, 4, 5)),
        (Layer("t3a", "conv3d", 32, 40), 5, 3, (0, 1, 2, 3, 4, 5)),
        (Layer("t3b", "conv3d", 64, 72), 10, 1, (0, 1, 2, 3, 4, 5)),
        (Layer("t3c", "conv3d", 32, 64), 12, 2, (0, 1, 2, 3, 4, 5)),
        (Layer("t4a", "conv3d", 32, 48, 4, 1, 0), 5, 3, (3, 4)),
        (Layer("t4b", "conv3d", 64, 20, 4, 1, 0), 8, 2, (3, 4)),
        (Layer("tda", "deconv3d", 32, 24, 4, 2, 1), 4, 3, (0, 1, 2)),
        (Layer("tdb", "deconv3d", 64, 72, 4, 2, 1), 8, 1, (0, 1, 2)),
    ]
    for l, n_in, B, forms in cases:
        ch = s3r.modules._HipChain([l], n_in, precision="fp32")
        s3r.seed_module(ch, 13)
        blk = oracle._Block(l).eval()
        blk.load_state_dict(getattr(ch, l.name).state_dict())
        ch.to(dev)
        x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), generator=torch.Generator().manual_seed(17))
        with torch.no_grad():
            want = blk(x).double()
        outs = {}
        ch.algo_override[l.name] = L.ALGO_DIRECT
        outs["direct"] = ch._run(x.to(dev)).clone()
        ch.algo_override[l.name] = L.ALGO_WINOGRAD
        for form in forms:
            ch.tile_override[l.name] = form
            outs[form] = ch._run(x.to(dev)).clone()
        ch.tile_override.pop(l.name)
        ch.algo_override.pop(l.name)
        outs["auto"] = ch._run(x.to(dev)).clone()
        for key, y in outs.items():
            assert y.shape == want.shape, (l.name, key, y.shape, want.shape)
            rel = float((y.cpu().double() - want).norm() / want.norm())
            assert rel < 1e-5, (l.name, n_in, key, rel)
        one_axis = [f for f in forms if f <= 2]
        two_axis = [f for f in forms if f >= 3]
        for group in (one_axis, two_axis):
            for f in group[1:]:
                assert torch.equal(outs[f], outs[group[0]]), (l.name, n_in, f, float((outs[f] - outs[group[0]]).abs().max()))
        assert any(torch.equal(outs["auto"], outs[k]) for k in outs if k != "auto"), (l.name, "auto matches no forced algorithm")
        # batch invariance of the library's own pick
        assert torch.equal(ch._run(x[B - 1:].to(dev))[0], outs["auto"][B - 1]), (l.name, "batch")


def test_two_axis_class_parallel_form_has_no_edge_limit(s3r, oracle):
    """A 3D layer whose four padded output slices do not fit the semi-fused finish kernel's 64 KiB of LDS (edge > 60) is planned
    — and, forced, launched — in the class-parallel form, whose flat finish kernel stages nothing (the launcher applied
    the semi-fused bound to both forms and failed with an opaque HIP error).  Edge 68, and edge 64 with the consumer's halo on the
    output (66^2 x 4 slices > 64 KiB); the semi-fused form itself stays S3R_ERR_INVALID there."""
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    Layer = spec.Layer
    a = Layer("ta", "conv3d", 32, 32)
    for layers, edge in (([a], 68), ([a, Layer("tb", "conv3d", 32, 32, 3, 2, 1)], 64)):
        ch = s3r.modules._HipChain(layers, edge, precision="fp32")
        s3r.seed_module(ch, 5)
        blocks = [oracle._Block(l).eval() for l in layers]
        for l, blk in zip(layers, blocks):
            blk.load_state_dict(getattr(ch, l.name).state_dict())
        ch.to(dev)
        x = torch.randn((1, 32) + (edge,) * 3, generator=torch.Generator().manual_seed(edge))
        with torch.no_grad():
            want = x
            for blk in blocks:
                want = blk(want)
            want = want.double()
        ch.algo_override["ta"] = L.ALGO_DIRECT
        direct = ch._run(x.to(dev)).clone()
        ch.algo_override["ta"] = L.ALGO_WINOGRAD
        outs = []
        for form in (3, 4):                                          # the two-axis algorithm: the library's plan, class-parallel
            ch.tile_override["ta"] = form
            outs.append(ch._run(x.to(dev)).clone())
        assert torch.equal(outs[0], outs[1]), edge
        for y in outs + [direct]:
            rel = float((y.cpu().double() - want).norm() / want.norm())
            assert rel < 1e-5, (edge, rel)
        ch.tile_override["ta"] = 5                                   # semi-fused: no such form at this edge
        with pytest.raises(s3r.S3RError):
            ch._run(x.to(dev))


def test_stem_that_writes_its_consumers_planes_at_other_sizes(s3r, oracle):
    """stem + e2 as a chain (the stem writes e2's six plane sets: `stem_wino_kernel`) against the same two layers run one by one
    (plain stem activation, padded by the chain, transformed by `wino_input_kernel`): bitwise, at render sizes other than the
    network's and with an odd number of images.  (8-bit renders take the same kernel with another sample type:
    tests/test_ingest_soak_gpu.py compares the two entries bitwise on the whole tower.)"""
    dev, spec = "cuda:0", s3r.arch_spec
    e1, e2 = spec.ENCODER[0], spec.ENCODER[1]
    for size, N in ((32, 3), (64, 5), (96, 2), (224, 1), (256, 1), (40, 2)):      # (256, 40: outside the fused kernel's limits)
        pair = s3r.modules._HipChain([e1, e2], size, precision="fp32")
        s3r.seed_module(pair, 31)
        first = s3r.modules._HipChain([e1], size, precision="fp32")
        second = s3r.modules._HipChain([e2], size // 2, precision="fp32")
        first.load_state_dict({k: v for k, v in pair.state_dict().items() if k.startswith("e1.")})
        second.load_state_dict({k: v for k, v in pair.state_dict().items() if k.startswith("e2.")})
        blocks = [oracle._Block(l).eval() for l in (e1, e2)]
        for l, blk in zip((e1, e2), blocks):
            blk.load_state_dict(getattr(pair, l.name).state_dict())
        for m in (pair, first, second):
            m.to(dev)
        u8 = torch.randint(0, 256, (N, 3, size, size), dtype=torch.uint8, generator=torch.Generator().manual_seed(size))
        x = s3r.data.renders_to_float(u8)
        with torch.no_grad():
            want = blocks[1](blocks[0](x)).double()
        got = pair._run(x.to(dev))
        step = second._run(first._run(x.to(dev)))
        assert torch.equal(got, step), (size, N, float((got - step).abs().max()))
        rel = float((got.cpu().double() - want).norm() / want.norm())
        assert rel < 1e-5, (size, rel)


def test_two_axis_conv2d_pairs_hand_over_transformed_planes(s3r, oracle):
    """Two two-axis Conv2d layers in a row over an edge that is a multiple of 4: the first one's finish kernel writes the second
    one's 36 plane sets (S3R_LAYOUT_WINO_HW) instead of its activation.  Against the two layers run one by one (plain
    activation, padded, transformed by `wino2p_input_kernel`): bitwise, for both launch forms of the producer, at the network's
    e6 -> e7 and at other sizes; an edge that is not a multiple of 4 takes the plain hand-off and must agree too."""
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    Layer = spec.Layer
    enc = {l.name: l for l in spec.ENCODER}
    for a, b, edge, B in ((enc["e6"], enc["e7"], 28, 3), (Layer("ha", "conv2d", 32, 64), Layer("hb", "conv2d", 64, 48), 8, 5),
                          (Layer("ha", "conv2d", 64, 32), Layer("hb", "conv2d", 32, 32), 12, 1),
                          (Layer("ha", "conv2d", 32, 32), Layer("hb", "conv2d", 32, 40), 10, 2)):
        pair = s3r.modules._HipChain([a, b], edge, precision="fp32")
        s3r.seed_module(pair, 41)
        first = s3r.modules._HipChain([a], edge, precision="fp32")
        second = s3r.modules._HipChain([b], edge, precision="fp32")
        first.load_state_dict({k: v for k, v in pair.state_dict().items() if k.startswith(a.name + ".")})
        second.load_state_dict({k: v for k, v in pair.state_dict().items() if k.startswith(b.name + ".")})
        blocks = [oracle._Block(l).eval() for l in (a, b)]
        for l, blk in zip((a, b), blocks):
            blk.load_state_dict(getattr(pair, l.name).state_dict())
        for m in (pair, first, second):
            m.to(dev)
        x = torch.randn((B, a.cin, edge, edge), generator=torch.Generator().manual_seed(edge))
        with torch.no_grad():
            want = blocks[1](blocks[0](x)).double()
        step = second._run(first._run(x.to(dev)))
        for form in (-1, 4, 5):                                      # the producer's launch form: the library's, class-parallel, semi-fused
            if form >= 0:
                pair.algo_override[a.name], pair.tile_override[a.name] = L.ALGO_WINOGRAD, form
            got = pair._run(x.to(dev))
            assert torch.equal(got, step), (a.name, edge, B, form, float((got - step).abs().max()))
        rel = float((step.cpu().double() - want).norm() / want.norm())
        assert rel < 1e-5, (edge, rel)


def test_three_axis_transposed_form_vs_oracle_and_invariants(s3r, oracle):
    """ConvTranspose3d k4 s2 p1 as F(2,2) along D, H AND W inside the parity classes (csrc/s3r_deconv_wino3.hip: 27 / 64 of the direct
    multiplications; algo = WINOGRAD, tile = 6): against the oracle block at 1e-5 on layer shapes that fill neither a 64-cout tile nor
    a 64-position tile, at edges 8 / 16 / 32, with and without the fused 1 x 1 x 1 head; deterministic; a sample's bits do not depend
    on its batch — nor on the launch form (serial / class-parallel over the depth class); AUTO takes the form from edge 16 up (the
    network's d3) and the two-axis form below (d2)."""
    dev, spec, L = "cuda:0", s3r.arch_spec, s3r._lib
    Layer = spec.Layer
    dec = {l.name: l for l in spec.DECODER}
    cases = [([Layer("ta", "deconv3d", 32, 64, 4, 2, 1)], 8, 1), ([Layer("tb", "deconv3d", 64, 72, 4, 2, 1)], 8, 3),
             ([Layer("tc", "deconv3d", 32, 24, 4, 2, 1)], 16, 2), ([Layer("td", "deconv3d", 16, 40, 4, 2, 1, True, "none")], 32, 1),
             ([dec["d2"]], 8, 2), ([dec["d3"]], 16, 2), ([dec["d3"], dec["d4"]], 16, 3)]
    for layers, n_in, B in cases:
        first = layers[0].name
        ch = s3r.modules._HipChain(layers, n_in, precision="fp32")
        s3r.seed_module(ch, 7)
        blocks = [oracle._Block(l).eval() for l in layers]
        for l, blk in zip(layers, blocks):
            blk.load_state_dict(getattr(ch, l.name).state_dict())
        ch.to(dev)
        x = torch.randn((B, layers[0].cin) + (n_in,) * 3, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            want = x
            for blk in blocks:
                want = blk(want)
            want = want.double()
        auto = ch._run(x.to(dev)).clone()
        ch.algo_override[first], ch.tile_override[first] = L.ALGO_WINOGRAD, 6
        three = ch._run(x.to(dev)).clone()
        assert torch.equal(three, ch._run(x.to(dev))), (first, "determinism")
        assert torch.equal(ch._run(x[B - 1:].to(dev))[0], three[B - 1]), (first, "batch")
        # launch forms: class-parallel over the depth class (tile 7: slabs + dwino3_finish_kernel) and serial (tile 8) give the
        # plan's bits (tile 6 picks between them by batch), with and without the fused head
        for code in (7, 8):
            ch.tile_override[first] = code
            assert torch.equal(ch._run(x.to(dev)), three), (first, n_in, "launch form", code)
            assert torch.equal(ch._run(x[:1].to(dev))[0], three[0]), (first, n_in, "launch form, batch 1", code)
        ch.tile_override[first] = 6
        rel = float((three.cpu().double() - want).norm() / want.norm())
        assert rel < 1e-5, (first, n_in, rel)
        if n_in >= 16:
            assert torch.equal(auto, three), (first, n_in, "AUTO: three axes from edge 16 up")
        elif layers[0].cin % 32 == 0:                                # (the two-axis form wants 32-channel K tiles)
            ch.tile_override[first] = -1                             # algo = WINOGRAD, the library's pick: the two-axis form
            two = ch._run(x.to(dev)).clone()
            assert float((two.cpu().double() - want).norm() / want.norm()) < 1e-5
            assert torch.equal(auto, two), (first, n_in, "AUTO: two axes below edge 16")
    bad = s3r.modules._HipChain([Layer("te", "deconv3d", 32, 32, 4, 2, 1)], 12, precision="fp32")      # edge 12: no such form
    s3r.seed_module(bad, 1)
    bad.to(dev)
    bad.algo_override["te"], bad.tile_override["te"] = L.ALGO_WINOGRAD, 6
    with pytest.raises(s3r.S3RError):
        bad._run(torch.randn(1, 32, 12, 12, 12, device=dev))
