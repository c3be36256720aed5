"""arch_spec — the ONE declarative layer table of this build.

BUILD-SPECIFIED, NOT THE REFERENCE'S.  The mounted reference (/root/reference) holds only
README.md + requirements.txt (README.md:5 — the model code lives on unmounted branches), so
the layer table of hzxie/Stereo-3D-Reconstruction is unknown.  This file fixes
an architecture that satisfies every constraint the mount + BASELINE.json evidence:

  * inputs  2 x (B,3,224,224) left/right RGB renders        (README.md:73-74, BASELINE.json)
  * a shared 2D conv tower over both views ("encoder")      (BASELINE.json north_star)
  * a bidirectional shift-and-diff disparity cost volume    (README.md:75-76, north_star)
  * a 3D-conv hourglass that emits a (B,32,32,32) occupancy
    grid in [0,1] ("decoder")                               (README.md:77, north_star)
  * a point head that emits (B,2048,3) for Stereo2Point     (README.md:36, BASELINE configs[3])
  * total parameters well under the ~77 M fp32 values the
    309 MB checkpoint bounds                                (README.md:35)

Every FLOP / byte figure quoted anywhere (bench.py roofline) is computed from this
table by the functions below, never typed by hand (SURVEY.md §7 Plan B step 1).

Layer record fields
  name   state_dict prefix inside its module ("e2" -> encoder.e2.conv.weight, encoder.e2.bn.*)
  op     "conv2d" | "conv3d" | "deconv3d" (ConvTranspose3d) | "linear"
  cin, cout, k, s, p          kernel / stride / padding (same on every spatial axis)
  bn     eval-mode BatchNorm after the conv (affine per channel, folded into the epilogue)
  act    "relu" | "sigmoid" | "none"
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

IMG_HW = 224          # input render size (BASELINE.json)
FEAT_C = 32           # encoder output channels per view
FEAT_HW = 28          # encoder output spatial size (stride 8)
MAX_DISP = 28         # disparity levels of the cost volume at 1/8 resolution (0..27)
VOX = 32              # output occupancy grid edge (README.md:77 ShapeNetVox32)
N_POINTS = 2048       # Stereo2Point cloud size (BASELINE.json configs[3])
LATENT_C = 512        # bottleneck channels (4^3 spatial)
BN_EPS = 1e-5


@dataclass(frozen=True)
class Layer:
    name: str
    op: str
    cin: int
    cout: int
    k: int = 3
    s: int = 1
    p: int = 1
    bn: bool = True
    act: str = "relu"
    # parameter-general layers (r05: what a reference layer table may hold beyond this build's own shapes; the network above uses
    # none of them).  dil: dilation; opad: ConvTranspose output_padding; act also "leaky_relu" | "elu" | "tanh", act_param its
    # slope / alpha (None: torch's default, 0.01 / 1.0); op also "deconv2d"
    dil: int = 1
    opad: int = 0
    act_param: float = None

    def as_dict(self):
        return asdict(self)


# ---- encoder: shared-weight 2D tower, applied to left and right --------------------------
ENCODER: Tuple[Layer, ...] = (
    Layer("e1", "conv2d", 3, 32, 3, 2, 1),      # 224 -> 112
    Layer("e2", "conv2d", 32, 64, 3, 1, 1),     # 112
    Layer("e3", "conv2d", 64, 64, 3, 2, 1),     # 112 -> 56
    Layer("e4", "conv2d", 64, 128, 3, 1, 1),    # 56
    Layer("e5", "conv2d", 128, 128, 3, 2, 1),   # 56 -> 28
    Layer("e6", "conv2d", 128, 256, 3, 1, 1),   # 28
    Layer("e7", "conv2d", 256, 256, 3, 1, 1),   # 28
    Layer("e8", "conv2d", 256, FEAT_C, 1, 1, 0),  # 1x1 feature compression
)

# ---- decoder: 3D hourglass over the (2*FEAT_C, D, H, W) cost volume ----------------------
DECODER_DOWN: Tuple[Layer, ...] = (
    Layer("v1", "conv3d", 2 * FEAT_C, 64, 3, 1, 1),   # 28^3
    Layer("v2", "conv3d", 64, 128, 3, 2, 1),          # 28 -> 14
    Layer("v3", "conv3d", 128, 128, 3, 1, 1),         # 14
    Layer("v4", "conv3d", 128, 256, 3, 2, 1),         # 14 -> 7
    Layer("v5", "conv3d", 256, 256, 3, 1, 1),         # 7
    Layer("v6", "conv3d", 256, LATENT_C, 4, 1, 0),    # 7 -> 4 (valid)
)
DECODER_UP: Tuple[Layer, ...] = (
    Layer("d1", "deconv3d", LATENT_C, 256, 4, 2, 1),  # 4 -> 8
    Layer("d2", "deconv3d", 256, 128, 4, 2, 1),       # 8 -> 16
    Layer("d3", "deconv3d", 128, 64, 4, 2, 1),        # 16 -> 32
    Layer("d4", "conv3d", 64, 1, 1, 1, 0, bn=False, act="sigmoid"),  # occupancy head
)
DECODER: Tuple[Layer, ...] = DECODER_DOWN + DECODER_UP

# ---- Stereo2Point head: MLP on the flattened 512x4^3 latent ------------------------------
POINT_HEAD: Tuple[Layer, ...] = (
    Layer("p1", "linear", LATENT_C * 4 ** 3, 1024, 1, 1, 0, bn=False, act="relu"),
    Layer("p2", "linear", 1024, 1024, 1, 1, 0, bn=False, act="relu"),
    Layer("p3", "linear", 1024, N_POINTS * 3, 1, 1, 0, bn=False, act="none"),
)


def out_size(layer: Layer, n: int) -> int:
    if layer.op in ("deconv3d", "deconv2d"):
        return (n - 1) * layer.s - 2 * layer.p + layer.dil * (layer.k - 1) + layer.opad + 1
    if layer.op == "linear":
        return 1
    return (n + 2 * layer.p - layer.dil * (layer.k - 1) - 1) // layer.s + 1


def ndim(layer: Layer) -> int:
    return {"conv2d": 2, "conv3d": 3, "deconv3d": 3, "deconv2d": 2, "linear": 0}[layer.op]


def act_param(layer: Layer) -> float:
    """Slope of leaky_relu / alpha of elu (torch's defaults when the layer does not say)."""
    if layer.act_param is not None:
        return float(layer.act_param)
    return {"leaky_relu": 0.01, "elu": 1.0}.get(layer.act, 0.0)


def trace(layers, n_in: int) -> List[Tuple[Layer, int, int]]:
    """[(layer, in_size, out_size)] walking a chain from spatial edge n_in."""
    rows, n = [], n_in
    for l in layers:
        m = out_size(l, n)
        rows.append((l, n, m))
        n = m
    return rows


def layer_macs(layer: Layer, n_in: int) -> int:
    """Mathematical multiply-adds of one layer for ONE sample (no padding / tile waste)."""
    d, n_out = ndim(layer), out_size(layer, n_in)
    if layer.op == "linear":
        return layer.cin * layer.cout
    if layer.op in ("deconv3d", "deconv2d"):   # every input voxel meets every kernel tap once
        return layer.cin * n_in ** d * layer.cout * layer.k ** d
    return layer.cout * n_out ** d * layer.cin * layer.k ** d


def layer_macs_interior(layer: Layer, n_in: int) -> int:
    """As layer_macs, but without the taps that meet zero PADDING (conv) or whose output falls outside the
    grid (transposed conv): the multiply-adds whose operands are real data.  SURVEY.md §8d's formulas (and
    layer_macs) count those border taps; the difference is 0-33 % per layer."""
    d = ndim(layer)
    if layer.op == "linear":
        return layer.cin * layer.cout
    n_out = out_size(layer, n_in)
    if layer.op == "deconv3d":   # (input i, tap t) pairs whose output i*s - p + t lies in [0, n_out)
        axis = sum(1 for i in range(n_in) for t in range(layer.k) if 0 <= i * layer.s - layer.p + t < n_out)
    else:                        # (output o, tap t) pairs whose input o*s - p + t lies in [0, n_in)
        axis = sum(1 for o in range(n_out) for t in range(layer.k) if 0 <= o * layer.s - layer.p + t < n_in)
    return layer.cin * layer.cout * axis ** d


def layer_params(layer: Layer) -> int:
    d = ndim(layer)
    n = layer.cin * layer.cout * layer.k ** d + layer.cout          # weight + bias
    if layer.bn:
        n += 4 * layer.cout                                        # gamma, beta, mean, var
    return n


def stage_table(stage: str):
    if stage == "encoder":
        return trace(ENCODER, IMG_HW)
    if stage == "decoder":
        return trace(DECODER, MAX_DISP)
    if stage == "decoder_down":
        return trace(DECODER_DOWN, MAX_DISP)
    if stage == "point_head":
        return trace(POINT_HEAD, 1)
    raise KeyError(stage)


def flops_per_pair(variant: str = "voxel") -> Dict[str, float]:
    """2*MACs per stereo pair, by stage.  Encoder runs twice (left and right)."""
    enc = 2 * 2 * sum(layer_macs(l, n) for l, n, _ in stage_table("encoder"))
    cv = 2 * FEAT_C * MAX_DISP * FEAT_HW * FEAT_HW       # one subtract per output element
    if variant == "voxel":
        dec = 2 * sum(layer_macs(l, n) for l, n, _ in stage_table("decoder"))
        head = 0
    else:
        dec = 2 * sum(layer_macs(l, n) for l, n, _ in stage_table("decoder_down"))
        head = 2 * sum(layer_macs(l, n) for l, n, _ in stage_table("point_head"))
    return {"encoder": float(enc), "cost_volume": float(cv), "decoder": float(dec),
            "point_head": float(head), "total": float(enc + cv + dec + head)}


def mfma_flops_per_pair(variant: str = "voxel", interior: bool = False) -> float:
    """FLOPs that run on the MFMA implicit-GEMM kernel (everything but e1, cost volume, and — for the point
    variant — the MLP head, which is a weight-streaming kernel; d4 runs fused inside d3's launch and is
    counted with it).  interior=True: without the taps that multiply padding zeros (layer_macs_interior)."""
    macs = layer_macs_interior if interior else layer_macs
    f = 0.0
    for l, n, _ in stage_table("encoder"):
        if l.name != "e1":
            f += 2 * 2 * macs(l, n)
    for l, n, _ in stage_table("decoder" if variant == "voxel" else "decoder_down"):
        f += 2 * macs(l, n)
    return f


def params_total(variant: str = "voxel") -> int:
    n = sum(layer_params(l) for l in ENCODER)
    if variant == "voxel":
        n += sum(layer_params(l) for l in DECODER)
    else:
        n += sum(layer_params(l) for l in DECODER_DOWN) + sum(layer_params(l) for l in POINT_HEAD)
    return n


def cost_volume_bytes_per_pair() -> int:
    """Algorithmic HBM bytes of the cost volume: read both feature maps once, write the volume."""
    feat = 2 * FEAT_C * FEAT_HW * FEAT_HW
    vol = 2 * FEAT_C * MAX_DISP * FEAT_HW * FEAT_HW
    return 4 * (feat + vol)


def describe() -> str:
    lines = ["stage    layer op        cin  cout k s p  in->out   MMAC/sample  params"]
    for stage in ("encoder", "decoder", "point_head"):
        for l, n, m in stage_table(stage):
            lines.append(f"{stage:8s} {l.name:5s} {l.op:9s} {l.cin:5d} {l.cout:5d} {l.k} {l.s} {l.p} "
                         f"{n:4d}->{m:<4d} {layer_macs(l, n) / 1e6:11.1f} {layer_params(l):9d}")
    for v in ("voxel", "point"):
        f = flops_per_pair(v)
        lines.append(f"{v}: " + ", ".join(f"{k}={x / 1e9:.3f} GFLOP" for k, x in f.items())
                     + f", params={params_total(v) / 1e6:.2f} M")
    return "\n".join(lines)


if __name__ == "__main__":
    print(describe())
