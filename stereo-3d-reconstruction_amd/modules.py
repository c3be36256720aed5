"""Host-side mirror of the reference's plugin surface for this path: ordinary `torch.nn.Module`s
named `encoder`, `cost_volume`, `decoder` (BASELINE.json north_star; the reference's own class
names live on unmounted branches, /root/reference/README.md:5) whose parameters are plain
nn.Conv/BatchNorm parameters — so `state_dict()` / `load_state_dict()` behave exactly like a stock
PyTorch model — but whose `forward` enqueues the hand-written HIP kernels of libs3r_hip.so through
its C-ABI.  PyTorch is plumbing here (device memory, streams); no torch conv / BN op runs in any
forward below, and nothing falls back to the CPU: without the HIP library every forward raises.

Forward-only (inference, eval-mode BatchNorm): outputs carry no autograd graph.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from . import arch_spec as spec

MAX_CHUNK = 256      # pairs per C-ABI call: keeps every activation < 2^31 elements / 4 GiB

# Per-layer descriptor overrides of tuning / diagnosis runs: layer name -> value, for every module of the process, while a
# `debug_overrides(...)` context is open.  The release path consults nothing else: no environment variable reaches a descriptor
# (r05 read S3R_TILE_<layer> / S3R_KSPLIT_<layer> / S3R_ALGO_<layer> on every forward).
_DEBUG = {"tile": {}, "ksplit": {}, "algo": {}}


class debug_overrides:
    """`with s3r.debug_overrides(tile={"v2": 2}, ksplit={"v4": 2}, algo={"e6": s3r.ALGO_DIRECT}): ...`

    Forces descriptor fields of named layers — the direct kernel's tile code, the split-K factor, the algorithm — for every
    forward enqueued inside the context.  These CHANGE RESULT BITS (another summation order / another algorithm): a tuning and
    diagnosis facility (tools/layer_bench.py, tools/ab_*.py), never set on a deployment path.  Nesting merges; leaving restores.
    A module captured in a HIP graph keeps the plan it was captured with."""

    def __init__(self, tile=None, ksplit=None, algo=None):
        self._new = {"tile": dict(tile or {}), "ksplit": dict(ksplit or {}), "algo": dict(algo or {})}
        for kind, d in self._new.items():
            for name, v in d.items():
                if not isinstance(name, str) or not isinstance(v, int):
                    raise TypeError(f"debug_overrides({kind}=...): layer name -> int, got {name!r}: {v!r}")

    def __enter__(self):
        self._saved = {k: dict(v) for k, v in _DEBUG.items()}
        for k, d in self._new.items():
            _DEBUG[k].update(d)
        return self

    def __exit__(self, *exc):
        for k in _DEBUG:
            _DEBUG[k].clear()
            _DEBUG[k].update(self._saved[k])
        return False

    @staticmethod
    def active() -> dict:
        """what is in force now ({} x 3 on a release path; bench.py marks a run under any of it as not the plain configuration)"""
        return {k: dict(v) for k, v in _DEBUG.items() if v}


def _to_channels_last_physical(x: torch.Tensor) -> torch.Tensor:
    """logical (B,C,...) tensor -> contiguous physical (B,...,C) tensor (no copy if already channels-last)."""
    nd = x.dim()
    return x.permute(0, *range(2, nd), 1).contiguous()


def _check_input_cl(x: torch.Tensor, name: str, shape_tail: Sequence[int], dtype=torch.bfloat16) -> torch.Tensor:
    """bf16 hand-off: validate a LOGICAL (B,C,...) tensor and return its PHYSICAL channels-last (B,...,C) form.  A
    tensor that already is channels-last in memory (what the bf16 modules hand each other: `_to_logical` views, also
    sliced along the batch) passes through as a view — no kernel runs; anything else is copied once."""
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {x.device}); this path has no CPU fallback")
    if x.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {x.dtype})")
    if tuple(x.shape[1:]) != tuple(shape_tail):
        raise RuntimeError(f"{name} must have shape (B, {', '.join(map(str, shape_tail))}), got {tuple(x.shape)}")
    nd = x.dim()
    phys = x.permute(0, *range(2, nd), 1)
    return phys if phys.is_contiguous() else phys.contiguous()


def _to_logical(x: torch.Tensor) -> torch.Tensor:
    """physical channels-last (B,...,C) -> logical (B,C,...) view (torch's channels_last convention)."""
    nd = x.dim()
    return x.permute(0, nd - 1, *range(1, nd - 1))


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_input(x: torch.Tensor, name: str, shape_tail: Sequence[int], dtype=torch.float32):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {x.device}); this path has no CPU fallback")
    if x.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {x.dtype})")
    if tuple(x.shape[1:]) != tuple(shape_tail):
        raise RuntimeError(f"{name} must have shape (B, {', '.join(map(str, shape_tail))}), got {tuple(x.shape)}")
    return x.contiguous()


def _check_render(x: torch.Tensor, name: str) -> torch.Tensor:
    """A batch of renders, (B,3,224,224): float32 in [0,1], or uint8 as a PNG decode yields them (the stem scales by
    1/255 as it reads, bit-identical to `x.float() / 255` on the host: `s3r_encoder_forward_u8`)."""
    dt = x.dtype if isinstance(x, torch.Tensor) and x.dtype == torch.uint8 else torch.float32
    x = _check_input(x, name, (3, spec.IMG_HW, spec.IMG_HW), dt)
    if x.data_ptr() % 16:           # the stems fetch whole rows by 16-byte LDS-DMA: a view at an odd storage offset is copied once
        x = x.clone()
    return x


@torch.no_grad()
def channels_last_to_f32(x: torch.Tensor) -> torch.Tensor:
    """Hand-off from the bf16 path to an fp32 consumer: a logical (B,C,...) bfloat16 tensor in channels-last memory (what
    the bf16 modules return) -> contiguous fp32 (B,C,...), exact, by `s3r_channels_last_to_f32` (no torch kernel)."""
    phys = _check_input_cl(x, "x", x.shape[1:])
    B, Cc = x.shape[0], x.shape[1]
    y = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    if B and y.numel():
        _lib.check(_lib.load().s3r_channels_last_to_f32(phys.data_ptr(), y.data_ptr(), B, Cc, y[0, 0].numel(),
                                                        _stream_ptr(x.device)), "channels_last_to_f32")
    return y


class _Workspace:
    """A chain's activation arena.  Every intermediate of the chain has its own region in it, stored with
    the zero halo the next layer's gather reads; the kernels write interiors only, so the
    arena is zeroed ONCE per (layout = batch size) — `fresh` tells the C-ABI call to do that — and then
    stays resident: nothing is re-allocated or re-zeroed on the steady-state path."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.layout = None
        self.pinned = False      # a captured HIP graph holds this buffer's address and layout

    def get(self, device, elems: int, layout):
        fresh = 0
        if self.buf is None or self.buf.device != device or self.buf.numel() < elems or self.layout != layout:
            if self.pinned:
                raise RuntimeError("this module's activation arena is captured in a HIP graph (GraphedForward) for "
                                   f"layout {self.layout}; run other batch shapes on another module instance")
        if self.buf is None or self.buf.device != device or self.buf.numel() < elems:
            self.buf = torch.empty(max(elems, 1), dtype=torch.float32, device=device)
            fresh = 1
        if self.layout != layout:
            self.layout, fresh = layout, 1
        return self.buf, fresh


class _Block(nn.Module):
    """Parameter holder for one arch_spec.Layer (conv [+ bn]); never called."""

    def __init__(self, layer: spec.Layer):
        super().__init__()
        self.layer = layer
        if layer.op == "conv2d":
            self.conv = nn.Conv2d(layer.cin, layer.cout, layer.k, layer.s, layer.p, dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm2d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "conv3d":
            self.conv = nn.Conv3d(layer.cin, layer.cout, layer.k, layer.s, layer.p, dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm3d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "deconv3d":
            self.conv = nn.ConvTranspose3d(layer.cin, layer.cout, layer.k, layer.s, layer.p, output_padding=layer.opad,
                                           dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm3d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "deconv2d":
            self.conv = nn.ConvTranspose2d(layer.cin, layer.cout, layer.k, layer.s, layer.p, output_padding=layer.opad,
                                           dilation=layer.dil, bias=True)
            self.bn = nn.BatchNorm2d(layer.cout, eps=spec.BN_EPS) if layer.bn else None
        elif layer.op == "linear":
            self.conv = nn.Linear(layer.cin, layer.cout, bias=True)
            self.bn = None
        else:
            raise ValueError(layer.op)

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: the enclosing HIP module runs the kernel")

    @torch.no_grad()
    def folded(self):
        """(scale, shift) of the fused epilogue y = conv_nobias(x) * scale + shift."""
        bias = self.conv.bias.detach().float()
        if self.bn is None:
            return None, bias.contiguous()
        bn = self.bn
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        shift = bn.bias.detach().float() + (bias - bn.running_mean.detach().float()) * scale
        return scale.contiguous(), shift.contiguous()


class _HipChain(nn.Module):
    """A sequence of arch_spec layers executed by one C-ABI stage call."""

    _entry = "s3r_chain_forward"

    def __init__(self, layers: Sequence[spec.Layer], in_size: int, tag_base: int = 0, precision: str = "fp32",
                 winograd=None):
        super().__init__()
        if precision not in _lib.DTYPE:
            raise ValueError(f"precision must be one of {list(_lib.DTYPE)}")
        if winograd not in (None, True, False):
            raise ValueError("winograd must be None (the library's geometry-only policy), True or False")
        self.precision = precision
        # fp32 convolution algorithm of every layer that has both kernels (include/s3r.h, s3r_algo): None = the library's
        # policy; False = the direct kernels only (what a latency-bound deployment at batches < 8 may prefer: DESIGN.md);
        # True = the Winograd kernel wherever the layer has one.  The two agree to fp32 rounding, not bit for bit, so the
        # choice belongs to the model, never to the batch.
        self.winograd = winograd
        self.algo_override: Dict[str, int] = {}      # layer name -> _lib.ALGO_*                    (tuning / tests)
        self._dtype = _lib.DTYPE[precision]
        self._layers = tuple(layers)
        self._in_size = in_size
        self._tag_base = tag_base
        self.names = [l.name for l in layers]
        for l in layers:
            self.add_module(l.name, _Block(l))
        self._packed = None          # (key, [(packed_w, scale, shift)])
        self._ws = _Workspace()
        self.tile_override: Dict[str, int] = {}      # layer name -> tile cfg + 16*gather width  (tuning)
        self.ksplit_override: Dict[str, int] = {}    # layer name -> split-K factor               (tuning)
        if self.training:
            self.eval()              # inference path: eval-mode BatchNorm is the only mode implemented

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("this build implements the forward/inference path only (eval-mode BatchNorm)")
        return super().train(False)

    # -- weight packing cache ------------------------------------------------------------------
    def _cache_key(self, device):
        # (the bf16 weight image depends on the matrix instruction the library packs for: S3R_BF16_MFMA, an A/B switch)
        items = [str(device), self.precision, os.environ.get("S3R_BF16_MFMA", "")]
        for t in list(self.parameters()) + list(self.buffers()):
            items.append((id(t), t._version, t.device.type))
        return tuple(items)

    def _sizes(self):
        rows, n = [], self._in_size
        for l in self._layers:
            m = spec.out_size(l, n)
            rows.append((n, m))
            n = m
        return rows

    @torch.no_grad()
    def _ensure_packed(self, device):
        key = self._cache_key(device)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        if self._ws.pinned:
            raise RuntimeError("parameters changed after this module was captured in a HIP graph (GraphedForward): "
                               "re-capture (build a new GraphedForward) after load_state_dict / .to()")
        lib = _lib.load()
        packed = []
        stream = _stream_ptr(device)
        for l, (n_in, _) in zip(self._layers, self._sizes()):
            blk: _Block = getattr(self, l.name)
            w = blk.conv.weight.detach()
            if w.device != device:
                raise RuntimeError(f"{l.name}: parameters are on {w.device}, input on {device}; call .to(device) first")
            w = w.float().contiguous()
            desc = _lib.make_desc(l, 1, n_in, dtype=self._dtype)
            n = C.c_int64(0)
            _lib.check(lib.s3r_conv_packed_elems(C.byref(desc), C.byref(n)), f"{l.name}: packed_elems")
            pw = torch.empty(n.value, dtype=torch.float32, device=device)
            _lib.check(lib.s3r_conv_pack_weights(C.byref(desc), w.data_ptr(), pw.data_ptr(), stream),
                       f"{l.name}: pack_weights")
            scale, shift = blk.folded()
            packed.append((pw, scale, shift, w))   # keep w alive until the pack kernel has run
        self._packed = (key, packed)
        return packed

    def _layer_array(self, batch: int, device, upto: Optional[str] = None, in_halo: int = 0, in_layout: int = 0):
        packed = self._ensure_packed(device)
        n_layers = len(self._layers) if upto is None else self.names.index(upto) + 1
        arr = (_lib.Layer * n_layers)()
        for i, (l, (n_in, _)) in enumerate(zip(self._layers[:n_layers], self._sizes())):
            pw, scale, shift, _ = packed[i]
            tile, ksplit = self._tile_ksplit_of(l)
            algo = self._algo_of(l)
            # only the chain's own input / output halos are the caller's to state (the output is always
            # a plain contiguous tensor); the library plans the intermediates
            arr[i].desc = _lib.make_desc(l, batch, n_in, tag=self._tag_base + i, tile=tile,
                                         in_halo=in_halo if i == 0 else 0, ksplit=ksplit,
                                         dtype=self._dtype, in_layout=in_layout if i == 0 else 0, algo=algo)
            arr[i].packed_w = pw.data_ptr()
            arr[i].scale = scale.data_ptr() if scale is not None else None
            arr[i].shift = shift.data_ptr() if shift is not None else None
        return arr, n_layers

    def _has_winograd_form(self, l: spec.Layer) -> bool:
        """Whether the LIBRARY has a Winograd form for this layer at the size it has in this chain (edge >= 4, a transposed
        layer's edge % 4 == 0, no sigmoid, ...: the rules live in s3r_api.hip, so the question is put to the scratch query of a
        descriptor with algo = WINOGRAD instead of being mirrored here)."""
        if self.precision != "fp32" or l.op not in ("conv2d", "conv3d", "deconv3d"):
            return False
        n_in = self._sizes()[self.names.index(l.name)][0]
        desc = _lib.make_desc(l, 1, n_in, in_halo=1 if l.op == "deconv3d" else l.p, dtype=self._dtype, algo=_lib.ALGO_WINOGRAD)
        return _lib.load().s3r_conv_scratch_elems(C.byref(desc)) >= 0

    def _tile_ksplit_of(self, l: spec.Layer):
        """(tile, ksplit) of a layer's descriptor: an open `debug_overrides` context, else this module's tuning tables
        (`autotune`), else the library's own choice (-1, 0)."""
        return (int(_DEBUG["tile"].get(l.name, self.tile_override.get(l.name, -1))),
                int(_DEBUG["ksplit"].get(l.name, self.ksplit_override.get(l.name, 0))))

    def _algo_of(self, l: spec.Layer) -> int:
        """s3r_algo of a layer's descriptor: a per-layer override, else the model's `winograd` switch (only layers that
        have the Winograd form can be asked for it), else AUTO."""
        if l.name in _DEBUG["algo"]:
            return int(_DEBUG["algo"][l.name])
        if l.name in self.algo_override:
            return int(self.algo_override[l.name])
        if self.winograd is None:
            return _lib.ALGO_AUTO
        if self.winograd and self._has_winograd_form(l):
            return _lib.ALGO_WINOGRAD
        return _lib.ALGO_DIRECT

    def wino_input_layout(self, batch: int) -> int:
        """The transformed layout the chain's first layer would read from its producer for a halo-1 input of this batch —
        `_lib.LAYOUT_WINO_H` (one-axis Winograd kernel), `_lib.LAYOUT_WINO_DH` (two-axis kernel) — or `_lib.LAYOUT_PLAIN` when
        it runs the direct kernel or the batch does not fit one transformed call (CostVolume.forward_wino / forward_wino2)."""
        if self.precision != "fp32" or batch <= 0:
            return _lib.LAYOUT_PLAIN
        l, (n_in, _) = self._layers[0], self._sizes()[0]
        tile, ksplit = self._tile_ksplit_of(l)
        desc = _lib.make_desc(l, batch, n_in, tile=tile, in_halo=1, ksplit=ksplit, dtype=self._dtype, algo=self._algo_of(l))
        return _lib.check(_lib.load().s3r_conv_wino_input_layout(C.byref(desc)), "wino_input_layout")

    def takes_wino_input(self, batch: int) -> bool:
        return self.wino_input_layout(batch) != _lib.LAYOUT_PLAIN

    def _out_shape(self, batch: int, n_layers: int):
        l = self._layers[n_layers - 1]
        m = self._sizes()[n_layers - 1][1]
        nd = spec.ndim(l)
        return (batch, l.cout) + (m,) * nd

    def _out_is_bf16(self, n_layers: int) -> bool:
        l = self._layers[n_layers - 1]
        return self.precision == "bf16" and not (l.cout == 1 and l.k == 1)      # the occupancy head writes fp32

    @torch.no_grad()
    def _run(self, x: torch.Tensor, upto: Optional[str] = None, in_halo: int = 0,
             x2: Optional[torch.Tensor] = None, in_layout: int = 0) -> torch.Tensor:
        """x: the chain input; with in_halo > 0 it is a halo-padded buffer (B, C, n+2h, ...) whose border
        is zero (internal hand-off from the cost-volume kernel), otherwise a plain contiguous tensor.
        x2 (encoder only): a second tensor of as many images — the chain runs over x's images, then x2's."""
        lib = _lib.load()
        device, batch = x.device, x.shape[0] + (x2.shape[0] if x2 is not None else 0)
        if in_layout in (_lib.LAYOUT_WINO_H, _lib.LAYOUT_WINO_DH):      # (6 | 36, B, C, ...): the transformed plane sets of the padded input
            batch = x.shape[1]
        if batch == 0:                 # same dtype / layout contract as a non-empty batch
            n0 = len(self._layers) if upto is None else self.names.index(upto) + 1
            shape = self._out_shape(0, n0)
            if self._out_is_bf16(n0):
                return _to_logical(torch.empty((0,) + shape[2:] + (shape[1],), dtype=torch.bfloat16, device=device))
            return torch.empty(shape, dtype=torch.float32, device=device)
        arr, n = self._layer_array(batch, device, upto, in_halo, in_layout)
        shape = self._out_shape(batch, n)
        if self._out_is_bf16(n):    # physical channels-last (B,...,C) bf16; handed back as a logical (B,C,...) view
            y = torch.empty((shape[0],) + shape[2:] + (shape[1],), dtype=torch.bfloat16, device=device)
        else:
            y = torch.empty(shape, dtype=torch.float32, device=device)
        need = _lib.check(lib.s3r_chain_workspace_elems(arr, n), "workspace query")
        # the arena's layout is the library's PLAN for this chain: batch, depth, input halo and every layer's
        # requested (tile, split-K) — those decide head fusion and where the split-K scratch starts.  Any change
        # re-zeroes the arena (fresh), so no region is ever read with another plan's bytes in its halo.
        cfg = tuple((arr[i].desc.tile, arr[i].desc.ksplit, arr[i].desc.algo) for i in range(n)) + (in_layout,)
        ws, fresh = self._ws.get(device, need, (batch, n, in_halo, need, cfg))
        u8 = x.dtype == torch.uint8
        if (u8 or x2 is not None) and self._entry != "s3r_encoder_forward":
            raise RuntimeError("8-bit renders / a (left, right) tensor pair are inputs of the encoder only")
        if x2 is not None and x2.dtype != x.dtype:
            raise RuntimeError("left and right renders must share one dtype")
        if x2 is not None or u8:                 # (any prefix of the tower: the entry takes a chain that starts with the stem)
            enc = lib.s3r_encoder_forward_u8 if u8 else lib.s3r_encoder_forward
            _lib.check(enc(arr, n, x.data_ptr(), x2.data_ptr() if x2 is not None else None, y.data_ptr(), ws.data_ptr(),
                           ws.numel(), fresh, _stream_ptr(device)), type(self).__name__)
        elif upto is None and self._entry == "s3r_encoder_forward":
            _lib.check(lib.s3r_encoder_forward(arr, n, x.data_ptr(), None, y.data_ptr(), ws.data_ptr(), ws.numel(), fresh,
                                               _stream_ptr(device)), type(self).__name__)
        else:
            entry = getattr(lib, self._entry if upto is None else "s3r_chain_forward")
            _lib.check(entry(arr, n, x.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), fresh, _stream_ptr(device)),
                       type(self).__name__)
        return _to_logical(y) if self._out_is_bf16(n) else y


    # -- measured per-layer configuration ----------------------------------------------------------
    _TUNE_TILES = (1, 2, 3, 7, 0)               # fp32 tile configurations (s3r_conv_glds.hip)
    _TUNE_TILES_BF16 = (1, 3, 9, 17, 22, 23)    # bf16: per-tap (128 / 128x128 / 32-ch K), row-reuse, plane-reuse.  (30 / 31,
                                                # plane-reuse over a parity-split input, are not candidates: forcing them
                                                # also switches the PRODUCER's output layout, a cost these per-layer
                                                # timings would not count; the hand-off measured slower)
    _TUNE_KSPLITS = (1, 2, 4, 8)

    def _mfma_layers(self):
        return [l for l in self._layers
                if l.op in ("conv2d", "conv3d", "deconv3d") and l.cin % 16 == 0 and not (l.cout == 1 and l.k == 1)]

    @torch.no_grad()
    def autotune(self, x: torch.Tensor, in_halo: int = 0, rounds: int = 3, log=None):
        """Pick every MFMA layer's (tile, split-K) by timing the candidates on `x` (HIP events around the
        layer's own launches, median of `rounds` runs of the whole chain) and keep the winners in
        tile_override / ksplit_override.  Split-K changes a sample's summation order, so an autotuned
        module is deterministic for a given batch size but no longer bit-identical ACROSS batch sizes."""
        chosen = {}
        for l in self._mfma_layers():
            tag = self._tag_base + self.names.index(l.name)
            bf16 = self.precision == "bf16"
            chunks = l.cin // (32 if bf16 else 16)
            best = None
            for ks in self._TUNE_KSPLITS:
                if chunks % ks:
                    continue
                for t in (self._TUNE_TILES_BF16 if bf16 else self._TUNE_TILES):
                    self.tile_override[l.name], self.ksplit_override[l.name] = t, ks
                    ms = []
                    try:
                        for r in range(rounds + 1):
                            _lib.profile_enable(8 * len(self._layers) + 8)
                            self._run(x, None, in_halo)
                            rec = _lib.profile_read(8 * len(self._layers) + 8)
                            _lib.profile_enable(0)
                            if r:
                                ms.append(sum(q["ms"] for q in rec if q["tag"] == tag and q["family"] == "conv_mfma"))
                    except _lib.S3RError:
                        _lib.profile_enable(0)
                        continue
                    ms.sort()
                    if best is None or ms[len(ms) // 2] < best[0]:
                        best = (ms[len(ms) // 2], t, ks)
            if best is None:
                self.tile_override.pop(l.name, None), self.ksplit_override.pop(l.name, None)
                continue
            self.tile_override[l.name], self.ksplit_override[l.name] = best[1], best[2]
            chosen[l.name] = {"tile": best[1], "ksplit": best[2], "ms": round(best[0], 4)}
            if log:
                log(f"autotune {l.name}: tile {best[1]} ksplit {best[2]} -> {best[0]:.4f} ms")
        return chosen


class Encoder(_HipChain):
    """Shared-weight 2D conv tower: (N,3,224,224) renders -> (N,32,28,28) features.

    The same weights serve both views: `forward_pair(left, right)` runs the tower once over both tensors (what
    Stereo2Voxel / Stereo2Point call); `forward(images)` takes any single batch of renders.
    """
    _entry = "s3r_encoder_forward"

    def __init__(self, precision: str = "fp32", winograd=None):
        super().__init__(spec.ENCODER, spec.IMG_HW, tag_base=100, precision=precision, winograd=winograd)

    def forward(self, images: torch.Tensor, upto: Optional[str] = None) -> torch.Tensor:
        """fp32: (N,32,28,28) contiguous.  bf16: (N,32,28,28) bfloat16 in channels_last memory format.
        `images`: float32 in [0,1] or uint8 (`_check_render`).
        `upto`: stop after the named layer (stage-by-stage checks), as on Decoder.forward."""
        x = _check_render(images, "images")
        return self._run(x, upto)

    def forward_pair(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        """The tower over the B left and the B right renders in ONE pass, read from their two tensors (the first
        kernel picks its source by image index): features (2B,32,28,28), left batch first.  Equal, bit for bit, to
        `forward(torch.cat([left, right]))` — without the concatenation copy."""
        left = _check_render(left, "left")
        right = _check_render(right, "right")
        if left.shape[0] != right.shape[0] or left.device != right.device or left.dtype != right.dtype:
            raise RuntimeError("left and right must be two batches of one size and dtype on one device")
        if left.shape[0] == 0:
            return self._run(left)
        return self._run(left, None, 0, right)


class CostVolume(nn.Module):
    """Bidirectional shift-and-diff disparity cost volume, (B,C,H,W) x2 -> (B,2C,D,H,W)."""

    def __init__(self, max_disp: int = spec.MAX_DISP, precision: str = "fp32"):
        super().__init__()
        self.max_disp = max_disp
        self.precision = precision
        self._padded: Optional[torch.Tensor] = None      # resident halo-padded volume (internal hand-off)
        self._planes: Optional[torch.Tensor] = None      # ... or its Winograd-transformed planes (forward_wino / forward_wino2)

    def _bf16(self, feat_left, feat_right, halo, resident):
        """bf16 path: logical (B,C,H,W) channels_last features -> physical (B,D+2h,H+2h,W+2h,2C) volume."""
        if feat_left.shape != feat_right.shape or feat_left.dim() != 4:
            raise RuntimeError("feature maps must both be (B,C,H,W)")
        fl = _check_input_cl(feat_left, "feat_left", feat_left.shape[1:])
        fr = _check_input_cl(feat_right, "feat_right", feat_left.shape[1:])
        B, H, W, Cc = fl.shape
        shape = (B, self.max_disp + 2 * halo, H + 2 * halo, W + 2 * halo, 2 * Cc)
        if resident:
            if self._padded is None or tuple(self._padded.shape) != shape or self._padded.device != fl.device \
                    or self._padded.dtype != torch.bfloat16:
                if getattr(self, "_pinned", False):
                    raise RuntimeError("the padded cost volume of this module is captured in a HIP graph for another shape")
                self._padded = torch.zeros(shape, dtype=torch.bfloat16, device=fl.device)
            vol = self._padded
        else:
            vol = torch.empty(shape, dtype=torch.bfloat16, device=fl.device)
        if B:
            _lib.check(_lib.load().s3r_cost_volume_forward_bf16(fl.data_ptr(), fr.data_ptr(), vol.data_ptr(), B, Cc,
                                                                self.max_disp, H, W, halo, _stream_ptr(fl.device)),
                       "cost_volume (bf16)")
        return vol

    @torch.no_grad()
    def forward_padded(self, feat_left: torch.Tensor, feat_right: torch.Tensor, halo: int = 1) -> torch.Tensor:
        """Internal hand-off to the decoder: the volume written straight into a resident
        (B,2C,D+2h,H+2h,W+2h) buffer whose zero halo the decoder's first 3D conv reads as its padding.
        The buffer is zeroed when (re)allocated; the kernel writes the interior only."""
        if self.precision == "bf16":
            return self._bf16(feat_left, feat_right, halo, resident=True)
        fl = _check_input(feat_left, "feat_left", feat_left.shape[1:])
        fr = _check_input(feat_right, "feat_right", feat_left.shape[1:])
        B, Cc, H, W = fl.shape
        shape = (B, 2 * Cc, self.max_disp + 2 * halo, H + 2 * halo, W + 2 * halo)
        if self._padded is None or tuple(self._padded.shape) != shape or self._padded.device != fl.device:
            if getattr(self, "_pinned", False):
                raise RuntimeError("the padded cost volume of this module is captured in a HIP graph for another shape")
            self._padded = torch.zeros(shape, dtype=torch.float32, device=fl.device)
        if B == 0:
            return self._padded
        _lib.check(_lib.load().s3r_cost_volume_forward(fl.data_ptr(), fr.data_ptr(), self._padded.data_ptr(), B, Cc,
                                                       self.max_disp, H, W, halo, _stream_ptr(fl.device)),
                   "cost_volume")
        return self._padded

    @torch.no_grad()
    def forward_wino(self, feat_left: torch.Tensor, feat_right: torch.Tensor) -> torch.Tensor:
        """fp32 internal hand-off to a decoder whose first 3D conv runs the Winograd kernel (`takes_wino_input`): the volume
        written directly as the six F(4,3)-along-H plane sets of its halo-1 padded form, (6,B,2C,D+2,H/4,W+2) — bit-identical
        to the consumer's own input transform, without the volume's round trip through HBM."""
        fl = _check_input(feat_left, "feat_left", feat_left.shape[1:])
        fr = _check_input(feat_right, "feat_right", feat_left.shape[1:])
        B, Cc, H, W = fl.shape
        shape = (6, B, 2 * Cc, self.max_disp + 2, H // 4, W + 2)      # F(4,3) along H: six plane sets, one row per four outputs
        if self._planes is None or tuple(self._planes.shape) != shape or self._planes.device != fl.device:
            if getattr(self, "_pinned", False):
                raise RuntimeError("the padded cost volume of this module is captured in a HIP graph for another shape")
            self._planes = torch.zeros(shape, dtype=torch.float32, device=fl.device)
        _lib.check(_lib.load().s3r_cost_volume_forward_wino(fl.data_ptr(), fr.data_ptr(), self._planes.data_ptr(), B, Cc,
                                                            self.max_disp, H, W, _stream_ptr(fl.device)), "cost_volume (wino)")
        return self._planes

    @torch.no_grad()
    def forward_wino2(self, feat_left: torch.Tensor, feat_right: torch.Tensor) -> torch.Tensor:
        """The same hand-off for a decoder whose first 3D conv runs the TWO-AXIS Winograd kernel: the 36 F(4,3) x F(4,3) plane
        sets of the halo-1 padded volume, (36,B,2C,D/4,H/4,W+2), bit-identical to the consumer's own input transform."""
        fl = _check_input(feat_left, "feat_left", feat_left.shape[1:])
        fr = _check_input(feat_right, "feat_right", feat_left.shape[1:])
        B, Cc, H, W = fl.shape
        shape = (36, B, 2 * Cc, self.max_disp // 4, H // 4, W + 2)
        if self._planes is None or tuple(self._planes.shape) != shape or self._planes.device != fl.device:
            if getattr(self, "_pinned", False):
                raise RuntimeError("the padded cost volume of this module is captured in a HIP graph for another shape")
            self._planes = torch.zeros(shape, dtype=torch.float32, device=fl.device)
        _lib.check(_lib.load().s3r_cost_volume_forward_wino2(fl.data_ptr(), fr.data_ptr(), self._planes.data_ptr(), B, Cc,
                                                             self.max_disp, H, W, _stream_ptr(fl.device)), "cost_volume (wino2)")
        return self._planes

    def forward_for(self, consumer, feat_left: torch.Tensor, feat_right: torch.Tensor):
        """(volume, in_layout) in whatever form `consumer` (a Decoder / VolumeEncoder) reads fastest for this batch."""
        layout = consumer.wino_input_layout(feat_left.shape[0])
        if layout == _lib.LAYOUT_WINO_H:
            return self.forward_wino(feat_left, feat_right), layout
        if layout == _lib.LAYOUT_WINO_DH:
            return self.forward_wino2(feat_left, feat_right), layout
        return self.forward_padded(feat_left, feat_right), _lib.LAYOUT_PLAIN

    @torch.no_grad()
    def forward(self, feat_left: torch.Tensor, feat_right: torch.Tensor) -> torch.Tensor:
        if self.precision == "bf16":      # logical (B,2C,D,H,W) view of the channels-last volume
            return _to_logical(self._bf16(feat_left, feat_right, 0, resident=False))
        if feat_left.shape != feat_right.shape or feat_left.dim() != 4:
            raise RuntimeError(f"feature maps must both be (B,C,H,W), got {tuple(feat_left.shape)} and "
                               f"{tuple(feat_right.shape)}")
        fl = _check_input(feat_left, "feat_left", feat_left.shape[1:])
        fr = _check_input(feat_right, "feat_right", feat_left.shape[1:])
        B, Cc, H, W = fl.shape
        vol = torch.empty((B, 2 * Cc, self.max_disp, H, W), dtype=torch.float32, device=fl.device)
        if B == 0:
            return vol
        _lib.check(_lib.load().s3r_cost_volume_forward(fl.data_ptr(), fr.data_ptr(), vol.data_ptr(), B, Cc,
                                                       self.max_disp, H, W, 0, _stream_ptr(fl.device)), "cost_volume")
        return vol


def _check_wino_planes(v: torch.Tensor, layout: int = _lib.LAYOUT_WINO_H) -> torch.Tensor:
    if layout == _lib.LAYOUT_WINO_DH:
        ncls, want = 36, (2 * spec.FEAT_C, spec.MAX_DISP // 4, spec.FEAT_HW // 4, spec.FEAT_HW + 2)
    else:
        ncls, want = 6, (2 * spec.FEAT_C, spec.MAX_DISP + 2, spec.FEAT_HW // 4, spec.FEAT_HW + 2)
    if not isinstance(v, torch.Tensor) or not v.is_cuda or v.dtype != torch.float32 or v.dim() != 6 or v.shape[0] != ncls or \
            tuple(v.shape[2:]) != want or not v.is_contiguous():
        raise RuntimeError(f"transformed volume must be a contiguous float32 HIP tensor ({ncls}, B, {', '.join(map(str, want))})")
    return v


class Decoder(_HipChain):
    """3D conv hourglass: cost volume (B,64,28,28,28) -> occupancy probabilities (B,32,32,32)."""
    _entry = "s3r_decoder_forward"

    def __init__(self, precision: str = "fp32", winograd=None):
        super().__init__(spec.DECODER, spec.MAX_DISP, tag_base=200, precision=precision, winograd=winograd)

    def forward(self, volume: torch.Tensor, upto: Optional[str] = None) -> torch.Tensor:
        tail = (2 * spec.FEAT_C, spec.MAX_DISP, spec.FEAT_HW, spec.FEAT_HW)
        if self.precision == "bf16":
            x = _check_input_cl(volume, "volume", tail)
        else:
            x = _check_input(volume, "volume", tail)
        y = self._run(x, upto)
        return y.squeeze(1) if upto is None or upto == self.names[-1] else y

    def forward_padded(self, volume_padded: torch.Tensor, halo: int = 1, in_layout: int = 0) -> torch.Tensor:
        """Decoder on the halo-padded volume CostVolume.forward_padded produced (no pad copy), or (in_layout =
        LAYOUT_WINO_H) on the transformed planes CostVolume.forward_wino produced."""
        if in_layout in (_lib.LAYOUT_WINO_H, _lib.LAYOUT_WINO_DH):
            return self._run(_check_wino_planes(volume_padded, in_layout), None, in_halo=1, in_layout=in_layout).squeeze(1)
        n = (spec.MAX_DISP + 2 * halo, spec.FEAT_HW + 2 * halo, spec.FEAT_HW + 2 * halo)
        if self.precision == "bf16":
            x = _check_input(volume_padded, "volume_padded", n + (2 * spec.FEAT_C,), torch.bfloat16)
        else:
            x = _check_input(volume_padded, "volume_padded", (2 * spec.FEAT_C,) + n)
        return self._run(x, None, in_halo=halo).squeeze(1)


class VolumeEncoder(_HipChain):
    """The down half of the hourglass alone (Stereo2Point): cost volume -> (B,512,4,4,4) latent."""
    _entry = "s3r_decoder_forward"

    def __init__(self, precision: str = "fp32", winograd=None):
        super().__init__(spec.DECODER_DOWN, spec.MAX_DISP, tag_base=200, precision=precision, winograd=winograd)

    def forward(self, volume: torch.Tensor) -> torch.Tensor:
        tail = (2 * spec.FEAT_C, spec.MAX_DISP, spec.FEAT_HW, spec.FEAT_HW)
        if self.precision == "bf16":
            return self._run(_check_input_cl(volume, "volume", tail))
        return self._run(_check_input(volume, "volume", tail))

    def forward_padded(self, volume_padded: torch.Tensor, halo: int = 1, in_layout: int = 0) -> torch.Tensor:
        if in_layout in (_lib.LAYOUT_WINO_H, _lib.LAYOUT_WINO_DH):
            return self._run(_check_wino_planes(volume_padded, in_layout), None, in_halo=1, in_layout=in_layout)
        n = (spec.MAX_DISP + 2 * halo, spec.FEAT_HW + 2 * halo, spec.FEAT_HW + 2 * halo)
        if self.precision == "bf16":            # physical (B,D+2h,H+2h,W+2h,2C) bf16 -> logical (B,512,4,4,4) bf16
            x = _check_input(volume_padded, "volume_padded", n + (2 * spec.FEAT_C,), torch.bfloat16)
        else:
            x = _check_input(volume_padded, "volume_padded", (2 * spec.FEAT_C,) + n)
        return self._run(x, None, in_halo=halo)


class PointHead(_HipChain):
    """MLP on the flattened latent: (B,512,4,4,4) -> (B,2048,3)."""

    def __init__(self):
        super().__init__(spec.POINT_HEAD, 1, tag_base=300)

    def forward(self, latent: torch.Tensor) -> torch.Tensor:
        x = _check_input(latent, "latent", (spec.LATENT_C, 4, 4, 4))
        return self._run(x.view(x.shape[0], spec.LATENT_C * 64)).view(x.shape[0], spec.N_POINTS, 3)   # (B may be 0)


class _DisparityMixin:
    """`disparity()` for the networks that hold an `encoder` and a `cost_volume`."""

    @torch.no_grad()
    def disparity(self, left: torch.Tensor, right: torch.Tensor, in_pixels: bool = True):
        """Predicted (left, right) disparity maps, (B,28,28) each: the winner-take-all read-out of the cost volume's
        shift-and-diff costs on this model's encoder features (`disparity_wta`).  in_pixels scales feature-resolution
        disparities to 224x224 render pixels (x8), the unit of the dataset's EXR ground truth."""
        left = _check_render(left, "left")
        right = _check_render(right, "right")
        if left.shape[0] != right.shape[0]:
            raise RuntimeError("left and right batch sizes differ")
        dls, drs = [], []
        for s in range(0, max(left.shape[0], 1), MAX_CHUNK):
            l, r = left[s:s + MAX_CHUNK], right[s:s + MAX_CHUNK]
            b = l.shape[0]
            feats = self.encoder.forward_pair(l, r)
            if feats.dtype != torch.float32:                     # bf16 path: channels-last bf16 -> fp32 NCHW (HIP kernel)
                feats = channels_last_to_f32(feats)
            dl, dr = disparity_wta(feats[:b], feats[b:], self.cost_volume.max_disp)
            dls.append(dl), drs.append(dr)
        dl = dls[0] if len(dls) == 1 else torch.cat(dls, 0)
        dr = drs[0] if len(drs) == 1 else torch.cat(drs, 0)
        scale = float(spec.IMG_HW // spec.FEAT_HW) if in_pixels else 1.0
        return dl * scale, dr * scale


class Stereo2Voxel(_DisparityMixin, nn.Module):
    """left,right (B,3,224,224) -> (B,32,32,32) occupancy.  state_dict keys: encoder.*, decoder.*

    precision="fp32": exact-fp32 MFMA path, NCHW (BASELINE configs[1]).  precision="bf16": bf16 MFMA path,
    channels-last bf16 activations with fp32 accumulation (BASELINE configs[2]); parameters stay fp32
    nn.Parameters either way (same state_dict), the bf16 images of the weights are made at pack time."""

    def __init__(self, precision: str = "fp32", winograd=None):
        """winograd: None = the library's policy (S3R_ALGO_AUTO: every layer that has a Winograd form runs it — one-axis F(4,3) on
        e2 / e4, two-axis F(4,3)^2 / F(2,4)^2 on the stride-1 layers with an edge <= 28, F(2,2)^2 inside the parity classes of the
        transposed layers; each measured faster than the direct kernel at B = 1, 4, 8 and 32); False = the direct
        kernels only; True = ask for the Winograd kernel on every layer the library has one for (direct elsewhere).
        Results differ between the settings at fp32 rounding level (never with the batch)."""
        super().__init__()
        self.precision = precision
        self.winograd = winograd
        self.encoder = Encoder(precision, winograd)
        self.cost_volume = CostVolume(precision=precision)
        self.decoder = Decoder(precision, winograd)
        self.eval()

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("forward/inference path only")
        return super().train(False)

    @torch.no_grad()
    def forward(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        """left, right: (B,3,224,224) float32 in [0,1] or uint8 (8-bit renders: scaled by 1/255 inside the first kernel)."""
        left = _check_render(left, "left")
        right = _check_render(right, "right")
        if left.shape[0] != right.shape[0]:
            raise RuntimeError("left and right batch sizes differ")
        outs = []
        for s in range(0, max(left.shape[0], 1), MAX_CHUNK):
            l, r = left[s:s + MAX_CHUNK], right[s:s + MAX_CHUNK]
            b = l.shape[0]
            feats = self.encoder.forward_pair(l, r)          # (2b,32,28,28): left batch, then right batch
            # (v1 on a Winograd kernel: the volume goes over in the transformed layout that kernel reads)
            vol, layout = self.cost_volume.forward_for(self.decoder, feats[:b], feats[b:])
            outs.append(self.decoder.forward_padded(vol, in_layout=layout))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    @torch.no_grad()
    def autotune(self, left: torch.Tensor, right: torch.Tensor, rounds: int = 3, log=None):
        """Measure-and-pick the per-layer kernel configuration on a representative batch (see
        _HipChain.autotune).  Returns {layer: {tile, ksplit, ms}}."""
        b = left.shape[0]
        images = torch.cat([left, right], 0)                 # (tuning only: one tensor for _HipChain.autotune)
        chosen = dict(self.encoder.autotune(images, rounds=rounds, log=log))
        feats = self.encoder(images)
        vol = self.cost_volume.forward_padded(feats[:b], feats[b:])
        chosen.update(self.decoder.autotune(vol, in_halo=1, rounds=rounds, log=log))
        return chosen


class Stereo2Point(_DisparityMixin, nn.Module):
    """left,right (B,3,224,224) -> (B,2048,3) point cloud.  Keys: encoder.*, decoder.*, point_head.*

    precision="bf16": the convolutional part (encoder, cost volume, v1-v6) on the bf16 MFMA path; the latent is handed
    to the point head as fp32 and the three linear layers stay fp32 (they stream 168 MB of weights: HBM-bound)."""

    def __init__(self, precision: str = "fp32", winograd=None):
        super().__init__()
        self.precision = precision
        self.winograd = winograd
        self.encoder = Encoder(precision, winograd)
        self.cost_volume = CostVolume(precision=precision)
        self.decoder = VolumeEncoder(precision, winograd)
        self.point_head = PointHead()
        self.eval()

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("forward/inference path only")
        return super().train(False)

    @torch.no_grad()
    def forward(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        left = _check_render(left, "left")
        right = _check_render(right, "right")
        if left.shape[0] != right.shape[0]:
            raise RuntimeError("left and right batch sizes differ")
        outs = []
        for s in range(0, max(left.shape[0], 1), MAX_CHUNK):
            l, r = left[s:s + MAX_CHUNK], right[s:s + MAX_CHUNK]
            b = l.shape[0]
            feats = self.encoder.forward_pair(l, r)
            vol, layout = self.cost_volume.forward_for(self.decoder, feats[:b], feats[b:])
            latent = self.decoder.forward_padded(vol, in_layout=layout)
            if latent.dtype != torch.float32:                      # bf16 channels-last view -> fp32 (B,512,4,4,4): HIP kernel
                latent = channels_last_to_f32(latent)
            outs.append(self.point_head(latent))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


@torch.no_grad()
def chamfer_distance(p: torch.Tensor, q: torch.Tensor):
    """(dist1 (B,N), dist2 (B,M), idx1 (B,N) int32, idx2 (B,M) int32): squared-L2 nearest neighbours."""
    if p.dim() != 3 or q.dim() != 3 or p.shape[-1] != 3 or q.shape[-1] != 3 or p.shape[0] != q.shape[0]:
        raise RuntimeError(f"chamfer_distance expects (B,N,3) and (B,M,3), got {tuple(p.shape)} and {tuple(q.shape)}")
    p = _check_input(p, "p", p.shape[1:])
    q = _check_input(q, "q", q.shape[1:])
    B, N, M = p.shape[0], p.shape[1], q.shape[1]
    if N == 0 or M == 0:
        raise RuntimeError("chamfer_distance needs non-empty point clouds")
    d1 = torch.empty((B, N), dtype=torch.float32, device=p.device)
    d2 = torch.empty((B, M), dtype=torch.float32, device=p.device)
    i1 = torch.empty((B, N), dtype=torch.int32, device=p.device)
    i2 = torch.empty((B, M), dtype=torch.int32, device=p.device)
    if B:
        _lib.check(_lib.load().s3r_chamfer_forward(p.data_ptr(), q.data_ptr(), d1.data_ptr(), d2.data_ptr(),
                                                   i1.data_ptr(), i2.data_ptr(), B, N, M, _stream_ptr(p.device)),
                   "chamfer")
    return d1, d2, i1, i2


class ChamferDistance(nn.Module):
    """Drop-in for the reference's extensions/chamfer_dist module (README.md:64-65): forward(p, q)
    returns mean(dist1) + mean(dist2).  (Whether the reference reduces with mean or sum, squared or
    not, is unknown — SURVEY.md §8a row 5; `chamfer_distance` exposes the unreduced tensors.)"""

    def forward(self, p, q):
        d1, d2, _, _ = chamfer_distance(p, q)
        return d1.mean() + d2.mean()


@torch.no_grad()
def voxel_iou(pred: torch.Tensor, gt: torch.Tensor, threshold: float = 0.5) -> torch.Tensor:
    """Per-sample IoU of (pred > th) vs (gt > th), computed on the device: (B,...) -> (B,)."""
    if pred.shape != gt.shape:
        raise RuntimeError("pred and gt shapes differ")
    pred = _check_input(pred, "pred", pred.shape[1:])
    gt = _check_input(gt, "gt", gt.shape[1:])
    B = pred.shape[0]
    out = torch.empty((B,), dtype=torch.float32, device=pred.device)
    if B:
        _lib.check(_lib.load().s3r_voxel_iou(pred.data_ptr(), gt.data_ptr(), float(threshold), out.data_ptr(), B,
                                             pred[0].numel(), _stream_ptr(pred.device)), "voxel_iou")
    return out


@torch.no_grad()
def disparity_wta(feat_l: torch.Tensor, feat_r: torch.Tensor, max_disp: int = spec.MAX_DISP):
    """Predicted left / right disparity maps (ground truth: disp_%02d_{l,r}.exr,
    /root/reference/README.md:75-76): winner-take-all over the shift-and-diff costs the cost volume holds, read
    straight from the two feature maps.  (B,C,H,W) x2 fp32 -> two (B,H,W) fp32 maps, integer-valued, in
    feature-resolution pixels; disp_l[b,h,w] = first argmin_d sum_c |L[b,c,h,w] - R[b,c,h,w-d]|, disp_r mirrored."""
    if feat_l.shape != feat_r.shape or feat_l.dim() != 4:
        raise RuntimeError("feat_l / feat_r must be two (B,C,H,W) tensors of one shape")
    feat_l = _check_input(feat_l.float(), "feat_l", feat_l.shape[1:])
    feat_r = _check_input(feat_r.float(), "feat_r", feat_r.shape[1:])
    B, Cc, H, W = feat_l.shape
    dl = torch.empty((B, H, W), dtype=torch.float32, device=feat_l.device)
    dr = torch.empty_like(dl)
    if B:
        _lib.check(_lib.load().s3r_disparity_wta(feat_l.data_ptr(), feat_r.data_ptr(), dl.data_ptr(), dr.data_ptr(), B,
                                                 Cc, H, W, int(max_disp), _stream_ptr(feat_l.device)), "disparity_wta")
    return dl, dr


@torch.no_grad()
def disparity_epe(pred: torch.Tensor, gt: torch.Tensor):
    """Per-sample end-point error mean|pred - gt| over the pixels whose ground truth is valid (finite and >= 0 — EXR
    disparity maps mark background with inf / negative values), reduced on the device: (B,...) x2 ->
    (epe (B,) fp32, valid pixel count (B,) int32)."""
    if pred.shape != gt.shape:
        raise RuntimeError("pred and gt shapes differ")
    pred = _check_input(pred, "pred", pred.shape[1:])
    gt = _check_input(gt, "gt", gt.shape[1:])
    B = pred.shape[0]
    epe = torch.empty((B,), dtype=torch.float32, device=pred.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=pred.device)
    if B:
        _lib.check(_lib.load().s3r_disparity_epe(pred.data_ptr(), gt.data_ptr(), epe.data_ptr(), cnt.data_ptr(), B,
                                                 pred[0].numel(), _stream_ptr(pred.device)), "disparity_epe")
    return epe, cnt


# the names BASELINE.json's north_star uses for the module API
encoder = Encoder
cost_volume = CostVolume
decoder = Decoder
