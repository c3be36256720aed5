"""ctypes binding of libs3r_hip.so (the C-ABI declared in include/s3r.h).

There is NO fallback: if the library is missing or a call fails, the product path raises.  The
oracle under oracle/ is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# S3R_LIB: another build of the SAME library (tools/ab_bench.sh times two builds on one device); never a fallback
LIB_PATH = os.environ.get("S3R_LIB") or os.path.join(_HERE, "csrc", "libs3r_hip.so")

OP_CONV, OP_DECONV, OP_LINEAR = 0, 1, 2
DTYPE = {"fp32": 0, "bf16": 1}
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "leaky_relu": 3, "elu": 4, "tanh": 5}
FAMILY = {0: "conv_mfma", 1: "stem", 2: "head", 3: "cost_volume", 4: "linear", 5: "chamfer", 6: "iou", 7: "pack",
          8: "pad_copy", 9: "disparity", 10: "aux"}      # aux: transform / finish passes, nested inside their layer's conv_mfma record
ABI_VERSION = 8
ALGO_AUTO, ALGO_DIRECT, ALGO_WINOGRAD = 0, 1, 2
ALGO = {None: 0, "auto": 0, "direct": 1, "winograd": 2, False: 1, True: 2}
RAN = {0: "direct", 1: "winograd-serial", 2: "winograd-class-parallel", 3: "winograd-dual", 4: "winograd-2axis", 5: "winograd-3axis", 6: "winograd-3axis-class-parallel"}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("op", "ndim", "batch", "cin", "cout", "in_size", "k", "stride", "pad", "act", "tag", "tile",
                 "in_halo", "out_halo", "ksplit", "dtype", "in_layout", "out_layout", "algo", "dilation", "out_pad")] + \
               [("act_param", C.c_float)]


class Layer(C.Structure):
    _fields_ = [("desc", ConvDesc), ("packed_w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


class ProfRecord(C.Structure):
    _fields_ = [("family", C.c_int32), ("tag", C.c_int32), ("ms", C.c_float), ("launches", C.c_int32),
                ("flops", C.c_double), ("bytes", C.c_double), ("exec_flops", C.c_double), ("algo", C.c_int32),
                ("reserved", C.c_int32)]


class S3RError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

# name -> (restype, argtypes); must list every symbol include/s3r.h declares (tests check this)
SIGNATURES = {
    "s3r_abi_version": (C.c_int, []),
    "s3r_last_error": (C.c_char_p, []),
    "s3r_conv_out_size": (C.c_int, [C.POINTER(ConvDesc)]),
    "s3r_conv_packed_elems": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int64)]),
    "s3r_conv_pack_weights": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3r_conv_scratch_elems": (C.c_int64, [C.POINTER(ConvDesc)]),
    "s3r_conv_forward": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p]),
    "s3r_chain_workspace_elems": (C.c_int64, [C.POINTER(Layer), C.c_int]),
    "s3r_chain_forward": (C.c_int, [C.POINTER(Layer), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_int, C.c_void_p]),
    "s3r_encoder_forward": (C.c_int, [C.POINTER(Layer), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_int, C.c_void_p]),
    "s3r_encoder_forward_u8": (C.c_int, [C.POINTER(Layer), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int, C.c_void_p]),
    "s3r_channels_last_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "s3r_decoder_forward": (C.c_int, [C.POINTER(Layer), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int, C.c_void_p]),
    "s3r_cost_volume_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p]),
    "s3r_cost_volume_forward_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p]),
    "s3r_cost_volume_forward_wino": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p]),
    "s3r_conv_wino_input_elems": (C.c_int64, [C.POINTER(ConvDesc)]),
    "s3r_conv_wino_input_layout": (C.c_int, [C.POINTER(ConvDesc)]),
    "s3r_cost_volume_forward_wino2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p]),
    "s3r_linear_scratch_elems": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "s3r_linear_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "s3r_chamfer_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "s3r_voxel_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "s3r_disparity_wta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "s3r_disparity_epe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "s3r_profile_enable": (C.c_int, [C.c_int]),
    "s3r_profile_reset": (C.c_int, []),
    "s3r_profile_detail": (C.c_int, [C.c_int]),
    "s3r_profile_read": (C.c_int, [C.POINTER(ProfRecord), C.c_int]),
}


def load():
    """Load (once) and return the ctypes handle; raises S3RError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise S3RError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C stereo-3d-reconstruction_amd/csrc`). There is no CPU fallback.")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise S3RError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.s3r_abi_version() != ABI_VERSION:
            raise S3RError("libs3r_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc < 0:
        msg = load().s3r_last_error().decode("utf-8", "replace")
        raise S3RError(f"{what}: {msg} (code {rc})")
    return rc


LAYOUT_PLAIN, LAYOUT_WINO_H, LAYOUT_WINO_DH, LAYOUT_WINO_HW = 0, 2, 3, 4


def make_desc(layer, batch, in_size, tag=0, tile=-1, in_halo=0, out_halo=0, ksplit=0, dtype=0, in_layout=0, out_layout=0,
              algo=0):
    """arch_spec.Layer -> ConvDesc.  algo: ALGO_AUTO (the library's geometry-only policy), ALGO_DIRECT, ALGO_WINOGRAD."""
    from . import arch_spec as _spec
    op = {"conv2d": OP_CONV, "conv3d": OP_CONV, "deconv3d": OP_DECONV, "deconv2d": OP_DECONV, "linear": OP_LINEAR}[layer.op]
    nd = _spec.ndim(layer)
    return ConvDesc(op, nd, batch, layer.cin, layer.cout, in_size, layer.k, layer.s, layer.p, ACT[layer.act], tag,
                    tile, in_halo, out_halo, ksplit, dtype, in_layout, out_layout, algo, getattr(layer, "dil", 1),
                    getattr(layer, "opad", 0), _spec.act_param(layer))


def profile_enable(max_records):
    check(load().s3r_profile_enable(int(max_records)), "profile_enable")


def profile_detail(level):
    """1: the transform / finish passes of the convolution layers get `aux` records of their own (nested in their layer's)."""
    check(load().s3r_profile_detail(int(level)), "profile_detail")


def profile_reset():
    check(load().s3r_profile_reset(), "profile_reset")


def profile_read(max_records=4096):
    buf = (ProfRecord * max_records)()
    n = check(load().s3r_profile_read(buf, max_records), "profile_read")
    return [dict(family=FAMILY.get(r.family, str(r.family)), tag=r.tag, ms=r.ms, launches=r.launches, flops=r.flops,
                 bytes=r.bytes, exec_flops=r.exec_flops, ran=RAN.get(r.algo, str(r.algo))) for r in buf[:n]]
