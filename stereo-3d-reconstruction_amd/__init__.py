"""stereo-3d-reconstruction_amd — MI355X (gfx950) forward path of a Stereo2Voxel / Stereo2Point network.

Scope: the data-parallel hot path only: stereo feature encoder, disparity cost
volume, 3D-conv voxel decoder, point head + Chamfer distance, behind the `encoder` / `cost_volume`
/ `decoder` nn.Module API.  The network architecture is BUILD-SPECIFIED (arch_spec.py) because the
reference's model code is not in the mount (/root/reference/README.md:5).

Importing this package never touches the GPU and never loads the oracle; the HIP library
(csrc/libs3r_hip.so) is loaded on first use and its absence is a hard error.
"""
from . import arch_spec, checkpoint, collate, data, evaluate, exr
from .graph import GraphedForward, PrefetchingLoader
from ._lib import ALGO_AUTO, ALGO_DIRECT, ALGO_WINOGRAD, S3RError, LIB_PATH, load as load_library, profile_enable, profile_read, profile_reset, profile_detail
from .init import seed_module, seeded_state_dict, synthetic_pairs
from .modules import (ChamferDistance, CostVolume, Decoder, Encoder, PointHead, Stereo2Point, Stereo2Voxel,
                      VolumeEncoder, chamfer_distance, cost_volume, debug_overrides, decoder, disparity_epe, disparity_wta,
                      encoder, voxel_iou)

__all__ = [
    "GraphedForward", "PrefetchingLoader", "arch_spec", "checkpoint", "collate", "data", "evaluate", "S3RError", "LIB_PATH", "load_library", "profile_enable", "profile_read", "profile_reset", "profile_detail",
    "seed_module", "seeded_state_dict", "synthetic_pairs",
    "Encoder", "CostVolume", "Decoder", "VolumeEncoder", "PointHead", "Stereo2Voxel", "Stereo2Point",
    "ChamferDistance", "chamfer_distance", "debug_overrides", "ALGO_AUTO", "ALGO_DIRECT", "ALGO_WINOGRAD", "voxel_iou", "disparity_wta", "disparity_epe", "encoder", "cost_volume", "decoder",
]
