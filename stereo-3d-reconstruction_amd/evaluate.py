"""Eval driver of the path: what `runner.py --test` does after building the model
(/root/reference/README.md:88-92) — forward over an eval list, thresholded IoU per sample, mean per
threshold — with the forward on the HIP modules and the IoU reduced on the device, so that multi-GPU
collation gathers a few scalars per sample instead of 32^3 grids.

The reference's threshold list and IoU formula are not in the mount; this build evaluates
IoU = |pred>t & gt| / |pred>t | gt| (binary ground truth) at t in THRESHOLDS (build-specified) and labels it so.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch

from . import collate
from .modules import chamfer_distance, disparity_epe, voxel_iou

THRESHOLDS = (0.2, 0.3, 0.4, 0.5)


def _device_batches(tensors: Sequence[torch.Tensor], b0: int, e0: int, batch: int, device):
    """[b0, e0) of a host-resident eval list as device batches.  Each batch's slices are page-locked IN PLACE
    (hipHostRegister, no copy), so they are DMA-able at PCIe rate (~55 GB/s here against 4.6 GB/s from pageable memory)
    and the next batch crosses on a side stream while the current one runs (graph.PrefetchingLoader).  A plain
    `.to(device)` per batch made this loop 10x slower than the forward.  Page-locking costs ~0.04 ms per MB — 0.12 s for a
    4 GB list of fp32 renders when done for the whole list before the first kernel starts (r02) — so it is done PER BATCH,
    one registration per slice (a copy must lie inside ONE registration: tools/reg_debug.py), by the loader's look-ahead:
    the pages of batch i+1 are locked while batch i-1 runs on the GPU.  (Measured, bf16 path, 3072 pairs at batch 256:
    31.9 k pairs/s with 8-bit renders, 14.1 k with fp32 renders — 13 ms of page-locking per batch against a 6.5 ms
    forward; a helper thread locking ahead of the loop made both worse, 21.0 k / 9.9 k.)"""
    from .graph import PrefetchingLoader
    if torch.device(device).type != "cuda" or any(t.is_cuda for t in tensors):
        for s in range(b0, e0, batch):
            yield tuple(t[s:min(e0, s + batch)].to(device) for t in tensors)
        return
    rt = torch.cuda.cudart()
    locked = []                                               # addresses registered so far
    lockable = [bool(t.numel()) and t.is_contiguous() and not t.is_pinned() for t in tensors]

    def host():
        for s in range(b0, e0, batch):
            e = min(e0, s + batch)
            out = []
            for i, t in enumerate(tensors):
                sl = t[s:e]
                if lockable[i] and sl.numel():
                    if int(rt.cudaHostRegister(sl.data_ptr(), sl.numel() * sl.element_size(), 0)) == 0:
                        locked.append(sl.data_ptr())
                    else:                                     # not lockable: the loader stages this tensor through its ring
                        lockable[i] = False
                        _clear_hip_error()
                out.append(sl)
            yield tuple(out)

    try:
        yield from PrefetchingLoader(host(), device)
    finally:
        torch.cuda.synchronize(device)
        for ptr in locked:
            rt.cudaHostUnregister(ptr)


def _clear_hip_error():
    """A failed hipHostRegister leaves its code in the runtime's sticky last-error slot, where the next unrelated torch
    call would find (and raise) it: read it out."""
    try:
        import ctypes
        ctypes.CDLL("libamdhip64.so").hipGetLastError()
    except OSError:                                           # pragma: no cover
        pass


def synthetic_eval_set(n: int, seed: int = 0, render_dtype: str = "float32") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """n synthetic (left, right, ground-truth 32^3 occupancy) triples: there is no dataset in this
    container (README.md:29 is a download link).  GT is a random axis-aligned box per sample.
    render_dtype="uint8": 8-bit renders (uniform 0..255), what a decoded PNG list looks like."""
    g = torch.Generator().manual_seed(seed)
    if render_dtype == "uint8":
        left = torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8)
        right = torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8)
    else:
        left, right = torch.rand(n, 3, 224, 224, generator=g), torch.rand(n, 3, 224, 224, generator=g)
    gt = torch.zeros(n, 32, 32, 32)
    lo = torch.randint(0, 12, (n, 3), generator=g)
    hi = lo + torch.randint(8, 20, (n, 3), generator=g)
    for i in range(n):
        gt[i, lo[i, 0]:hi[i, 0], lo[i, 1]:hi[i, 1], lo[i, 2]:hi[i, 2]] = 1.0
    return left, right, gt


@torch.no_grad()
def test_dataset(model, ds, batch: int = 32, thresholds: Sequence[float] = THRESHOLDS, device="cuda", group=None,
                 workers: int = 0) -> Dict[str, object]:
    """test_net over a data.StereoShapeNet: each rank decodes and evaluates only its shard_bounds slice of the item
    list, host batches cross PCIe on a copy stream while the previous batch runs (graph.PrefetchingLoader), and the
    per-sample IoUs are all-gathered in list order.  A dataset built with_disparity=True also yields the end-point error
    of the model's disparity read-out against the EXR ground truth, pooled over valid pixels (see test_disparity)."""
    import torch.distributed as dist
    from . import data as _data
    from .graph import PrefetchingLoader
    dist_on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if dist_on else 1
    rank = dist.get_rank(group) if dist_on else 0
    total = len(ds)
    b0, e0 = collate.shard_bounds(total, world, rank)
    ious = torch.empty((e0 - b0, len(thresholds)), dtype=torch.float32, device=device)
    with_disp = bool(getattr(ds, "with_disparity", False))
    rows = torch.zeros((e0 - b0, 4), dtype=torch.float64, device=device)          # epe_l, n_l, epe_r, n_r
    host = _data.batches(ds, batch, range(b0, e0), workers)
    done = 0
    for item in PrefetchingLoader(host, device):
        l, r, g = item[:3]
        pred = model(l, r)
        for j, t in enumerate(thresholds):
            ious[done:done + l.shape[0], j] = voxel_iou(pred, g, t)
        if with_disp:
            dl, dr = model.disparity(l, r)
            el, nl = disparity_epe(dl, _data.downsample_disparity(item[3], dl.shape[-1]))
            er, nr = disparity_epe(dr, _data.downsample_disparity(item[4], dr.shape[-1]))
            rows[done:done + l.shape[0]] = torch.stack([el.double(), nl.double(), er.double(), nr.double()], 1)
        done += l.shape[0]
    if dist_on:
        ious = collate.all_gather_ragged(ious, total, group)
        if with_disp:
            rows = collate.all_gather_ragged(rows, total, group)
    mean = ious.mean(0).cpu().tolist() if total else [float("nan")] * len(thresholds)
    out = {"samples": total, "thresholds": list(thresholds), "mean_iou": mean, "per_sample": ious.cpu()}
    # per-taxonomy breakdown (what a ShapeNet test log lists): mean IoU per threshold and sample count per category
    per_tax: Dict[str, Dict[str, object]] = {}
    cpu = out["per_sample"]
    for tax in sorted({it[0] for it in getattr(ds, "items", [])}):
        idx = [i for i, it in enumerate(ds.items) if it[0] == tax]
        per_tax[tax] = {"samples": len(idx), "mean_iou": cpu[idx].mean(0).tolist()}
    out["per_taxonomy"] = per_tax
    if with_disp:
        rows = rows.cpu()
        nl, nr = rows[:, 1].sum().item(), rows[:, 3].sum().item()
        out["epe_left"] = (rows[:, 0] * rows[:, 1]).sum().item() / nl if nl else float("nan")
        out["epe_right"] = (rows[:, 2] * rows[:, 3]).sum().item() / nr if nr else float("nan")
    return out


@torch.no_grad()
def test_point_net(model, left: torch.Tensor, right: torch.Tensor, gt_clouds: torch.Tensor, batch: int = 32, device="cuda",
                   group=None) -> Dict[str, object]:
    """Stereo2Point's metric (the reference's one native op, extensions/chamfer_dist, README.md:59-66): per-sample
    Chamfer distance mean_i min_j |p_i - q_j|^2 + mean_j min_i |p_i - q_j|^2 between the predicted (N,2048,3) cloud and
    the ground-truth (N,M,3) cloud, on the device; with torch.distributed initialised the per-sample scalars of every
    rank's shard are all-gathered in list order.  (Squared / mean is this build's stated choice — SURVEY.md §8a row 5.)"""
    import torch.distributed as dist
    dist_on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if dist_on else 1
    rank = dist.get_rank(group) if dist_on else 0
    total = left.shape[0]
    b0, e0 = collate.shard_bounds(total, world, rank)
    cd = torch.empty((e0 - b0,), dtype=torch.float32, device=device)
    done = 0
    for l, r, g in _device_batches((left, right, gt_clouds), b0, e0, batch, device):
        d1, d2, _, _ = chamfer_distance(model(l, r), g)
        cd[done:done + l.shape[0]] = d1.mean(1) + d2.mean(1)
        done += l.shape[0]
    if dist_on:
        cd = collate.all_gather_ragged(cd, total, group)
    return {"samples": total, "mean_chamfer": cd.mean().item() if total else float("nan"), "per_sample": cd.cpu()}


@torch.no_grad()
def test_disparity(model, left: torch.Tensor, right: torch.Tensor, disp_l_gt: torch.Tensor, disp_r_gt: torch.Tensor,
                   batch: int = 32, device="cuda", group=None) -> Dict[str, object]:
    """End-point error of the model's predicted left / right disparity (Stereo2Voxel.disparity, render pixels) against
    (N,28,28) ground-truth maps at feature resolution in render pixels (the dataset's
    disp_%02d_{l,r}.exr, /root/reference/README.md:75-76, downsampled by the caller; invalid pixels: inf / negative).
    EPE is pooled over all valid pixels of the list: per-sample (sum, count) pairs are reduced on the device and, with
    torch.distributed initialised, all-gathered (rank order = list order)."""
    import torch.distributed as dist
    dist_on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if dist_on else 1
    rank = dist.get_rank(group) if dist_on else 0
    total = left.shape[0]
    b0, e0 = collate.shard_bounds(total, world, rank)
    rows = torch.zeros((e0 - b0, 4), dtype=torch.float64, device=device)        # epe_l, n_l, epe_r, n_r
    done = 0
    for l, r, gl, gr in _device_batches((left, right, disp_l_gt, disp_r_gt), b0, e0, batch, device):
        dl, dr = model.disparity(l, r)
        el, nl = disparity_epe(dl, gl)
        er, nr = disparity_epe(dr, gr)
        rows[done:done + l.shape[0]] = torch.stack([el.double(), nl.double(), er.double(), nr.double()], 1)
        done += l.shape[0]
    if dist_on:
        rows = collate.all_gather_ragged(rows, total, group)
    rows = rows.cpu()
    nl, nr = rows[:, 1].sum().item(), rows[:, 3].sum().item()
    epe_l = (rows[:, 0] * rows[:, 1]).sum().item() / nl if nl else float("nan")
    epe_r = (rows[:, 2] * rows[:, 3]).sum().item() / nr if nr else float("nan")
    return {"samples": total, "epe_left": epe_l, "epe_right": epe_r, "valid_left": int(nl), "valid_right": int(nr),
            "per_sample": rows}


@torch.no_grad()
def test_net(model, left: torch.Tensor, right: torch.Tensor, gt: torch.Tensor, batch: int = 32,
             thresholds: Sequence[float] = THRESHOLDS, device="cuda", group=None) -> Dict[str, object]:
    """Mean IoU per threshold over the eval list.  With torch.distributed initialised every rank
    evaluates its shard_bounds slice and the per-sample IoUs are all-gathered (rank order = list order)."""
    import torch.distributed as dist
    dist_on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if dist_on else 1
    rank = dist.get_rank(group) if dist_on else 0
    total = left.shape[0]
    b0, e0 = collate.shard_bounds(total, world, rank)
    ious = torch.empty((e0 - b0, len(thresholds)), dtype=torch.float32, device=device)
    done = 0
    for l, r, g in _device_batches((left, right, gt), b0, e0, batch, device):
        pred = model(l, r)
        for j, t in enumerate(thresholds):
            # the device kernel thresholds both operands at t; GT is binary {0,1}, so gt > t == (gt == 1)
            ious[done:done + l.shape[0], j] = voxel_iou(pred, g, t)
        done += l.shape[0]
    if dist_on:
        ious = collate.all_gather_ragged(ious, total, group)
    mean = ious.mean(0).cpu().tolist() if total else [float("nan")] * len(thresholds)
    return {"samples": total, "thresholds": list(thresholds), "mean_iou": mean, "per_sample": ious.cpu()}
