#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: Stereo2Voxel forward throughput (stereo pairs/s), batch 32 per GPU,
224x224 stereo pair -> 32^3 voxels, fp32, synthetic inputs, random-init weights (BUILD-SPECIFIED
architecture, arch_spec.py — the reference's model code is not in the mount).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--variant voxel|point]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one forward of the hot path over one resident batch of B pairs per GPU (inputs already in
HBM).  N>1: one process per GPU, the batch is sharded (weak scaling: B pairs per rank), and each step
ends with the one exchange the path has — an RCCL all-gather of the (B,32,32,32) predictions for eval
collation.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer exchange between the per-GPU
# processes fails with `hipIpcGetMemHandle: invalid argument` (must be set before the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cpus():
    """CPUs this process may actually run on: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(budget_s=12.0):
    """The oracle (this build's PyTorch-CPU restatement, NOT the reference) timed on the host cores."""
    import torch
    import s3r
    from oracle import s2v_oracle as O
    ncpu = usable_cpus()
    log(f"cpu_baseline: os.cpu_count()={os.cpu_count()} usable={ncpu}")
    torch.set_num_threads(ncpu)
    m = O.OracleStereo2Voxel().eval()
    s3r.seed_module(m, 0)
    B = 2                                           # BASELINE.json configs[0]
    left, right = s3r.synthetic_pairs(B, seed=0)
    with torch.no_grad():
        m(left, right)                              # warm-up (allocations, MKL-DNN primitive cache)
        times = []
        t_end = time.perf_counter() + budget_s
        while time.perf_counter() < t_end and len(times) < 1000:
            t0 = time.perf_counter()
            m(left, right)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "stereo pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (torch CPU fp32, eval, no_grad) Stereo2Voxel forward, batch {B}, median of "
                      f"{len(times)} iterations (~{sum(times):.1f} s of CPU work)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="stereo pairs per GPU per step")
    ap.add_argument("--variant", default="voxel", choices=["voxel", "point"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32: exact-fp32 MFMA path (the headline, BASELINE configs[1]); bf16: bf16 MFMA path, "
                         "channels-last bf16 activations (configs[2], quoted at --batch 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the ~20 kernels of a step eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--include-h2d", action="store_true",
                    help="PCIe-inclusive variant: every step copies its batch host->device first (never the headline)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to exercise the N>1 code on one GPU)")
    ap.add_argument("--same-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (with --backend gloo) so the N>1 path runs on a 1-GPU box")
    ap.add_argument("--autotune", action="store_true",
                    help="time tile / split-K candidates per layer in warm-up instead of using the library's table")
    args = ap.parse_args()

    import torch
    import s3r
    spec = s3r.arch_spec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        log(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}` "
            f"(WORLD_SIZE={world})")
        sys.exit(2)
    dist = None
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    B = args.batch
    prec = "bf16" if args.dtype == "bf16" else "fp32"
    model = s3r.Stereo2Voxel(prec) if args.variant == "voxel" else s3r.Stereo2Point(prec)
    s3r.seed_module(model, 0)
    model.to(dev)
    left, right = s3r.synthetic_pairs(B, seed=1000 + rank)      # random data (never zeros: DVFS, rule 25)
    left, right = left.to(dev), right.to(dev)
    gathered = None
    out_shape = (B, 32, 32, 32) if args.variant == "voxel" else (B, spec.N_POINTS, 3)
    if world > 1:
        # eval collation is double-buffered: step k's all-gather (RCCL, its own stream) runs under step k+1's forward
        gathered = [torch.empty((world * B,) + out_shape[1:], dtype=torch.float32, device=dev) for _ in range(2)]
        staged = [torch.empty(out_shape, dtype=torch.float32, device=dev) for _ in range(2)]
        pending = [None, None]
        step_no = [0]

    host_l = host_r = None
    if args.include_h2d:
        host_l, host_r = left.cpu().pin_memory(), right.cpu().pin_memory()

    tuned = None
    if args.autotune and args.variant == "voxel":
        # untimed warm-up work: each MFMA layer's (tile, split-K) is picked by measurement on this batch
        tuned = model.autotune(left, right, rounds=3, log=log if rank == 0 else None)

    # The step's ~20 launches have no host-side data dependence: capture once, replay per step (hipGraph).
    # Kernel-level HIP-event profiling needs eager launches (events are not captured), so the timed region
    # runs the graph and a second, untimed eager pass afterwards feeds the roofline.
    graphed = None
    if not args.no_graph:
        try:
            graphed = s3r.GraphedForward(model, B, dev)
            graphed.left.copy_(left)
            graphed.right.copy_(right)
        except Exception as e:                       # never lose the measurement to a capture problem
            log(f"rank {rank}: HIP-graph capture failed ({type(e).__name__}: {e}); falling back to eager launches")
            graphed = None
            torch.cuda.synchronize()

    gt_cloud = None
    if args.variant == "point":       # BASELINE configs[3]: Stereo2Point forward + the Chamfer-distance kernel
        g = torch.Generator().manual_seed(77 + rank)
        gt_cloud = torch.rand(B, spec.N_POINTS, 3, generator=g).to(dev)

    def step():
        if host_l is not None:
            (graphed.left if graphed else left).copy_(host_l, non_blocking=True)
            (graphed.right if graphed else right).copy_(host_r, non_blocking=True)
        y = graphed() if graphed else model(left, right)
        if gt_cloud is not None:
            s3r.chamfer_distance(y, gt_cloud)
        if world > 1:
            k = step_no[0] & 1
            step_no[0] += 1
            if pending[k] is not None:
                pending[k].wait()                         # buffer k's previous collation (two steps ago) is done
            staged[k].copy_(y)                            # y is the graph's static output: the next replay overwrites it
            pending[k] = dist.all_gather_into_tensor(gathered[k], staged[k], async_op=True)   # RCCL over xGMI
        return y

    def drain():
        if world > 1:
            for k in range(2):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None

    if world > 1:          # build the RCCL communicator outside the timed region even when --warmup 0
        dist.all_gather_into_tensor(gathered[0], torch.zeros(out_shape, dtype=torch.float32, device=dev))
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()

    profiling = not args.no_profile
    if profiling and graphed is None:
        s3r.profile_enable(64 * args.steps + 64)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                               # every step's collation has completed inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    records = []
    if profiling:
        if graphed is not None:      # per-kernel HIP events: the same K steps again, launched eagerly (untimed)
            s3r.profile_enable(64 * args.steps + 64)
            for _ in range(args.steps):
                yy = model(left, right)
                if gt_cloud is not None:
                    s3r.chamfer_distance(yy, gt_cloud)
            torch.cuda.synchronize()
        records = s3r.profile_read(64 * args.steps + 64)
        s3r.profile_enable(0)

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        fl = spec.flops_per_pair(args.variant)
        # ---- roofline of the dominant kernel family: the fp32-MFMA implicit-GEMM convolution
        roof = None
        if records:
            fam = {}
            for r in records:
                f = fam.setdefault(r["family"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0, "launches": 0})
                f["ms"] += r["ms"]; f["flops"] += r["flops"]; f["bytes"] += r["bytes"]; f["n"] += 1
                f["launches"] += r["launches"]
            per_layer = {}
            for r in records:
                if r["family"] == "conv_mfma":
                    e = per_layer.setdefault(r["tag"], {"ms": 0.0, "flops": 0.0, "n": 0})
                    e["ms"] += r["ms"]; e["flops"] += r["flops"]; e["n"] += 1
            names = {100 + i: l.name for i, l in enumerate(spec.ENCODER)}
            names.update({200 + i: l.name for i, l in enumerate(spec.DECODER)})
            log("kernel family          launches   ms/step   TFLOP/s    GB/s(algorithmic)")
            for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
                log(f"  {k:20s} {f['n']:8d} {f['ms'] / args.steps:9.3f} {f['flops'] / f['ms'] / 1e9 if f['ms'] else 0:9.2f} "
                    f"{f['bytes'] / f['ms'] / 1e6 if f['ms'] else 0:9.1f}")
            log("conv_mfma per layer:   ms/launch   TFLOP/s   frac of fp32 MFMA peak")
            for tag, e in sorted(per_layer.items()):
                tf = e["flops"] / e["ms"] / 1e9
                pk = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS
                log(f"  {names.get(tag, tag)!s:6s} {e['ms'] / e['n']:12.4f} {tf:9.2f} {tf / pk:8.3f}")
            c = fam.get("conv_mfma")
            if c and c["ms"] > 0:
                # dominant kernel = conv_glds_kernel (one template, 16 launches per step): algorithmic FLOPs
                # per launch / average launch duration, both over the timed region's launches (HIP events
                # recorded by the library on the stream it launches on)
                achieved = c["flops"] / c["ms"] / 1e9            # TFLOP/s
                traffic = pmc_util = pmc_clock = None
                bf = args.dtype == "bf16"
                tpath = os.path.join(ROOT, "profiles", "traffic_r01_bf16.json" if bf else "traffic_r01.json")
                # the committed counters are those of the default fp32 run (B=32) / the bf16 run at B=256 only
                if os.path.exists(tpath) and args.variant == "voxel" and B == (256 if bf else 32):
                    try:
                        tj = json.load(open(tpath))
                        traffic = tj.get("hbm_bytes_per_launch")                         # rocprofv3 PMC, per launch
                        pmc_util, pmc_clock = tj.get("mfma_utilisation_pmc"), tj.get("shader_clock_ghz_pmc")
                    except Exception:
                        traffic = None
                peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
                roof = {"bound": "mfma",
                        "kernel": "conv_bf16{,r,p}_kernel (v_mfma_f32_32x32x16_bf16 implicit-GEMM conv, channels-last, LDS-DMA; "
                                  "per-tap / row-reuse / plane-reuse gathers)" if bf
                        else "conv_glds_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit-GEMM conv, LDS-DMA operand staging)",
                        "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic,
                        # from the committed rocprofv3 SQ pass of this command (profiles/traffic_*.json): MFMA pipe
                        # cycles / (1024 SIMDs x elapsed shader cycles), and the shader clock the chip held
                        "mfma_utilisation_pmc": round(pmc_util, 4) if pmc_util else None,
                        "shader_clock_ghz_pmc": round(pmc_clock, 3) if pmc_clock else None,
                        "layers_per_step": c["n"] // args.steps,
                        "launches_per_step": c["launches"] // args.steps,
                        "algorithmic_gflop_per_launch": round(c["flops"] / c["launches"] / 1e9, 3),
                        "avg_launch_ms": round(c["ms"] / c["launches"], 5),
                        "algorithmic_gflop_per_step": round(c["flops"] / args.steps / 1e9, 3),
                        "kernel_ms_per_step": round(c["ms"] / args.steps, 4)}
        out = {
            "metric": f"stereo pairs/s forward (batch {B}, 224x224 -> 32^3 voxel)" if args.variant == "voxel"
                      else f"stereo pairs/s forward (batch {B}, 224x224 -> 2048-pt cloud)",
            "value": round(value, 2), "unit": "stereo pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" + (" (host->device copy of every batch inside the step)" if args.include_h2d else ""),
            "config": {"workload": f"Stereo2{'Voxel' if args.variant == 'voxel' else 'Point'} forward, batch={B} per GPU, "
                                   f"224x224 RGB stereo pair, {'bf16 MFMA path' if args.dtype == 'bf16' else 'fp32'}, "
                                   f"random-init weights, build-specified arch_spec "
                                   f"({fl['total'] / 1e9:.2f} GFLOP/pair)",
                       "per_gpu_batch": B, "global_batch": world * B,
                       "parallelism": f"batch-sharded x{world}, RCCL all-gather of predictions every step (overlapped "
                                      f"with the next step's forward)" if world > 1 else "single GPU"},
            "end_to_end_tflops": round(value * fl["total"] / 1e12, 3),
            "launch": "eager" if graphed is None else "hipGraph replay (1 launch per step); per-kernel HIP-event timing "
                      "for `roofline` taken on an eager re-run of the same K steps right after the timed region",
            "autotuned": {k: [v["tile"], v["ksplit"]] for k, v in tuned.items()} if tuned else None,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
