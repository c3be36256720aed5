#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: Stereo2Voxel forward throughput (stereo pairs/s), batch 32 per GPU,
224x224 stereo pair -> 32^3 voxels, fp32, synthetic inputs, random-init weights (BUILD-SPECIFIED
architecture, arch_spec.py — the reference's model code is not in the mount).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--variant voxel|point] [--dtype f32|bf16]

A "step" is one forward of the hot path over one resident batch of B pairs per GPU (inputs already in
HBM).  N>1: one process per GPU, the batch is sharded (weak scaling: B pairs per rank), and each step
ends with the one exchange the path has — an RCCL all-gather of the (B,32,32,32) predictions for eval
collation.  `python bench.py --gpus N` launches its own N ranks (child `torch.distributed.run`, started
before anything touches the GPU); under an external launcher (WORLD_SIZE set) it is one of the ranks.
Rank 0 prints ONE JSON line.

Besides the headline (BASELINE configs[1]) the same process measures, untimed for `value`, the two other
single-GPU configurations BASELINE.json names — configs[2] (bf16 MFMA path, batch 256) and configs[3]
(Stereo2Point + Chamfer, batch 32) — and reports them under `secondary` (N=1 only; --no-secondary skips).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
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
PEAK_VALU_TLANE_INSTR = 78.6      # fp32 VALU issue: 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz (= the 157.3 TFLOP/s vector peak / 2 FLOPs per FMA)
ACHIEVABLE_HBM_GBS = 6300.0       # what a float4 copy kernel reaches on this chip (MI355X_MICROARCH.md: 6.29 TB/s measured, 79 %)
SCHEMA = 5                        # of the JSON line.  5 (r05): + `roofline.aux`, `gemm_ms_per_step`, `frac_gemm_only`, `frac_useful`.
                                  # 4 (r04): `roofline.frac` = EXECUTED MFMA FLOPs / family kernel time / peak; through r03 (no
                                  # schema key) `frac` charged the direct form's FLOPs, which r04+ reports as `speedup_vs_direct_form`
                                  # (alias `frac_credited`): BENCH_r03 and BENCH_r04+ lines do not compare on `frac`


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


def csrc_sha256():
    """Hash of the kernel sources the loaded library was built from (csrc/*.hip, *.h, include/s3r.h): the key that
    ties a committed rocprofv3 counter summary (profiles/traffic_*.json) to the code it was measured on."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "stereo-3d-reconstruction_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "stereo-3d-reconstruction_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "s3r.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _time_oracle(O, s3r, torch, B, budget_s):
    m = O.OracleStereo2Voxel().eval()
    s3r.seed_module(m, 0)
    left, right = s3r.synthetic_pairs(B, seed=0)
    with torch.no_grad():
        m(left, right)                              # warm-up (allocations, MKL-DNN primitive cache)
        times = []
        t_end = time.perf_counter() + budget_s
        while (time.perf_counter() < t_end or len(times) < 3) and len(times) < 1000:
            t0 = time.perf_counter()
            m(left, right)
            times.append(time.perf_counter() - t0)
    times.sort()
    return B / times[len(times) // 2], len(times), sum(times)


def cpu_baseline(batch=32, budget_s=12.0):
    """The oracle (this build's PyTorch-CPU restatement, NOT the reference) timed on the host cores: the headline's
    own workload (batch 32) and BASELINE.json configs[0] (batch 2)."""
    import torch
    import s3r
    from oracle import s2v_oracle as O
    ncpu = usable_cpus()
    log(f"cpu_baseline: os.cpu_count()={os.cpu_count()} usable={ncpu}")
    torch.set_num_threads(ncpu)
    v, n, s = _time_oracle(O, s3r, torch, batch, budget_s)
    v2, n2, s2 = _time_oracle(O, s3r, torch, 2, budget_s / 2)
    return {"value": round(v, 3), "unit": "stereo pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (torch CPU fp32, eval, no_grad) Stereo2Voxel forward, batch {batch}, median of "
                      f"{n} iterations (~{s:.1f} s of CPU work)",
            "batch2": {"value": round(v2, 3), "sample": f"same oracle at batch 2 (BASELINE.json configs[0]), median of "
                                                        f"{n2} iterations (~{s2:.1f} s)"}}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD process (never exec:
    this process stays the parent and has not touched the GPU), relay rank 0's JSON line (the child shares our
    stdout) and return the child's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={max(1, args.gpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: launching " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    return subprocess.run(cmd, env=env).returncode


def family_table(records, steps):
    fam = {}
    for r in records:
        f = fam.setdefault(r["family"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0, "launches": 0})
        f["ms"] += r["ms"]; f["flops"] += r["flops"]; f["bytes"] += r["bytes"]; f["n"] += 1
        f["launches"] += r["launches"]
    return fam


def pmc_summary(variant, dtype, B, plain_run):
    """Counters from the committed rocprofv3 passes of THIS configuration (tools/profile.sh ->
    tools/summarize_profile.py -> profiles/traffic_<tag>.json).  They are emitted only when that file records the
    hash of the kernel sources this run was built from and the run is the profiled configuration (no autotune, no
    tile overrides): otherwise traffic is null — a stale constant is not a measurement."""
    if not plain_run:
        return None
    want = csrc_sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
        try:
            tj = json.load(open(path))
        except Exception:
            continue
        if "--renders u8" in tj.get("bench_args", ""):          # (a plain run has fp32 renders: not the 8-bit-entry pass)
            continue
        if tj.get("csrc_sha256") == want and tj.get("dtype") == dtype and tj.get("batch") == B and \
                tj.get("variant", "voxel") == variant:
            return {"file": os.path.relpath(path, ROOT), "csrc_sha256": want,
                    "hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
                    "mfma_utilisation_pmc": tj.get("mfma_utilisation_pmc"),
                    "shader_clock_ghz_pmc": tj.get("shader_clock_ghz_pmc")}
    return None


def roofline_of(records, steps, dtype, variant, B, spec, plain_run, quiet=False, eager_step_ms=None):
    """Roofline of the dominant kernel family (the MFMA implicit-GEMM convolution): the FLOPs its launches EXECUTE on the
    matrix cores / their HIP-event durations (events recorded by the library on the stream it launches on).  The library
    reports per launch record what ran (direct kernel or a Winograd form) and the FLOPs that form executes; `frac` is that
    executed rate over the peak (<= 1 by construction: the MFMA pipe's utilisation), `frac_credited` the same time charged
    with the direct form's algorithmic FLOPs (SURVEY 8d's count: what a layer is WORTH, which a Winograd kernel delivers
    with 1/2 .. 9/16 of the multiplications)."""
    if not records:
        return None, {}
    fam = family_table(records, steps)
    per_layer = {}
    for r in records:
        if r["family"] == "conv_mfma":
            e = per_layer.setdefault(r["tag"], {"ms": 0.0, "flops": 0.0, "exec": 0.0, "n": 0, "ran": r.get("ran", "direct"),
                                                "launches": 0, "aux_ms": 0.0, "aux_bytes": 0.0, "aux_n": 0})
            e["ms"] += r["ms"]; e["flops"] += r["flops"]; e["exec"] += r.get("exec_flops", r["flops"]); e["n"] += 1
            e["launches"] += r["launches"]; e["bytes"] = e.get("bytes", 0.0) + r["bytes"]
    for r in records:       # the transform / finish passes of a layer, recorded INSIDE its conv_mfma record (same tag)
        if r["family"] == "aux" and r["tag"] in per_layer:
            e = per_layer[r["tag"]]
            e["aux_ms"] += r["ms"]; e["aux_bytes"] += r["bytes"]; e["aux_n"] += 1
    names = {100 + i: l.name for i, l in enumerate(spec.ENCODER)}
    names.update({200 + i: l.name for i, l in enumerate(spec.DECODER)})
    edge = {l.name: n_in for layers, n0 in ((spec.ENCODER, spec.IMG_HW), (spec.DECODER, spec.MAX_DISP)) for l, n_in, _ in spec.trace(layers, n0)}

    def useful(tag, ran):
        """Share of a layer's executed MFMA FLOPs that lands on outputs the layer has: a Winograd group is 4 outputs per
        transformed axis, so an edge that is not a multiple of 4 computes a partial last group (v3: 16 rows for 14 on both
        axes, v5: 8 for 7)."""
        n = edge.get(names.get(tag), 0)
        if not n or not ran.startswith("winograd") or names[tag].startswith("d") or names[tag] == "v6":
            return 1.0
        f = n / (4.0 * ((n + 3) // 4))
        return f * f if ran == "winograd-2axis" else f
    bf = dtype == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
    if not quiet:
        log(f"[{variant} {dtype} B={B}] kernel family  launches   ms/step   TFLOP/s    GB/s(algorithmic)")
        for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
            log(f"  {k:20s} {f['n']:8d} {f['ms'] / steps:9.3f} {f['flops'] / f['ms'] / 1e9 if f['ms'] else 0:9.2f} "
                f"{f['bytes'] / f['ms'] / 1e6 if f['ms'] else 0:9.1f}")
        log(f"conv_mfma per layer:   ms/layer  launches  executed TFLOP/s  frac of {'bf16' if bf else 'fp32'} MFMA peak   "
            f"direct-form TFLOP/s   aux passes: n  ms   TB/s   GEMM-only frac   what ran")
        for tag, e in sorted(per_layer.items()):
            tf, tfe = e["flops"] / e["ms"] / 1e9, e["exec"] / e["ms"] / 1e9
            gemm = max(e["ms"] - e["aux_ms"], 1e-9)
            log(f"  {names.get(tag, tag)!s:6s} {e['ms'] / e['n']:12.4f} {e['launches'] // e['n']:6d} {tfe:12.2f} {tfe / peak:12.3f} "
                f"{tf:18.2f}   {e['aux_n'] // e['n']:10d} {e['aux_ms'] / e['n']:6.4f} "
                f"{(e['aux_bytes'] / e['aux_ms'] / 1e9 if e['aux_ms'] else 0.0):6.2f} {e['exec'] / gemm / 1e9 / peak:12.3f}   {e['ran']}")
    aux = fam.pop("aux", None)      # (nested inside conv_mfma: reported under roofline.aux, never added to a sum of families)
    # every layer against its OWN roof: the matrix pipe (executed FLOPs over the peak) or HBM (algorithmic bytes — input + output +
    # weights, each once — over the 6.3 TB/s a streaming kernel reaches): the larger fraction names the bound
    layers = {}
    for tag, e in sorted(per_layer.items()):
        fm = e["exec"] / e["ms"] / 1e9 / peak
        fh = e.get("bytes", 0.0) / e["ms"] / 1e6 / ACHIEVABLE_HBM_GBS
        layers[names.get(tag, str(tag))] = {"ms": round(e["ms"] / e["n"], 4), "bound": "hbm" if fh > fm else "mfma",
                                            "frac_mfma": round(fm, 3), "frac_hbm": round(fh, 3),
                                            "algorithmic_gbs": round(e.get("bytes", 0.0) / e["ms"] / 1e6, 1)}
    if not quiet:
        log("per layer against its own roof (mfma: executed FLOPs / peak; hbm: algorithmic bytes / 6.3 TB/s): " +
            "  ".join(f"{k} {v['bound']} {max(v['frac_mfma'], v['frac_hbm']):.2f}" for k, v in layers.items()))
    kernels = {k: {"ms_per_step": round(f["ms"] / steps, 4),
                   "tflops": round(f["flops"] / f["ms"] / 1e9, 2) if f["ms"] else None,
                   "algorithmic_gbs": round(f["bytes"] / f["ms"] / 1e6, 1) if f["ms"] else None,
                   "launches_per_step": f["launches"] // steps} for k, f in fam.items()}
    # the other kernel families against THEIR roofs: the streaming ones (stem, cost volume, the point head's linear
    # layers: 168 MB of weights read once) against the 6.3 TB/s a streaming kernel reaches; Chamfer against VALU issue — the oracle's
    # formula is 8 fp32 instructions per pair and direction (3 sub, 3 mul, 2 add, no FMA: three roundings each) = the record's
    # `flops`, against the 78.6 T lane-instructions/s the SIMDs issue (the fp32 vector peak counted in instructions, not FMAs)
    for k, v in kernels.items():
        f = fam[k]
        if not f["ms"]:
            continue
        if k in ("stem", "cost_volume", "linear", "head", "pad_copy", "iou", "disparity"):
            v["roofline"] = {"bound": "hbm", "achieved": round(f["bytes"] / f["ms"] / 1e6, 1), "peak": ACHIEVABLE_HBM_GBS, "unit": "GB/s",
                             "frac": round(f["bytes"] / f["ms"] / 1e6 / ACHIEVABLE_HBM_GBS, 4)}
        elif k == "chamfer":
            v["roofline"] = {"bound": "valu_issue", "achieved": round(f["flops"] / f["ms"] / 1e9, 2), "peak": PEAK_VALU_TLANE_INSTR,
                             "unit": "T lane-instructions/s", "frac": round(f["flops"] / f["ms"] / 1e9 / PEAK_VALU_TLANE_INSTR, 4),
                             "note": "8 instructions per pair and direction (no FMA in the oracle's formula) + the running minimum "
                                     "and index select (2-3 more per pair: not in the count), so ~0.75 is this formula's ceiling"}
    c = fam.get("conv_mfma")
    if not c or c["ms"] <= 0:
        return None, kernels
    # the family's kernel time step by step (records arrive in launch order, the same number per step): what the
    # average launch duration below averages over
    conv = [r["ms"] for r in records if r["family"] == "conv_mfma"]
    per = len(conv) // steps if steps else 0
    step_sums = [sum(conv[i * per:(i + 1) * per]) for i in range(steps)] if per and per * steps == len(conv) else []
    if step_sums and not quiet:
        log("conv_mfma kernel ms per eager step, in order: " + " ".join(f"{v:.3f}" for v in step_sums))
    step_sums.sort()
    executed = sum(e["exec"] for e in per_layer.values())
    executed_useful = sum(e["exec"] * useful(t, e["ran"]) for t, e in per_layer.items())
    aux_ms = aux["ms"] if aux else 0.0
    gemm_ms = max(c["ms"] - aux_ms, 1e-9)
    credited = c["flops"] / c["ms"] / 1e9            # TFLOP/s on the direct form's count
    achieved = executed / c["ms"] / 1e9              # TFLOP/s the matrix cores execute
    # §8d's formulas count the taps that multiply padding zeros; without them (arch_spec.layer_macs_interior)
    ratio = spec.mfma_flops_per_pair(variant, interior=True) / spec.mfma_flops_per_pair(variant)
    pmc = pmc_summary(variant, dtype, B, plain_run)
    all_ms = sum(f["ms"] for f in fam.values())      # (`aux` was taken out above: its time is inside conv_mfma's)
    roof = {"bound": "mfma",
            "kernel": "conv_bf16{,r,p}_kernel (bf16 MFMA implicit-GEMM conv, channels-last, LDS-DMA; "
                      "per-tap / row-reuse / plane-reuse gathers)" if bf
            else "the fp32 v_mfma_f32_32x32x2_f32 implicit-GEMM convolution family, LDS-DMA operand staging: conv_glds_kernel / "
                 "conv_glds_dual_kernel (direct form) + the Winograd class kernel wino_kernel / wino_dual_kernel with its transform and "
                 "finish passes (wino_input / wino_diff / wino_finish, wino2_input / wino2p_input, wino2s_finish / wino2p_finish / "
                 "wino2_finish_flat).  One-axis forms: F(4,3) along H (1/2 of the direct multiplications), F(2,2) along D and H inside the "
                 "parity classes of a transposed convolution (9/16).  Two-axis forms on every stride-1 layer with an edge <= 28: F(4,3)^2 "
                 "over D, H (v1, v3, v5) or H, W (e6, e7): 1/4; F(2,4)^2 (v6): 25/64.  Launch forms (serial, class-parallel, dual, "
                 "semi-fused) are bit-identical per algorithm",
            "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            # executed minus the Winograd groups that lie beyond an edge (v3: 16 rows for 14, v5: 8 for 7): the pipe's USEFUL work
            "frac_useful": round(executed_useful / c["ms"] / 1e9 / peak, 4),
            # the same executed FLOPs over the family's time WITHOUT its transform / difference / finish / combine passes
            # (`aux` below, timed in a second eager pass of the same K steps): the utilisation of the matrix pipe inside the
            # kernels that use it
            "gemm_ms_per_step": round(gemm_ms / steps, 4),
            "frac_gemm_only": round(executed / gemm_ms / 1e9 / peak, 4),
            # the family's HBM-bound passes, graded against their own roof: algorithmic bytes (what each pass must read and
            # write) over its HIP-event time, against the 6.3 TB/s a streaming kernel reaches on this chip
            "aux": ({"bound": "hbm", "ms_per_step": round(aux_ms / steps, 4), "passes_per_step": aux["n"] // steps,
                     "algorithmic_gb_per_step": round(aux["bytes"] / steps / 1e9, 4),
                     "achieved": round(aux["bytes"] / aux_ms / 1e6, 1) if aux_ms else None, "peak": ACHIEVABLE_HBM_GBS, "unit": "GB/s",
                     "frac": round(aux["bytes"] / aux_ms / 1e6 / ACHIEVABLE_HBM_GBS, 4) if aux_ms else None,
                     "share_of_family_time": round(aux_ms / c["ms"], 4)} if aux else None),
            # NOT a roofline fraction: the same kernel time charged with the DIRECT form's FLOP count over the peak =
            # how much faster than a direct-form kernel AT THE PEAK the family runs; > 1 because Winograd forms skip multiplications
            "achieved_credited": round(credited, 3),
            "speedup_vs_direct_form_at_peak": round(credited / peak, 4),
            "frac_credited": round(credited / peak, 4),                       # (r04's name for the line above, kept for the driver)
            "frac_credited_border_excluded": round(credited / peak * ratio, 4),
            "executed_over_algorithmic_mfma_flops": round(executed / c["flops"], 4),
            "what_ran": {names.get(t, str(t)): e["ran"] for t, e in sorted(per_layer.items())},
            "layers": layers,
            "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
            "pmc_source": pmc,
            # `peak` is the data-sheet figure at 2.4 GHz; the profiled box ran its kernels at pmc.shader_clock_ghz_pmc: the same
            # executed rate over the peak AT THAT CLOCK is what the counters' busy-cycle ratio (mfma_utilisation_pmc) measures
            "frac_at_pmc_clock": round(achieved / peak * 2.4 / pmc["shader_clock_ghz_pmc"], 4)
            if pmc and pmc.get("shader_clock_ghz_pmc") else None,
            "layers_per_step": c["n"] // steps,
            "launches_per_step": c["launches"] // steps,
            "executed_gflop_per_launch": round(executed / c["launches"] / 1e9, 3),
            "algorithmic_gflop_per_launch": round(c["flops"] / c["launches"] / 1e9, 3),
            "avg_launch_ms": round(c["ms"] / c["launches"], 5),
            "executed_gflop_per_step": round(executed / steps / 1e9, 3),
            "algorithmic_gflop_per_step": round(c["flops"] / steps / 1e9, 3),
            "kernel_ms_per_step": round(c["ms"] / steps, 4),
            "all_kernels_ms_per_step": round(all_ms / steps, 4),
            "eager_step_ms": round(eager_step_ms, 4) if eager_step_ms else None,
            "kernel_ms_per_step_spread": ({"min": round(step_sums[0], 4), "median": round(step_sums[len(step_sums) // 2], 4),
                                           "max": round(step_sums[-1], 4)} if step_sums else None)}
    return roof, kernels


def eager_records(s3r, torch, model, left, right, gt_cloud, steps, detail=False):
    """Per-kernel HIP events cannot bracket kernels inside a graph replay: run the same K steps eagerly (untimed).
    The chip needs a few steps of uninterrupted work to settle after the host-side pause that precedes this pass (the
    first three or four eager steps' kernels ran 2-14 % longer than the rest: clocks ramping back up), so six steps go
    first and the recording starts behind them WITHOUT draining the queue — what is averaged is the steady state the
    timed region runs in.  Returns (records, ms per eager step): the second from one event pair around the K recorded
    steps on the same stream, so that sum(kernel ms) <= step ms can be read off the line itself."""
    def one():
        yy = model(left, right)
        if gt_cloud is not None:
            s3r.chamfer_distance(yy, gt_cloud)
    s3r.profile_detail(1 if detail else 0)          # (detail: + one record per transform / finish pass, nested in its layer's)
    s3r.profile_enable(128 * steps + 64)
    for _ in range(6):
        one()
    s3r.profile_reset()                              # (host-side only: the queue stays full)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize()
    records = s3r.profile_read(128 * steps + 64)
    s3r.profile_enable(0)
    s3r.profile_detail(0)
    return records, e0.elapsed_time(e1) / steps


def aux_records(s3r, torch, model, left, right, gt_cloud, steps):
    """A second eager pass with one record per aux pass (input transforms, difference tensors, finish / combine kernels).  Kept
    apart from the pass that times the layers: every extra event pair between two kernels of a layer costs queue time (r05: 19
    pairs a step made the family's kernel time read 5 % longer)."""
    recs, _ = eager_records(s3r, torch, model, left, right, gt_cloud, steps, detail=True)
    return [r for r in recs if r["family"] == "aux"]


# environment switches that change which kernel (or which variant of one) runs without changing the kernel sources: a run
# under any of them is not the configuration the committed counter passes were taken on
_KERNEL_ENV = ("S3R_LIB", "S3R_ABL", "S3R_BF16_MFMA", "S3R_STEM_MFMA", "S3R_DEEP_RING", "S3R_NO_TAIL_CUT",
               "S3R_NO_DUAL", "S3R_WINO", "S3R_DWINO_MAT", "S3R_WINO_FORM", "S3R_WINO2_FORM", "S3R_WINO2_MAX_EDGE", "S3R_STEM_WINO",
               "S3R_WINO_HANDOFF")      # (tools/README.md holds the table; tests/test_abi_cpu.py checks it against the sources)


def kernel_env_overrides(s3r=None):
    """... plus whatever an open s3r.debug_overrides(...) context forces (`--override`): per-layer descriptor fields"""
    found = sorted(k for k in os.environ if k.startswith(_KERNEL_ENV))
    if s3r is not None:
        found += [f"{kind}.{name}={v}" for kind, d in sorted(s3r.debug_overrides.active().items()) for name, v in sorted(d.items())]
    return found


def parse_overrides(items):
    """--override tile.v2=2 ksplit.v4=2 algo.e6=1  ->  kwargs of s3r.debug_overrides"""
    kw = {"tile": {}, "ksplit": {}, "algo": {}}
    for it in items or []:
        try:
            key, val = it.split("=")
            kind, name = key.split(".")
            kw[kind][name] = int(val)
        except (ValueError, KeyError):
            sys.exit(f"--override wants tile|ksplit|algo.<layer>=<int>, got {it!r}")
    return kw


def secondary_config(s3r, torch, dev, variant, dtype, B, steps, warmup, plain_run):
    """One more single-GPU configuration of BASELINE.json, measured the way the headline is (graph replay of K
    steps over one resident batch between synchronisations), in this process, after the headline's timed region."""
    spec = s3r.arch_spec
    prec = "bf16" if dtype == "bf16" else "fp32"
    model = s3r.Stereo2Voxel(prec) if variant == "voxel" else s3r.Stereo2Point(prec)
    s3r.seed_module(model, 0)
    model.to(dev)
    left, right = s3r.synthetic_pairs(B, seed=2000)
    left, right = left.to(dev), right.to(dev)
    gt_cloud = None
    if variant == "point":
        gt_cloud = torch.rand(B, spec.N_POINTS, 3, generator=torch.Generator().manual_seed(78)).to(dev)
    graphed = s3r.GraphedForward(model, B, dev)
    graphed.left.copy_(left)
    graphed.right.copy_(right)

    def step():
        y = graphed()
        if gt_cloud is not None:
            s3r.chamfer_distance(y, gt_cloud)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    records, eager_ms = eager_records(s3r, torch, model, left, right, gt_cloud, steps)
    records = records + aux_records(s3r, torch, model, left, right, gt_cloud, steps)
    roof, kernels = roofline_of(records, steps, dtype, variant, B, spec, plain_run=plain_run, eager_step_ms=eager_ms)
    fl = spec.flops_per_pair(variant)
    out = {"workload": f"Stereo2{'Voxel' if variant == 'voxel' else 'Point'} forward"
                       f"{' + Chamfer distance vs a (B,2048,3) cloud' if variant == 'point' else ''}, batch={B}, "
                       f"{'bf16 MFMA path' if dtype == 'bf16' else 'fp32'}, 1 GPU",
           "value": round(B * steps / elapsed, 2), "unit": "stereo pairs/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * elapsed / steps, 4), "dtype": dtype,
           "end_to_end_tflops": round(B * steps / elapsed * fl["total"] / 1e12, 3),
           "roofline": roof, "kernels": kernels}
    del graphed, model
    torch.cuda.empty_cache()
    return out


def in_flight_config(s3r, torch, dev, B, steps, warmup, streams=3):
    """The headline workload with `streams` independent batches in flight: each HIP stream replays its own model's graph (own
    arena, same weights) back to back, nothing joins them until the end.  Kernels of different batches share the chip — a finish /
    transform pass (HBM-bound) beside another batch's class GEMM (MFMA-bound) — which is how a serving loop would run the path;
    never part of `value`, whose steps run one after the other on one stream."""
    models = [s3r.Stereo2Voxel("fp32") for _ in range(streams)]
    s3r.seed_module(models[0], 0)
    for m in models[1:]:
        m.load_state_dict(models[0].state_dict())
    graphs, sts = [], []
    for i, m in enumerate(models):
        m.to(dev)
        g = s3r.GraphedForward(m, B, dev)
        l, r = s3r.synthetic_pairs(B, seed=3000 + i)
        g.left.copy_(l.to(dev)); g.right.copy_(r.to(dev))
        graphs.append(g); sts.append(torch.cuda.Stream(device=dev))
    cur = torch.cuda.current_stream(dev)

    def run(n):
        for st in sts:
            st.wait_stream(cur)
        for _ in range(n):
            for g, st in zip(graphs, sts):
                with torch.cuda.stream(st):
                    g()
        for st in sts:
            cur.wait_stream(st)

    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    out = {"workload": f"Stereo2Voxel forward, batch={B}, fp32, 1 GPU, {streams} independent batches in flight on {streams} HIP streams",
           "value": round(B * steps * streams / elapsed, 2), "unit": "stereo pairs/s", "steps": steps * streams, "streams": streams,
           "ms_per_step": round(1e3 * elapsed / (steps * streams), 4), "dtype": "f32"}
    del graphs, models
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="stereo pairs per GPU per step (--batch 256 --gpus 8 is "
                                                          "BASELINE configs[4]'s 8 x 256)")
    ap.add_argument("--variant", default="voxel", choices=["voxel", "point"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32: exact-fp32 MFMA path (the headline, BASELINE configs[1]); bf16: bf16 MFMA path, "
                         "channels-last bf16 activations (configs[2], quoted at --batch 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[2] / configs[3] measurements that follow the headline at N=1")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the ~20 kernels of a step eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--include-h2d", action="store_true",
                    help="PCIe-inclusive variant: every step copies its batch host->device first (never the headline)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to exercise the N>1 code on one GPU)")
    ap.add_argument("--same-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (with --backend gloo) so the N>1 path runs on a 1-GPU box")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N>1 code path (init_process_group, warm-up all-gather, all-reduce, RCCL version query, "
                         "per-step collation) even with one rank: a world-size-1 RCCL rehearsal on a 1-GPU box")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="pairs per step over ALL ranks (overrides --batch: per-GPU batch = global / N); "
                         "--global-batch 2048 --gpus 8 is BASELINE configs[4]")
    ap.add_argument("--renders", default="f32", choices=["f32", "u8"],
                    help="dtype of the resident input renders: f32 (SURVEY 8d's torch.rand pairs, the headline) or u8 "
                         "(8-bit renders, scaled by 1/255 inside the first kernel)")
    ap.add_argument("--autotune", action="store_true",
                    help="time tile / split-K candidates per layer in warm-up instead of using the library's table")
    ap.add_argument("--override", nargs="*", default=[], metavar="KIND.LAYER=INT",
                    help="tuning / diagnosis only: force descriptor fields of named layers for the whole run through "
                         "s3r.debug_overrides (tile.v2=2 ksplit.v4=2 algo.e6=1); the line is marked as not the plain configuration")
    args = ap.parse_args()
    forced = parse_overrides(args.override)

    if args.same_device and args.backend == "nccl" and max(args.gpus, int(os.environ.get("WORLD_SIZE", "1"))) > 1:
        sys.exit("--same-device puts every rank on cuda:0, which RCCL refuses (one communicator rank per device): "
                 "use --backend gloo with it")
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))              # nothing has touched the GPU in this process

    import torch
    import s3r
    spec = s3r.arch_spec
    if any(forced.values()):
        s3r.debug_overrides(**forced).__enter__()      # for the whole process: every model of this run is built under it

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        log(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        sys.exit(2)
    dist = None
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist       # (--force-dist: the same code with a communicator of one rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    B = args.batch
    if args.global_batch:
        if args.global_batch % world:
            log(f"--global-batch {args.global_batch} does not divide over {world} ranks")
            sys.exit(2)
        B = args.global_batch // world
    prec = "bf16" if args.dtype == "bf16" else "fp32"
    model = s3r.Stereo2Voxel(prec) if args.variant == "voxel" else s3r.Stereo2Point(prec)
    s3r.seed_module(model, 0)
    model.to(dev)
    left, right = s3r.synthetic_pairs(B, seed=1000 + rank)      # random data (never zeros: DVFS, rule 25)
    if args.renders == "u8":                                    # the same pairs quantised to 8 bits
        left, right = (left * 255.0).round().to(torch.uint8), (right * 255.0).round().to(torch.uint8)
    left, right = left.to(dev), right.to(dev)
    gathered = None
    out_shape = (B, 32, 32, 32) if args.variant == "voxel" else (B, spec.N_POINTS, 3)
    n_ranks_seen = 1
    if dist_on:
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
            graphed = s3r.GraphedForward(model, B, dev, input_dtype=left.dtype)
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
        if dist_on:
            k = step_no[0] & 1
            step_no[0] += 1
            if pending[k] is not None:
                pending[k].wait()                         # buffer k's previous collation (two steps ago) is done
            staged[k].copy_(y)                            # y is the graph's static output: the next replay overwrites it
            pending[k] = dist.all_gather_into_tensor(gathered[k], staged[k], async_op=True)   # RCCL over xGMI
        return y

    def drain():
        if dist_on:
            for k in range(2):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None

    if dist_on:            # build the RCCL communicator outside the timed region even when --warmup 0
        dist.all_gather_into_tensor(gathered[0], torch.zeros(out_shape, dtype=torch.float32, device=dev))
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)                              # every rank the launcher started is really in the job
        n_ranks_seen = int(one.item())
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()

    profiling = not args.no_profile
    if profiling and graphed is None:
        s3r.profile_enable(128 * args.steps + 64)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                               # every step's collation has completed inside the timed region
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    records = []
    if profiling and graphed is None:                     # eager launches: the timed region's own kernel events
        records = s3r.profile_read(128 * args.steps + 64)
        s3r.profile_enable(0)

    # per-step spread (untimed for `value`): the same K steps once more, each bracketed by its own pair of events on
    # the launch stream — the timed region above is one interval, this shows what it averages over
    spread = None
    if not dist_on:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evs:
            a.record()
            step()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        spread = {"min": round(ts[0], 4), "median": round(ts[len(ts) // 2], 4), "max": round(ts[-1], 4)}

    eager_ms = None
    if profiling and graphed is not None:                 # per-kernel HIP events: the same K steps again, eagerly (untimed)
        records, eager_ms = eager_records(s3r, torch, model, left, right, gt_cloud, args.steps)
        records = records + aux_records(s3r, torch, model, left, right, gt_cloud, args.steps)

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        fl = spec.flops_per_pair(args.variant)
        overrides = kernel_env_overrides(s3r)
        plain_run = not args.autotune and not args.no_graph and not overrides and not args.include_h2d and \
            args.renders == "f32"
        roof, kernels = roofline_of(records, args.steps, args.dtype, args.variant, B, spec, plain_run, eager_step_ms=eager_ms)
        rccl = None
        if dist_on and args.backend == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        out = {
            "schema": SCHEMA,
            "metric": f"stereo pairs/s forward (batch {B}, 224x224 -> 32^3 voxel)" if args.variant == "voxel"
                      else f"stereo pairs/s forward (batch {B}, 224x224 -> 2048-pt cloud)",
            "value": round(value, 2), "unit": "stereo pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "step_ms_spread": spread, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" + (" (host->device copy of every batch inside the step)" if args.include_h2d else ""),
            "config": {"workload": f"Stereo2{'Voxel' if args.variant == 'voxel' else 'Point'} forward, batch={B} per GPU, "
                                   f"224x224 RGB stereo pair ({'8-bit' if args.renders == 'u8' else 'fp32'} renders), "
                                   f"{'bf16 MFMA path' if args.dtype == 'bf16' else 'fp32'}, "
                                   f"random-init weights, build-specified arch_spec "
                                   f"({fl['total'] / 1e9:.2f} GFLOP/pair)",
                       "per_gpu_batch": B, "global_batch": world * B,
                       "parallelism": f"batch-sharded x{world}, RCCL all-gather of predictions every step (overlapped "
                                      f"with the next step's forward)" if dist_on else "single GPU"},
            "n_ranks_seen": n_ranks_seen, "collective_backend": (args.backend if dist_on else None),
            "kernel_env_overrides": overrides or None,
            "rccl_version": rccl,
            "end_to_end_tflops": round(value * fl["total"] / 1e12, 3),
            "launch": "eager" if graphed is None else "hipGraph replay (1 launch per step); per-kernel HIP-event timing "
                      "for `roofline` taken on an eager re-run of the same K steps right after the timed region",
            "autotuned": {k: [v["tile"], v["ksplit"]] for k, v in tuned.items()} if tuned else None,
            "roofline": roof, "kernels": kernels,
        }
        if roof and eager_ms:
            # two clocks: `ms_per_step` times K graph replays (no host launch gaps); the per-kernel events come from an EAGER re-run, where
            # each record's interval runs from the previous kernel's end to its own end — launch gap included — so their sum tiles the
            # eager step (<= eager_step_ms) and may exceed the graph step by what the graph saves in gaps
            roof["graph_step_ms"] = round(ms_per_step, 4)
            roof["eager_minus_graph_step_ms"] = round(eager_ms - ms_per_step, 4)
        if not dist_on and not args.no_secondary and args.variant == "voxel" and args.dtype == "f32" and \
                not args.include_h2d and args.renders == "f32":
            # BASELINE.json configs[2] and configs[3], measured in the same process (never part of `value`)
            graphed = None                           # (its buffers stay with the model; the secondaries build their own)
            sec = {}
            k2 = max(5, min(args.steps, 10))
            # (fp32_b256: the per-rank workload of configs[4]'s 8 x 256 — the N = 1 anchor a 1 -> 8 weak-scaling curve divides by)
            for name, (v, d, b) in {"bf16_b256": ("voxel", "bf16", 256), "point_b32": ("point", "f32", 32),
                                    "fp32_b256": ("voxel", "f32", 256)}.items():
                try:
                    sec[name] = secondary_config(s3r, torch, dev, v, d, b, k2, max(2, min(args.warmup, 3)),
                                                 plain_run=not overrides)
                except Exception as e:
                    log(f"secondary {name} failed: {type(e).__name__}: {e}")
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}
            try:
                sec["in_flight3_b32"] = in_flight_config(s3r, torch, dev, 32, k2, 2)
            except Exception as e:
                log(f"secondary in_flight3_b32 failed: {type(e).__name__}: {e}")
                sec["in_flight3_b32"] = {"error": f"{type(e).__name__}: {e}"}
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline and not args.force_dist:
            out["cpu_baseline"] = cpu_baseline(32)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
