#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory (rocprofv3 CSVs) into profiles/<tag>_*.

  python tools/summarize_profile.py gpurun_out/prof_r01a r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats summary, s3r kernels first),
profiles/<tag>_pmc_per_dispatch.csv (one row per s3r dispatch of one bench step with FETCH_SIZE /
WRITE_SIZE / SQ counters joined by dispatch order) and profiles/traffic_<tag>.json (HBM bytes per
step for the conv_mfma family, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md §HBM).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


FAMILY = "conv_glds,wino_kernel,wino_dual,wino_finish,wino_input,wino_diff"   # the fp32 conv family: conv_glds_kernel, conv_glds_dual_kernel and the Winograd pair
                                            # (input transform + class kernel); third argument overrides it (bf16: "conv_bf16")


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def in_family(name):
    return any(f in name for f in FAMILY.split(","))


def short(name):
    n = name.replace("void ", "").replace("s3r::", "")
    return n.split("(")[0]


def main():
    global FAMILY
    src, tag = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3:
        FAMILY = sys.argv[3]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    # ---- --stats summary
    st = rows(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0])
    fam_calls = sum(int(r["Calls"]) for r in st if in_family(r["Name"]))
    fam_ns = sum(int(r["TotalDurationNs"]) for r in st if in_family(r["Name"]))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        # the dominant kernel is ONE template (conv_glds_kernel) launched 16x per step in several tile
        # instantiations: its family row is what bench.py's roofline (avg launch duration) is checked against
        w.writerow([FAMILY.split(",")[0] + "<*> (all instantiations" + (" + " + " + ".join(FAMILY.split(",")[1:]) if "," in FAMILY else "") + ")", fam_calls, fam_ns, f"{fam_ns / max(fam_calls, 1):.1f}", "", "", ""])
        for r in st:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    # ---- per-dispatch counters: join the three PMC passes by (kernel name, occurrence index)
    per = defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if not files:
            continue
        seen = defaultdict(int)
        last_dispatch = None
        for r in rows(files[0]):
            if "s3r::" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"]:
                continue
            key_d = (r["Kernel_Name"], r["Dispatch_Id"])
            if key_d != last_dispatch:
                seen[r["Kernel_Name"]] += 1
                last_dispatch = key_d
            k = (short(r["Kernel_Name"]), seen[r["Kernel_Name"]] - 1)
            per[k][r["Counter_Name"]] = float(r["Counter_Value"])
            per[k]["grid"] = r["Grid_Size"]
            per[k]["vgpr"] = r["VGPR_Count"]
            per[k]["agpr"] = r["Accum_VGPR_Count"]
            per[k]["lds"] = r["LDS_Block_Size"]
            if sub == "pmc_sq":
                per[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    names = sorted({c for v in per.values() for c in v})
    with open(os.path.join(out, f"{tag}_pmc_per_dispatch.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "occurrence"] + names)
        for (k, i), v in sorted(per.items()):
            w.writerow([k, i] + [v.get(c, "") for c in names])
    # ---- traffic per forward pass; the PMC passes' forward count = number of stem_kernel dispatches
    steps = max(1, sum(1 for (k, _), v in per.items() if k.startswith("stem_") and "FETCH_SIZE" in v))
    fam = [v for (k, _), v in per.items() if in_family(k)]
    launches = len(fam)
    fetch = sum(v.get("FETCH_SIZE", 0) for v in fam) * 1024
    write = sum(v.get("WRITE_SIZE", 0) for v in fam) * 1024
    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in fam)
    gui = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in fam)
    sq_ns = sum(v.get("ns", 0) for v in fam)                  # durations of the SQ-counter pass
    # SQ_VALU_MFMA_BUSY_CYCLES = MFMA pipe cycles summed over the 1024 SIMDs (64 per v_mfma_f32_32x32x2_f32, 32 per
    # v_mfma_f32_32x32x16_bf16: it reproduces the algorithmic MFMA count); GRBM_GUI_ACTIVE sums the 8 XCDs' active
    # cycles.  utilisation = busy / (1024 x elapsed shader cycles); the clock held = elapsed cycles / elapsed time.
    cycles = gui / 8.0
    prof = {}
    try:      # written on the GPU box by tools/profile.sh: hash of the kernel sources that ran + the bench arguments
        prof = json.load(open(os.path.join(src, "profiled.json")))
    except Exception:
        pass
    bargs = prof.get("args", "").split()
    info = {
        "source": src, "steps_profiled": steps, "kernel": FAMILY, "launches_profiled": launches,
        "csrc_sha256": prof.get("csrc_sha256"), "bench_args": prof.get("args"),
        "dtype": bargs[bargs.index("--dtype") + 1] if "--dtype" in bargs else "f32",
        "batch": int(bargs[bargs.index("--batch") + 1]) if "--batch" in bargs else 32,
        "variant": bargs[bargs.index("--variant") + 1] if "--variant" in bargs else "voxel",
        "fetch_bytes_per_step_raw": fetch / steps,
        "fetch_bytes_per_step_x2": 2 * fetch / steps,
        "write_bytes_per_step": write / steps,
        "hbm_bytes_per_step": (2 * fetch + write) / steps,
        "hbm_bytes_per_launch": (2 * fetch + write) / max(launches, 1),
        "stats_avg_launch_ns": fam_ns / max(fam_calls, 1),
        "note": "FETCH_SIZE is in KiB and is doubled per MI355X_MICROARCH.md §HBM (gfx950 reports 1/2 of a wide "
                "coalesced stream; this kernel's loads are 16-byte-per-lane LDS-DMA on most layers); WRITE_SIZE as read",
        "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
        "mfma_utilisation_pmc": busy / (1024.0 * cycles) if cycles else None,
        "shader_clock_ghz_pmc": cycles / sq_ns if sq_ns else None,
    }
    with open(os.path.join(out, f"traffic_{tag}.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(json.dumps(info, indent=1))


if __name__ == "__main__":
    main()
