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


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def short(name):
    n = name.replace("void ", "").replace("s3r::", "")
    return n.split("(")[0]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out, exist_ok=True)
    # ---- --stats summary
    st = rows(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0])
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
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
    # ---- traffic per bench step (profile.sh runs 1 warm-up + 3 timed steps = 4 identical steps)
    steps = 4
    fetch = sum(v.get("FETCH_SIZE", 0) for (k, _), v in per.items() if k.startswith("conv_mfma")) * 1024 / steps
    write = sum(v.get("WRITE_SIZE", 0) for (k, _), v in per.items() if k.startswith("conv_mfma")) * 1024 / steps
    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for (k, _), v in per.items() if k.startswith("conv_mfma"))
    gui = sum(v.get("GRBM_GUI_ACTIVE", 0) for (k, _), v in per.items() if k.startswith("conv_mfma"))
    info = {
        "source": src, "steps_profiled": steps,
        "conv_mfma_fetch_bytes_per_step_raw": fetch,
        "conv_mfma_fetch_bytes_per_step_x2": 2 * fetch,
        "conv_mfma_write_bytes_per_step": write,
        "conv_mfma_hbm_bytes_per_step": 2 * fetch + write,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports 1/2 of a coalesced stream; the "
                "dword gather pattern of this kernel is uncalibrated, so raw is kept beside it)",
        "conv_mfma_SQ_VALU_MFMA_BUSY_CYCLES": busy, "conv_mfma_GRBM_GUI_ACTIVE": gui,
    }
    with open(os.path.join(out, f"traffic_{tag}.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(json.dumps(info, indent=1))


if __name__ == "__main__":
    main()
