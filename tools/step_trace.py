#!/usr/bin/env python3
"""One forward step as the sequence of its kernels: reads a rocprofv3 --kernel-trace CSV of bench.py, cuts the dispatch list at
the stem kernel and prints the median duration of every position of the step (name, us), in launch order.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary
    python3 tools/step_trace.py /tmp/tr"""
import csv, glob, statistics, sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
cuts = [i for i, (n, _) in enumerate(ev) if "stem" in n]
steps = [ev[a:b] for a, b in zip(cuts, cuts[1:])]
length = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == length][-8:]
total = 0.0
for k in range(length):
    us = statistics.median(s[k][1] for s in steps)
    total += us
    name = steps[0][k][0].replace("s3r::", "").replace("void ", "")
    print(f"{k:3d} {us:8.1f} us  {name[:110]}")
print(f"sum {total:.1f} us over {length} kernels ({len(steps)} steps)")
