#!/usr/bin/env python3
"""Start / end of the last forward's conv dispatches from a rocprofv3 --kernel-trace CSV directory: do the
remainder launches of a cut layer (s3r_conv_glds.hip, plan_tail_cut) run beside their bulk launch?
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --no-graph --steps 3 ...
    python3 tools/kernel_overlap.py /tmp/kt"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'conv_glds' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
t0 = int(rows[-n]['Start_Timestamp'])
for r in rows[-n:]:
    nm = r['Kernel_Name'].split('conv_glds_kernel')[1][:22]
    print(f"{nm:24s} grid {int(r['Grid_Size_X']):8d} queue {r.get('Queue_Id', '?'):>3s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} "
          f"end {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us")
