#!/bin/bash
# PMC passes over tools/stem_bench.py (run on the GPU box): where the bf16 stem's time goes.  Every rocprofv3 call is
# under its own timeout (a counter group the profiler cannot schedule has hung for the whole gpurun limit before).
OUT=/root/repo/gpurun_out/stem_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 /root/repo/tools/stem_bench.py --rounds 2 > $OUT/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('/root/repo/gpurun_out/stem_pmc/g*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'stem' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} last={v[-1]:.5g}")
PY
find $OUT -type f ! -name "*.log" -delete
