#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the bench command.
# Summaries land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.
# Usage: bash tools/profile.sh <tag> [extra bench.py arguments, e.g. --dtype bf16 --batch 256]
TAG=${1:-r02}
shift
EXTRA="$*"
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
# what was profiled: the hash of the kernel sources of THIS snapshot (bench.py emits the PMC-derived fields only when
# the tree it runs from hashes to the same value) and the bench arguments
python3 -c "import sys; sys.path.insert(0, '/root/repo'); import bench, json; print(json.dumps({'csrc_sha256': bench.csrc_sha256(), 'args': '$EXTRA'}))" > $OUT/profiled.json
cd /tmp && export TMPDIR=/tmp
CMD="python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $EXTRA"   # 13 identical steps
PMC_CMD="python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $EXTRA"  # 4 identical steps (counters are per dispatch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PMC_CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $PMC_CMD > $OUT/pmc_sq.log 2>&1
# keep what travels back small: the per-dispatch CSVs only
find $OUT -name "*.csv" | head -50
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" -delete
du -sh $OUT
