#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the bench command.
# Summaries land in gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
# Usage: bash tools/profile.sh <tag> [extra bench.py arguments, e.g. --dtype bf16 --batch 256]
TAG=${1:-r01}
shift
EXTRA="$*"
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA"   # 13 identical steps, default configuration
PMC_CMD="python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA"  # 4 identical steps (counters are per dispatch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PMC_CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $PMC_CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -50
du -sh $OUT
