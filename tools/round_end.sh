#!/bin/bash
# Runs on the GPU box at the end of a round: the GPU suite, the profiled configurations, the end-to-end runner and the batch sweep.
#   bash tools/round_end.sh r06          (round tag: names profiles/<tag>_* and gpurun_out/<tag>_final/)
set -x
R=${1:?usage: round_end.sh <round tag, e.g. r06>}
OUT=gpurun_out/${R}_final
mkdir -p ${OUT}
python -m pytest tests -m gpu -x -q > ${OUT}/test_gpu.txt 2>&1
bash tools/profile_all.sh ${R} > ${OUT}/profile_all.log 2>&1
python runner.py --test --samples 4096 --batch 32 > ${OUT}/runner_f32.json 2> ${OUT}/runner_f32.err
python runner.py --test --samples 8192 --batch 256 --precision bf16 > ${OUT}/runner_bf16.json 2> ${OUT}/runner_bf16.err
python runner.py --test --samples 4096 --batch 32 --renders f32 > ${OUT}/runner_f32_f32renders.json 2> /dev/null
python bench.py --include-h2d --no-secondary --no-cpu-baseline > ${OUT}/bench_h2d.json 2> /dev/null
python bench.py --include-h2d --renders u8 --no-secondary --no-cpu-baseline > ${OUT}/bench_h2d_u8.json 2> /dev/null
for b in 1 2 4 8 16 64 256; do python bench.py --no-secondary --no-cpu-baseline --batch $b > ${OUT}/bench_b$b.json 2>/dev/null; done
