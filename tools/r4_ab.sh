#!/bin/bash
# A/B of two builds on one box: $1 = alt .so, $2.. = layers
ALT=$1; shift
mkdir -p gpurun_out/r4ab
for rep in 1 2; do
  timeout 300 python tools/layer_bench.py --algo 2 --batch 32 --layers $1 --tiles=-1 --rounds 7 > gpurun_out/r4ab/main_r$rep.log 2>&1
  S3R_LIB=$ALT timeout 300 python tools/layer_bench.py --algo 2 --batch 32 --layers $1 --tiles=-1 --rounds 7 > gpurun_out/r4ab/alt_r$rep.log 2>&1
done
for rep in 1 2; do
  python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4ab/bench_main_r$rep.json 2> gpurun_out/r4ab/bench_main_r$rep.err
  S3R_LIB=$ALT python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4ab/bench_alt_r$rep.json 2> gpurun_out/r4ab/bench_alt_r$rep.err
done
grep -h "|" gpurun_out/r4ab/main_r*.log | sed 's/^/main /'; grep -h "|" gpurun_out/r4ab/alt_r*.log | sed 's/^/alt  /'
for f in gpurun_out/r4ab/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; done
grep -h "^  d[123]" gpurun_out/r4ab/bench_main_r1.err gpurun_out/r4ab/bench_alt_r1.err
