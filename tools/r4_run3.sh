#!/bin/bash
mkdir -p gpurun_out/r4e
for rep in 1 2; do for M in 0 1; do
  S3R_DWINO_MAT=$M timeout 300 python tools/layer_bench.py --algo 2 --batch 32 --layers d1,d2,d3 --tiles=-1 --rounds 7 > gpurun_out/r4e/mat${M}_r$rep.log 2>&1
done; done
for B in 32 8 2; do
  timeout 600 python tools/layer_bench.py --algo 2 --batch $B --layers e2,e4,e6,e7,v1,v3,v5,d1,d2,d3 --tiles=-1,0,1,2 --rounds 5 > gpurun_out/r4e/forms_b$B.log 2>&1
done
grep -h "^d[123] *|" gpurun_out/r4e/mat*.log
