#!/bin/bash
# round-4 experiment driver (runs on the GPU box): Winograd launch forms per layer at several batches
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_wino_gpu.py -x -q -m gpu > gpurun_out/r4b/pytest_wino.log 2>&1
tail -5 gpurun_out/r4b/pytest_wino.log
for B in 32 8 4 1; do
  timeout 600 python tools/layer_bench.py --algo 2 --batch $B --layers e2,e4,e6,e7,v1,v3,d1,d2,d3 --tiles 0,8,1,5,2,6 --rounds 5 > gpurun_out/r4b/forms_b$B.log 2>&1
  timeout 300 python tools/layer_bench.py --algo 1 --batch $B --layers e2,e4,e6,e7,v1,v3,v5,d1,d2,d3 --rounds 5 > gpurun_out/r4b/direct_b$B.log 2>&1
done
S3R_WINO_R4=1 timeout 600 python tools/layer_bench.py --algo 2 --batch 32 --layers v3,v5 --tiles 0,8,1,5,2,6 --rounds 5 > gpurun_out/r4b/r4_b32.log 2>&1
timeout 300 python tools/layer_bench.py --algo 2 --batch 32 --layers v5 --tiles 0,8,1,5 --rounds 5 > gpurun_out/r4b/v5_r2_b32.log 2>&1
grep -h "BEST\|^!!" gpurun_out/r4b/*.log | head -80
