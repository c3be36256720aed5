#!/bin/bash
mkdir -p gpurun_out/r4i
timeout 1200 python -m pytest tests/test_wino_gpu.py -x -q -m gpu > gpurun_out/r4i/pytest_wino.log 2>&1; tail -5 gpurun_out/r4i/pytest_wino.log
timeout 600 python tools/layer_bench.py --algo 2 --batch 32 --layers v1,v3,v5 --tiles=2,4,5 --rounds 5 2>&1 | grep -v BEST | grep "|" | sed 's/ TF  *[0-9.]*  *\[/ [/g; s/v0 k0: //g'
timeout 600 python tools/layer_bench.py --algo 2 --batch 32 --layers v6 --tiles=3 --rounds 5 2>&1 | grep "|" | sed 's/v0 k0: //g'
timeout 600 python tools/layer_bench.py --algo 1 --batch 32 --layers v5,v6 --rounds 5 2>&1 | grep "|" | sed 's/v0 k0: //g'
for B in 32 1 4; do python bench.py --no-secondary --no-cpu-baseline --batch $B > gpurun_out/r4i/b$B.json 2> gpurun_out/r4i/b$B.err; python -c "
import json; d=json.load(open('gpurun_out/r4i/b$B.json')); r=d['roofline']; print($B, d['value'], d['ms_per_step'], r['frac'], r['frac_credited'], r['kernel_ms_per_step'])"; done
grep -E "^  (v3|v5|v6) " gpurun_out/r4i/b32.err
