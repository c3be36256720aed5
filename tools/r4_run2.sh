#!/bin/bash
mkdir -p gpurun_out/r4c
timeout 1200 python -m pytest tests/test_wino_gpu.py tests/test_quantization_gpu.py -x -q -m gpu > gpurun_out/r4c/pytest_wino.log 2>&1
tail -5 gpurun_out/r4c/pytest_wino.log
for B in 32 1 2 4 8 16; do python bench.py --no-secondary --no-cpu-baseline --batch $B > gpurun_out/r4c/b$B.json 2> gpurun_out/r4c/b$B.err; done
for B in 32 8 4 1; do
  timeout 600 python tools/layer_bench.py --algo 2 --batch $B --layers e2,e4,e6,e7,v1,v3,v5,d1,d2,d3 --tiles -1,0,1,2 --rounds 5 > gpurun_out/r4c/forms_b$B.log 2>&1
done
python - <<'PY'
import json
for B in (32,1,2,4,8,16):
    try:
        d=json.load(open(f'gpurun_out/r4c/b{B}.json')); r=d['roofline']
        print(B, d['value'], d['ms_per_step'], r['frac'], r['frac_credited'], r['kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['eager_step_ms'])
    except Exception as e: print(B, 'ERR', e)
PY
