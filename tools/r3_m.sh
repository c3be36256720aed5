#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
for i in 1 2 3; do
S3R_DUAL_MODEL=0 python bench.py --no-cpu-baseline --no-secondary > $O/bench_old_$i.json 2> $O/bench_old_$i.err
python bench.py --no-cpu-baseline --no-secondary > $O/bench_new_$i.json 2> $O/bench_new_$i.err
done
for f in $O/bench_*.json; do python -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);r=d['roofline'];print('$f',d['value'],d['ms_per_step'],d['step_ms_spread']['median'],r['frac'],r['launches_per_step'],r['kernel_ms_per_step'])"; done
grep -A17 "per layer" $O/bench_old_2.err | awk '{print $1,$2}' | tr '\n' ' '; echo; grep -A17 "per layer" $O/bench_new_2.err | awk '{print $1,$2}' | tr '\n' ' '
python -m pytest tests/test_quantization_gpu.py tests/test_parity_gpu.py -x -q -k "quant or tail or batch32 or each_layer or tile or invariance or odd" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -n 3 $O/pytest.txt
