#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_quantization_gpu.py tests/test_parity_gpu.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
for i in 1 2 3; do
S3R_NO_DUAL=1 python bench.py --no-cpu-baseline --no-secondary > $O/bench_two_$i.json 2> $O/bench_two_$i.err
python bench.py --no-cpu-baseline --no-secondary > $O/bench_dual_$i.json 2> $O/bench_dual_$i.err
done
tail -n 3 $O/pytest.txt
for f in $O/bench_*.json; do python -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);r=d['roofline'];print('$f',d['value'],d['ms_per_step'],d['step_ms_spread']['median'],r['frac'],r['launches_per_step'],r['kernel_ms_per_step'])"; done
grep -A17 "per layer" $O/bench_two_2.err | awk '{print $1,$2}' | tr '\n' ' '; echo; grep -A17 "per layer" $O/bench_dual_2.err | awk '{print $1,$2}' | tr '\n' ' '
