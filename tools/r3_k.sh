#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-secondary > $O/bench_$i.json 2> $O/bench_$i.err
done
python bench.py > $O/bench_full.json 2> $O/bench_full.err
for f in $O/bench_*.json; do python -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);r=d['roofline'];print('$f',d['value'],d['ms_per_step'],r['frac'],r['kernel_ms_per_step'],r['kernel_ms_per_step_spread'])"; done
