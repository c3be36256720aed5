#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_base_$i.json 2> $O/bench_base_$i.err
S3R_TILE_e4=4 S3R_TILE_d1=7 S3R_TILE_e2=7 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_tuned_$i.json 2> $O/bench_tuned_$i.err
done
for f in $O/bench_*.json; do python -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print('$f',d['value'],d['ms_per_step'],d['step_ms_spread'],d['roofline']['frac'])"; done
