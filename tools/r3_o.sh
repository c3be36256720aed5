#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-secondary > $O/bench_base_$i.json 2> $O/bench_base_$i.err
S3R_TILE_e5=0 S3R_TILE_v2=7 S3R_TILE_v3=7 python bench.py --no-cpu-baseline --no-secondary > $O/bench_A_$i.json 2> $O/bench_A_$i.err
S3R_TILE_e5=0 S3R_TILE_v2=7 S3R_TILE_v3=7 S3R_TILE_e3=1 S3R_TILE_d2=7 python bench.py --no-cpu-baseline --no-secondary > $O/bench_B_$i.json 2> $O/bench_B_$i.err
done
for f in $O/bench_*.json; do python -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);r=d['roofline'];print('$f',d['value'],d['ms_per_step'],d['step_ms_spread']['median'],r['frac'],r['launches_per_step'],r['kernel_ms_per_step'])"; done
for t in base_2 A_2 B_2; do grep -A17 "per layer" $O/bench_$t.err | awk '{print $1,$2}' | tr '\n' ' '; echo; done
