#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --autotune > $O/bench_autotune.json 2> $O/bench_autotune.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err
python -m pytest tests/test_parity_gpu.py tests/test_quantization_gpu.py -x -q > $O/pytest_parity.txt 2>&1; echo "rc=$?" >> $O/pytest_parity.txt
grep autotune $O/bench_autotune.err; for f in $O/bench_*.json; do python -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print('$f',d['value'],d['ms_per_step'],d['step_ms_spread'],d['roofline']['frac'],d['autotuned'])"; done; tail -n 3 $O/pytest_parity.txt
