#!/bin/bash
# round-4 driver: the 2D two-axis form (H, W) on e6 / e7 — tests, per-layer A/B of the launch forms, whole-model bench
out=gpurun_out/r4k
mkdir -p $out
timeout 1500 python -m pytest tests/test_wino_gpu.py -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee $out/rc.txt
tail -5 $out/pytest.log
timeout 600 python tools/layer_bench.py --algo 2 --batch 32 --layers e6,e7,e4 --tiles=2,4,5 --rounds 5 2>&1 | grep -v BEST | grep "|" | sed 's/ TF  *[0-9.]*  *\[/ [/g; s/v0 k0: //g' | tee $out/layers.txt
for B in 32 1 4; do python bench.py --no-secondary --no-cpu-baseline --batch $B > $out/b$B.json 2> $out/b$B.err; python -c "
import json; d=json.load(open('$out/b$B.json')); r=d['roofline']; print($B, d['value'], d['ms_per_step'], r['frac'], r['frac_credited'], r['kernel_ms_per_step'])"; done
grep -E "^  (e[0-9]) " $out/b32.err
