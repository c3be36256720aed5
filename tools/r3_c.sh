#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
python tools/eval_breakdown.py --precision bf16 --batch 256 --samples 3072 > $O/breakdown_bf16.txt 2>&1
python tools/eval_breakdown.py --precision fp32 --batch 32 --samples 1024 > $O/breakdown_fp32.txt 2>&1
python tools/layer_bench.py --dtype bf16 --batch 256 --shapes 32,16 --tiles -1 --rounds 5 > $O/lb_all_shapes.txt 2>&1
cat $O/breakdown_bf16.txt $O/breakdown_fp32.txt; grep -v BEST $O/lb_all_shapes.txt
