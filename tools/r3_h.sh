#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "rc=$?" >> $O/pytest_all.txt
python tools/layer_bench.py --layers e2,e4,e6,e7,v5,v6,d1 --tiles 0,1,2,3,4,5,6,7 --variants 0 --rounds 5 > $O/lb_fp32_tiles.txt 2>&1
python tools/layer_bench.py --layers v5,v6,d1,v4 --tiles 3,7,6,0 --variants 0 --ksplits 1,2,4,8 --rounds 5 > $O/lb_fp32_ksplit.txt 2>&1
for r in u8 f32; do
  python runner.py --test --precision bf16 --batch 256 --samples 3072 --renders $r > $O/runner_bf16_$r.json 2> $O/runner_bf16_$r.err
  python runner.py --test --precision fp32 --batch 32 --samples 1024 --renders $r > $O/runner_fp32_$r.json 2> $O/runner_fp32_$r.err
done
python runner.py --test --precision bf16 --batch 256 --samples 8192 --renders u8 > $O/runner_bf16_u8_8k.json 2> $O/runner_bf16_u8_8k.err
python runner.py --test --precision fp32 --batch 32 --samples 4096 --renders u8 > $O/runner_fp32_u8_4k.json 2> $O/runner_fp32_u8_4k.err
tail -n 5 $O/pytest_all.txt; grep -v BEST $O/lb_fp32_tiles.txt | cut -c1-900; cat $O/runner_*.json | cut -c1-300
