#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -x -q -k "wsplit or each_layer or every_tile or stage_by_stage or batch32 or golden or handoff" > $O/pytest_wsplit.txt 2>&1; echo "rc=$?" >> $O/pytest_wsplit.txt
python tools/layer_bench.py --layers e3,e5,v2,v4 --tiles 0,1,2,3,7 --variants 1 --rounds 5 > $O/lb_plain.txt 2>&1
python tools/layer_bench.py --layers e3,e5,v2,v4 --tiles 0,1,2,3,5,7 --variants 1,4 --rounds 5 --wsplit > $O/lb_wsplit.txt 2>&1
python tools/layer_bench.py --layers v4 --tiles 3,7,6 --variants 0 --ksplits 1,2,4,8 --rounds 5 --wsplit > $O/lb_wsplit_v4.txt 2>&1
python tools/layer_bench.py --layers e2,e4,v1,v3 --tiles -1 --rounds 5 > $O/lb_producers.txt 2>&1
for i in 1 2; do
S3R_WSPLIT=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_plain_$i.json 2> $O/bench_plain_$i.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_wsplit_$i.json 2> $O/bench_wsplit_$i.err
done
for r in u8 f32; do
  python runner.py --test --precision bf16 --batch 256 --samples 3072 --renders $r > $O/runner_bf16_$r.json 2> $O/runner_bf16_$r.err
  python runner.py --test --precision fp32 --batch 32 --samples 1024 --renders $r > $O/runner_fp32_$r.json 2> $O/runner_fp32_$r.err
done
tail -3 $O/pytest_wsplit.txt; grep -v BEST $O/lb_plain.txt $O/lb_wsplit.txt $O/lb_wsplit_v4.txt; cat $O/runner_*.json
