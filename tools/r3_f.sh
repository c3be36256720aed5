#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests/test_bf16_gpu.py -x -q -k "rows_kernel" > $O/pytest_rows.txt 2>&1; echo "rc=$?" >> $O/pytest_rows.txt
python tools/layer_bench.py --dtype bf16 --batch 256 --layers e2 --tiles 22,40 --shapes 32,16 --rounds 7 > $O/lb_e2.txt 2>&1
python tools/layer_bench.py --dtype bf16 --batch 128 --layers e2 --tiles 22,40 --rounds 7 > $O/lb_e2_b128.txt 2>&1
python tools/layer_bench.py --dtype bf16 --batch 64 --layers e2 --tiles 22,40 --rounds 7 > $O/lb_e2_b64.txt 2>&1
for i in 1 2; do
S3R_ROWS=0 python bench.py --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_norows_$i.json 2> $O/bench_norows_$i.err
python bench.py --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_rows_$i.json 2> $O/bench_rows_$i.err
done
python -m pytest tests/test_bf16_gpu.py tests/test_ingest_soak_gpu.py -x -q > $O/pytest_bf16_ingest.txt 2>&1; echo "rc=$?" >> $O/pytest_bf16_ingest.txt
python -m pytest tests/test_parity_gpu.py -x -q -k "wsplit or linear or stereo2point or eval or dataset" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt
for r in u8 f32; do
  python runner.py --test --precision bf16 --batch 256 --samples 3072 --renders $r > $O/runner_bf16_$r.json 2> $O/runner_bf16_$r.err
  python runner.py --test --precision fp32 --batch 32 --samples 1024 --renders $r > $O/runner_fp32_$r.json 2> $O/runner_fp32_$r.err
done
tail -3 $O/pytest_rows.txt; grep -v BEST $O/lb_e2*.txt; tail -3 $O/pytest_bf16_ingest.txt $O/pytest_sel.txt; cat $O/runner_*.json | cut -c1-300
