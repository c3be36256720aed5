#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_ingest_soak_gpu.py -x -q -k "eval or u8_through" > $O/pytest_ingest.txt 2>&1; echo "rc=$?" >> $O/pytest_ingest.txt
python -m pytest tests/test_parity_gpu.py tests/test_runner_gpu.py -x -q -k "eval or dataset or runner or prefetch" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt
for r in u8 f32; do
  python runner.py --test --precision bf16 --batch 256 --samples 3072 --renders $r > $O/runner_bf16_$r.json 2> $O/runner_bf16_$r.err
  python runner.py --test --precision fp32 --batch 32 --samples 1024 --renders $r > $O/runner_fp32_$r.json 2> $O/runner_fp32_$r.err
done
python runner.py --test --precision bf16 --batch 256 --samples 8192 --renders u8 > $O/runner_bf16_u8_8k.json 2> $O/runner_bf16_u8_8k.err
python runner.py --test --precision fp32 --batch 32 --samples 4096 --renders u8 > $O/runner_fp32_u8_4k.json 2> $O/runner_fp32_u8_4k.err
python bench.py --dtype bf16 --batch 256 --renders u8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_bf16_u8.json 2> $O/bench_bf16_u8.err
python bench.py --renders u8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_fp32_u8.json 2> $O/bench_fp32_u8.err
python bench.py --renders u8 --include-h2d --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_fp32_u8_h2d.json 2> $O/bench_fp32_u8_h2d.err
python bench.py --include-h2d --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_fp32_f32_h2d.json 2> $O/bench_fp32_f32_h2d.err
tail -n 3 $O/pytest_ingest.txt $O/pytest_sel.txt; cat $O/runner_*.json | cut -c1-300
