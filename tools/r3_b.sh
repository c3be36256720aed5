#!/bin/bash
# round-3 GPU batch b: ingestion tests, eval-loop rates, bf16 MFMA-shape A/B inside the whole step
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests/test_ingest_soak_gpu.py -x -q > $O/pytest_ingest.txt 2>&1; echo "ingest rc=$?" >> $O/pytest_ingest.txt
python -m pytest tests/test_parity_gpu.py tests/test_bf16_gpu.py tests/test_runner_gpu.py -x -q -k "stem or dataset or errors or graph or disparity or runner or odd or point" > $O/pytest_sel.txt 2>&1; echo "sel rc=$?" >> $O/pytest_sel.txt
for r in u8 f32; do
  python runner.py --test --precision bf16 --batch 256 --samples 3072 --renders $r > $O/runner_bf16_$r.json 2> $O/runner_bf16_$r.err
  python runner.py --test --precision fp32 --batch 32 --samples 1024 --renders $r > $O/runner_fp32_$r.json 2> $O/runner_fp32_$r.err
done
for i in 1 2; do for sh in 32 16; do
  S3R_BF16_MFMA=$sh python bench.py --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_bf16_sh${sh}_$i.json 2> $O/bench_bf16_sh${sh}_$i.err
done; done
tail -3 $O/pytest_ingest.txt $O/pytest_sel.txt; cat $O/runner_*.json
