set -x
mkdir -p gpurun_out/r5final
python -m pytest tests -m gpu -x -q > gpurun_out/r5final/test_gpu.txt 2>&1
bash tools/profile_all.sh r05 > gpurun_out/r5final/profile_all.log 2>&1
python runner.py --test --samples 4096 --batch 32 > gpurun_out/r5final/runner_f32.json 2> gpurun_out/r5final/runner_f32.err
python runner.py --test --samples 8192 --batch 256 --precision bf16 > gpurun_out/r5final/runner_bf16.json 2> gpurun_out/r5final/runner_bf16.err
python runner.py --test --samples 4096 --batch 32 --renders f32 > gpurun_out/r5final/runner_f32_f32renders.json 2> /dev/null
python bench.py --include-h2d --no-secondary --no-cpu-baseline > gpurun_out/r5final/bench_h2d.json 2> /dev/null
python bench.py --include-h2d --renders u8 --no-secondary --no-cpu-baseline > gpurun_out/r5final/bench_h2d_u8.json 2> /dev/null
for b in 1 2 4 8 16 64 256; do python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5final/bench_b$b.json 2>/dev/null; done
