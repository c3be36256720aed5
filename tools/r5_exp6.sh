set -x
mkdir -p gpurun_out/r5f
python -m pytest tests -m gpu -x -q > gpurun_out/r5f/test_gpu.txt 2>&1
python bench.py > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err
python runner.py --test --samples 4096 --batch 32 > gpurun_out/r5f/runner_f32.json 2> gpurun_out/r5f/runner_f32.err
python runner.py --test --samples 8192 --batch 256 --precision bf16 > gpurun_out/r5f/runner_bf16.json 2> gpurun_out/r5f/runner_bf16.err
python bench.py --include-h2d --no-secondary --no-cpu-baseline > gpurun_out/r5f/bench_h2d.json 2> gpurun_out/r5f/bench_h2d.err
python bench.py --include-h2d --renders u8 --no-secondary --no-cpu-baseline > gpurun_out/r5f/bench_h2d_u8.json 2> gpurun_out/r5f/bench_h2d_u8.err
