set -x
mkdir -p gpurun_out/r5n
python -m pytest tests -m gpu -x -q > gpurun_out/r5n/test_gpu.txt 2>&1
for i in 1 2 3; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5n/bench_new$i.json 2> gpurun_out/r5n/bench_new$i.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5n/bench_base$i.json 2> gpurun_out/r5n/bench_base$i.err
done
