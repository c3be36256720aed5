set -x
mkdir -p gpurun_out/r5j
python -m pytest tests/test_wino_gpu.py tests/test_quantization_gpu.py -x -q > gpurun_out/r5j/test_wino.txt 2>&1
python -m pytest tests/test_parity_gpu.py -x -q -k "not bench" > gpurun_out/r5j/test_parity.txt 2>&1
for i in 1 2 3; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5j/bench_new$i.json 2> gpurun_out/r5j/bench_new$i.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5j/bench_base$i.json 2> gpurun_out/r5j/bench_base$i.err
done
for b in 1 4 8; do python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5j/bench_new_b$b.json 2>/dev/null; S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5j/bench_base_b$b.json 2>/dev/null; done
