mkdir -p gpurun_out/r5ag
python -m pytest tests/test_wino_gpu.py tests/test_parity_gpu.py tests/test_quantization_gpu.py -x -q -m gpu > gpurun_out/r5ag/test.txt 2>&1
for b in 1 2 4 32; do
python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5ag/auto_b$b.json 2> gpurun_out/r5ag/auto_b$b.err
done
