mkdir -p gpurun_out/r5aj
python -m pytest tests/test_wino_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r5aj/test.txt 2>&1
for i in 1 2 3; do python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5aj/new$i.json 2> gpurun_out/r5aj/new$i.err; done
python bench.py --no-secondary --no-cpu-baseline --batch 1 > gpurun_out/r5aj/b1.json 2> gpurun_out/r5aj/b1.err
