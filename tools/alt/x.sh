mkdir -p gpurun_out/r5ah
python -m pytest tests/test_wino_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "cost_volume or form_invariant or stage_by_stage or batch32" > gpurun_out/r5ah/test.txt 2>&1
for b in 1 2 4 8 16 32; do
python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5ah/auto_b$b.json 2> gpurun_out/r5ah/auto_b$b.err
done
