set -x
mkdir -p gpurun_out/r5g
python -m pytest tests/test_wino_gpu.py -x -q > gpurun_out/r5g/test_wino.txt 2>&1
python -m pytest tests/test_parity_gpu.py -x -q -k "not bench" > gpurun_out/r5g/test_parity.txt 2>&1
bash tools/ab_bench.sh gpurun_out/r5g/ab --algo 2 --layers d1,d2,d3 --tiles=-1 --rounds 5 > gpurun_out/r5g/ab.txt 2>&1
for i in 1 2; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5g/bench_new$i.json 2> gpurun_out/r5g/bench_new$i.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5g/bench_base$i.json 2> gpurun_out/r5g/bench_base$i.err
done
