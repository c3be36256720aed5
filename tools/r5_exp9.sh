set -x
mkdir -p gpurun_out/r5i
python -m pytest tests/test_wino_gpu.py -x -q > gpurun_out/r5i/test_wino.txt 2>&1
bash tools/ab_bench.sh gpurun_out/r5i/ab --algo 2 --layers e6,e7,v1 --tiles=-1 --rounds 5 > gpurun_out/r5i/ab.txt 2>&1
for i in 1 2 3; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5i/bench_new$i.json 2> gpurun_out/r5i/bench_new$i.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5i/bench_base$i.json 2> gpurun_out/r5i/bench_base$i.err
done
