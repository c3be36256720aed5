set -x
mkdir -p gpurun_out/r5h
bash tools/ab_bench.sh gpurun_out/r5h/ab --algo 2 --layers d1,d2,d3 --tiles=-1 --rounds 5 > gpurun_out/r5h/ab.txt 2>&1
for i in 1 2 3; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5h/bench_new$i.json 2> gpurun_out/r5h/bench_new$i.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5h/bench_base$i.json 2> gpurun_out/r5h/bench_base$i.err
done
