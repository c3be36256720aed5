set -x
mkdir -p gpurun_out/r5o
timeout 300 python tools/r5_d3test.py 2>&1 | grep -v amdgpu > gpurun_out/r5o/d3test.txt
timeout 600 python tools/layer_bench.py --algo 2 --layers d2,d3 --tiles=0,2,6 --rounds 5 2>&1 | grep -v amdgpu > gpurun_out/r5o/lb.txt
for i in 1 2; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5o/bench_a$i.json 2> gpurun_out/r5o/bench_a$i.err
S3R_ALGO_d2=2 S3R_TILE_d2=6 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5o/bench_b$i.json 2> gpurun_out/r5o/bench_b$i.err
done
