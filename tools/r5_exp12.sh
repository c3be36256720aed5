set -x
mkdir -p gpurun_out/r5l
timeout 600 python tools/layer_bench.py --algo 2 --layers d2,d3 --tiles=-1,6 --rounds 5 2>&1 | grep -v amdgpu > gpurun_out/r5l/lb.txt
for i in 1 2; do
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5l/bench_two$i.json 2> gpurun_out/r5l/bench_two$i.err
S3R_ALGO_d3=2 S3R_TILE_d3=6 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5l/bench_three$i.json 2> gpurun_out/r5l/bench_three$i.err
S3R_ALGO_d3=2 S3R_TILE_d3=6 S3R_ALGO_d2=2 S3R_TILE_d2=6 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5l/bench_three23_$i.json 2> gpurun_out/r5l/bench_three23_$i.err
done
