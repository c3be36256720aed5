set -x
mkdir -p gpurun_out/r5m
for b in 1 2 4 8 16; do
python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5m/two_b$b.json 2>/dev/null
S3R_ALGO_d3=2 S3R_TILE_d3=6 python bench.py --no-secondary --no-cpu-baseline --batch $b > gpurun_out/r5m/three_b$b.json 2>/dev/null
done
