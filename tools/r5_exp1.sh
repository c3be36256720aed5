set -x
mkdir -p gpurun_out/r5a
python bench.py --no-secondary > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
python tools/layer_bench.py --algo 2 --layers d1,d2,d3 --tiles=-1,0,1,2 --rounds 5 > gpurun_out/r5a/lb_deconv.txt 2>&1
python tools/layer_bench.py --algo 2 --layers v1,v3,v5,e6,e7 --tiles=3,4,5 --rounds 5 > gpurun_out/r5a/lb_two.txt 2>&1
python tools/layer_bench.py --algo 2 --layers e2,e4 --tiles=-1,0,1,2 --rounds 5 > gpurun_out/r5a/lb_one.txt 2>&1
for b in 16 24 31 32 33 40 48 64; do python tools/layer_bench.py --algo 2 --layers v1,d3,e7 --tiles=-1 --rounds 3 --batch $b 2>&1 | grep -v BEST | grep -v TOTAL | sed "s/^/B=$b /" >> gpurun_out/r5a/lb_batch.txt; done
