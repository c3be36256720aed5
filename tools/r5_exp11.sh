set -x
mkdir -p gpurun_out/r5k
timeout 600 python tools/r5_d3test.py > gpurun_out/r5k/d3test.txt 2>&1
timeout 600 python tools/layer_bench.py --algo 2 --layers d2,d3 --tiles=-1,6 --rounds 5 > gpurun_out/r5k/lb.txt 2>&1
