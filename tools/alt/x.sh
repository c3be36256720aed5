mkdir -p gpurun_out/r5t
export S3R_LIB=$PWD/tools/alt/abl.so
python tools/layer_bench.py --layers e2,e4,e6,e7,v1,v3,v5,v6,d1,d2,d3 --algo 2 --tiles=-1 --rounds 3 > gpurun_out/r5t/rand_w.log 2>&1
python tools/layer_bench.py --layers e2,e4,e6,e7,v1,v3,v5,v6,d1,d2,d3 --algo 2 --tiles=-1 --rounds 3 --zeros > gpurun_out/r5t/zeros_w.log 2>&1
python tools/layer_bench.py --layers e3,e5,v2,v4 --algo 1 --tiles=-1 --rounds 3 > gpurun_out/r5t/rand_d.log 2>&1
python tools/layer_bench.py --layers e3,e5,v2,v4 --algo 1 --tiles=-1 --rounds 3 --zeros > gpurun_out/r5t/zeros_d.log 2>&1
