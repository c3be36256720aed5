mkdir -p gpurun_out/r4f
timeout 900 python tools/layer_bench.py --algo 1 --batch 32 --layers e3,e5,v2,v4,v6,e8 --tiles=-1,0,1,2,3,6,7 --ksplits 0,1,2,4,8 --rounds 3 > gpurun_out/r4f/direct_tiles.log 2>&1
grep -v "^!!" gpurun_out/r4f/direct_tiles.log | grep BEST
