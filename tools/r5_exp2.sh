set -x
mkdir -p gpurun_out/r5b
(cd /tmp && rocprofv3 -L > /root/repo/gpurun_out/r5b/counters.txt 2>&1)
for a in 0 3 4 5; do
  S3R_LIB=tools/alt/abl.so S3R_ABL=$a python tools/layer_bench.py --algo 2 --layers e2,e7,v1,v3,d2,d3 --tiles=-1 --rounds 5 2>&1 | grep -v "BEST\|amdgpu\|^!!" | sed "s/^/ABL=$a /" >> gpurun_out/r5b/abl.txt
done
python -m pytest tests/test_wino_gpu.py -k "edge_limit" -x -q > gpurun_out/r5b/test_edge.txt 2>&1
