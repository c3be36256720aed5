set -x
mkdir -p gpurun_out/r5e
export S3R_LIB=tools/alt/abl.so
for rep in 1 2; do
for a in 0 6 3; do
  S3R_ABL=$a python tools/layer_bench.py --algo 2 --layers e2,e4,v1,v3,v5,d1,d2,d3 --tiles=-1 --rounds 5 2>&1 | grep -v "BEST\|amdgpu\|^!!" | sed "s/^/ABL=$a /" >> gpurun_out/r5e/abl.txt
done
done
