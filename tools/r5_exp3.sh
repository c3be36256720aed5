set -x
mkdir -p gpurun_out/r5c
export S3R_LIB=tools/alt/abl.so
for l in v1 e7 e2 d3 d2 v3; do
  S3R_ABL=7 python tools/timeline.py --wino --layer $l 2>&1 | grep -v amdgpu >> gpurun_out/r5c/timeline.txt
done
S3R_ABL=7 python tools/timeline.py --layer e3 2>&1 | grep -v amdgpu >> gpurun_out/r5c/timeline.txt
S3R_ABL=7 python tools/timeline.py --layer v2 2>&1 | grep -v amdgpu >> gpurun_out/r5c/timeline.txt
for a in 0 1 3; do
  S3R_ABL=$a python tools/layer_bench.py --algo 2 --layers e2,e7,v1,v3,d2,d3 --tiles=-1 --rounds 5 2>&1 | grep -v "BEST\|amdgpu\|^!!" | sed "s/^/ABL=$a /" >> gpurun_out/r5c/abl.txt
done
