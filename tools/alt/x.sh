mkdir -p gpurun_out/r5ai
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ai/def.json 2> gpurun_out/r5ai/def.err
S3R_WINO2_MAX_EDGE=56 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ai/edge56.json 2> gpurun_out/r5ai/edge56.err
S3R_DWINO_MAT=0 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ai/mat0.json 2> gpurun_out/r5ai/mat0.err
S3R_DWINO_MAT=1 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ai/mat1.json 2> gpurun_out/r5ai/mat1.err
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ai/def2.json 2> gpurun_out/r5ai/def2.err
