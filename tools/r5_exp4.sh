set -x
mkdir -p gpurun_out/r5d
python -m pytest tests/test_wino_gpu.py -x -q > gpurun_out/r5d/test_wino.txt 2>&1
python -m pytest tests/test_parity_gpu.py -x -q > gpurun_out/r5d/test_parity.txt 2>&1
bash tools/ab_bench.sh gpurun_out/r5d/ab --algo 2 --layers e2,e4,e6,e7,v1,v3,v5,v6,d1 --tiles=-1 --rounds 5 > gpurun_out/r5d/ab.txt 2>&1
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5d/bench_new.json 2> gpurun_out/r5d/bench_new.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5d/bench_base.json 2> gpurun_out/r5d/bench_base.err
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5d/bench_new2.json 2> gpurun_out/r5d/bench_new2.err
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5d/bench_base2.json 2> gpurun_out/r5d/bench_base2.err
