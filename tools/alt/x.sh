mkdir -p gpurun_out/r5ab
for i in 1 2 3; do
S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ab/base$i.json 2> gpurun_out/r5ab/base$i.err
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ab/new$i.json 2> gpurun_out/r5ab/new$i.err
done
