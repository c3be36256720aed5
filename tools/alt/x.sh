mkdir -p gpurun_out/r5z
for v in base exp1 exp2; do S3R_LIB=$PWD/tools/alt/$v.so python tools/alt/hash.py > gpurun_out/r5z/hash_$v.log 2>&1; done
for i in 1 2 3; do for v in base exp1 exp2; do
 S3R_LIB=$PWD/tools/alt/$v.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5z/$v$i.json 2> gpurun_out/r5z/$v$i.err
done; done
