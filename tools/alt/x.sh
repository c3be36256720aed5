mkdir -p gpurun_out/r5ac
for v in base nst2 nst4; do S3R_LIB=$PWD/tools/alt/$v.so python tools/alt/hash.py > gpurun_out/r5ac/hash_$v.log 2>&1; done
for i in 1 2 3; do for v in base nst2 nst4; do
S3R_LIB=$PWD/tools/alt/$v.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5ac/$v$i.json 2> gpurun_out/r5ac/$v$i.err
done; done
