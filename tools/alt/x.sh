mkdir -p gpurun_out/r5u
for w in 0 1 2; do S3R_WINO_WG8=$w python tools/alt/hash.py > gpurun_out/r5u/hash$w.log 2>&1; done
for i in 1 2 3; do for w in 0 1 2; do
 S3R_WINO_WG8=$w python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5u/b$w$i.json 2> gpurun_out/r5u/b$w$i.err
done; done
