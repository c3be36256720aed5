mkdir -p gpurun_out/r5v
for w in 0 1; do S3R_WINO_FOLD=$w python tools/alt/hash.py > gpurun_out/r5v/hash$w.log 2>&1; done
for i in 1 2 3; do for w in 0 1; do
 S3R_WINO_FOLD=$w python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5v/b$w$i.json 2> gpurun_out/r5v/b$w$i.err
done; done
