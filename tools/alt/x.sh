mkdir -p gpurun_out/r5w
for s in 4 6 8; do python tools/two_stream_exp.py --batch 32 --streams $s --stagger-ms 0.9 --steps 30 > gpurun_out/r5w/n$s.log 2>&1; done
