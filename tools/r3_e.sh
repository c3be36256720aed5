#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
python tools/reg_debug.py > $O/reg_debug.txt 2>&1
python -m pytest tests/test_parity_gpu.py -x -q -k "wsplit or linear or stereo2point or each_layer or stage_by_stage or batch32" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt
python tools/point_bench.py > $O/point_bench_wgk.txt 2>&1
S3R_LINEAR_WGK=0 python tools/point_bench.py > $O/point_bench_old.txt 2>&1
for i in 1 2 3; do
S3R_WSPLIT=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_plain_$i.json 2> $O/bench_plain_$i.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_wsplit_$i.json 2> $O/bench_wsplit_$i.err
done
python bench.py --variant point --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_point.json 2> $O/bench_point.err
cat $O/reg_debug.txt; tail -3 $O/pytest_sel.txt; cat $O/point_bench_wgk.txt $O/point_bench_old.txt
