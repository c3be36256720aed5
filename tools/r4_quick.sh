#!/bin/bash
# round-4 driver: Winograd tests, then the kernels of one step in order
out=gpurun_out/r4k
mkdir -p $out
timeout 1500 python -m pytest tests/test_wino_gpu.py -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?"
tail -4 $out/pytest.log
bash tools/r4_steptrace.sh $*
