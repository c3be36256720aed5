#!/bin/bash
# the kernels of one B = 32 fp32 step, in order (tools/step_trace.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $* > /tmp/tr.log 2>&1
python3 $R/tools/step_trace.py /tmp/tr | tee $R/gpurun_out/step_trace.txt
