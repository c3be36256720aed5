#!/bin/bash
# per-kernel times of the 2D two-axis form on e6 / e7 (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r4k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for t in ${FORMS:-4 5}; do
  rm -rf /tmp/prof_t$t
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t$t -- python3 $R/tools/layer_bench.py --algo 2 --batch 32 --layers ${LAYERS:-e6,e7} --tiles=$t --rounds 3 > $out/prof_t$t.log 2>&1
  f=$(find /tmp/prof_t$t -name "*kernel_stats.csv" | head -1)
  echo "== form $t ($f)"; head -12 "$f" | cut -c1-220
  [ -n "$f" ] && cp "$f" $out/plane_t${t}_kernel_stats.csv
done
