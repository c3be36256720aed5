#!/bin/bash
# A/B two builds of libs3r_hip.so on ONE device (device clocks differ by up to 12 % across the pool, so
# numbers from different gpurun calls do not compare): tools/alt/base.so vs the in-tree build, alternating.
#   bash tools/ab_bench.sh <outdir> [layer_bench args...]
# tools/alt/base.so is the baseline build (git-ignored): e.g.
#   git archive HEAD stereo-3d-reconstruction_amd/csrc include | tar x -C /tmp/base && make -C /tmp/base/stereo-3d-reconstruction_amd/csrc
#   mkdir -p tools/alt && cp /tmp/base/stereo-3d-reconstruction_amd/csrc/libs3r_hip.so tools/alt/base.so
# (two builds of IDENTICAL source differ by up to 4 % per layer with this method: treat < 5 % as noise)
OUT=$1; shift
mkdir -p $OUT
for i in 1 2 3; do
  S3R_LIB=$PWD/tools/alt/base.so python tools/layer_bench.py "$@" > $OUT/base$i.log 2>&1
  python tools/layer_bench.py "$@" > $OUT/new$i.log 2>&1
done
python - "$OUT" <<'PY'
import re, sys, glob
out = sys.argv[1]
def best(f):
    r = {}
    for l in open(f):
        m = re.match(r"(\w+)\s+BEST tile (\S+) vec \S+ ksplit (\S+): ([\d.]+) ms", l)
        if m: r[m.group(1)] = (float(m.group(4)), m.group(2), m.group(3))
    return r
b = [best(f) for f in sorted(glob.glob(out + "/base*.log"))]
n = [best(f) for f in sorted(glob.glob(out + "/new*.log"))]
tb = tn = 0.0
for k in b[0]:
    if not all(k in x for x in b + n): continue
    mb = min(x[k][0] for x in b); mn = min(x[k][0] for x in n)
    tb += mb; tn += mn
    print(f"{k:4s} base {mb:.4f} (t{b[0][k][1]} k{b[0][k][2]})  new {mn:.4f} (t{n[0][k][1]} k{n[0][k][2]})  new/base {mn / mb:.3f}")
print(f"SUM  base {tb:.4f}  new {tn:.4f}  new/base {tn / max(tb, 1e-9):.3f}")
PY
