#!/bin/bash
# A/B two builds of libs3r_hip.so on ONE device at MODEL level: tools/alt/base.so vs the in-tree build, alternating bench.py runs
# (headline only).  Prints ms per step and the per-layer table's rows named in $LAYERS for each run.
#   bash tools/ab_model.sh <outdir> [bench.py args...]        LAYERS="d3 v1" bash tools/ab_model.sh gpurun_out/ab
OUT=$1; shift
mkdir -p $OUT
for i in 1 2 3; do
  S3R_LIB=$PWD/tools/alt/base.so python bench.py --no-secondary --no-cpu-baseline "$@" > $OUT/base$i.json 2> $OUT/base$i.err
  python bench.py --no-secondary --no-cpu-baseline "$@" > $OUT/new$i.json 2> $OUT/new$i.err
done
python - "$OUT" "${LAYERS:-}" <<'PY'
import json, sys, glob
out, layers = sys.argv[1], sys.argv[2].split()
for kind in ("base", "new"):
    rows = [json.load(open(f)) for f in sorted(glob.glob(f"{out}/{kind}*.json"))]
    ms = [r["ms_per_step"] for r in rows]
    line = f"{kind:5s} ms/step {' '.join(f'{m:.4f}' for m in ms)}  min {min(ms):.4f}"
    for l in layers:
        v = [r["roofline"]["layers"][l]["ms"] for r in rows if r.get("roofline")]
        line += f"  {l} {min(v):.4f}"
    fam = [r["roofline"]["kernel_ms_per_step"] for r in rows if r.get("roofline")]
    print(line + f"  family {min(fam):.4f}")
PY
