#!/bin/bash
# bench the in-tree library under several environment settings: tools/ab_env.sh <tag> "VAR=1 VAR2=2" "..." ...
TAG=${1:-abenv}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for e in "" "$@"; do
  i=$((i+1))
  env $e timeout 600 python bench.py --steps 10 --warmup 3 --cpu-frames 0 > $OUT/$i.json 2> $OUT/$i.err
  python - "$OUT/$i.json" "[$e]" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "jod", d["jod"], d.get("kernel_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
