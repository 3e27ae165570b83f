#!/bin/bash
# A/B of library variants on one box: tools/ab_bench.sh <tag> [variant.so ...]   (run through gpurun from the repo root)
# Prints the per-kernel milliseconds of the bench step (WORKLOAD, default 4k64) for the in-tree library and for each variant.
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
WL=${WORKLOAD:-4k64}
run() { # name, lib
  CVVDP_DEV_KNOBS=1 CVVDP_LIB=$2 timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-frames 0 --no-power-probe $BENCH_ARGS > $OUT/$1.json 2> $OUT/$1.err
  python - "$OUT/$1.json" "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "Mpix/s", d["value"], "jod", d["jod"], d.get("kernel_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run intree ""
for v in "$@"; do run $(basename $v .so) $PWD/$v; done
run intree2 ""
