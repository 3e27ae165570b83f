#!/bin/bash
# A/B of development knobs on one box: tools/ab_env.sh <tag> "<bench args>" "VAR=val ..." ["VAR=val ..." ...]   (through gpurun, repo root)
# Runs bench.py on the development library (make dev: libcvvdp_hip_dev.so reads the knobs of kernels.h::dev_knob) once per setting.
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for setting in "$@"; do
  i=$((i+1))
  env CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/colorvideovdp_amd/libcvvdp_hip_dev.so $setting timeout 900 python bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-power-probe $ARGS > $OUT/$i.json 2> $OUT/$i.err
  python - "$OUT/$i.json" "$setting" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "| ms/step", d["ms_per_step"], "Mpix/s", d["value"], "jod", d["jod"], d.get("kernel_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
