#!/bin/bash
# temporal stage at several frame rates, both sample formats, in-tree library against variants: tools/fps_sweep.sh <tag> [variant.so ...]
TAG=${1:-fps}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "" "$@"; do
  name=$(basename ${lib:-intree} .so)
  for dt in f32 u8; do
    for fps in 24 30 60 90 120; do
      CVVDP_DEV_KNOBS=1 CVVDP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --fps $fps --dtype $dt --steps 5 --warmup 2 --cpu-frames 0 > $OUT/${name}_${dt}_$fps.json 2>/dev/null
      python - "$OUT/${name}_${dt}_$fps.json" "$name $dt $fps" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "step", d["ms_per_step"], "jod", d["jod"], "fir", d["kernel_ms_per_step"]["temporal_fir"], "frac_8TBs", d["kernel_roofline"]["temporal_fir"]["frac_of_8TBs"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    done
  done
done
