#!/bin/bash
# round 6b, final build: the remaining workload lines (configs[3] on one GPU, planar Y'CbCr, 120 fps) -> gpurun_out/final/r06b_more.jsonl
O=gpurun_out/final; mkdir -p $O; : > $O/r06b_more.jsonl
for a in "--workload 4k1024 --steps 2 --warmup 1" "--dtype yuv420p8" "--dtype yuv420p10" "--fps 120" "--fps 120 --dtype u8" "--workload 4k256 --steps 2 --warmup 1"; do
  timeout 600 python bench.py $a --cpu-frames 0 >> $O/r06b_more.jsonl 2>> $O/more.err
done
python - <<'PY'
import json
for l in open("gpurun_out/final/r06b_more.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["config"]["workload"][:75], d["ms_per_step"], d["value"])
PY
