#!/bin/bash
# round 6b, final build: the default bench line (counter traffic quoted: the stamp of profiles/traffic.json must match) and the lines of the
# workloads the heat-map kernels touch.  Through gpurun from the repo root -> gpurun_out/final/
O=gpurun_out/final; mkdir -p $O
timeout 900 python bench.py > $O/r06b_bench.json 2> $O/bench.err
: > $O/r06b_bench_other_workloads.jsonl
timeout 600 python bench.py --workload 8k256pq --heatmap-sink device --cpu-frames 0 --steps 3 --warmup 1 >> $O/r06b_bench_other_workloads.jsonl 2>> $O/bench.err
timeout 600 python bench.py --workload 8k256pq --cpu-frames 0 --steps 3 --warmup 1 >> $O/r06b_bench_other_workloads.jsonl 2>> $O/bench.err
timeout 600 python bench.py --workload fhd64 --cpu-frames 0 >> $O/r06b_bench_other_workloads.jsonl 2>> $O/bench.err
timeout 600 python bench.py --dtype u8 --cpu-frames 0 >> $O/r06b_bench_other_workloads.jsonl 2>> $O/bench.err
timeout 600 python bench.py --workload 4k256 --dtype u8 --cpu-frames 0 --steps 2 --warmup 1 >> $O/r06b_bench_other_workloads.jsonl 2>> $O/bench.err
timeout 600 python tools/heatmap_bench.py 4k 32 > $O/r06b_heatmap_bench.txt 2>&1
timeout 600 python tools/heatmap_bench.py 8k 24 >> $O/r06b_heatmap_bench.txt 2>&1
python - <<'PY'
import json
for p in ("gpurun_out/final/r06b_bench.json", "gpurun_out/final/r06b_bench_other_workloads.jsonl"):
    for l in open(p):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["config"]["workload"][:60], d["ms_per_step"], d["value"], (d.get("roofline") or {}).get("bound"), (d.get("roofline") or {}).get("traffic"))
PY
grep -v amdgpu.ids gpurun_out/final/r06b_heatmap_bench.txt
