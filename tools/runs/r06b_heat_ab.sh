#!/bin/bash
# round 6b: the GPU tests that touch the heat-map finishing kernels, then the A/B of configs[4] (heat map consumed on the GPU, and delivered
# to the host) against variants/heat_old.so (the library with the previous heatmap.hip, tools/build_variant.sh).  Through gpurun, repo root.
mkdir -p gpurun_out/heat1
timeout 1200 python -m pytest tests -x -q -m gpu -k "heat or sink or distogram or configs4_as_stated or 8k_pq_full" --durations=8 > gpurun_out/heat1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/heat1/pytest.log
tail -14 gpurun_out/heat1/pytest.log
WORKLOAD=8k256pq BENCH_ARGS="--heatmap-sink device" bash tools/ab_bench.sh heat1 variants/heat_old.so > gpurun_out/heat1/ab.txt 2>&1
cat gpurun_out/heat1/ab.txt
