#!/bin/bash
# round 6b: select-free range of the context image in the level-0 heat band kernels (kernels.h PositiveRange): heat-map tests, then the A/B of
# configs[4] against the library built before the change (variants/lib_before_range.so).  Through gpurun, repo root.
mkdir -p gpurun_out/range
timeout 1200 python -m pytest tests -x -q -m gpu -k "heat or sink or distogram or 8k_pq_full" > gpurun_out/range/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/range/pytest.log
tail -4 gpurun_out/range/pytest.log
WORKLOAD=8k256pq BENCH_ARGS="--heatmap-sink device" bash tools/ab_bench.sh range variants/lib_before_range.so > gpurun_out/range/ab.txt 2>&1
cat gpurun_out/range/ab.txt
