#!/bin/bash
# round 6b: the raw heat map on the row-tile walk with the hardware log2 / exp2 pair: tests that touch it, then the whole-clip heat-map bench
mkdir -p gpurun_out/raw
timeout 900 python -m pytest tests -x -q -m gpu -k "heat or sink or raw or golden" > gpurun_out/raw/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/raw/pytest.log
tail -4 gpurun_out/raw/pytest.log
timeout 600 python tools/heatmap_bench.py 4k 32 2>&1 | grep -v amdgpu.ids > gpurun_out/raw/heatmap_bench.txt
timeout 600 python tools/heatmap_bench.py 8k 24 2>&1 | grep -v amdgpu.ids >> gpurun_out/raw/heatmap_bench.txt
cat gpurun_out/raw/heatmap_bench.txt
