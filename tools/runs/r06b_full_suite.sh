#!/bin/bash
# the whole GPU suite with its durations record (through gpurun, repo root) -> gpurun_out/suite/pytest_gpu.log
mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -q -m gpu --durations=30 > gpurun_out/suite/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/suite/pytest_gpu.log
tail -45 gpurun_out/suite/pytest_gpu.log
