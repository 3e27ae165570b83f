#!/bin/bash
# round 6b: randomised sweep of the final build against the oracle (heat-map cases run the rewritten finishing kernels) -> gpurun_out/fuzz/
mkdir -p gpurun_out/fuzz
timeout 900 python tools/fuzz_shapes.py 160 61 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz/seed61.txt
timeout 900 python tools/fuzz_shapes.py 160 62 fuse_heat 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz/seed62_fuse_heat.txt
for f in gpurun_out/fuzz/*.txt; do echo "$f: $(grep -c '^ok' $f) ok, $(grep -c '^BAD' $f) BAD, heat-map cases $(grep -c 'heat frac' $f)"; grep '^BAD' $f | cut -c1-220; done
