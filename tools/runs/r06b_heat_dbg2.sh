#!/bin/bash
# round 6b: bisect of the stage at which the two heat-map finishing kernels part (debug variants of heatmap.hip)
for n in 1 2 3 4 5 6 7; do CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/variants/heat_dbg$n.so python tools/runs/r06b_heat_dbg2.py 2>&1 | grep -v amdgpu.ids; done
