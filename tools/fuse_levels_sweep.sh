#!/bin/bash
for wl in "4k64" "fhd64" "4k256 --dtype u8"; do
for F in 0 1 2 3 4 6; do
  CVVDP_FUSE_LEVELS=$F CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/colorvideovdp_amd/libcvvdp_hip_dev.so python bench.py --workload $wl --steps 8 --warmup 3 --cpu-frames 0 --gen gpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', 'F=$F', 'ms', d['ms_per_step'], d['kernel_ms_per_step'])"
done; done
