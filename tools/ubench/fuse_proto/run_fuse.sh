#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms', d['ms_per_step'], 'jod', d['jod'], d['kernel_ms_per_step'])"; }
python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | show intree
CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/variants/fuse.so python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | show fuse_lib_knob_off
CVVDP_FUSE_PROTO=1 CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/variants/fuse.so python bench.py --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | show fuse_proto_on
CVVDP_FUSE_PROTO=1 CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/variants/fuse.so python bench.py --steps 10 --warmup 3 --cpu-frames 0 --workload 4k256 --dtype u8 --gen gpu 2>/dev/null | show fuse_proto_on_4k256u8
python bench.py --steps 10 --warmup 3 --cpu-frames 0 --workload 4k256 --dtype u8 --gen gpu 2>/dev/null | show intree_4k256u8
