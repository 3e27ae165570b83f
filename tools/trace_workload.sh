#!/bin/bash
# rocprofv3 kernel trace of one bench workload: tools/trace_workload.sh <tag> "<bench args>"  -> gpurun_out/trace/<tag>_kernel_trace.txt
set -u
TAG=$1; ARGS=$2
R=$(pwd); OUT=$R/gpurun_out/trace; mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$TAG -o kt -- bash -c "cd $R && python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-power-probe --gen gpu $ARGS" > $OUT/kt_$TAG.log 2>&1 )
python $R/tools/rocpd_summary.py "$(find $OUT/kt_$TAG -name '*.db' | head -1)" > $OUT/${TAG}_kernel_trace.txt
rm -rf $OUT/kt_$TAG
grep "cvvdp::" $OUT/${TAG}_kernel_trace.txt | head -30 | cut -c1-200
