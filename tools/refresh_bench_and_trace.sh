#!/bin/bash
# The two files that have to come from the same box (bench.py's event timing of the dominant kernel against rocprofv3's):
#   tools/refresh_bench_and_trace.sh r02      -> gpurun_out/refresh/<tag>_kernel_trace.txt, <tag>_bench.json
set -u
# the library that is measured must be the one these sources build (a variant experiment can leave an older .so in the tree: round 6)
# (by modification time: the objects of the build do not travel to the GPU box, so `make -q` cannot answer there)
NEWER=$(find colorvideovdp_amd/csrc -maxdepth 1 \( -name "*.hip" -o -name "*.h" -o -name "*.cpp" \) -newer colorvideovdp_amd/libcvvdp_hip.so)
[ -z "$NEWER" ] || { echo "colorvideovdp_amd/libcvvdp_hip.so is older than its sources ($NEWER): run make first" >&2; exit 1; }
TAG=${1:-r05}
R=$(pwd)
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --gen gpu"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- bash -c "cd $R && $BENCH" > $OUT/kt.log 2>&1 )
python $R/tools/rocpd_summary.py "$(find $OUT/kt -name '*.db' | head -1)" > $OUT/${TAG}_kernel_trace.txt
rm -rf $OUT/kt
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
grep -E "k_band4s(_edge)?\(" $OUT/${TAG}_kernel_trace.txt | grep -E " (5376|768) " | tail -2
python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
