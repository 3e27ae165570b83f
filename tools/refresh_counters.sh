#!/bin/bash
# The part of tools/refresh_profiles.sh that has to be redone whenever a kernel source changes (run from the repo root through gpurun):
# kernel trace, HBM traffic counters (separate --pmc passes), SQ counters, the stamp of the sources they were measured on, and the
# default bench line from the same box.    tools/refresh_counters.sh r03   -> gpurun_out/refresh/<tag>_*
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
summ() { python $R/tools/rocpd_summary.py "$(find $1 -name '*.db' | head -1)"; }
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- bash -c "cd $R && $BENCH" > $OUT/kt.log 2>&1 )
summ $OUT/kt > $OUT/${TAG}_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc -- bash -c "cd $R && $BENCH" > $OUT/pmc_$c.log 2>&1 )
  summ $OUT/pmc_$c > $OUT/${TAG}_pmc_$(echo $c | tr A-Z a-z).txt
done
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python -c "import bench, json; print(json.dumps(bench.code_stamp()))" > $OUT/${TAG}_stamp.json
tools/sq_counters.sh > $OUT/${TAG}_pmc_sq_counters_body.txt 2> $OUT/sq.err
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
grep -E "k_band4s(_edge)?\(" $OUT/${TAG}_kernel_trace.txt | grep -E " (5376|768) " | tail -2
cat $OUT/${TAG}_bench.json | cut -c1-400
