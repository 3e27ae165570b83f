#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#   tools/refresh_profiles.sh r01
# Writes everything under gpurun_out/refresh/ (merged back by gpurun); copy the *.txt/*.json into profiles/.
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile"
summ() { python $R/tools/rocpd_summary.py "$(find $1 -name '*.db' | head -1)"; }
# 1. kernel trace (timing only)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- bash -c "cd $R && $BENCH" > $OUT/kt.log 2>&1 )
summ $OUT/kt > $OUT/${TAG}_kernel_trace.txt
# 2. HBM traffic counters, one pass each (never combined with other trace domains)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc -- bash -c "cd $R && $BENCH" > $OUT/pmc_$c.log 2>&1 )
  summ $OUT/pmc_$c > $OUT/${TAG}_pmc_$(echo $c | tr A-Z a-z).txt
done
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE     # the rocpd databases are large; keep the summaries
# 3. bench lines
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
: > $OUT/${TAG}_bench_other_workloads.jsonl
timeout 900 python bench.py --workload fhd64 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --workload 4k256 --cpu-frames 0 --steps 2 --warmup 1 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --dtype u8 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --dtype yuv420p8 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --dtype yuv420p10 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
# 3b. heat-map path (configs[4] shape per GPU) and the shard-halo cost
timeout 900 python tools/heatmap_bench.py 4k 32 > $OUT/${TAG}_heatmap_bench.txt 2>&1
timeout 900 python tools/heatmap_bench.py 8k 24 >> $OUT/${TAG}_heatmap_bench.txt 2>&1
timeout 600 python tools/shard_halo_bench.py > $OUT/${TAG}_shard_halo_bench.txt 2>&1
# 4. GPU test log
timeout 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1
tail -3 $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_bench.json
