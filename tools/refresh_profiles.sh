#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#   tools/refresh_profiles.sh r02
# Writes everything under gpurun_out/refresh/ (merged back by gpurun); copy the *.txt/*.json into profiles/ and run
# tools/make_traffic_json.py <tag>.
set -u
# the library that is measured must be the one these sources build (a variant experiment can leave an older .so in the tree: round 6)
# (by modification time: the objects of the build do not travel to the GPU box, so `make -q` cannot answer there)
NEWER=$(find colorvideovdp_amd/csrc -maxdepth 1 \( -name "*.hip" -o -name "*.h" -o -name "*.cpp" \) -newer colorvideovdp_amd/libcvvdp_hip.so)
[ -z "$NEWER" ] || { echo "colorvideovdp_amd/libcvvdp_hip.so is older than its sources ($NEWER): run make first" >&2; exit 1; }
TAG=${1:-r05}
R=$(pwd)
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --gen gpu"
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
python -c "import bench, json; print(json.dumps(bench.code_stamp()))" > $OUT/${TAG}_stamp.json     # what these counters were measured on
# 2b. SQ counters of the three dominant kernels
tools/sq_counters.sh > $OUT/${TAG}_pmc_sq_counters_body.txt 2> $OUT/sq.err
# 3. bench lines: the default line (BASELINE metric clip, reference-checked), the other configurations, other input formats
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
: > $OUT/${TAG}_bench_other_workloads.jsonl
timeout 900 python bench.py --workload fhd64 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --workload 4k256 --cpu-frames 0 --steps 2 --warmup 1 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --workload 4k256 --dtype u8 --cpu-frames 0 --steps 2 --warmup 1 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --workload 4k1024 --cpu-frames 0 --steps 2 --warmup 1 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --workload 8k256pq --cpu-frames 0 --steps 2 --warmup 1 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --workload 8k256pq --heatmap-sink device --cpu-frames 0 --steps 2 --warmup 1 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --dtype u8 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --dtype yuv420p8 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --dtype yuv420p10 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
# 120 fps: 31-tap temporal filters (k_fir_rot<., 31>, 30 halo frames); kernel_roofline.temporal_fir of this line is its roofline figure
timeout 900 python bench.py --fps 120 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
timeout 900 python bench.py --fps 120 --dtype u8 --cpu-frames 0 >> $OUT/${TAG}_bench_other_workloads.jsonl 2>> $OUT/bench.err
# 3b. the multi-rank path on this one GPU: 2 ranks over gloo sharing the device (shard plan, halo frames, gather, rank-0 line),
#     five runs per sharded workload, 120 s limit each: every run must print its JSON line (tests/test_bench_multirank.py)
: > $OUT/${TAG}_two_ranks_one_gpu_gloo.log
for w in 4k64 4k1024; do
  for i in 1 2 3 4 5; do
    echo "== $w run $i  $(date +%T)" >> $OUT/${TAG}_two_ranks_one_gpu_gloo.log
    CVVDP_BENCH_BACKEND=gloo CVVDP_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port $((29611 + i)) bench.py --gpus 2 --workload $w --steps 2 --warmup 1 --cpu-frames 0 $([ $w = 4k1024 ] && echo --frames 256) \
      2>&1 | grep -v "amdgpu.ids" >> $OUT/${TAG}_two_ranks_one_gpu_gloo.log
    echo "== exit ${PIPESTATUS[0]}" >> $OUT/${TAG}_two_ranks_one_gpu_gloo.log
  done
done
# 3c. heat-map path (whole-clip tensor), shard-halo cost, other shapes
timeout 900 python tools/heatmap_bench.py 4k 32 > $OUT/${TAG}_heatmap_bench.txt 2>&1
timeout 900 python tools/heatmap_bench.py 8k 24 >> $OUT/${TAG}_heatmap_bench.txt 2>&1
( timeout 600 python tools/features_bench.py 3840x2160; timeout 600 python tools/features_bench.py 1920x1080 ) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_features_bench.txt
timeout 600 python tools/shard_halo_bench.py > $OUT/${TAG}_shard_halo_bench.txt 2>&1
timeout 600 tools/shape_bench.sh 2560x1440 1920x1080 1366x768 1360x768 854x480 848x480 > $OUT/${TAG}_shape_bench.txt 2>&1
# 3d. the two wave layouts of the fused band kernel side by side, shader clock / socket power under the step
timeout 300 python tools/ab_layout.py 4k64 10 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_ab_layout.txt
timeout 120 python tools/clock_power_probe.py 4 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_clock_power_probe.txt
# 4. GPU test log
timeout 1800 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1
tail -3 $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_bench.json
