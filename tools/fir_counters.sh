#!/bin/bash
# SQ counters of the temporal kernel at 120 fps (31 taps) and 60 fps (17 taps): which resource holds k_fir_rot back.
# Separate --pmc passes with the kernel trace only.  Run from the repo root through gpurun:   tools/fir_counters.sh > gpurun_out/fir_counters.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$OLDPWD}
for fps in 120 60; do
  echo "== fps $fps (fp32 input, 4K x 64)"
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAVES SQ_INST_LEVEL_VMEM" \
             "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rm -rf /tmp/fc$fps$i
    ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/fc$fps$i -o fc -- python bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile --no-power-probe --gen gpu --fps $fps > /tmp/fc$fps$i.log 2>&1 )
    db=$(find /tmp/fc$fps$i -name "*.db" | head -1)
    if [ -z "$db" ]; then echo "   (pass $i failed: $(tail -1 /tmp/fc$fps$i.log | cut -c1-160))"; continue; fi
    python $R/tools/rocpd_summary.py $db | grep -A400 "PMC counters" | grep -E "k_fir_rot" | cut -c1-44,78-200
  done
done
