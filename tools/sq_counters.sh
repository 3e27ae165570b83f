#!/bin/bash
# SQ counters of the three dominant kernels (profiles/<tag>_pmc_sq_counters.txt body).  Separate --pmc passes with the
# kernel trace only (never combined with other trace domains).  Run from the repo root through gpurun:
#   tools/sq_counters.sh > gpurun_out/sq_summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$OLDPWD}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/sq$i -o sq$i -- python bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile > /tmp/sq$i.log 2>&1 )
  db=$(find /tmp/sq$i -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $db | grep -A400 "PMC counters" | grep -E "k_band4s\(|k_band4s_edge\(|k_band4f<4, (0|1)>|k_band4<4, false, false, false, false>|k_fir_rot<3, 17>|k_reduce2" | awk '$0 ~ / 5376 / || $0 ~ / 1152 / || $0 ~ / 768 / || $0 ~ / 384 / || $0 ~ / 256 / || $0 ~ / 6144 / || $0 ~ / 1536 / || $0 ~ / 512 / || $0 ~ / 64800 / || $0 ~ / 9216 /'
done
