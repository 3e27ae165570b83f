#!/bin/bash
# SQ counters of the level-0 band kernel for the in-tree library and for variants: tools/sq_band.sh [variant.so ...]
# (separate --pmc passes with the kernel trace only; run from the repo root through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for lib in "" "$@"; do
  name=${lib:-intree}; [ -n "$lib" ] && lib=$R/$lib
  echo "== $(basename $name .so)"
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
    i=$((i+1)); rm -rf /tmp/sqb$i
    ( cd $R && CVVDP_DEV_KNOBS=1 CVVDP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/sqb$i -o sq -- python bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile > /tmp/sqb$i.log 2>&1 )
    db=$(find /tmp/sqb$i -name "*.db" | head -1)
    python $R/tools/rocpd_summary.py $db | grep -E "${SQ_KERNEL:-k_band4s}" | awk -v wg="${SQ_WG:-5376}" '$0 ~ (" " wg " ")' | awk '{printf "%-26s %18.0f\n", $(NF-3), $NF}'
  done
done
