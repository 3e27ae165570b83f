#!/bin/bash
# Instruction-cache counters of the band kernels for the in-tree library and for variant libraries (one rocprofv3 --pmc pass each, kernel
# trace only):  tools/icache_counters.sh [variant.so ...]   (through gpurun, from the repo root)
R=$(pwd); export TMPDIR=/tmp
run() { # tag, lib ("" = in-tree)
  rm -rf /tmp/ic_$1
  ( cd /tmp && env ${2:+CVVDP_DEV_KNOBS=1 CVVDP_LIB=$2} timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES \
      -d /tmp/ic_$1 -o ic -- bash -c "cd $R && python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-power-probe --gen gpu" > /tmp/ic_$1.log 2>&1 )
  echo "== $1"
  python $R/tools/rocpd_summary.py "$(find /tmp/ic_$1 -name '*.db' | head -1)" | grep -A400 "PMC counters" | grep -E "k_band4s|k_fir_rot<3, 17>" | awk '$0 ~ / (5376|6144|768|64800) /' | cut -c1-60,78-200
}
run intree ""
for v in "$@"; do run $(basename $v .so) $R/$v; done
