#!/bin/bash
# quick HBM-traffic check of the cvvdp kernels: FETCH_SIZE and WRITE_SIZE passes (kernel trace + one counter each)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
LIBV=${1:+$R/$1}
for c in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf /tmp/pq_$c
  ( cd $R && CVVDP_DEV_KNOBS=1 CVVDP_LIB=$LIBV timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pq_$c -o pmc -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile > /tmp/pq_$c.log 2>&1 )
  python $R/tools/rocpd_summary.py "$(find /tmp/pq_$c -name '*.db' | head -1)" | grep -A40 "PMC counters" | grep -E "$c" | head -12
done
