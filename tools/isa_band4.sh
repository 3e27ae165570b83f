#!/bin/bash
# Compile band4.hip to gfx950 assembly (/tmp/isa/band4_new.s) and print the register budget of its four kernels.
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -x hip --cuda-device-only \
  -S /root/repo/colorvideovdp_amd/csrc/band4.hip -o /tmp/isa/band4_new.s 2>&1 | grep -v "hip-link"
grep "\.vgpr_count\|\.sgpr_spill\|vgpr_spill\|\.name:.*band4" /tmp/isa/band4_new.s | paste - - - - | sed 's/ \+/ /g'
awk '/^_ZN5cvvdp7k_band4ILi4ELb0ELb0ELb0EEEvNS_8BandArgsE:/,/\.end_amdhsa_kernel/' /tmp/isa/band4_new.s > /tmp/isa/k4n.s
