#!/bin/bash
# Timing / A-B variants of the library: tools/build_variant.sh <name> <source.hip> [-DFLAG ...]
# Compiles ONE source of colorvideovdp_amd/csrc with extra flags and links it with the in-tree objects of the other sources into
# variants/<name>.so (git-ignored; travels to the GPU box).  Load it with CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/variants/<name>.so
# (tools/ab_bench.sh, tools/sq_band.sh).  Run `make` in csrc first.
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/colorvideovdp_amd/csrc
mkdir -p $R/variants/obj
BAND=""; case $SRC in band*) BAND="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $BAND "$@" -I$C -x hip -c $C/$SRC -o $R/variants/obj/$NAME.o
OBJS=$(ls $C/build/*.hip.o $C/build/*.cpp.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/$NAME.so $OBJS $R/variants/obj/$NAME.o
echo "variants/$NAME.so"
