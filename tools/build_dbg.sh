#!/bin/bash
# Debug / timing build of ONE source file next to the shipped objects:
#   tools/build_dbg.sh <tag> <file.hip> <extra flags...>     ->  tools/_dbg/libamtx_<tag>.so   (load it with AMTX_LIB_PATH)
# e.g. tools/build_dbg.sh convtiming conv.hip -DAMTX_CONV_TIMING
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; SRC=$2; shift 2
mkdir -p $R/tools/_dbg
python -c "from amt_tools_amd.build import build; build(verbose=False)"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function"
case $SRC in spec.hip|convg.hip|cqt_dec.hip) FL="$FL -fno-slp-vectorize";; esac
BASE=${SRC%.*}
/opt/rocm/bin/hipcc $FL "$@" -x hip -c $R/amt_tools_amd/csrc/$SRC -o $R/tools/_dbg/${BASE}_$TAG.o
OBJS=$(ls $R/amt_tools_amd/csrc/*.o | grep -v "/${BASE}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_dbg/libamtx_$TAG.so $OBJS $R/tools/_dbg/${BASE}_$TAG.o
echo $R/tools/_dbg/libamtx_$TAG.so
