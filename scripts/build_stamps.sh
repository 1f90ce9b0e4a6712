#!/bin/bash
# the product build with phase stamps in the one-launch dual/price kernel (scripts/dual_stamps.py): only kernels_consensus.hip differs
# usage: scripts/build_stamps.sh [output .so]   (after scripts/fastbuild.sh: the other objects come from build/obj)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/build/obj
OUT=${1:-$ROOT/scripts/tmp/libdopf_dstamps.so}
mkdir -p $ROOT/scripts/tmp
(cd $ROOT/decentralopf.jl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-value -Wno-unused-result -DDOPF_DUAL_STAMPS kernels_consensus.hip -o $O/kernels_consensus_stamps.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $O/dopf_api.o $O/dopf_comm.o $O/dopf_central.o $O/kernels_central.o $O/kernels_agents.o $O/kernels_consensus_stamps.o -ldl -lpthread
ls -la $OUT
