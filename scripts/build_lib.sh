#!/bin/bash
# builds csrc/libdopf_hip.so (or $1) the way __graft_entry__.build() does; extra hipcc flags in $DOPF_HIPCC_FLAGS
set -e
cd "$(dirname "$0")/../decentralopf.jl_amd/csrc"
OUT=${1:-libdopf_hip.so}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result $DOPF_HIPCC_FLAGS -o $OUT \
    dopf_api.hip dopf_comm.hip dopf_central.hip kernels_central.hip kernels_agents.hip kernels_consensus.hip -ldl -lpthread
