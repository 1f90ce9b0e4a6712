#!/bin/bash
# product library + the counters build used by scripts/stats_*.py + the product build with block start/end stamps (stats_blocks.py)
# + the product build with phase stamps in the one-launch dual/price kernel (dual_stamps.py)
set -e
cd "$(dirname "$0")/../decentralopf.jl_amd/csrc"
mkdir -p ../../scripts/tmp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $F -o libdopf_hip.so dopf_api.hip dopf_comm.hip dopf_central.hip kernels_central.hip kernels_agents.hip kernels_consensus.hip -ldl -lpthread &
/opt/rocm/bin/hipcc $F -DDOPF_STATS -o ../../scripts/tmp/libdopf_stats.so dopf_api.hip dopf_comm.hip dopf_central.hip kernels_central.hip kernels_agents.hip kernels_consensus.hip -ldl -lpthread &
/opt/rocm/bin/hipcc $F -DDOPF_BLOCK_STAMPS -o ../../scripts/tmp/libdopf_stamps.so dopf_api.hip dopf_comm.hip dopf_central.hip kernels_central.hip kernels_agents.hip kernels_consensus.hip -ldl -lpthread &
/opt/rocm/bin/hipcc $F -DDOPF_DUAL_STAMPS -o ../../scripts/tmp/libdopf_dstamps.so dopf_api.hip dopf_comm.hip dopf_central.hip kernels_central.hip kernels_agents.hip kernels_consensus.hip -ldl -lpthread &
wait
