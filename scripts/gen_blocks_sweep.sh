#!/bin/bash
# experiment: number of streaming generator blocks in the fused launch (config2, settled state)
# (the DOPF_* tuning knobs exist only in a library built with -DDOPF_EXPERIMENTS: scripts/fastbuild.sh -DDOPF_EXPERIMENTS builds
#  every object with it when given as DOPF_HIPCC_FLAGS to scripts/build_lib.sh)
cd $GRAFT_REPO_ROOT
L=decentralopf.jl_amd/csrc/libdopf_hip.so
for nb in 192 199 256 320 400 512 768; do
  echo "genBlocks $nb: $(DOPF_GEN_BLOCKS=$nb timeout -k 10 120 python scripts/gu_sweep.py $L 1536 2>&1 | grep items)"
done
echo "no streaming: $(DOPF_NO_GEN_STREAM=1 timeout -k 10 120 python scripts/gu_sweep.py $L 1536 2>&1 | grep items)"
