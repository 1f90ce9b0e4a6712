#!/bin/bash
# experiment: number of streaming generator blocks in the fused launch (config2, settled state)
cd $GRAFT_REPO_ROOT
L=decentralopf.jl_amd/csrc/libdopf_hip.so
for nb in 192 199 256 320 400 512 768; do
  echo "genBlocks $nb: $(DOPF_GEN_BLOCKS=$nb timeout -k 10 120 python scripts/gu_sweep.py $L 1536 2>&1 | grep items)"
done
echo "no streaming: $(DOPF_NO_GEN_STREAM=1 timeout -k 10 120 python scripts/gu_sweep.py $L 1536 2>&1 | grep items)"
