#!/bin/bash
# every profile of a round in one go (GPU box): kernel traces + FETCH/WRITE passes of the default command per workload, the trace of
# the driver's command, SQ counters of the agent kernels. Summaries under gpurun_out/prof_* and gpurun_out/pmc_* (raw dumps stay in /tmp).
cd $GRAFT_REPO_ROOT
for w in config2 config4 config3-share config4x2 config1; do
  bash scripts/profile.sh $w > gpurun_out/prof_$w.txt 2>&1 && echo "profile $w done" || echo "profile $w FAILED"
done
bash scripts/profile.sh config2 driver > gpurun_out/prof_config2_driver.txt 2>&1 && echo "profile config2 driver done"
for w in config4 config2 config3-share; do
  bash scripts/prof_pmc.sh $w > gpurun_out/pmc_$w.txt 2>&1 && echo "pmc $w done" || echo "pmc $w FAILED"
done
