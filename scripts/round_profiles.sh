#!/bin/bash
# GPU box: every rocprofv3 pass of the round (kernel traces + windows, FETCH/WRITE, SQ counters) and the two bench lines.
# Afterwards, on the host: python scripts/collect_profiles.py rNN config1 config2 config2_driver config4 config4x2 config3-share config3
#                          python scripts/collect_valu.py rNN config4 config2 config3-share
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
for w in config2 config1 config4 config4x2 config3-share config3; do
  scripts/profile.sh $w > gpurun_out/prof_$w.txt 2>&1 || echo "profile $w failed"
done
scripts/profile.sh config2 driver > gpurun_out/prof_config2_driver.txt 2>&1 || echo "profile config2 driver failed"
for w in config4 config2 config3-share; do
  scripts/prof_pmc.sh $w > gpurun_out/pmc_$w.txt 2>&1 || echo "pmc $w failed"
done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench/driver.json 2> gpurun_out/bench/driver.err
python bench.py > gpurun_out/bench/default.json 2> gpurun_out/bench/default.err
python - <<PY
import json
for f in ("driver","default"):
    d=json.loads(open("gpurun_out/bench/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "ms/step %.4f value %.3e frac %.3f kernel_ms %.4f steady %.4f whole %.3f"%(d["ms_per_step"], d["value"], r["frac"], r["kernel_ms"], r["steady_state"]["kernel_ms"], r["whole_iteration"]["frac"]), "errors", d.get("errors"))
    print("  also", [(a["workload"], round(a["ms_per_step"],4)) for a in d.get("also",[])])
    print("  ttr", d.get("time_to_1e-3_residual"))
PY
