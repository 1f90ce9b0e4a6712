#!/bin/bash
# the driver's command and the default command, one line each, key numbers printed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $1 > gpurun_out/r3/b_driver.json 2> gpurun_out/r3/b_driver.err
python bench.py --no-cpu-baseline --no-also > gpurun_out/r3/b_default.json 2> gpurun_out/r3/b_default.err
python - <<PY
import json
for f in ("gpurun_out/r3/b_driver.json","gpurun_out/r3/b_default.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "value %.3e ms/step %.4f frac %.3f kernel_ms %.4f (events raw %.4f, overhead %.4f) steady %.4f whole-iteration frac %.3f" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_ms"], r["kernel_ms_events_raw"], r["event_overhead_ms"], r["steady_state"]["kernel_ms"], r["whole_iteration"]["frac"]))
    print("  kernels_ms", {k: round(v,4) for k,v in d["kernels_ms"].items()})
    print("  parts", {k: (round(v["kernel_ms"],4) if isinstance(v, dict) else v) for k,v in r.get("parts_as_separate_launches",{}).items()})
    t=d.get("time_to_1e-3_residual"); print("  ttr", t["seconds"], t["iterations"], t.get("relative_gap_to_central_lp"))
    print("  also", [(a["workload"], round(a["ms_per_step"],4), round(a.get("hbm_traffic_frac_of_peak",0),3)) for a in d.get("also",[])])
    print("  errors", d.get("errors"))
PY
