#!/bin/bash
# GPU box: the lean-body fuzzer, then config2 (driver command + default) and config4 with the lean body and with the general one
# usage: scripts/ab_lean.sh <tag> [fuzz cases]
cd $GRAFT_REPO_ROOT
T=$1; N=${2:-60}
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 400 python scripts/fuzz_lean.py $N 7 > $O/fl_$T.txt 2>&1; tail -1 $O/fl_$T.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-side > $O/${T}_driver.json 2> $O/${T}_driver.err
python bench.py --no-cpu-baseline --no-also --no-side > $O/${T}_default.json 2> $O/${T}_default.err
python bench.py --no-cpu-baseline --no-also --no-side --flags 16384 > $O/${T}_default_gen.json 2> $O/${T}_default_gen.err
python bench.py --workload config4 --no-cpu-baseline --no-also --no-side > $O/${T}_c4.json 2> $O/${T}_c4.err
python bench.py --workload config4 --no-cpu-baseline --no-also --no-side --flags 16384 > $O/${T}_c4_gen.json 2> $O/${T}_c4_gen.err
python - <<PY
import json
for f in ("driver","default","default_gen","c4","c4_gen"):
    try:
        d=json.loads(open("$O/${T}_%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.ljust(12), "ms/step %.4f frac %.3f kernel_ms %.4f steady %.4f"%(d["ms_per_step"], r["frac"], r["kernel_ms"], r.get("steady_state",{}).get("kernel_ms",0)), {k: round(v,4) for k,v in d["kernels_ms"].items() if k in ("gen_ms","sto_ms")}, d.get("solver_failures"), d.get("errors"))
    except Exception as e: print(f, "failed", e)
PY
