#!/bin/bash
# PMC counters for the storage / generator kernels (separate pass from any tracing, as the pool requires)
#   prof_pmc.sh <workload> [extra bench.py flags, e.g. "--flags 16" = separate generator / storage launches]
set -e
cd /tmp && export TMPDIR=/tmp
W=${1:-config4}
X=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$W
RAW=/tmp/dopf_pmc_$W          # raw counter dumps stay on the box (gpurun_out/ is copied back only below 64 MiB)
rm -rf $RAW; mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
ARGS="bench.py --workload $W --no-side --steps 400 --warmup 400 --timed-iters 2 $X"      # ~2 000 launches per kernel: the settled state dominates the means
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $RAW/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $RAW/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections, json
summary = {}
for p in ("p1","p2"):
    for f in glob.glob("$RAW/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            if any(x in k for x in ("k_sto", "k_gen", "k_reduce", "k_agents", "k_net_agents", "k_slack", "k_dual")):
                print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
                summary.setdefault(k.split("::")[-1], {}).update({c: v / cnt[(k, c)] for c, v in d.items()})
                summary[k.split("::")[-1]]["launches_averaged"] = max(cnt[(k, c)] for c in d)
json.dump(summary, open("$OUT/summary.json", "w"), indent=1)
PY
