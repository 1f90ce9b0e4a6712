#!/bin/bash
# one PMC pass (vector / scalar / LDS instruction counts per wave) of the x-update kernels of a workload
#   valu_quick.sh <workload> <tag> [extra bench.py flags]
set -e
cd /tmp && export TMPDIR=/tmp
W=${1:-config4}; TAG=${2:-x}; X=${3:-}
RAW=/tmp/dopf_vq_${W}_$TAG; rm -rf $RAW; mkdir -p $RAW $GRAFT_REPO_ROOT/gpurun_out/r4
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $RAW -- python3 bench.py --workload $W --no-side --no-cpu-baseline --no-also --steps 300 --warmup 300 --timed-iters 2 $X > gpurun_out/r4/vq_${W}_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$RAW/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-50:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if any(x in k for x in ("k_sto", "k_agents", "k_gen", "k_net")):
            m = {c: v / cnt[(k, c)] for c, v in d.items()}
            w = max(m.get("SQ_WAVES", 1), 1)
            print("$W $TAG", k, "launches", cnt[(k, "SQ_WAVES")], "waves %.0f VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f cyc/VALU %.2f wave-cycles/wave %.0f" % (w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, 4 * m["SQ_ACTIVE_INST_VALU"] / max(m["SQ_INSTS_VALU"], 1), 4 * m["SQ_WAVE_CYCLES"] / w))
PY
