#!/bin/bash
# rocprofv3 profile of bench.py for one workload (kernel_stats.csv = ALL launches of the process, kernel_stats_windows.csv =
# the launches bench.py's events time); summaries land in gpurun_out/prof_<workload>/ and are
# copied into profiles/ by hand (profiles/ is tracked, gpurun_out/ is scratch).
#   pass 1: --kernel-trace --stats          per-kernel durations (must agree with bench.py's HIP-event numbers)
#   pass 2: --pmc FETCH_SIZE                HBM read traffic   } separate passes, no tracing mixed in
#   pass 3: --pmc WRITE_SIZE                HBM write traffic  } (MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -e
W=${1:-config2}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
ARGS="bench.py --workload $W --no-cpu-baseline --no-also"      # bench.py defaults: 48 warm-up + 400 timed + 32 event-timed iterations
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
python3 - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs") if k in r} for r in rows]
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
# the same windows bench.py's events cover, from the kernel trace: launch order is engine 1 = W+K graph iterations,
# engine 2 (replay) = W graph iterations + K event-timed ones, of which the last `timed_iters` are the steady state
line = json.loads(open("$OUT/trace.json").read().strip().splitlines()[-1])
W, K, tail = line["warmup"], line["steps"], 32
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows = sorted((r for r in csv.DictReader(open(f)) if "dopf::" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    win = []
    for k, d in by.items():
        if len(d) < 2 * (W + K):
            continue
        reg, st = d[2 * W + K: 2 * W + 2 * K], d[2 * W + 2 * K - tail: 2 * W + 2 * K]
        win.append({"Name": k, "window": "timed-region replay", "Calls": len(reg), "AverageNs": sum(reg) / len(reg), "MinNs": min(reg), "MaxNs": max(reg)})
        win.append({"Name": k, "window": "steady state (last %d)" % tail, "Calls": len(st), "AverageNs": sum(st) / len(st), "MinNs": min(st), "MaxNs": max(st)})
    out["kernel_stats_windows"] = win
    with open("$OUT/kernel_stats_windows.csv", "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=["Name", "window", "Calls", "AverageNs", "MinNs", "MaxNs"]); wr.writeheader(); wr.writerows(win)
pmc = {}
for name in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            pmc.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
out["pmc"] = pmc
out["bench_line"] = json.loads(open("$OUT/trace.json").read().strip().splitlines()[-1])
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for r in out.get("kernel_stats", []): print(r)
for k, d in pmc.items(): print(k, d)
PY
