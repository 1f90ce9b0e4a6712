#!/bin/bash
# rocprofv3 profile of bench.py for one workload; summaries land in gpurun_out/prof_<workload>[_driver]/ and are copied into
# profiles/ by scripts/collect_profiles.py (profiles/ is tracked, gpurun_out/ is scratch).
#   profile.sh <workload> [driver]     "driver" = the driver's command (--steps 20 --warmup 5) instead of bench.py's defaults
#   pass 1: --kernel-trace --stats          per-kernel durations (kernel_stats.csv = ALL launches of the process;
#                                           kernel_stats_windows.csv = the launches bench.py's HIP events time)
#   pass 2: --pmc FETCH_SIZE                HBM read traffic   } separate passes, no tracing mixed in
#   pass 3: --pmc WRITE_SIZE                HBM write traffic  } (MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -e
W=${1:-config2}
MODE=${2:-default}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$W
ARGS="bench.py --workload $W --no-side"      # bench.py defaults: 48 warm-up + 400 timed iterations
if [ "$MODE" = "driver" ]; then OUT=${OUT}_driver; ARGS="$ARGS --gpus 1 --steps 20 --warmup 5"; fi
RAW=/tmp/dopf_prof_${W}_$MODE      # raw traces / counter dumps stay on the box (gpurun_out/ is copied back only below 64 MiB)
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err
if [ "$MODE" != "driver" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/write -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
fi
python3 - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$RAW/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs") if k in r} for r in rows]
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
# the windows bench.py's events cover, from the kernel trace. Launch order of the x-update kernel: engine 1 = W+K graph
# iterations (the timed region); engine 2 (replay) = W graph iterations, K event-timed ones (= roofline.kernel_ms), then graph
# iterations up to iteration max(W+K, 200) and timed_iters event-timed ones (= roofline.steady_state)
line = json.loads(open("$OUT/trace.json").read().strip().splitlines()[-1])
W, K, tail = line["warmup"], line["steps"], 32
gap = max(W + K, 200) - (W + K)
for f in glob.glob("$RAW/trace/**/*kernel_trace.csv", recursive=True):
    rows = sorted((r for r in csv.DictReader(open(f)) if "dopf::" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    win = []
    for k, d in by.items():
        if len(d) < 2 * (W + K) + gap + tail:
            continue
        reg = d[2 * W + K: 2 * W + 2 * K]
        st = d[2 * W + 2 * K + gap: 2 * W + 2 * K + gap + tail]
        g1 = d[W: W + K]
        win.append({"Name": k, "window": "timed region, graph replay (engine 1, iterations %d..%d)" % (W + 1, W + K), "Calls": len(g1), "AverageNs": sum(g1) / len(g1), "MinNs": min(g1), "MaxNs": max(g1)})
        win.append({"Name": k, "window": "timed region, event-timed replay (= roofline.kernel_ms)", "Calls": len(reg), "AverageNs": sum(reg) / len(reg), "MinNs": min(reg), "MaxNs": max(reg)})
        win.append({"Name": k, "window": "steady state (%d iterations from iteration %d on = roofline.steady_state)" % (tail, max(W + K, 200) + 1), "Calls": len(st), "AverageNs": sum(st) / len(st), "MinNs": min(st), "MaxNs": max(st)})
    out["kernel_stats_windows"] = win
    with open("$OUT/kernel_stats_windows.csv", "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=["Name", "window", "Calls", "AverageNs", "MinNs", "MaxNs"]); wr.writeheader(); wr.writerows(win)
pmc = {}
for name in ("fetch", "write"):
    for f in glob.glob("$RAW/%s/**/*counter_collection.csv" % name, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            pmc.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
out["pmc"] = pmc
out["bench_line"] = line
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for r in out.get("kernel_stats_windows", []):
    if "k_agents" in r["Name"] or "k_gen" in r["Name"] or "k_sto" in r["Name"]: print(r)
r = line["roofline"]; print("bench line: ms/step %.4f kernel_ms %.4f steady %.4f frac %.3f" % (line["ms_per_step"], r["kernel_ms"], r["steady_state"]["kernel_ms"], r["frac"]))
for k, d in pmc.items(): print(k, d)
PY
