#!/bin/bash
# per-launch durations of the x-update kernel over a long settled window: how many launches exceed 1.5x the mean, and when
#   spikes.sh <workload> [extra bench flags]
cd /tmp && export TMPDIR=/tmp
W=${1:-config2}; X=${2:-}
RAW=/tmp/dopf_spk_$W; rm -rf $RAW; mkdir -p $RAW
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $RAW -- python3 bench.py --workload $W --no-side --no-cpu-baseline --no-also --steps 600 --warmup 48 --timed-iters 2 $X > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$RAW/**/*kernel_trace.csv", recursive=True):
    rows = sorted((r for r in csv.DictReader(open(f)) if "dopf::" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows: by[r["Kernel_Name"].split("(")[0][-44:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in by.items():
        if len(d) < 600: continue
        w = d[48:648]                      # engine 1: iterations 49..648
        m = sum(w) / len(w)
        big = [(i + 49, round(x, 1)) for i, x in enumerate(w) if x > 1.5 * m]
        print("$W", k, "mean %.1f us, %d of %d launches > 1.5x mean:" % (m, len(big), len(w)), big[:40])
        srt = sorted(w); print("   p50 %.1f p90 %.1f p99 %.1f max %.1f" % (srt[len(w)//2], srt[int(len(w)*0.9)], srt[int(len(w)*0.99)], srt[-1]))
PY
