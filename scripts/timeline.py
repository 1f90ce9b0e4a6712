"""Gaps between consecutive kernels of the iteration chain, from a rocprofv3 --kernel-trace CSV.
usage: python scripts/timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dopf::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    ka = a["Kernel_Name"].split("(")[0].replace("void dopf::", "").replace("dopf::", "")
    kb = b["Kernel_Name"].split("(")[0].replace("void dopf::", "").replace("dopf::", "")
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000:
        gap[(ka, kb)].append(g)
    dur[ka].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
for k, v in dur.items():
    v.sort(); print(f"dur  {k:40s} n={len(v):5d} median {v[len(v)//2]/1e3:7.2f} us  mean {sum(v)/len(v)/1e3:7.2f}")
for k, v in gap.items():
    v.sort(); print(f"gap  {k[0]:28s} -> {k[1]:28s} n={len(v):5d} median {v[len(v)//2]/1e3:7.2f} us  mean {sum(v)/len(v)/1e3:7.2f}")
