"""profiles/<tag>_valu.json from scripts/prof_pmc.sh's counter summaries (gpurun_out/pmc_<workload>/summary.json) and the steady-state
kernel durations of scripts/profile.sh (profiles/<tag>_<workload>_kernel_stats_windows.csv): per kernel the SQ counters, mean per
launch, and what they say against the VALU roofline.
  VALU-busy fraction = SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) * 4 / (1024 SIMDs) / (duration * 2.4 GHz)
  (the share of the launch's SIMD-cycles in which a SIMD issues a vector instruction; 2.4 GHz = the chip's maximum clock, so a lower
  bound of the share at the clock actually held)
usage: python scripts/collect_valu.py r03 config4 config2"""
import csv, json, os, sys
tag, workloads = sys.argv[1], sys.argv[2:]
out_path = f"profiles/{tag}_valu.json"
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
out["_note"] = __doc__
for w in workloads:
    s = json.load(open(f"gpurun_out/pmc_{w}/summary.json"))
    dur = {}
    f = f"profiles/{tag}_{w}_kernel_stats_windows.csv"
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            if r["window"].startswith("steady"):
                dur[r["Name"].split("(")[0].split("::")[-1]] = float(r["AverageNs"])
    rec = {}
    for k, c in s.items():
        d = dur.get(k)
        e = {x: c[x] for x in c}
        if d and "SQ_ACTIVE_INST_VALU" in c:
            e["steady_state_duration_ns"] = d
            e["valu_busy_frac_at_2.4GHz"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (d * 2.4)
            e["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / max(c.get("SQ_WAVES", 1.0), 1.0)
            e["cycles_per_valu_instruction"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / max(c["SQ_INSTS_VALU"], 1.0)
        rec[k] = e
    out[w] = rec
json.dump(out, open(out_path, "w"), indent=1)
print("updated", out_path)
