"""Copy the rocprofv3 summaries of scripts/profile.sh from gpurun_out/ (scratch) into profiles/ (tracked).
usage: python scripts/collect_profiles.py r01 config1 config2 config4"""
import json, os, shutil, sys

tag, workloads = sys.argv[1], sys.argv[2:]
pmc_path = f"profiles/{tag}_pmc.json"
lines_path = f"profiles/{tag}_bench_lines_under_rocprof.json"
pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
lines = json.load(open(lines_path)) if os.path.exists(lines_path) else {}
pmc["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, scripts/profile.sh = the default bench.py "
                "command), mean per launch, unit KiB. On gfx950 FETCH_SIZE counts half the bytes of a streaming read "
                "(MI355X_MICROARCH.md, HBM): HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.")
for w in workloads:
    d = f"gpurun_out/prof_{w}"
    if w.endswith("_driver"):          # the driver's command: kernel trace only
        sd = json.load(open(f"{d}/summary.json"))
        shutil.copy(f"{d}/kernel_stats.csv", f"profiles/{tag}_{w}_kernel_stats.csv")
        shutil.copy(f"{d}/kernel_stats_windows.csv", f"profiles/{tag}_{w}_kernel_stats_windows.csv")
        lines[w] = sd["bench_line"]
        continue
    s = json.load(open(f"{d}/summary.json"))
    shutil.copy(f"{d}/kernel_stats.csv", f"profiles/{tag}_{w}_kernel_stats.csv")
    if os.path.exists(f"{d}/kernel_stats_windows.csv"):
        shutil.copy(f"{d}/kernel_stats_windows.csv", f"profiles/{tag}_{w}_kernel_stats_windows.csv")
    pmc[w] = {k.split("::")[-1]: {"FETCH_SIZE": v.get("FETCH_SIZE", {}).get("mean"), "WRITE_SIZE": v.get("WRITE_SIZE", {}).get("mean"),
                                  "launches_averaged": v.get("FETCH_SIZE", v.get("WRITE_SIZE", {})).get("n")}
              for k, v in s["pmc"].items() if "dopf::" in k}
    lines[w] = s["bench_line"]
json.dump(pmc, open(pmc_path, "w"), indent=1)
json.dump(lines, open(lines_path, "w"), indent=1)
print("updated", pmc_path, lines_path)
