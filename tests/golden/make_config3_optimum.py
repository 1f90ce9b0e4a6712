#!/usr/bin/env python3
"""Central-LP optimum of BASELINE configs[3] at its full size (synthetic 118-node/186-line network, 100 000 agents x 168):
units aggregated per (node, cost[, level/power ratio]) — exact for the LP (central_lp.aggregate_by_node; checked here on
the 1/8 share against its un-aggregated optimum in synthetic_optima.json) — then the nodal-injection LP in HiGHS."""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import synth
from central_lp import aggregate_by_node, solve_central_nodal

path = os.path.join(HERE, "synthetic_optima.json")
out = json.load(open(path))
t0 = time.time()
share = synth.baseline_config(3, scale=0.125)
r = solve_central_nodal(aggregate_by_node(share))
print("share, aggregated:", r["objective"], "un-aggregated:", out["config3-share"]["objective"], f"{time.time() - t0:.0f} s", flush=True)
assert abs(r["objective"] - out["config3-share"]["objective"]) <= 1e-8 * r["objective"]
t0 = time.time()
pp = synth.baseline_config(3)
agg = aggregate_by_node(pp)
r = solve_central_nodal(agg)
dt = time.time() - t0
out["config3"] = {"G": pp.G, "S": pp.S, "T": pp.T, "N": pp.N, "L": pp.L, "objective": r["objective"], "seed": synth.SEED,
                  "how": f"synth.baseline_config(3); units aggregated per (node, cost) to {agg.G}+{agg.S} (exact for the LP, verified on the share), "
                         f"nodal-injection LP in HiGHS, {dt:.0f} s"}
print("config3", r["objective"], f"{dt:.0f} s")
json.dump(out, open(path, "w"), indent=1)
