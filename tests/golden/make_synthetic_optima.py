#!/usr/bin/env python3
"""Central-LP optima (SciPy/HiGHS, tests/central_lp.py = src/opf_central_reference.jl:21-53 restated) of the
seed-stable synthetic BASELINE configurations, so that the GPU tests can check "converged objective within
1e-3 of the central optimum" without solving a 4.8-million-variable LP on the GPU box (config2: ~45 s here)."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import dopf_pkg  # noqa: E402

dopf_pkg.load()
from decentralopf_jl_amd import synth  # noqa: E402
from central_lp import aggregate_copper_plate, solve_central, solve_central_nodal  # noqa: E402

out = {"source": "scipy.optimize.linprog(method='highs') on synth.baseline_config(i); objective = sum mc*P + sum mc*(D+C)"}
for idx in (1, 2):
    pp = synth.baseline_config(idx)
    t0 = time.time()
    r = solve_central(pp)
    out[f"config{idx}"] = {"G": pp.G, "S": pp.S, "T": pp.T, "objective": r["objective"], "seed": synth.SEED}
    print(idx, r["objective"], f"{time.time() - t0:.1f}s")
# config4 (1M agents x 24 = 24 million LP variables): units with identical parameters aggregated (exact on the copper
# plate, central_lp.aggregate_copper_plate; the same aggregation reproduces the two full solves above to the digit)
for idx in (1, 2, 4):
    pp = synth.baseline_config(idx)
    r = solve_central(aggregate_copper_plate(pp))
    if f"config{idx}" in out:
        assert abs(r["objective"] - out[f"config{idx}"]["objective"]) <= 1e-9 * r["objective"], idx
    else:
        out[f"config{idx}"] = {"G": pp.G, "S": pp.S, "T": pp.T, "objective": r["objective"], "seed": synth.SEED,
                               "how": "aggregated LP (one unit per parameter class)"}
    print(idx, "aggregated", r["objective"])
# one GPU's share of the 118-node network configuration (12 500 agents x 168): nodal-injection formulation, ~2 min
pp = synth.baseline_config(3, scale=0.125)
r = solve_central_nodal(pp)
out["config3-share"] = {"G": pp.G, "S": pp.S, "T": pp.T, "N": pp.N, "L": pp.L, "objective": r["objective"], "seed": synth.SEED,
                        "how": "synth.baseline_config(3, scale=0.125); LP with explicit nodal injections (central_lp.solve_central_nodal)"}
print("config3-share", r["objective"])
json.dump(out, open(os.path.join(HERE, "synthetic_optima.json"), "w"), indent=1)
