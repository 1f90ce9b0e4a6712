#!/usr/bin/env python3
"""Makes the golden fixtures under tests/golden/ from the reference's shipped result dumps.

Source (data files, MIT licence of the reference): /root/reference/results/
  TNS_{generators,storages,duals}.csv           gamma = 0.3, flow weight 10   (550 iterations)
  big_gamma_{...}.csv                           gamma = 0.5                    (757 iterations)
  wrong_weight_{...}.csv                        flow weight 0.15               (674 iterations)
written by export_results (src/helpers/output.jl:1-85): dual row i = the dual USED in
iteration i (row 1 = zeros); storage columns are charge, discharge.

Only a subsample of the iterations is kept (values verbatim, full double precision). Run here,
in the build container; the GPU box has no /root/reference and only reads the .json files.
"""
import csv
import json
import os

SRC = "/root/reference/results"
OUT = os.path.dirname(os.path.abspath(__file__))
GENS = ["pv", "wind", "coal", "gas"]          # order of src/cases/three_node.jl:13-18
T, L = 2, 3

CASES = {
    "TNS": dict(params=dict(gamma=0.3, w_flow=10.0),
                keep=list(range(1, 13)) + [25, 50, 75, 100, 150, 200, 250, 300, 350, 400, 450]
                + list(range(470, 482)) + [500, 549, 550]),
    "big_gamma": dict(params=dict(gamma=0.5, w_flow=10.0),
                      keep=list(range(1, 13)) + [50, 100, 200, 300, 400, 500, 600, 700, 750, 756, 757]),
    "wrong_weight": dict(params=dict(gamma=0.3, w_flow=0.15),
                         keep=list(range(1, 13)) + [50, 100, 200, 300, 400, 500, 600, 650, 673, 674]),
}


def main():
    for name, spec in CASES.items():
        keep = set(spec["keep"])
        rec = {k: dict(P=[[None] * T for _ in GENS], C=[None] * T, D=[None] * T, lam=[None] * T,
                       mu=[[None] * T for _ in range(L)], rho=[[None] * T for _ in range(L)])
               for k in keep}
        n_iter = 0
        for r in csv.DictReader(open(f"{SRC}/{name}_generators.csv")):
            k = int(r["iteration"])
            n_iter = max(n_iter, k)
            if k in keep:
                rec[k]["P"][GENS.index(r["generator"])][int(r["timestep"]) - 1] = float(r["generation"])
        for r in csv.DictReader(open(f"{SRC}/{name}_storages.csv")):
            k = int(r["iteration"])
            if k in keep:
                rec[k]["C"][int(r["timestep"]) - 1] = float(r["charge"])
                rec[k]["D"][int(r["timestep"]) - 1] = float(r["discharge"])
        for r in csv.DictReader(open(f"{SRC}/{name}_duals.csv")):
            k = int(r["iteration"])
            if k not in keep:
                continue
            t = int(r["timestep"]) - 1
            v = float(r["value"])
            if r["dual"] == "lambda":
                rec[k]["lam"][t] = v
            else:
                rec[k]["mue" == r["dual"] and "mu" or "rho"][int(r["line"]) - 1][t] = v
        out = dict(
            source=f"rockstaedt/DecentralOPF.jl results/{name}_{{generators,storages,duals}}.csv",
            case="src/cases/three_node.jl", params=spec["params"], n_iterations_in_source=n_iter,
            generators=GENS, layout="P[g][t], C[t], D[t], lam[t], mu[l][t], rho[l][t]; duals of row k "
                                   "are the ones USED by iteration k",
            iterations={str(k): rec[k] for k in sorted(keep)})
        with open(os.path.join(OUT, f"{name}.json"), "w") as f:
            json.dump(out, f, indent=0, separators=(",", ":"))
        print(name, n_iter, "iterations in source;", len(keep), "kept")


if __name__ == "__main__":
    main()
