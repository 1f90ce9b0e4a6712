"""Host-side mirror of the reference's driver API (admm.py, network.py, synth.py), CPU only.
The compute backend here is the oracle library — test infrastructure standing in for the GPU so
that the marshalling, history and CSV logic is covered without one."""
import csv
import os

import numpy as np
import pytest

import decentralopf_jl_amd as pkg
from decentralopf_jl_amd import synth


def test_types_and_pack(three_node):
    nodes, lines, gens, stos, pp = three_node
    assert (pp.N, pp.L, pp.T, pp.G, pp.S) == (3, 3, 2, 4, 1)
    assert pp.demand.tolist() == [[10, 250], [50, 70], [120, 200]]
    assert pp.gen_node.tolist() == [0, 1, 2, 0] and pp.sto_node.tolist() == [0]
    assert pp.f_max.tolist() == [20, 45, 70]
    kw = pp.engine_kwargs()
    assert kw["demand"].tolist() == [10, 50, 120, 250, 70, 200]            # column-major [n + N*t]
    assert np.allclose(kw["ptdf"][:3], [-0.4, -0.6, 0.4])                  # column n = 0 first


def test_run_with_history_and_nodal_price(oracle_api, three_node, thesis):
    nodes, lines, gens, stos, _ = three_node
    admm = pkg.ADMM(0.3, nodes, gens, stos, lines, backend=oracle_api, backend_mode=1)
    pkg.run(admm)
    assert admm.convergence.all and admm.iteration == 476
    assert len(admm.results) == 476 and len(admm.lambdas) == 477          # duals pushed before the test
    np_used = pkg.get_nodal_price(admm, admm.iteration)
    assert np.abs(np_used - np.asarray(thesis["admm"]["nodal_price"])).max() < 6e-5
    # post-update duals do NOT reproduce Table 17
    np_after = pkg.get_nodal_price(admm, admm.iteration + 1)
    assert abs(np_after[0, 1] - (-81.9756)) > 2e-4
    r = admm.results[-1]
    assert abs(r.total_costs - 14034.5056) < 1e-3
    assert np.abs(r.of(gens[2]).generation - [4.9976, 219.9834]).max() < 6e-5
    assert np.abs(r.of(stos[0]).level - [10, 0]).max() < 1e-3
    assert len(admm.convergence.lambda_res) == 475                       # no test at iteration 1


def test_run_without_history_matches(oracle_api, three_node):
    nodes, lines, gens, stos, _ = three_node
    a = pkg.run(pkg.ADMM(0.3, nodes, gens, stos, lines, backend=oracle_api, backend_mode=1, record=False))
    assert a.convergence.all and a.iteration == 476 and len(a.results) == 1
    assert abs(a.results[0].total_costs - 14034.5056) < 1e-3


def test_iteration_cap(oracle_api, three_node):
    nodes, lines, gens, stos, _ = three_node
    a = pkg.run(pkg.ADMM(0.5, nodes, gens, stos, lines, backend=oracle_api, backend_mode=1, max_iters=40))
    assert not a.convergence.all and a.iteration == 41 and len(a.results) == 40


def test_export_results_layout_matches_reference_dump(oracle_api, three_node, golden, tmp_path):
    """Same columns, row order and values as results/TNS_*.csv (src/helpers/output.jl:1-85)."""
    nodes, lines, gens, stos, _ = three_node
    admm = pkg.ADMM(0.3, nodes, gens, stos, lines, backend=oracle_api, backend_mode=1, eps=0.0, max_iters=12)
    pkg.run(admm)
    pkg.export_results(admm, "TNS", parent_dir=str(tmp_path) + os.sep)
    rows = list(csv.reader(open(tmp_path / "TNS_duals.csv")))
    assert rows[0] == ["iteration", "dual", "timestep", "line", "value"]
    assert rows[1] == ["1", "lambda", "1", "", "0.0"]
    n_it = admm.iteration if admm.iteration <= len(admm.results) else len(admm.results)
    assert len(rows) == 1 + n_it * 2 * (1 + 3 + 3)
    assert rows[1 + 2 * n_it][:4] == ["1", "rho", "1", "1"]               # lambda block, then rho, then mue
    gold = golden["TNS"]["iterations"]
    for r in rows[1:]:
        k, name, t, l, v = int(r[0]), r[1], int(r[2]) - 1, r[3], float(r[4])
        if str(k) in gold:
            g = gold[str(k)]
            want = g["lam"][t] if name == "lambda" else g["mu" if name == "mue" else "rho"][int(l) - 1][t]
            assert abs(v - want) < 1e-5
    rows = list(csv.reader(open(tmp_path / "TNS_generators.csv")))
    assert rows[0] == ["iteration", "generator", "timestep", "generation"]
    assert rows[1][:3] == ["1", "pv", "1"] and abs(float(rows[1][3]) - 36.25469304166821) < 1e-6
    rows = list(csv.reader(open(tmp_path / "TNS_storages.csv")))
    assert rows[0] == ["iteration", "storage", "timestep", "charge", "discharge"]
    assert rows[1][:3] == ["1", "battery", "1"] and abs(float(rows[1][3]) - 10.0) < 1e-6


def test_synthetic_cases_are_seed_stable():
    a, b = synth.baseline_config(1), synth.baseline_config(1)
    assert (a.G, a.S, a.T, a.N, a.L) == (1000, 100, 24, 1, 0)
    assert np.array_equal(a.gen_mc, b.gen_mc) and np.array_equal(a.demand, b.demand)
    assert a.gen_mc.min() >= 1 and a.gen_mc.max() <= 60 and a.gen_pmax.min() >= 10 and a.gen_pmax.max() <= 300
    assert np.array_equal(a.sto_emax, 2 * a.sto_pmax) and np.all(a.demand == np.round(a.demand))
    c = synth.baseline_config(3, scale=0.01)
    assert (c.N, c.L, c.T) == (118, 186, 168) and np.all(np.diff(c.gen_node) >= 0)
    assert c.ptdf.shape == (186, 118) and np.abs(c.ptdf[:, 0]).max() == 0       # slack column is zero


def test_shard_partitions_agents():
    pp = synth.synthetic_case(23, 7, 4, N=3, L=3, seed=1)
    parts = [pp.shard(r, 4) for r in range(4)]
    assert sum(p.G for p in parts) == 23 and sum(p.S for p in parts) == 7
    assert np.array_equal(np.concatenate([p.gen_mc for p in parts]), pp.gen_mc)
    assert all(p.meta["n_agents_global"] == 30 for p in parts)


def test_central_reference_reproduces_the_thesis_tables(three_node, thesis):
    """decentralopf.jl_amd/central.py = src/opf_central_reference.jl: objective 14035, dispatch, flows, system price and
    nodal prices of thesis Tables 8-16 (printed pp. 49-52)."""
    nodes, lines, gens, stos, pp = three_node
    r = pkg.central_reference(nodes, gens, stos, lines)
    c = thesis["central"]
    assert abs(r.objective - c["objective"]) < 1e-6
    assert np.abs(r.generation - np.asarray(c["P"])).max() < 1e-6
    assert np.abs(r.discharge[0] - np.asarray(c["D"])).max() < 1e-6 and np.abs(r.charge[0] - np.asarray(c["C"])).max() < 1e-6
    assert np.abs(r.level[0] - np.asarray(c["E"])).max() < 1e-6
    assert np.abs(r.line_utilization - np.asarray(c["flows"])).max() < 1e-6
    assert np.abs(r.injection - np.asarray(c["injection"])).max() < 1e-6
    assert np.abs(r.system_price - np.asarray(c["lambda"])).max() < 1e-6
    assert np.abs(r.nodal_price - np.asarray(c["nodal_price"])).max() < 0.051          # the thesis prints one decimal
    # the decentral run lands on it (sign of the prices flipped by construction, thesis p. 52)
    t = thesis["admm"]
    assert np.abs(-np.asarray(t["nodal_price"]) - r.nodal_price).max() < 0.05
    # and the unit-wise formulation of the tests gives the same optimum
    from central_lp import solve_central
    assert abs(solve_central(pp)["objective"] - r.objective) < 1e-6
