"""The CPU oracle against everything the reference ships that pins this path:
the three dumped trajectories, the iteration count, the thesis tables. CPU only."""
import numpy as np
import pytest

from helpers import follow_golden, make_engine

CASES = {"TNS": dict(), "big_gamma": dict(gamma=0.5), "wrong_weight": dict(w_flow=0.15)}


@pytest.mark.parametrize("mode", [0, 1], ids=["literal-qp", "exact"])
@pytest.mark.parametrize("name", list(CASES))
def test_trajectory_matches_reference_dump(oracle_api, three_node, golden, name, mode):
    pp = three_node[4]
    gold = golden[name]
    assert gold["params"]["gamma"] == CASES[name].get("gamma", 0.3)
    e = make_engine(oracle_api, pp, mode=mode, eps=0.0, **CASES[name])
    # Gurobi's barrier tolerance shows as 1e-9..6e-6 noise in the dumps
    follow_golden(e, gold, atol_primal=1e-5, atol_dual=1e-5)


@pytest.mark.parametrize("mode", [0, 1], ids=["literal-qp", "exact"])
def test_converges_at_iteration_476_and_matches_thesis(oracle_api, three_node, thesis, mode):
    pp = three_node[4]
    e = make_engine(oracle_api, pp, mode=mode)
    done, conv = e.iterate(5000)
    lam_res, mu_res, rho_res, it = e.get_residuals()
    assert conv and done == 476 and it == 476
    assert max(lam_res, mu_res, rho_res) < 1e-3
    P, D, C, E = e.get_primal()
    inj, aU, aK, flow, cost = e.get_consensus()
    t = thesis["admm"]
    assert np.abs(P - np.asarray(t["P"])).max() < 6e-5          # printed to 4 decimals
    assert np.abs(flow - np.asarray(t["flows"])).max() < 6e-5
    assert abs(cost - t["total_costs"]) < 1e-3
    c = thesis["central"]
    assert abs(cost - c["objective"]) / c["objective"] < 1e-3   # north-star tolerance
    assert np.abs(P - np.asarray(c["P"])).max() / 220 < 2e-3    # per-mille table match
    assert np.abs(D - np.asarray(c["D"])).max() < 1e-3 and np.abs(C - np.asarray(c["C"])).max() < 1e-3
    assert np.abs(E - np.asarray(c["E"])).max() < 1e-3
    # nodal price from the duals the last solve USED (src/opf_admm_decentral.jl:9): thesis Table 17
    assert np.abs(e.get_nodal_price(0) - np.asarray(t["nodal_price"])).max() < 6e-5
    lam_used = e.get_duals_used()[0]
    assert np.abs(lam_used - np.asarray(t["lambda"])).max() < 6e-5
    # avg_U at convergence = the example matrix of thesis eq. (40)
    assert np.abs(aU - np.asarray([[30.0005, 0.0], [90.0011, 25.0034], [0.0, 69.9865]])).max() < 6e-4


@pytest.mark.parametrize("name", ["big_gamma", "wrong_weight"])
def test_negative_controls_do_not_converge(oracle_api, three_node, name):
    """gamma = 0.5 and flow weight gamma/2 must NOT converge (thesis section 4.2)."""
    e = make_engine(oracle_api, three_node[4], mode=1, max_iters=750, **CASES[name])
    done, conv = e.iterate(2000)
    assert not conv and done == 750
    lam_res, mu_res, rho_res, it = e.get_residuals()
    assert it == 751 and max(lam_res, mu_res, rho_res) > 1.0
