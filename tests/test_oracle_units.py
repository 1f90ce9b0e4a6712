"""Unit tests of the oracle's building blocks and of the exact mode against the literal mode."""
import ctypes as C

import numpy as np
import pytest

from central_lp import solve_central
from decentralopf_jl_amd import synth
from helpers import make_engine, max_diff, state_of, storage_kkt_violation

dp = C.POINTER(C.c_double)


def _qp(api, Q, c, A, b, lb, ub):
    n = len(c)
    m = 0 if A is None else A.shape[0]
    x = np.zeros(n)
    y = np.zeros(max(m, 1))
    it = C.c_int32(0)
    arr = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    ptr = lambda a: None if a is None else a.ctypes.data_as(dp)
    Q, c, A, b, lb, ub = map(arr, (Q, c, A, b, lb, ub))
    rc = api.qp_solve(n, m, ptr(Q), ptr(c), ptr(A), ptr(b), ptr(lb), ptr(ub), ptr(x), ptr(y), C.byref(it))
    assert rc == 0
    return x


def test_qp_solver_known_answers(oracle_api):
    x = _qp(oracle_api, np.eye(1), [-3.0], None, None, [0.0], [2.0])
    assert abs(x[0] - 2.0) < 1e-9
    x = _qp(oracle_api, np.diag([1.0, 0.0]), [-3.0, 1.0], np.array([[1.0, -1.0]]), [0.5], [0.0, 0.0], [np.inf, 5.0])
    assert np.abs(x - [2.0, 1.5]).max() < 1e-9
    # random strictly convex box QP against projected gradient to high accuracy
    rng = np.random.default_rng(0)
    M = rng.normal(size=(6, 6))
    Q = M @ M.T + 0.5 * np.eye(6)
    c = rng.normal(size=6) * 5
    lb, ub = -np.ones(6), np.ones(6)
    x = _qp(oracle_api, Q, c, None, None, lb, ub)
    z = np.zeros(6)
    step = 1.0 / np.linalg.eigvalsh(Q).max()
    for _ in range(20000):
        z = np.clip(z - step * (Q @ z + c), lb, ub)
    assert np.abs(x - z).max() < 1e-8


def test_ptdf_three_node(oracle_api, three_node):
    pp = three_node[4]
    want = np.array([[-0.4, 0.2, 0.0], [-0.6, -0.2, 0.0], [0.4, 0.8, 0.0]])
    assert np.abs(pp.ptdf - want).max() < 1e-12
    frm = np.array([1, 2, 1], dtype=np.int32)
    to = np.array([0, 0, 2], dtype=np.int32)
    sus = np.array([1.0, 1.0, 2.0])
    out = np.zeros(9)
    rc = oracle_api.calculate_ptdf(3, 3, frm.ctypes.data_as(C.POINTER(C.c_int32)), to.ctypes.data_as(C.POINTER(C.c_int32)),
                                   sus.ctypes.data_as(dp), 2, out.ctypes.data_as(dp))
    assert rc == 0
    assert np.abs(out.reshape(3, 3).T - want).max() < 1e-12


def test_central_lp_three_node(three_node, thesis):
    r = solve_central(three_node[4])
    c = thesis["central"]
    assert abs(r["objective"] - c["objective"]) < 1e-6
    assert np.abs(r["P"] - np.asarray(c["P"])).max() < 1e-6
    assert np.abs(np.abs(r["lam"]) - 30.0).max() < 1e-6


SYNTH = [
    ("copper-6", dict(n_gen=20, n_sto=5, T=6), dict(gamma=0.05), 30, 1e-9),
    ("copper-12-storage-heavy", dict(n_gen=10, n_sto=8, T=12, seed=3), dict(gamma=0.1), 30, 1e-9),
    ("net-4x5", dict(n_gen=12, n_sto=4, T=5, N=4, L=5, seed=5, fmax_factor=0.7, fmax_min=5), dict(gamma=0.1), 25, 1e-6),
    ("net-5x7", dict(n_gen=15, n_sto=6, T=8, N=5, L=7, seed=11, fmax_factor=0.5, fmax_min=5), dict(gamma=0.3), 20, 1e-6),
]


@pytest.mark.parametrize("name,case,params,iters,tol", SYNTH, ids=[s[0] for s in SYNTH])
def test_exact_mode_equals_literal_qp(oracle_api, name, case, params, iters, tol):
    """Slack elimination + price-threshold recursion (mode 1) against the literally assembled QPs."""
    pp = synth.synthetic_case(**case)
    a = make_engine(oracle_api, pp, mode=0, eps=0.0, **params)
    b = make_engine(oracle_api, pp, mode=1, eps=0.0, **params)
    for k in range(iters):
        a.iterate(1)
        # one-step comparison: start b from a's previous state so that differences do not compound
        b.iterate(1)
        sa, sb = state_of(a), state_of(b)
        worst, where = max_diff(sa, sb)
        assert worst < tol, (k, where, worst)
        b.set_state(P=sa["P"], D=sa["D"], C_=sa["C"], avg_U=sa["avg_U"], avg_K=sa["avg_K"], lam=sa["lam"],
                    mu=sa["mu"], rho=sa["rho"], iteration=a.get_residuals()[3])


def test_agent_slacks_sum_to_average(oracle_api, three_node):
    pp = three_node[4]
    e = make_engine(oracle_api, pp, mode=0)
    e.iterate(7)
    L, T = pp.L, pp.T
    U = np.zeros(L * T)
    K = np.zeros(L * T)
    sumU = np.zeros(L * T)
    for a in range(pp.G + pp.S):
        assert oracle_api.get_agent_slacks(e._ctx, a, U.ctypes.data_as(dp), K.ctypes.data_as(dp)) == 0
        sumU += U
    aU = e.get_consensus()[1]
    assert np.abs(sumU.reshape(T, L).T / 5 - aU).max() < 1e-12      # averages divide by ALL agents


def test_storage_edge_cases(oracle_api):
    """T = 1, a storage that cannot move (pmax = 0), one that cannot store (emax = 0)."""
    pp = synth.synthetic_case(6, 3, 1, seed=9)
    for mode in (0, 1):
        make_engine(oracle_api, pp, mode=mode, gamma=0.05).iterate(5)
    pp = synth.synthetic_case(8, 4, 6, seed=10)
    pp.sto_pmax[0] = 0.0
    pp.sto_emax[1] = 0.0
    a = make_engine(oracle_api, pp, mode=0, eps=0.0, gamma=0.05)
    b = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=0.05)
    a.iterate(15)
    b.iterate(15)
    sa, sb = state_of(a), state_of(b)
    assert max_diff(sa, sb)[0] < 1e-7
    assert np.abs(sb["D"][0]).max() == 0 and np.abs(sb["C"][0]).max() == 0
    assert np.abs(sb["E"][1]).max() < 1e-9


def test_exact_storage_satisfies_kkt_certificate(oracle_api):
    """The independent optimality certificate used by the full-size GPU tests, validated on the oracle."""
    pp = synth.synthetic_case(40, 25, 24, seed=21)
    gamma = 0.01
    e = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=gamma)
    for k in range(12):
        before = state_of(e)
        e.iterate(1)
        after = state_of(e)
        s_prev = before["inj"].sum(axis=0)
        lam_used = e.get_duals_used()[0]
        theta = lam_used[None, :] + gamma * (s_prev[None, :] - (before["D"] - before["C"]))
        v = storage_kkt_violation(pp, np.arange(pp.S), before["D"], before["C"], after["D"], after["C"], after["E"],
                                  theta, gamma)
        assert v < 1e-6, (k, v)
    # and the certificate does reject a perturbed (non-optimal) point
    bad = after["D"].copy()
    bad[:, 3] = np.clip(bad[:, 3] + 0.5, 0, pp.sto_pmax)
    E_bad = np.cumsum(after["C"] - bad, axis=1)
    assert storage_kkt_violation(pp, np.arange(pp.S), before["D"], before["C"], bad, after["C"], E_bad, theta, gamma) > 1e-3


def test_synthetic_admm_reaches_central_optimum(oracle_api):
    """Config-1-shaped case (scaled down): gamma ~ 1/A converges to the LP optimum within 1e-3."""
    pp = synth.synthetic_case(100, 10, 24)
    A = pp.G + pp.S
    e = make_engine(oracle_api, pp, mode=1, gamma=1.0 / A, max_iters=4000)
    done, conv = e.iterate(4000)
    assert conv
    cost = e.get_consensus()[4]
    opt = solve_central(pp)["objective"]
    assert abs(cost - opt) / opt < 1e-3


def test_sharding_is_a_partition_of_a_sum(oracle_api):
    """n_shards in {1,2,4}: local_update + summed consensus buffers + apply_consensus == one shard."""
    from decentralopf_jl_amd.sharded import host_consensus_view
    pp = synth.synthetic_case(30, 9, 8, N=3, L=3, seed=4, fmax_factor=0.8, fmax_min=5)
    ref = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=0.05)
    ref.iterate(12)
    want = state_of(ref)
    for world in (2, 4):
        shards = [pp.shard(r, world) for r in range(world)]
        engs = [make_engine(oracle_api, sh, mode=1, eps=0.0, gamma=0.05, n_agents_global=pp.G + pp.S) for sh in shards]
        views = [host_consensus_view(e) for e in engs]
        for _ in range(12):
            for e in engs:
                e.local_update()
            total = np.sum(views, axis=0)
            for v in views:
                v[:] = total
            for e in engs:
                e.apply_consensus()
        got = [state_of(e) for e in engs]
        for k in ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow", "cost"):
            for g in got:
                assert np.abs(g[k] - want[k]).max() < 1e-9, (world, k)
        P = np.concatenate([g["P"] for g in got])
        D = np.concatenate([g["D"] for g in got])
        assert np.abs(P - want["P"]).max() < 1e-9 and np.abs(D - want["D"]).max() < 1e-9


def test_oracle_exact_mode_ignores_noise_level_ptdf_entries(oracle_api):
    """A 4-node network whose PTDF holds entries of 1e-16 (rounding noise of the inverse): their kinks sit at 1e19,
    where the exact mode's tables used to lose all precision (all generators of a node flipped to pmax at some
    timesteps). Found by scripts/fuzz_parity.py — HIP and the literal mode agreed, the exact mode did not."""
    from decentralopf_jl_amd import synth
    from helpers import make_engine, state_of, max_diff
    pp = synth.synthetic_case(n_gen=31, n_sto=0, T=30, N=4, L=3, seed=153374, fmax_factor=0.5, fmax_min=1.0)
    assert 0 < np.abs(pp.ptdf)[np.abs(pp.ptdf) > 0].min() < 1e-12          # the case really has such entries
    params = dict(gamma=0.03225806451612903, w_flow=0.1, eps=0.0)
    lit = make_engine(oracle_api, pp, mode=0, **params)
    exa = make_engine(oracle_api, pp, mode=1, **params)
    for _ in range(4):
        lit.iterate(1)
        exa.iterate(1)
        sl, se = state_of(lit), state_of(exa)
        assert max_diff(sl, se, keys=[k for k in sl if k != "cost"])[0] < 1e-6
        exa.set_state(P=sl["P"], D=sl["D"], C_=sl["C"], avg_U=sl["avg_U"], avg_K=sl["avg_K"], lam=sl["lam"], mu=sl["mu"],
                      rho=sl["rho"], iteration=lit.get_residuals()[3])


def test_aggregated_central_lp_equals_the_full_one():
    """tests/central_lp.aggregate_copper_plate (how the 1M-agent optimum is obtained): same optimum as the full LP."""
    from decentralopf_jl_amd import synth
    from central_lp import aggregate_copper_plate
    pp = synth.synthetic_case(300, 60, 12, seed=77)
    ag = aggregate_copper_plate(pp)
    assert ag.G <= 60 and ag.S < pp.S
    full, agg = solve_central(pp)["objective"], solve_central(ag)["objective"]
    assert abs(full - agg) <= 1e-9 * full
    with pytest.raises(ValueError):
        aggregate_copper_plate(synth.synthetic_case(10, 2, 3, N=3, L=2, seed=1))


def test_nodal_central_lp_equals_the_unit_formulation():
    """tests/central_lp.solve_central_nodal (explicit nodal injections: how the 118-node share's optimum is obtained)."""
    from decentralopf_jl_amd import synth
    from central_lp import solve_central_nodal
    for case in (dict(n_gen=40, n_sto=8, T=6, N=5, L=6, seed=3, fmax_factor=0.8, fmax_min=5), dict(n_gen=30, n_sto=5, T=8, seed=2)):
        pp = synth.synthetic_case(**case)
        a, b = solve_central(pp)["objective"], solve_central_nodal(pp)["objective"]
        assert abs(a - b) <= 1e-9 * abs(a)
