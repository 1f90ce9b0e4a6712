"""Parity of the HIP path (libdopf_hip.so, through the C ABI) against the oracle, the reference's
golden dumps and size-independent optimality properties. Needs a real MI355X: pytest -m gpu."""
import numpy as np
import pytest

from central_lp import solve_central
from decentralopf_jl_amd import _capi, synth
from helpers import follow_golden, make_engine, max_diff, state_of, storage_kkt_violation

pytestmark = pytest.mark.gpu

CASES = {"TNS": dict(), "big_gamma": dict(gamma=0.5), "wrong_weight": dict(w_flow=0.15)}


# ---- the reference's own data ---------------------------------------------------------------------

@pytest.mark.parametrize("name", list(CASES))
def test_hip_follows_reference_dump(hip_api, three_node, golden, name):
    e = make_engine(hip_api, three_node[4], eps=0.0, **CASES[name])
    follow_golden(e, golden[name], atol_primal=1e-5, atol_dual=1e-5)      # Gurobi noise floor ~6e-6
    assert e.solver_failures() == 0


@pytest.mark.parametrize("flags", [0, _capi.F_NO_GRAPH, _capi.F_OVERLAP_AGENTS], ids=["graph", "eager", "two-streams"])
def test_hip_three_node_converges_like_the_thesis(hip_api, three_node, thesis, flags):
    e = make_engine(hip_api, three_node[4], flags=flags)
    done, conv = e.iterate(5000)                                          # one call, stop test on the device
    lam_res, mu_res, rho_res, it = e.get_residuals()
    assert conv and done == 476 and it == 476 and max(lam_res, mu_res, rho_res) < 1e-3
    P, D, C, E = e.get_primal()
    inj, aU, aK, flow, cost = e.get_consensus()
    t, c = thesis["admm"], thesis["central"]
    assert np.abs(P - np.asarray(t["P"])).max() < 6e-5
    assert np.abs(flow - np.asarray(t["flows"])).max() < 6e-5
    assert abs(cost - c["objective"]) / c["objective"] < 1e-3             # north-star tolerance
    assert np.abs(P - np.asarray(c["P"])).max() / 220 < 2e-3              # per-mille table match
    assert np.abs(e.get_nodal_price(0) - np.asarray(t["nodal_price"])).max() < 6e-5
    assert e.iterate(10) == (0, True)                                     # frozen after convergence


@pytest.mark.parametrize("name", ["big_gamma", "wrong_weight"])
def test_hip_negative_controls(hip_api, three_node, name):
    e = make_engine(hip_api, three_node[4], max_iters=750, **CASES[name])
    done, conv = e.iterate(2000)
    assert not conv and done == 750 and e.get_residuals()[3] == 751


# ---- against the oracle on seeded synthetic cases ----------------------------------------------------

SYNTH = [
    # name, case, params, oracle mode, iterations, tolerance (fp64; 1e-9 on copper plate)
    ("three-node-literal", None, dict(), 0, 60, 1e-7),
    ("copper-T1", dict(n_gen=9, n_sto=4, T=1, seed=9), dict(gamma=0.05), 1, 20, 1e-9),
    ("copper-T6", dict(n_gen=20, n_sto=5, T=6), dict(gamma=0.05), 1, 40, 1e-9),
    ("copper-T24-lps8", dict(n_gen=100, n_sto=30, T=24, seed=2), dict(gamma=0.01), 1, 40, 1e-9),
    ("copper-T23-odd", dict(n_gen=300, n_sto=40, T=23, seed=9), dict(gamma=0.003), 1, 25, 1e-9),
    ("copper-T40-lps16", dict(n_gen=50, n_sto=20, T=40, seed=5), dict(gamma=0.02), 1, 15, 1e-9),
    ("copper-T96-lps32", dict(n_gen=50, n_sto=20, T=96, seed=3), dict(gamma=0.02), 1, 15, 1e-9),
    ("copper-T168-lps64", dict(n_gen=50, n_sto=20, T=168, seed=4), dict(gamma=0.02), 1, 8, 1e-9),
    ("copper-T250-nch4", dict(n_gen=20, n_sto=6, T=250, seed=6), dict(gamma=0.02), 1, 4, 1e-9),
    ("copper-T400-nch8", dict(n_gen=20, n_sto=6, T=400, seed=7), dict(gamma=0.02), 1, 3, 1e-9),
    ("copper-T600-gen-only", dict(n_gen=30, n_sto=0, T=600, seed=8), dict(gamma=0.02), 1, 5, 1e-9),
    ("copper-gamma1-diverging", dict(n_gen=60, n_sto=12, T=24, seed=12), dict(gamma=1.0), 1, 25, 1e-7),
    ("copper-multinode", dict(n_gen=40, n_sto=10, T=12, N=5, seed=13), dict(gamma=0.02), 1, 20, 1e-9),
    ("copper-30-nodes-two-level-reduce", dict(n_gen=200, n_sto=40, T=24, N=30, seed=33), dict(gamma=0.004), 1, 15, 1e-9),
    ("copper-200-nodes-dual-per-timestep", dict(n_gen=600, n_sto=60, T=24, N=200, seed=34), dict(gamma=0.0015), 1, 8, 1e-9),
    ("net-4x5", dict(n_gen=12, n_sto=4, T=5, N=4, L=5, seed=5, fmax_factor=0.7, fmax_min=5), dict(gamma=0.1), 1, 40, 1e-8),
    ("net-4x5-literal", dict(n_gen=12, n_sto=4, T=5, N=4, L=5, seed=5, fmax_factor=0.7, fmax_min=5), dict(gamma=0.1), 0, 15, 1e-6),
    ("net-6x9", dict(n_gen=60, n_sto=15, T=24, N=6, L=9, seed=7, fmax_factor=0.6, fmax_min=5), dict(gamma=0.05), 1, 30, 1e-8),
    ("net-30x50-empty-nodes", dict(n_gen=25, n_sto=6, T=8, N=30, L=50, seed=17, fmax_factor=0.6, fmax_min=5), dict(gamma=0.05), 1, 10, 1e-8),
    ("net-2x1-300-agents-per-node", dict(n_gen=560, n_sto=40, T=6, N=2, L=1, seed=29, fmax_factor=0.5, fmax_min=5), dict(gamma=0.004, w_flow=0.002), 1, 12, 1e-8),
    ("net-3x3-scaled-flow-weight", dict(n_gen=200, n_sto=30, T=12, N=3, L=3, seed=31, fmax_factor=0.8, fmax_min=5), dict(gamma=0.004, w_flow=0.0013), 1, 25, 1e-8),
    ("net-4x3-noise-level-ptdf-entries", dict(n_gen=31, n_sto=0, T=30, N=4, L=3, seed=153374, fmax_factor=0.5, fmax_min=1.0), dict(gamma=0.03225806451612903, w_flow=0.1), 0, 6, 1e-6),
    ("net-6x5-storages-w0.1", dict(n_gen=15, n_sto=22, T=8, N=6, L=5, seed=866744, fmax_factor=0.8, fmax_min=1.0), dict(gamma=0.05, w_flow=0.1), 1, 12, 1e-8),
    ("storages-only", dict(n_gen=0, n_sto=12, T=24, seed=14), dict(gamma=0.05), 1, 10, 1e-9),
    ("net-100x300-more-lines-than-one-block (k_dual_t + k_price_t)", dict(n_gen=150, n_sto=20, T=24, N=100, L=300, seed=23, fmax_factor=0.7, fmax_min=20), dict(gamma=0.01), 1, 4, 1e-7),
    ("net-118x186-config3-shape", dict(n_gen=900, n_sto=90, T=12, N=118, L=186, seed=19, fmax_factor=0.7, fmax_min=20), dict(gamma=0.002), 1, 4, 1e-7),
    ("net-4x5-T73-generator-blocks-of-three-column-passes", dict(n_gen=60, n_sto=8, T=73, N=4, L=5, seed=47, fmax_factor=0.7, fmax_min=5), dict(gamma=0.02), 1, 8, 1e-8),
    ("net-4x5-T1", dict(n_gen=30, n_sto=6, T=1, N=4, L=5, seed=48, fmax_factor=0.7, fmax_min=5), dict(gamma=0.03), 1, 12, 1e-8),
    ("net-4x5-T250-scan-kernel-only", dict(n_gen=20, n_sto=6, T=250, N=4, L=5, seed=50, fmax_factor=0.7, fmax_min=5), dict(gamma=0.03), 1, 4, 1e-8),
    ("net-40x60-T250-wide-and-long", dict(n_gen=60, n_sto=8, T=250, N=40, L=60, seed=51, fmax_factor=0.8, fmax_min=5), dict(gamma=0.02), 1, 3, 1e-7),
    ("net-4x5-T2", dict(n_gen=30, n_sto=6, T=2, N=4, L=5, seed=49, fmax_factor=0.7, fmax_min=5), dict(gamma=0.03), 1, 12, 1e-8),
    ("net-30x45-T96-three-launch-chain (k_net_agents, k_slack, k_dual_price_t1024 with the slack sums)",
     dict(n_gen=120, n_sto=24, T=96, N=30, L=45, seed=41, fmax_factor=0.7, fmax_min=10), dict(gamma=0.01), 1, 5, 1e-7),
]


@pytest.mark.parametrize("name,case,params,mode,iters,tol", SYNTH, ids=[s[0] for s in SYNTH])
def test_hip_one_step_parity(hip_api, oracle_api, three_node, name, case, params, mode, iters, tol):
    """Both sides start every iteration from the SAME state (the oracle's), so the number checked is
    the error of one x-update + consensus + dual update, not its amplification by the dynamics."""
    pp = three_node[4] if case is None else synth.synthetic_case(**case)
    h = make_engine(hip_api, pp, eps=0.0, **params)
    o = make_engine(oracle_api, pp, mode=mode, eps=0.0, **params)
    for k in range(iters):
        h.iterate(1)
        o.iterate(1)
        sh, so = state_of(h), state_of(o)
        scale = max(1.0, float(np.abs(so["lam"]).max()))
        worst, where = max_diff(sh, so, keys=[x for x in sh if x != "cost"])
        assert worst <= tol * scale, (k, where, worst)
        assert abs(sh["cost"][0] - so["cost"][0]) <= 1e-9 * max(1.0, abs(so["cost"][0]))
        h.set_state(P=so["P"], D=so["D"], C_=so["C"], avg_U=so["avg_U"], avg_K=so["avg_K"], lam=so["lam"],
                    mu=so["mu"], rho=so["rho"], iteration=o.get_residuals()[3])
    assert h.solver_failures() == 0


def test_hip_free_running_trajectory(hip_api, oracle_api):
    """No state resets: 200 iterations side by side on a convergent case stay within 1e-8."""
    pp = synth.synthetic_case(100, 10, 24)
    g = 1.0 / (pp.G + pp.S)
    h = make_engine(hip_api, pp, eps=0.0, gamma=g)
    o = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=g)
    h.iterate(200)
    o.iterate(200)
    assert max_diff(state_of(h), state_of(o), keys=["P", "D", "C", "E", "lam", "inj"])[0] < 1e-8


def test_hip_free_running_network_on_the_three_launch_chain(hip_api, oracle_api):
    """30 nodes / 45 lines x 96 timesteps (consensus state beyond the one-block dual kernel: generators and storages in one
    launch, node sums and their changes from k_slack, slack sums formed by the dual/price kernel), no state resets: 30
    iterations side by side with the oracle's exact mode — through the iterations in which lines are flagged and agents are
    walked one by one, and the ones after."""
    pp = synth.synthetic_case(120, 24, 96, N=30, L=45, seed=43, fmax_factor=0.8, fmax_min=10)
    A = pp.G + pp.S
    kw = dict(eps=0.0, gamma=1.0 / A, w_flow=0.3 / A)
    h = make_engine(hip_api, pp, **kw)
    o = make_engine(oracle_api, pp, mode=1, **kw)
    assert h.iterate_timed(1)["agents_fused"] == 1
    o.iterate(1)
    for n in (1, 3, 10, 15):
        h.iterate(n)
        o.iterate(n)
        sh, so = state_of(h), state_of(o)
        scale = max(1.0, float(np.abs(so["lam"]).max()))
        worst, where = max_diff(sh, so, keys=[x for x in sh if x != "cost"])
        assert worst <= 1e-7 * scale, (n, where, worst)
    assert h.solver_failures() == 0


def test_hip_warm_start_is_exact_and_used(hip_api, oracle_api):
    """The warm-start storage kernel (previous contact structure + KKT certificate) against the cold scan
    kernel and the oracle over a free run; and it must really carry the load in steady state."""
    pp = synth.synthetic_case(400, 120, 24, seed=41)
    g = 1.0 / (pp.G + pp.S)
    warm = make_engine(hip_api, pp, eps=0.0, gamma=g)
    cold = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=_capi.F_NO_WARM_START)
    ora = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=g)
    for n in (1, 1, 8, 40, 100):
        warm.iterate(n)
        cold.iterate(n)
        ora.iterate(n)
        a, b, c = state_of(warm), state_of(cold), state_of(ora)
        assert max_diff(a, b, keys=["P", "D", "C", "E", "lam", "inj"])[0] < 1e-8
        assert max_diff(a, c, keys=["P", "D", "C", "E", "lam", "inj"])[0] < 1e-8
    solved, left = warm.warm_start_stats()
    assert solved + left == pp.S and solved >= 0.5 * pp.S, (solved, left)
    assert cold.warm_start_stats() == (0, pp.S)
    assert warm.solver_failures() == 0 and cold.solver_failures() == 0
    # T = 96 (32 lanes x 3) and T = 168 (64 x 3) variants of the same kernel
    for T, seed in ((96, 42), (168, 43)):
        pp = synth.synthetic_case(100, 40, T, seed=seed)
        g = 1.0 / (pp.G + pp.S)
        warm = make_engine(hip_api, pp, eps=0.0, gamma=g)
        ora = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=g)
        warm.iterate(60)
        ora.iterate(60)
        assert max_diff(state_of(warm), state_of(ora), keys=["P", "D", "C", "E", "lam", "inj"])[0] < 1e-8
        assert warm.warm_start_stats()[0] >= 0.5 * pp.S


def test_hip_edge_cases(hip_api, oracle_api):
    pp = synth.synthetic_case(8, 4, 6, seed=10)
    pp.sto_pmax[0] = 0.0          # cannot move
    pp.sto_emax[1] = 0.0          # cannot store
    pp.gen_pmax[2] = 0.0          # cannot produce
    h = make_engine(hip_api, pp, eps=0.0, gamma=0.05)
    o = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=0.05)
    h.iterate(15)
    o.iterate(15)
    sh, so = state_of(h), state_of(o)
    assert max_diff(sh, so)[0] < 1e-8
    assert np.abs(sh["D"][0]).max() == 0 and np.abs(sh["C"][0]).max() == 0 and np.abs(sh["P"][2]).max() == 0
    assert np.abs(sh["E"][1]).max() < 1e-9


def test_hip_rejects_bad_input(hip_api):
    pp = synth.synthetic_case(4, 2, 3)
    kw = pp.engine_kwargs()
    kw["gen_node"] = np.asarray([0, 0, 5, 0], dtype=np.int32)
    with pytest.raises(_capi.DopfError, match="gen_node"):
        _capi.Engine(hip_api, params=_capi.default_params(), **kw)
    with pytest.raises(_capi.DopfError, match="T <= 512"):
        make_engine(hip_api, synth.synthetic_case(4, 2, 600))
    with pytest.raises(_capi.DopfError, match="positive"):
        make_engine(hip_api, pp, gamma=0.0)


def test_hip_is_bitwise_reproducible(hip_api):
    """Fixed-order reductions: two runs give identical bits, with and without the graph."""
    pp = synth.synthetic_case(3000, 300, 24, seed=31)
    # (the fused and the separate launches group their partial sums differently: two families)
    NT = _capi.F_NO_TAIL_FUSE
    for family in ((0, 0, _capi.F_NO_GRAPH), (_capi.F_NO_FUSE, _capi.F_NO_FUSE, _capi.F_NO_FUSE | _capi.F_NO_GRAPH),
                   (NT, NT, NT | _capi.F_NO_GRAPH), (NT | _capi.F_NO_FUSE, NT | _capi.F_NO_FUSE, _capi.F_OVERLAP_AGENTS)):
        outs = []
        for flags in family:
            e = make_engine(hip_api, pp, eps=0.0, gamma=1.0 / 3300, flags=flags)
            e.iterate(40)
            outs.append(state_of(e))
        for o in outs[1:]:
            for k in outs[0]:
                assert np.array_equal(outs[0][k], o[k]), (family, k)


@pytest.mark.parametrize("case", ["network", "copper plate", "copper plate, 3 nodes"])
def test_hip_sharded_contexts_equal_one(hip_api, case):
    """Two contexts on one GPU, consensus buffers summed by hand: the N > 1 arithmetic without RCCL."""
    import torch
    if case == "network":
        pp = synth.synthetic_case(300, 40, 24, N=3, L=3, seed=4, fmax_factor=0.8, fmax_min=5)
    elif case == "copper plate":
        pp = synth.synthetic_case(3000, 400, 24, seed=4)
    else:
        pp = synth.synthetic_case(300, 40, 24, N=3, L=0, seed=4)
    A = pp.G + pp.S
    ref = make_engine(hip_api, pp, eps=0.0, gamma=0.01)
    ref.iterate(12)
    want = state_of(ref)
    engs = [make_engine(hip_api, pp.shard(r, 2), eps=0.0, gamma=0.01, n_agents_global=A) for r in range(2)]
    bufs = [torch.zeros(e.consensus_size(), dtype=torch.float64, device="cuda") for e in engs]
    for e, b in zip(engs, bufs):
        e.bind_consensus(b.data_ptr())
    for _ in range(12):
        for e in engs:
            e.local_update()
        for e in engs:
            e.sync()
        total = bufs[0] + bufs[1]
        bufs[0].copy_(total)
        bufs[1].copy_(total)
        torch.cuda.synchronize()
        for e in engs:
            e.apply_consensus()
        for e in engs:
            e.sync()
    got = [state_of(e) for e in engs]
    for k in ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow", "cost"):
        for g in got:
            if want[k].size == 0:
                continue
            assert np.abs(g[k] - want[k]).max() <= 1e-9 * max(1.0, np.abs(want[k]).max()), k
    assert np.abs(np.concatenate([g["P"] for g in got]) - want["P"]).max() < 1e-9


# ---- BASELINE.json's full sizes: size-independent properties -----------------------------------------

FULL = [("config1-1100x24", 1, 1.0), ("config2-50k-x96", 2, 1.0), ("config4-1M-x24", 4, 1.0)]


@pytest.mark.parametrize("name,idx,scale", FULL, ids=[f[0] for f in FULL])
def test_hip_full_size_properties(hip_api, name, idx, scale):
    pp = synth.baseline_config(idx, scale=scale)
    A = pp.G + pp.S
    gamma = 1.0 / A
    e = make_engine(hip_api, pp, eps=0.0, gamma=gamma)
    e.iterate(30)
    before = state_of(e)
    e.iterate(1)
    after = state_of(e)
    lam_used = e.get_duals_used()[0]
    s_prev = before["inj"].sum(axis=0)
    # generators: closed form of SURVEY.md section 9.4, vectorised over all (g,t)
    want = np.clip(before["P"] - (pp.gen_mc[:, None] + lam_used[None, :] + gamma * s_prev[None, :]) / (1 + gamma),
                   0.0, pp.gen_pmax[:, None])
    assert np.abs(after["P"] - want).max() < 1e-9
    # storages: bounds, level recursion, and the KKT optimality certificate for every storage
    D, C, E = after["D"], after["C"], after["E"]
    assert D.min() >= 0 and C.min() >= 0 and (D - pp.sto_pmax[:, None]).max() <= 0 and (C - pp.sto_pmax[:, None]).max() <= 0
    assert E.min() >= -1e-9 and (E - pp.sto_emax[:, None]).max() <= 1e-9
    assert np.abs(np.cumsum(C - D, axis=1) - E).max() < 1e-9
    theta = lam_used[None, :] + gamma * (s_prev[None, :] - (before["D"] - before["C"]))
    assert storage_kkt_violation(pp, np.arange(pp.S), before["D"], before["C"], D, C, E, theta, gamma) < 1e-6
    # consensus: the reduced injection equals the sum of what the agents report; lambda step
    inj = -pp.demand + (after["P"].sum(axis=0) + (D - C).sum(axis=0))[None, :]
    # (rounding of a sum scales with the magnitude of the summands, not of the small residual)
    mag = after["P"].sum(axis=0).max() + pp.demand.max()
    assert np.abs(after["inj"] - inj).max() <= 1e-11 * mag + 1e-9
    assert np.abs(after["lam"] - (lam_used + gamma * after["inj"].sum(axis=0))).max() < 1e-9 * max(1.0, np.abs(after["lam"]).max())
    cost = float(pp.gen_mc @ after["P"].sum(axis=1) + pp.sto_mc @ (D + C).sum(axis=1))
    assert abs(after["cost"][0] - cost) <= 1e-10 * cost
    assert e.solver_failures() == 0


def test_hip_full_size_continuous_storage_parameters(hip_api):
    """config2's size with storages drawn from continuous ranges (marginal cost, power and the level/power ratio all
    different per storage): no two storages share a contact structure, unlike the 48 lock-stepped integer types of the
    synthetic generator. Every storage must carry a KKT certificate every iteration and none may reach the scan kernel's
    iteration cap."""
    pp = synth.baseline_config(2)
    rng = np.random.default_rng(77)
    pp.sto_mc = rng.uniform(0.5, 3.5, pp.S)
    pp.sto_pmax = rng.uniform(5.0, 20.0, pp.S)
    pp.sto_emax = pp.sto_pmax * rng.uniform(0.7, 4.0, pp.S)
    gamma = 1.0 / (pp.G + pp.S)
    e = make_engine(hip_api, pp, eps=0.0, gamma=gamma)
    left_total = 0
    for n in (1, 1, 1, 3, 9, 25, 60):
        e.iterate(n - 1)
        before = state_of(e)
        e.iterate(1)
        after = state_of(e)
        lam_used = e.get_duals_used()[0]
        s_prev = before["inj"].sum(axis=0)
        D, C, E = after["D"], after["C"], after["E"]
        assert E.min() >= -1e-9 and (E - pp.sto_emax[:, None]).max() <= 1e-9
        assert np.abs(np.cumsum(C - D, axis=1) - E).max() < 1e-9
        theta = lam_used[None, :] + gamma * (s_prev[None, :] - (before["D"] - before["C"]))
        assert storage_kkt_violation(pp, np.arange(pp.S), before["D"], before["C"], D, C, E, theta, gamma) < 1e-6
        left_total += e.warm_start_stats()[1]
    assert e.solver_failures() == 0
    assert left_total <= 0.01 * pp.S, left_total          # the active-set body certifies (almost) everything itself


@pytest.mark.parametrize("case", ["config1", "config2/5", "config4/5", "340k+2k x24 (fused and row skipping)"])
def test_hip_fused_agents_match_separate_launches(hip_api, case):
    """k_agents (one launch for generators + storages) against the separate kernels: same iterates up to the
    rounding of differently grouped partial sums."""
    pp = {"config1": lambda: synth.baseline_config(1), "config2/5": lambda: synth.baseline_config(2, scale=0.2),
          "config4/5": lambda: synth.baseline_config(4, scale=0.2)}.get(case, lambda: synth.synthetic_case(340000, 2000, 24))()
    g = 1.0 / (pp.G + pp.S)
    a = make_engine(hip_api, pp, eps=0.0, gamma=g)
    b = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=_capi.F_NO_FUSE)
    assert a.iterate_timed(1)["agents_fused"] == 1 and b.iterate_timed(1)["agents_fused"] == 0
    for n in (1, 5, 40, 150):
        a.iterate(n)
        b.iterate(n)
        for x, y in zip(a.get_primal(), b.get_primal()):
            assert np.abs(x - y).max() <= 1e-9 * (1.0 + np.abs(y).max())
        la, lb = a.get_duals()[0], b.get_duals()[0]
        assert np.abs(la - lb).max() <= 1e-10 * (1.0 + np.abs(lb).max())
    assert a.solver_failures() == 0 and b.solver_failures() == 0


TAIL = [
    ("config1 (fused launch, T=24)", lambda: synth.baseline_config(1), 0),
    ("config2/5 (fused launch, streaming generator blocks, T=96)", lambda: synth.baseline_config(2, scale=0.2), 0),
    ("config4/5 (generator launch + storage launch, row skipping)", lambda: synth.baseline_config(4, scale=0.2), 0),
    ("separate launches of a small grid", lambda: synth.synthetic_case(3000, 300, 24, seed=31), _capi.F_NO_FUSE),
    ("generators only, T=600 (512-thread blocks carry the tail)", lambda: synth.synthetic_case(500, 0, 600, seed=8), 0),
    ("storages only", lambda: synth.synthetic_case(0, 700, 48, seed=14), 0),
    ("T=168 (one wave per storage)", lambda: synth.synthetic_case(900, 120, 168, seed=4), 0),
]


@pytest.mark.parametrize("name,make,flags", TAIL, ids=[t[0] for t in TAIL])
def test_hip_tail_in_the_launch_matches_the_three_launch_chain(hip_api, name, make, flags):
    """One node, no lines: the iteration's consensus sums, dual step and stop test finished inside the x-update launch
    (integer accumulators, last block's tail) against k_reduce + k_dual_price_small (DOPF_F_NO_TAIL_FUSE): same iterates
    up to the rounding of the sums (fixed point vs fp64 in a fixed order), same stopping iteration, and two runs of the
    one-launch form are bit-identical, graph or eager."""
    pp = make()
    g = 1.0 / (pp.G + pp.S)
    a = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=flags)
    b = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=flags | _capi.F_NO_TAIL_FUSE)
    a2 = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=flags | _capi.F_NO_GRAPH)
    assert a.iterate_timed(1)["tail_fused"] == 1 and b.iterate_timed(1)["tail_fused"] == 0 and a2.iterate_timed(1)["tail_fused"] == 1
    for n in (1, 5, 40, 150):
        a.iterate(n)
        b.iterate(n)
        a2.iterate(n)
        sa, sb, s2 = state_of(a), state_of(b), state_of(a2)
        for k in sa:
            if sa[k].size == 0:
                continue
            assert np.array_equal(sa[k], s2[k]), k
            assert np.abs(sa[k] - sb[k]).max() <= 1e-9 * (1.0 + np.abs(sb[k]).max()), (n, k)
        assert a.get_residuals()[3] == b.get_residuals()[3] == 2 + sum(x for x in (1, 5, 40, 150) if x <= n)
    assert a.solver_failures() == 0 and b.solver_failures() == 0
    # the stop test fires at the same iteration and freezes the state
    c = make_engine(hip_api, pp, gamma=g, flags=flags, max_iters=5000)
    d = make_engine(hip_api, pp, gamma=g, flags=flags | _capi.F_NO_TAIL_FUSE, max_iters=5000)
    (nc, cc), (nd, cd) = c.iterate(5000), d.iterate(5000)
    assert (nc, cc) == (nd, cd) and cc
    assert c.iterate(10) == (0, True)
    assert abs(c.get_consensus()[4] - d.get_consensus()[4]) <= 1e-9 * abs(d.get_consensus()[4])


@pytest.mark.parametrize("scale", [2.0 ** 24, 2.0 ** 27, 2.0 ** 28, 2.0 ** 29, 2.0 ** 32], ids=lambda x: f"pmax x 2^{int(np.log2(x))}")
def test_hip_fixed_point_sums_at_the_admission_edge(hip_api, scale):
    """The one-launch iteration adds the blocks' sums as fixed-point integers whose binary point comes from the problem's
    bounds (k = 52 - ceil(log2(sum pmax)) fraction bits; dopf_create admits the tail only while k >= 8). Units scaled so that
    k sits on that edge and on both sides of it (12, 9, 8 | 7, 4): either the tail is in the launch, and then
      * the consensus sum it forms is the EXACT sum of the agents' injections (math.fsum of the primal arrays the same context
        returns) to within the resolution it promises — half a unit of 2^-k per contributing block (< 800 blocks) — while the
        fp64 chain (DOPF_F_NO_TAIL_FUSE) is within the rounding of its own additions;
      * the two chains' iterates agree to 1e-9 of the quantities being summed (sum pmax). (At these magnitudes the imbalance is
        the difference of 13-digit numbers and the dual step multiplies what is left by gamma: neither chain can hold lambda
        to 1e-9 of ITS size: they agree to 1e-7 ... 1e-4 of it here — asserted at 1e-3, a sanity bound; the statement about the
        fixed-point sums is the first one);
    or (k < 8) the library has put the case on the fp64 chain by itself."""
    import math
    pp = synth.synthetic_case(300, 30, 24, seed=77)
    for arr in (pp.gen_pmax, pp.sto_pmax, pp.sto_emax, pp.demand):
        arr *= scale
    for arr in (pp.gen_mc, pp.sto_mc):       # (cheap units: the cost accumulator — |cost| <= T sum |mc| pmax in 53 bits — must not be
        arr *= 2.0 ** -6                     # what keeps the tail out before the injection sums' binary point reaches the edge)
    total = float(pp.gen_pmax.sum() + pp.sto_pmax.sum())
    k = 52 - int(np.ceil(np.log2(1.0 + total)))
    kc = 52 - int(np.ceil(np.log2(1.0 + pp.T * float((np.abs(pp.gen_mc) * pp.gen_pmax).sum() + (np.abs(pp.sto_mc) * 2.0 * pp.sto_pmax).sum()))))
    assert {2.0 ** 24: 12, 2.0 ** 27: 9, 2.0 ** 28: 8, 2.0 ** 29: 7, 2.0 ** 32: 4}[scale] == k and (kc >= 0 or k < 8)
    g = 1.0 / (pp.G + pp.S)
    a = make_engine(hip_api, pp, eps=0.0, gamma=g)
    b = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=_capi.F_NO_TAIL_FUSE)
    fused = a.iterate_timed(1)["tail_fused"]
    b.iterate(1)
    assert fused == (1 if k >= 8 else 0), (k, fused)
    ulp = float(np.spacing(total))

    def exact_gap(e):           # |consensus injection - exact sum of the agents' injections|, worst timestep
        P, D, C, _ = e.get_primal()
        inj = e.get_consensus()[0]
        want = [math.fsum(P[:, t]) + math.fsum(D[:, t]) - math.fsum(C[:, t]) - float(pp.demand[:, t].sum()) for t in range(pp.T)]
        return float(np.abs(inj.sum(axis=0) - np.asarray(want)).max())
    for n in (1, 6, 40):
        a.iterate(n)
        b.iterate(n)
        assert exact_gap(b) <= (pp.G + pp.S) * ulp, (k, n, exact_gap(b))
        assert exact_gap(a) <= ((800 * 2.0 ** -(k + 1) + 2 * ulp) if fused else (pp.G + pp.S) * ulp), (k, n, exact_gap(a))
        sa, sb = state_of(a), state_of(b)
        for key in sa:
            if sa[key].size == 0:
                continue
            d = float(np.abs(sa[key] - sb[key]).max())
            if key in ("lam",):
                assert d <= 1e-3 * (1.0 + np.abs(sb[key]).max()), (k, n, key, d)
            else:
                assert d <= 1e-9 * total, (k, n, key, d)
    assert a.solver_failures() == 0 and b.solver_failures() == 0


PERSIST = [
    ("T24", lambda: synth.synthetic_case(1000, 100, 24, seed=20220720)),                 # config1's shape (8 lanes x 3 steps)
    ("T96", lambda: synth.synthetic_case(6000, 600, 96, seed=3)),                        # config2's shape (32 x 3), several passes per block
    ("T48 mixed storages", lambda: synth.synthetic_case(3000, 500, 48, seed=8)),
]


@pytest.mark.parametrize("name,make", PERSIST, ids=[t[0] for t in PERSIST])
def test_hip_persistent_iterations_match_the_launch_per_iteration_chain(hip_api, name, make):
    """DOPF_F_PERSIST (csrc/agents_persist.h): up to 16 iterations per launch — the grid stays, the tail block publishes the new
    prices to the other blocks — against one launch per iteration: the same kernels' arithmetic in the same order, so every
    array is bit-identical, after calls of any length (1, 4 and 16 iterations per launch, mixed), through the cold start (storage
    blocks that hand storages to the scan body), and the stop test fires at the same iteration and freezes the state."""
    pp = make()
    if "mixed" in name:
        rng = np.random.default_rng(5)
        pp.sto_emax = pp.sto_pmax * rng.choice([0.7, 1.0, 2.0, 3.3], size=pp.S)
    g = 1.0 / (pp.G + pp.S)
    a = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=_capi.F_PERSIST)
    b = make_engine(hip_api, pp, eps=0.0, gamma=g)
    ta, tb = a.iterate_timed(1), b.iterate_timed(1)
    assert ta["persist"] == 1 and tb["persist"] == 0 and ta["tail_fused"] == 1 and ta["sto_lean"] == 1
    for n in (1, 2, 5, 16, 37, 64, 3):
        a.iterate(n)
        b.iterate(n)
        sa, sb = state_of(a), state_of(b)
        for k in sa:
            if sa[k].size:
                assert np.array_equal(sa[k], sb[k]), (n, k, float(np.abs(sa[k] - sb[k]).max()))
        assert a.get_residuals() == b.get_residuals()
    assert a.solver_failures() == 0 and b.solver_failures() == 0
    c = make_engine(hip_api, pp, gamma=g, flags=_capi.F_PERSIST, max_iters=5000)
    d = make_engine(hip_api, pp, gamma=g, max_iters=5000)
    rc, rd = c.iterate(5000), d.iterate(5000)
    assert rc == rd and rc[1]
    assert c.iterate(10) == (0, True)
    sc, sd = state_of(c), state_of(d)
    for k in sc:
        if sc[k].size:
            assert np.array_equal(sc[k], sd[k]), k


SLACK_IN_DUAL = [
    ("30 nodes / 45 lines, limits binding", lambda: synth.synthetic_case(500, 50, 168, N=30, L=45, seed=11, fmax_factor=1.0, fmax_min=20), 0.3),
    ("30 nodes / 45 lines, literal flow weight", lambda: synth.synthetic_case(500, 50, 168, N=30, L=45, seed=12), None),
    ("118 nodes / 186 lines (2 % of configs[3])", lambda: synth.baseline_config(3, scale=0.02), 0.3),
    ("118 nodes / 186 lines, literal flow weight", lambda: synth.baseline_config(3, scale=0.02), None),
    # more than 32 partial rows per node (the quiet chain's old limit): the dual/price kernel stages ~1 100 rows per timestep in LDS
    ("30 nodes / 45 lines, 40 rows per node", lambda: synth.synthetic_case(3000, 300, 96, N=30, L=45, seed=13, fmax_factor=1.0, fmax_min=20), 0.3),
]


@pytest.mark.parametrize("name,make,wf", SLACK_IN_DUAL, ids=[t[0] for t in SLACK_IN_DUAL])
def test_hip_slack_sums_in_the_dual_launch_are_bit_identical(hip_api, monkeypatch, name, make, wf):
    """Networks on the one-launch dual/price kernel (single-GPU chain): generators and storages in one launch (256-thread
    generator blocks, k_net_agents), the node sums leave from k_slack and the slack sums of timestep t are formed by the dual
    block of t (no k_reduce launch) — against the five-launch chain (DOPF_F_NO_TAIL_FUSE | DOPF_F_NO_FUSE: 512-thread
    generator kernel, storage kernel, k_slack, k_reduce, dual): every array the C ABI exposes is bit-identical at every step,
    through the cold start (lines whose switch point lies inside a node's window: the agent-by-agent sums) and after.
    (Same generator items on both sides: the one-launch form would take half as many, twice as large.)"""
    pp = make()
    A = pp.G + pp.S
    kw = dict(eps=0.0, gamma=1.0 / A)
    if wf is not None:
        kw["w_flow"] = wf / A
    a = make_engine(hip_api, pp, flags=_capi.F_NET_SMALL_ITEMS, **kw)
    b = make_engine(hip_api, pp, flags=_capi.F_NO_TAIL_FUSE | _capi.F_NO_FUSE, **kw)
    assert a.iterate_timed(1)["agents_fused"] == 1 and b.iterate_timed(1)["agents_fused"] == 0
    for n in (1, 3, 20, 100):
        a.iterate(n)
        b.iterate(n)
        sa, sb = state_of(a), state_of(b)
        for k in sa:
            if k == "cost":              # (a block's cost is a butterfly per wave, then the waves in order: 4 waves against 8)
                assert abs(sa[k][0] - sb[k][0]) <= 1e-13 * abs(sb[k][0])
            elif sa[k].size:
                assert np.array_equal(sa[k], sb[k]), (n, k, float(np.abs(sa[k] - sb[k]).max()))
    assert a.solver_failures() == 0 and b.solver_failures() == 0
    assert np.abs(sa["avg_U"]).max() > 0 or np.abs(sa["avg_K"]).max() > 0 or wf is not None
    # the quiet chain (no k_slack launch while no line is flagged): allowed on a, never on b; with the scaled flow weight the flags are
    # gone by now and a runs it — more single iterations, so that a line flagged again parks it in front of the host's eyes
    import ctypes as C
    qa, qb = (C.c_int64 * 3)(), (C.c_int64 * 3)()
    assert hip_api.lib.dopf_debug_quiet(a._ctx, qa) == 0 and hip_api.lib.dopf_debug_quiet(b._ctx, qb) == 0
    assert qa[0] == 1 and qb[0] == 0 and qb[1] == 0 and qb[2] == 0
    for n in (1, 1, 2, 7, 16, 33):
        a.iterate(n)
        b.iterate(n)
        sa, sb = state_of(a), state_of(b)
        for k in sa:
            if k != "cost" and sa[k].size:
                assert np.array_equal(sa[k], sb[k]), ("quiet", n, k)
        assert a.get_residuals() == b.get_residuals()
    hip_api.lib.dopf_debug_quiet(a._ctx, qa)
    print(name, "quiet at the end:", qa[1], "parked:", qa[2])
    if wf is not None and "118" in name:
        assert qa[1] == 1


@pytest.mark.parametrize("idx,steps", [(4, (1, 7, 60)), (2, (1, 2, 3, 9, 25, 110)), (1, (1, 5, 40, 90))],
                         ids=["config4: the generator launch", "config2: streaming generator blocks of the one-launch iteration", "config1"])
def test_hip_row_skipping_is_bit_identical(hip_api, idx, steps):
    """The generator sweeps that skip rows of P parked on a bound against the full sweep: 1M agents x 24 (a launch of its own,
    gen_pair_skip_body) and the streaming generator blocks of the one-launch iteration (config2, config1: gen_pair_stream_skip) —
    every array the ABI exposes, bit for bit, through the cold start and after; and the state words hold what P holds."""
    pp = synth.baseline_config(idx)
    g = 1.0 / (pp.G + pp.S)
    a = make_engine(hip_api, pp, eps=0.0, gamma=g)
    b = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=_capi.F_NO_ROW_SKIP)
    for n in steps:
        a.iterate(n)
        b.iterate(n)
        sa, sb = state_of(a), state_of(b)
        for k in sa:
            if sa[k].size:
                assert np.array_equal(sa[k], sb[k]), (n, k)
        assert a.get_consensus()[4] == b.get_consensus()[4]
        assert a.get_residuals() == b.get_residuals()
    Pa = a.get_primal()[0]
    sat = ((Pa == 0).all(axis=1) | (Pa == pp.gen_pmax[:, None]).all(axis=1)).mean()
    assert sat > 0.3            # the skipping path really had rows to skip
    assert a.solver_failures() == 0 and b.solver_failures() == 0


def test_hip_config2_time_to_residual_and_optimum(hip_api):
    """BASELINE config 2 (50k agents x 96): gamma = 1/A reaches the 1e-3 residual in a few hundred iterations
    and the cost sits within 1e-3 (in fact ~1e-5) of the central LP optimum (tests/golden/synthetic_optima.json)."""
    from conftest import load_golden
    pp = synth.baseline_config(2)
    opt = load_golden("synthetic_optima")["config2"]
    assert (opt["G"], opt["S"], opt["T"]) == (pp.G, pp.S, pp.T)
    e = make_engine(hip_api, pp, gamma=1.0 / (pp.G + pp.S), max_iters=20000)
    done, conv = e.iterate(20000)
    assert conv and done < 2000
    assert abs(e.get_consensus()[4] - opt["objective"]) / opt["objective"] < 1e-3
    assert e.solver_failures() == 0


def test_hip_config4_reaches_central_optimum(hip_api):
    """BASELINE config 4 (1M agents x 24): the 1e-3 residual is reached in ~650 iterations and the cost sits within
    1e-3 (in fact 4e-6) of the central LP optimum (aggregated LP, tests/golden/synthetic_optima.json)."""
    from conftest import load_golden
    pp = synth.baseline_config(4)
    opt = load_golden("synthetic_optima")["config4"]
    assert (opt["G"], opt["S"], opt["T"]) == (pp.G, pp.S, pp.T)
    e = make_engine(hip_api, pp, gamma=1.0 / (pp.G + pp.S), max_iters=5000)
    done, conv = e.iterate(5000)
    assert conv and done < 1500
    assert abs(e.get_consensus()[4] - opt["objective"]) / opt["objective"] < 1e-3
    assert e.solver_failures() == 0


def test_hip_network_share_reaches_central_optimum(hip_api):
    """One GPU's share of BASELINE config 3 (118 nodes, 186 lines, 12 500 agents x 168), flow weight 0.3/A: converges
    in ~670 iterations to within 1e-3 (in fact 7e-7) of the central LP optimum (tests/golden/synthetic_optima.json)."""
    from conftest import load_golden
    pp = synth.baseline_config(3, scale=0.125)
    opt = load_golden("synthetic_optima")["config3-share"]
    assert (opt["G"], opt["S"], opt["T"], opt["N"], opt["L"]) == (pp.G, pp.S, pp.T, pp.N, pp.L)
    A = pp.G + pp.S
    e = make_engine(hip_api, pp, gamma=1.0 / A, w_flow=0.3 / A, max_iters=5000)
    done, conv = e.iterate(5000)
    assert conv and done < 1500
    inj, aU, aK, flow, cost = e.get_consensus()
    assert abs(cost - opt["objective"]) / opt["objective"] < 1e-3
    assert (np.abs(flow) - pp.f_max[:, None]).max() <= 1e-6 * pp.f_max.max()
    assert e.solver_failures() == 0


def test_hip_config3_full_size(hip_api):
    """BASELINE configs[3] at its real size on ONE GPU: synthetic 118-node / 186-line network, 100 000 agents x 168.
    Size-independent properties of one iteration (closed-form slack sums against an agent-by-agent evaluation, flows,
    dual step, storage certificates), then the run to the 1e-3 residual and its cost against the central LP optimum
    (tests/golden/make_config3_optimum.py: units aggregated per node and cost, exact for the LP)."""
    from conftest import load_golden
    pp = synth.baseline_config(3)
    opt = load_golden("synthetic_optima")["config3"]
    assert (opt["G"], opt["S"], opt["T"], opt["N"], opt["L"]) == (pp.G, pp.S, pp.T, pp.N, pp.L) == (90910, 9090, 168, 118, 186)
    A = pp.G + pp.S
    g, wf = 1.0 / A, 0.3 / A
    e = make_engine(hip_api, pp, gamma=g, w_flow=wf, max_iters=5000)
    e.iterate(20)
    before = state_of(e)
    lam0, mu0, rho0 = e.get_duals()
    e.iterate(1)
    after = state_of(e)
    # consensus sums from what the agents report: injection, flows = ptdf . injection
    inj = -pp.demand.copy()
    np.add.at(inj, pp.gen_node, after["P"])
    np.add.at(inj, pp.sto_node, after["D"] - after["C"])
    mag = np.abs(pp.demand).max() + 1.0
    assert np.abs(after["inj"] - inj).max() <= 1e-10 * mag * 100
    assert np.abs(after["flow"] - pp.ptdf @ after["inj"]).max() <= 1e-9 * mag
    # mean slacks: U_a = max(0, (g a - w2 (f + h d_a - F)) / (w2 + g)) summed agent by agent
    w2 = 2 * wf
    dG = after["P"] - before["P"]
    dS = (after["D"] - after["C"]) - (before["D"] - before["C"])
    sumU = np.zeros((pp.L, pp.T))
    sumK = np.zeros((pp.L, pp.T))
    for n in range(pp.N):
        d = np.concatenate([dG[pp.gen_node == n], dS[pp.sto_node == n]], axis=0)          # (agents at n, T)
        if d.shape[0] == 0:
            continue
        fl = before["flow"][None, :, :] + pp.ptdf[:, n][None, :, None] * d[:, None, :]      # (a, L, T)
        sumU += np.maximum(0.0, (g * before["avg_U"][None] - w2 * (fl - pp.f_max[None, :, None])) / (w2 + g)).sum(axis=0)
        sumK += np.maximum(0.0, (g * before["avg_K"][None] + w2 * (fl + pp.f_max[None, :, None])) / (w2 + g)).sum(axis=0)
    assert np.abs(after["avg_U"] - sumU / A).max() <= 1e-9 * max(1.0, np.abs(sumU / A).max())
    assert np.abs(after["avg_K"] - sumK / A).max() <= 1e-9 * max(1.0, np.abs(sumK / A).max())
    # dual step (update_duals.jl:1-39) from the reported consensus state
    assert np.abs(after["lam"] - (lam0 + g * after["inj"].sum(axis=0))).max() <= 1e-9 * max(1.0, np.abs(after["lam"]).max())
    mu1 = (mu0 + g * (after["flow"] + after["avg_U"] - pp.f_max[:, None])) * (after["avg_U"] <= 1e-2)
    rho1 = (rho0 + g * (after["avg_K"] - after["flow"] - pp.f_max[:, None])) * (after["avg_K"] <= 1e-2)
    assert np.abs(after["mu"] - mu1).max() <= 1e-12 + 1e-9 * np.abs(mu1).max() and np.abs(after["rho"] - rho1).max() <= 1e-12 + 1e-9 * np.abs(rho1).max()
    # storages: bounds and level recursion
    D, C, E = after["D"], after["C"], after["E"]
    assert D.min() >= 0 and C.min() >= 0 and (D - pp.sto_pmax[:, None]).max() <= 0 and (C - pp.sto_pmax[:, None]).max() <= 0
    assert E.min() >= -1e-9 and (E - pp.sto_emax[:, None]).max() <= 1e-9 and np.abs(np.cumsum(C - D, axis=1) - E).max() < 1e-9
    # the run to the stop test, and its optimum
    done, conv = e.iterate(5000)
    assert conv and done < 1500
    inj, aU, aK, flow, cost = e.get_consensus()
    assert abs(cost - opt["objective"]) / opt["objective"] < 1e-3
    assert (np.abs(flow) - pp.f_max[:, None]).max() <= 1e-6 * pp.f_max.max()
    assert e.solver_failures() == 0


def test_hip_config1_reaches_central_optimum(hip_api):
    """BASELINE config 1 (1000 gens + 100 storages x 24): converges (gamma = 1/A) to the LP optimum."""
    pp = synth.baseline_config(1)
    A = pp.G + pp.S
    e = make_engine(hip_api, pp, gamma=1.0 / A, max_iters=5000)
    done, conv = e.iterate(5000)
    assert conv and done < 2000
    opt = solve_central(pp)["objective"]
    assert abs(e.get_consensus()[4] - opt) / opt < 1e-3


@pytest.mark.parametrize("fmax_factor,congested", [(1.5, False), (1.0, True)])
def test_hip_network_reaches_central_optimum(hip_api, fmax_factor, congested):
    """12 nodes / 18 lines, 330 agents x 12: with the flow-consensus weight scaled like gamma (0.3/A; the reference's
    literal 10 flips all agents between their bounds at this size) the run converges to the central LP optimum
    (tests/central_lp.py). With binding line limits (fmax = the merit-order flows) the masked mu/rho update keeps
    moving above the 1e-3 stop test, but cost and flows are at the LP optimum."""
    pp = synth.synthetic_case(300, 30, 12, N=12, L=18, seed=23, fmax_factor=fmax_factor, fmax_min=20)
    A = pp.G + pp.S
    opt = solve_central(pp)["objective"]
    e = make_engine(hip_api, pp, gamma=1.0 / A, w_flow=0.3 / A, eps=1e-3, max_iters=4000)
    done, conv = e.iterate(4000)
    inj, aU, aK, flow, cost = e.get_consensus()
    assert abs(cost - opt) / opt < 1e-3
    # (congested: the iterate keeps wandering around the optimum, up to ~10 % over the limit of a 20 MW line)
    assert ((np.abs(flow) - pp.f_max[:, None]) / pp.f_max[:, None]).max() < (0.2 if congested else 1e-6)
    assert np.abs(inj.sum(axis=0)).max() < 1e-3 * pp.demand.sum(axis=0).max()
    if congested:
        assert e.get_duals()[1].max() > 0.1                 # a line limit binds: its multiplier is up
    else:
        assert conv and done < 1500
    assert e.solver_failures() == 0


def _random_storage_case(rng, T):
    """Copper-plate case with adversarial storage parameters and a random feasible previous iterate."""
    G, S = int(rng.integers(3, 12)), int(rng.integers(4, 24))
    pp = synth.synthetic_case(G, S, T, seed=int(rng.integers(1, 10**6)))
    kind = rng.integers(0, 5, size=S)
    pp.sto_mc[:] = np.where(kind == 0, 0.0, rng.integers(0, 40, size=S))          # no dead band / wide dead band
    pp.sto_pmax[:] = rng.integers(1, 60, size=S).astype(float)
    pp.sto_emax[:] = np.where(kind == 1, pp.sto_pmax * 0.3,                          # fills in a fraction of a step
                              np.where(kind == 2, pp.sto_pmax * 3 * T, pp.sto_pmax * rng.integers(1, 6, size=S)))
    P = rng.uniform(0, 1, size=(G, T)) * pp.gen_pmax[:, None]
    C = np.zeros((S, T))
    D = np.zeros((S, T))
    for s in range(S):                       # random walk inside the level band (so that (D, C) is feasible)
        e = 0.0
        for t in range(T):
            c = rng.uniform(0, min(pp.sto_pmax[s], pp.sto_emax[s] - e)) if rng.random() < 0.5 else 0.0
            d = rng.uniform(0, min(pp.sto_pmax[s], e + c)) if rng.random() < 0.5 else 0.0
            C[s, t], D[s, t] = c, d
            e += c - d
    lam = rng.normal(0, 25, size=T) - 15.0
    return pp, P, D, C, lam


@pytest.mark.parametrize("T,n_cases", [(5, 12), (24, 10), (33, 6), (96, 4), (150, 3)])
def test_hip_storage_kernels_on_random_states(hip_api, oracle_api, T, n_cases):
    """Cold scan kernel (first step after set_state) and warm-start kernel (the steps after) against the
    oracle's exact mode — and against the literally assembled QPs on the short horizons — from random
    feasible states with awkward storages: zero marginal cost, tiny or never-binding level bands,
    strong and weak penalties, non-unit prox weight."""
    rng = np.random.default_rng(1000 + T)
    for i in range(n_cases):
        pp, P, D, C, lam = _random_storage_case(rng, T)
        params = dict(gamma=float(rng.choice([0.003, 0.05, 0.3, 1.0])), w_prox=float(rng.choice([1.0, 0.25, 4.0])), eps=0.0)
        h = make_engine(hip_api, pp, **params)
        engines = [h, make_engine(oracle_api, pp, mode=1, **params)]
        if T <= 5:
            engines.append(make_engine(oracle_api, pp, mode=0, **params))     # the literal QPs as a third opinion
        for e in engines:
            e.set_state(P=P, D=D, C_=C, lam=lam, iteration=7)
        for step in range(4):                 # step 0: cold scan; steps 1-3: warm start where its certificate holds
            for e in engines:
                e.iterate(1)
            st = [state_of(e) for e in engines]
            worst, where = max_diff(st[0], st[1], keys=["D", "C", "E", "P", "lam", "inj"])
            assert worst < 1e-8, (T, i, step, where, worst)
            if T <= 5:                        # interior-point noise of the literal mode: 1e-7 .. 1e-6 per step
                worst, where = max_diff(st[0], st[2], keys=["D", "C", "E", "P", "lam", "inj"])
                assert worst < 2e-5, (T, i, step, where, worst)
        assert h.solver_failures() == 0


@pytest.mark.parametrize("shape", ["fused launch", "storage kernel of big grids", "network"])
def test_hand_over_from_the_active_set_body_to_the_scan_body(hip_api, shape):
    """DOPF_F_DEBUG_LEAVE makes the active-set storage body leave every third storage to the scan body, in the same launch
    (copper plates) or through the non-inlined call (networks). Both are exact solvers of the same QP: the run must equal the
    ordinary one, and the statistics must show that the scan body really served its share."""
    if shape == "fused launch":
        pp = synth.synthetic_case(700, 90, 96, seed=5)
        g, kw = 1.0 / 790, {}
    elif shape == "storage kernel of big grids":
        pp = synth.synthetic_case(900, 120, 24, seed=9)
        g, kw = 1.0 / 1020, dict(flags=_capi.F_NO_FUSE)
    else:
        pp = synth.synthetic_case(400, 60, 24, N=5, L=6, seed=12, fmax_factor=0.8, fmax_min=5)
        g, kw = 0.01, dict(w_flow=0.3 / 460)
    ref = make_engine(hip_api, pp, eps=0.0, gamma=g, **kw)
    fl = kw.pop("flags", 0) | _capi.F_DEBUG_LEAVE
    e = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=fl, **kw)
    for k in (1, 3, 12):
        ref.iterate(k)
        e.iterate(k)
        assert e.warm_start_stats()[1] >= pp.S // 3 and ref.warm_start_stats()[1] < pp.S // 3
        a, b = state_of(e), state_of(ref)
        for key in a:
            if a[key].size:
                assert np.abs(a[key] - b[key]).max() <= 1e-8 * max(1.0, np.abs(b[key]).max()), (shape, key, k)
    assert e.solver_failures() == 0


def test_no_access_past_the_end_of_any_device_array(hip_api):
    """DOPF_GUARD=1 puts every device array at the end of its own mapping: an access past an array's end kills the process
    with a GPU memory fault instead of reading a neighbour. Ragged shapes, shards and whole runs in a child process
    (tests/guard_worker.py; the first case is the one scripts/fuzz_sharded.py tripped over in round 2)."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard_worker.py")
    r = subprocess.run([sys.executable, worker], env=dict(os.environ, DOPF_GUARD="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "guard worker: ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
