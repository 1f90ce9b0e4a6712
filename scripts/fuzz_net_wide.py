"""Random networks whose consensus state is beyond the one-block dual kernel (max(N, L) * T > 4096: k_net_agents, k_slack,
k_dual_price_t1024 with the slack sums and the tables, up to 256 nodes / 256 lines; beyond that k_dual_t + k_price_t), HIP against
the oracle's exact mode: a few one-step comparisons from the oracle's state, then a short free run of both.
usage: python scripts/fuzz_net_wide.py [n_cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, state_of, max_diff
import __graft_entry__ as ge
hip = _capi.CApi(os.environ["DOPF_LIB"], "dopf_") if os.environ.get("DOPF_LIB") else _capi.hip_api()
from oracle.binding import OracleApi, set_threads
ora = OracleApi(ge.ORACLE_LIB)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst_all, bad, fused = 0.0, 0, 0
t0 = time.time()
for k in range(n_cases):
    N = int(rng.choice([12, 30, 64, 65, 118, 128, 129, 200, 256, 300]))
    L = int(min(rng.integers(N - 1, 2 * N), rng.choice([256, 256, 400])))
    T = int(rng.choice([24, 48, 73, 96, 168]))
    while max(N, L) * T <= 4096:
        T *= 2
    T = min(T, 192)
    G, S = int(rng.integers(20, 160)), int(rng.integers(2, 30))
    case = dict(n_gen=G, n_sto=S, T=T, N=N, L=L, seed=int(rng.integers(1, 10**6)), fmax_factor=float(rng.choice([0.6, 0.8, 1.0, 1.5])),
                fmax_min=float(rng.choice([1, 5, 20])))
    try:
        pp = synth.synthetic_case(**case)
    except ValueError:
        continue
    A = G + S
    params = dict(gamma=float(rng.choice([1.0 / A, 0.3 / A, 3.0 / A, 0.05])), w_flow=float(rng.choice([10.0, 0.1, 0.3 / A, 1.0 / A])))
    h = make_engine(hip, pp, eps=0.0, **params)
    o = make_engine(ora, pp, mode=1, eps=0.0, **params)
    set_threads(o, 8)
    fused += h.iterate_timed(1)["agents_fused"]
    o.iterate(1)
    w_case, ok = 0.0, True
    def check(tag):
        global w_case, ok
        sh, so = state_of(h), state_of(o)
        scale = max(1.0, float(np.abs(so["lam"]).max()), float(np.abs(so["mu"]).max()))
        worst, where = max_diff(sh, so, keys=[x for x in sh if x != "cost"])
        w_case = max(w_case, worst / scale)
        if worst > 1e-6 * scale:
            print("MISMATCH", case, params, tag, where, worst, flush=True)
            ok = False
        return so
    check("iteration 1")
    for it in range(3):                      # one-step comparisons from the oracle's state
        if not ok: break
        so = state_of(o)
        h.set_state(P=so["P"], D=so["D"], C_=so["C"], avg_U=so["avg_U"], avg_K=so["avg_K"], lam=so["lam"], mu=so["mu"], rho=so["rho"],
                    iteration=o.get_residuals()[3])
        h.iterate(1); o.iterate(1)
        check(f"one step {it}")
    for it in range(4):                      # free run
        if not ok: break
        h.iterate(2); o.iterate(2)
        check(f"free run {it}")
    if not ok or h.solver_failures():
        bad += 1
        if h.solver_failures(): print("SOLVER FAILURES", case, params, h.solver_failures(), flush=True)
    worst_all = max(worst_all, w_case)
    h.close(); o.close()
    if k % 5 == 4:
        print(f"{k+1} cases, worst relative difference {worst_all:.2e}, bad {bad}, one launch for all agents in {fused}, {time.time()-t0:.0f}s", flush=True)
print(f"done: {n_cases} cases, worst {worst_all:.2e}, bad {bad}")
