"""Random small cases, one-step parity HIP vs oracle (exact mode), far more seeds than the test-suite carries.
usage: python scripts/fuzz_parity.py [n_cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, state_of, max_diff
import __graft_entry__ as ge
hip = _capi.CApi(os.environ["DOPF_LIB"], "dopf_") if os.environ.get("DOPF_LIB") else _capi.hip_api()
from oracle.binding import OracleApi
ora = OracleApi(ge.ORACLE_LIB)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
MODE = int(os.environ.get("ORACLE_MODE", "1"))      # 0: the literal mode (interior point, keep the cases small), 1: exact
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst_all, bad = 0.0, 0
t0 = time.time()
for k in range(n_cases):
    net = rng.random() < 0.6
    N = int(rng.integers(2, 9)) if net else int(rng.choice([1, 1, 3]))
    L = int(rng.integers(N - 1, 2 * N + 1)) if net else 0
    T = int(rng.choice([2, 3, 5, 7, 8, 12, 24, 30, 48, 96, 168])) if MODE else int(rng.choice([2, 3, 5, 8, 12]))
    big = MODE and rng.random() < 0.15
    G, S = int(rng.integers(1, 400 if big else (80 if MODE else 25))), int(rng.integers(0, 80 if big else (25 if MODE else 10)))
    case = dict(n_gen=G, n_sto=S, T=T, N=N, L=L, seed=int(rng.integers(1, 10**6)))
    if net:
        case.update(fmax_factor=float(rng.choice([0.5, 0.8, 1.0, 1.5])), fmax_min=float(rng.choice([1, 5, 20])))
    try:
        pp = synth.synthetic_case(**case)
    except ValueError:
        continue
    A = G + S
    params = dict(gamma=float(rng.choice([1.0 / A, 0.3 / A, 3.0 / A, 0.05, 0.3])), w_flow=float(rng.choice([10.0, 1.0, 0.1, 1.0 / A, 0.1 / A])))
    h = make_engine(hip, pp, eps=0.0, **params)
    o = make_engine(ora, pp, mode=MODE, eps=0.0, **params)
    iters = int(rng.integers(5, 30))
    w_case = 0.0
    for it in range(iters):
        h.iterate(1); o.iterate(1)
        sh, so = state_of(h), state_of(o)
        scale = max(1.0, float(np.abs(so["lam"]).max()), float(np.abs(so["mu"]).max()) if so["mu"].size else 0.0)
        worst, where = max_diff(sh, so, keys=[x for x in sh if x != "cost"])
        w_case = max(w_case, worst / scale)
        if worst > (1e-6 if MODE else 1e-5) * scale:
            bad += 1
            print("MISMATCH", case, params, "iteration", it, where, worst, flush=True)
            break
        h.set_state(P=so["P"], D=so["D"], C_=so["C"], avg_U=so["avg_U"], avg_K=so["avg_K"], lam=so["lam"], mu=so["mu"], rho=so["rho"],
                    iteration=o.get_residuals()[3])
    if h.solver_failures():
        bad += 1
        print("SOLVER FAILURES", case, params, h.solver_failures(), flush=True)
    worst_all = max(worst_all, w_case)
    h.close(); o.close()
    if k % 20 == 19:
        print(f"{k+1} cases, worst relative one-step difference {worst_all:.2e}, bad {bad}, {time.time()-t0:.0f}s", flush=True)
print(f"done: {n_cases} cases, worst {worst_all:.2e}, bad {bad}")
