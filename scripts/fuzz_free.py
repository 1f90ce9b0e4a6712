"""Random cases run FREELY side by side (no state resets, so the warm-start kernel and its certificate carry the
storages after the first iterations): HIP vs oracle exact mode, compared after every iteration.
usage: python scripts/fuzz_free.py [n_cases] [seed]"""
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
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst_all, bad, warm_used, left_total = 0.0, 0, 0, 0
t0 = time.time()
for k in range(n_cases):
    net = rng.random() < 0.35
    N = int(rng.integers(2, 7)) if net else 1
    L = int(rng.integers(N - 1, 2 * N + 1)) if net else 0
    T = int(rng.choice([4, 8, 12, 24, 24, 36, 48, 96]))
    G, S = int(rng.integers(5, 150)), int(rng.integers(3, 60))
    case = dict(n_gen=G, n_sto=S, T=T, N=N, L=L, seed=int(rng.integers(1, 10**6)))
    if net:
        case.update(fmax_factor=float(rng.choice([1.0, 1.5])), fmax_min=20.0)
    try:
        pp = synth.synthetic_case(**case)
    except ValueError:
        continue
    if rng.random() < 0.5:                      # vary the storage shapes: not only emax = 2 pmax
        pp.sto_emax = pp.sto_pmax * rng.choice([0.7, 1.0, 2.0, 3.3, 5.0], size=S)
    A = G + S
    params = dict(gamma=float(rng.choice([1.0, 0.3])) / A, w_flow=(float(rng.choice([0.1, 0.03])) / A if net else 10.0))   # convergent settings:
    # in a 2-cycle (gamma = 2/A, or the literal flow weight) rounding differences double every iteration
    h = make_engine(hip, pp, eps=0.0, **params)
    o = make_engine(ora, pp, mode=1, eps=0.0, **params)
    iters = int(rng.integers(40, 160))
    w_case = 0.0
    for it in range(iters):
        h.iterate(1); o.iterate(1)
        sh, so = state_of(h), state_of(o)
        scale = max(1.0, float(np.abs(so["lam"]).max()))
        worst, where = max_diff(sh, so, keys=["P", "D", "C", "E", "lam", "mu", "rho", "inj"])
        w_case = max(w_case, worst / scale)
        left_total += h.warm_start_stats()[1]
        if worst > 1e-5 * scale:
            bad += 1
            print("MISMATCH", case, params, "emax/pmax", sorted(set(np.round(pp.sto_emax / pp.sto_pmax, 2).tolist())), "iteration", it, where, worst, flush=True)
            break
    ok, fb = h.warm_start_stats()
    warm_used += ok > 0
    if h.solver_failures():
        bad += 1
        print("SOLVER FAILURES", case, params, h.solver_failures(), flush=True)
    worst_all = max(worst_all, w_case)
    h.close(); o.close()
    if k % 10 == 9:
        print(f"{k+1} cases, worst relative difference along the runs {worst_all:.2e}, bad {bad}, warm start carried {warm_used} cases at the end, {time.time()-t0:.0f}s", flush=True)
print(f"done: {n_cases} cases, worst {worst_all:.2e}, bad {bad}, storage solves left to the scan kernel over all iterations: {left_total}")
