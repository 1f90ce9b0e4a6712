"""CPU only: the oracle's two modes against each other on random small cases (literal = the reference's QPs by an
interior-point method, exact = the slack-eliminated solve), one step at a time from the literal mode's state.
usage: python scripts/fuzz_oracle.py [n_cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, state_of, max_diff
import __graft_entry__ as ge
from oracle.binding import OracleApi
ora = OracleApi(ge.ORACLE_LIB)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst_all, bad, fails = 0.0, 0, 0
t0 = time.time()
for k in range(n_cases):
    net = rng.random() < 0.7
    N = int(rng.integers(2, 6)) if net else 1
    L = int(rng.integers(N - 1, N + 3)) if net else 0
    T = int(rng.choice([2, 3, 5, 8, 12]))
    G, S = int(rng.integers(1, 25)), int(rng.integers(0, 10))
    case = dict(n_gen=G, n_sto=S, T=T, N=N, L=L, seed=int(rng.integers(1, 10**6)))
    if net:
        case.update(fmax_factor=float(rng.choice([0.5, 0.8, 1.0, 1.5])), fmax_min=float(rng.choice([1, 5, 20])))
    try:
        pp = synth.synthetic_case(**case)
    except ValueError:
        continue
    A = G + S
    params = dict(gamma=float(rng.choice([1.0 / A, 0.3 / A, 0.05, 0.3])), w_flow=float(rng.choice([10.0, 1.0, 0.1, 1.0 / A])))
    a = make_engine(ora, pp, mode=0, eps=0.0, **params)
    b = make_engine(ora, pp, mode=1, eps=0.0, **params)
    for it in range(int(rng.integers(3, 12))):
        try:
            a.iterate(1)
        except _capi.DopfError as e:
            fails += 1
            print("LITERAL FAILED", case, params, "iteration", it, e, flush=True)
            break
        b.iterate(1)
        sa, sb = state_of(a), state_of(b)
        scale = max(1.0, float(np.abs(sa["lam"]).max()), float(np.abs(sa["mu"]).max()) if sa["mu"].size else 0.0)
        worst, where = max_diff(sa, sb, keys=[x for x in sa if x != "cost"])
        worst_all = max(worst_all, worst / scale)
        if worst > 1e-5 * scale:
            bad += 1
            print("MISMATCH", case, params, "iteration", it, where, worst, flush=True)
            break
        b.set_state(P=sa["P"], D=sa["D"], C_=sa["C"], avg_U=sa["avg_U"], avg_K=sa["avg_K"], lam=sa["lam"], mu=sa["mu"], rho=sa["rho"],
                    iteration=a.get_residuals()[3])
    a.close(); b.close()
print(f"done: {n_cases} cases, worst relative one-step difference {worst_all:.2e}, mismatches {bad}, literal failures {fails}, {time.time()-t0:.0f}s")
