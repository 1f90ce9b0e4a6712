"""The lean copper-plate storage body (csrc/sto_lean.h: horizons that fill the lane groups — 8, 16, 24, 48, 96 steps) against
the oracle's exact mode AND against the general active-set body (DOPF_F_STO_GENERAL) on the same cases: free runs from the
zero state (many contact-set rounds in the first iterations, one round per solve later), compared after every call.
usage: python scripts/fuzz_lean.py [n_cases] [seed]"""
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
F_GENERAL = 16384
worst_o, worst_g, bad, left_total = 0.0, 0.0, 0, 0
t0 = time.time()
for k in range(n_cases):
    T = int(rng.choice([8, 16, 24, 24, 48, 96, 96, 5, 12, 20, 30, 36, 72, 120, 168]))       # full and ragged horizons, 8 to 64 lanes per storage
    G, S = int(rng.integers(5, 150)), int(rng.integers(3, 90))
    if rng.random() < 0.1:
        S = int(rng.integers(300, 1200))          # several storage items, more than one pass per block
    case = dict(n_gen=G, n_sto=S, T=T, N=1, L=0, seed=int(rng.integers(1, 10**6)))
    pp = synth.synthetic_case(**case)
    shape = "2"
    if rng.random() < 0.6:                      # vary the storage shapes: not only emax = 2 pmax
        pp.sto_emax = pp.sto_pmax * rng.choice([0.7, 1.0, 2.0, 3.3, 5.0], size=S)
        shape = "mixed"
    if rng.random() < 0.3:                      # continuous parameters: no two kinks coincide
        pp.sto_pmax = pp.sto_pmax * rng.uniform(0.6, 1.4, size=S)
        pp.sto_mc = pp.sto_mc * rng.uniform(0.5, 1.5, size=S)
        shape += "+continuous"
    A = G + S
    params = dict(gamma=float(rng.choice([1.0, 0.3, 1.5])) / A, w_flow=10.0)
    flags = int(rng.choice([0, 0, 16, 4096, 16 | 4096]))       # fused or separate launches, tail in the launch or not
    h = make_engine(hip, pp, eps=0.0, flags=flags, **params)
    g = make_engine(hip, pp, eps=0.0, flags=flags | F_GENERAL, **params)
    o = make_engine(ora, pp, mode=1, eps=0.0, **params)
    iters = int(rng.integers(30, 120))
    w_o = w_g = 0.0
    it = 0
    while it < iters:
        n = 1 if it < 12 or rng.random() < 0.5 else int(rng.integers(2, 6))
        h.iterate(n); g.iterate(n); o.iterate(n)
        it += n
        sh, sg, so = state_of(h), state_of(g), state_of(o)
        scale = max(1.0, float(np.abs(so["lam"]).max()))
        keys = ["P", "D", "C", "E", "lam", "inj"]
        wo, where_o = max_diff(sh, so, keys=keys)
        wg, where_g = max_diff(sh, sg, keys=keys)
        w_o, w_g = max(w_o, wo / scale), max(w_g, wg / scale)
        left_total += h.warm_start_stats()[1]
        if wo > 1e-6 * scale or wg > 1e-6 * scale:
            bad += 1
            print("MISMATCH", case, params, "flags", flags, "shape", shape, "iteration", it, "vs oracle", where_o, wo,
                  "vs general body", where_g, wg, flush=True)
            break
    if h.solver_failures():
        bad += 1
        print("SOLVER FAILURES", case, params, h.solver_failures(), flush=True)
    if max(w_o, w_g) > float(os.environ.get("FUZZ_REPORT", "1e-7")):
        print("LARGE", case, params, "flags", flags, "shape", shape, "iterations", iters, "vs oracle %.2e vs general body %.2e" % (w_o, w_g), "left", left_total, flush=True)
    worst_o, worst_g = max(worst_o, w_o), max(worst_g, w_g)
    h.close(); g.close(); o.close()
    if k % 10 == 9:
        print(f"{k+1} cases, worst relative difference vs oracle {worst_o:.2e}, vs the general body {worst_g:.2e}, bad {bad}, {time.time()-t0:.0f}s", flush=True)
print(f"done: {n_cases} cases, worst {worst_o:.2e} (oracle) {worst_g:.2e} (general body), bad {bad}, storage solves left to the scan body over all iterations: {left_total}")
