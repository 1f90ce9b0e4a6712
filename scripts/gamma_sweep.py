"""Penalty sweep (BASELINE configs[4]: "1M agents x 24 timesteps, rho penalty sweep"): iterations and wall time to the
1e-3 residual, iteration rate, for gamma = m/A.  usage: python scripts/gamma_sweep.py [config index = 4] [cap = 6000]"""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
pp = synth.baseline_config(idx)
A = pp.G + pp.S
rows = []
for m in (0.03, 0.1, 0.3, 1.0, 1.5, 2.0, 3.0, 10.0, 0.3 * A, 1.0 * A):      # the last two: the reference's 0.3 and BASELINE's "1.0"
    e = _capi.Engine(api, params=_capi.default_params(gamma=m / A, eps=1e-3, max_iters=cap), **pp.engine_kwargs())
    e.iterate(0)
    t0 = time.perf_counter(); done, conv = 0, False
    while not conv and done < cap:
        d, conv = e.iterate(min(64, cap - done)); done += d
        if d == 0: break
    dt = time.perf_counter() - t0
    r = e.get_residuals()
    rows.append(dict(gamma_times_A=m, gamma=m / A, iterations=done, converged=bool(conv), seconds=dt, iters_per_sec=done / dt,
                     lam_residual=r[0], total_cost=e.get_consensus()[4], warm_start=e.warm_start_stats()))
    print(json.dumps(rows[-1]), flush=True)
    e.close()
