import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from central_lp import solve_central
api = _capi.hip_api()
for idx, mult in [(1, 1.0), (2, 1.0), (2, 4.0), (4, 1.0), (4, 4.0)]:
    pp = synth.baseline_config(idx); A = pp.G + pp.S
    e = _capi.Engine(api, params=_capi.default_params(gamma=mult / A, max_iters=60000), **pp.engine_kwargs())
    e.iterate(16); 
    t0 = time.perf_counter(); done, conv = e.iterate(60000); dt = time.perf_counter() - t0
    r = e.get_residuals(); cost = e.get_consensus()[4]
    print(f"config{idx} gamma={mult}/A: converged={conv} iterations={r[3]} time={dt:.3f}s res={r[0]:.2e} cost={cost:.6e} warm={e.warm_start_stats()}", flush=True)
    if idx in (1, 2):
        t0 = time.perf_counter(); opt = solve_central(pp)["objective"]; print("   central LP", opt, "rel diff", abs(cost - opt) / opt, f"({time.perf_counter()-t0:.1f}s)", flush=True)
