import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
pp = synth.baseline_config(3, scale=0.125); A = pp.G + pp.S
for mult in (1.0, 0.25):
    e = _capi.Engine(api, params=_capi.default_params(gamma=mult / A, eps=1e-3, max_iters=3000), **pp.engine_kwargs())
    for n in (1, 9, 40, 150, 300, 500):
        t0 = time.perf_counter(); done, conv = e.iterate(n); dt = time.perf_counter() - t0
        r = e.get_residuals()
        print(f"gamma={mult}/A it={r[3]} conv={conv} res=({r[0]:.2e},{r[1]:.2e},{r[2]:.2e}) warm={e.warm_start_stats()} {1e3*dt/max(done,1):.2f} ms/it fails={e.solver_failures()}", flush=True)
