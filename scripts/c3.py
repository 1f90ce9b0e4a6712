import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
pp = synth.baseline_config(3, scale=0.125); A = pp.G + pp.S
for mult, wf in ((0.1, 10.0), (0.02, 10.0), (0.1, 1000.0)):
    e = _capi.Engine(api, params=_capi.default_params(gamma=mult / A, w_flow=wf, eps=1e-3, max_iters=4000), **pp.engine_kwargs())
    for n in (10, 90, 400, 1500, 2000):
        t0 = time.perf_counter(); done, conv = e.iterate(n); dt = time.perf_counter() - t0
        r = e.get_residuals()
        print(f"gamma={mult}/A w_flow={wf} it={r[3]} conv={conv} res=({r[0]:.2e},{r[1]:.2e},{r[2]:.2e}) warm={e.warm_start_stats()} {1e3*dt/max(done,1):.2f} ms/it", flush=True)
        if conv: break
