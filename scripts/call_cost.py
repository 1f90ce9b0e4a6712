"""Fixed cost of a dopf_iterate call (config2, settled): status read-back alone (n = 0), calls of 1, 4, 16, 20 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
pp = bench.make_problem(synth, "config2"); A = pp.G + pp.S
e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / A, eps=0.0), **pp.engine_kwargs())
e.iterate(300)
for n in (0, 1, 4, 16, 20, 64, 400):
    best = 1e9
    for _ in range(50 if n < 100 else 10):
        t0 = time.perf_counter(); e.iterate(n); best = min(best, time.perf_counter() - t0)
    print(f"iterate({n}): {best*1e6:.1f} us per call" + (f" = {best*1e6/n:.2f} us/iteration" if n else ""), flush=True)
