"""Network case (config3 shape at a 1/8 share): per-kernel times and residual history for a few penalties."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
pp = synth.baseline_config(3, scale=0.125)
A = pp.G + pp.S
print("agents", A, "T", pp.T, "N", pp.N, "L", pp.L, flush=True)
for arg in sys.argv[1:] or ['1,10']:
    mult, wf = (float(x) for x in arg.split(','))
    e = _capi.Engine(api, params=_capi.default_params(gamma=mult / A, w_flow=wf / A, eps=1e-3, max_iters=100000), **pp.engine_kwargs())
    t0 = time.perf_counter(); done = 0
    for k in range(12):
        d, conv = e.iterate(101); done += d
        r = e.get_residuals()
        print(f"gamma={mult}/A w_flow={wf}/A it {done}: res {r[0]:.3e} {r[1]:.3e} {r[2]:.3e} cost {e.get_consensus()[4]:.6e} warm {e.warm_start_stats()} {time.perf_counter()-t0:.1f}s", flush=True)
        if k % 4 == 3: print('   ', {k2: round(v * 1e3, 1) for k2, v in e.iterate_timed(4).items() if k2.endswith('_ms')}, flush=True)
        if conv: break
    e.close()
