"""Long runs without a stop test: the one-launch iteration (tail block polling) and the quiet chain over hundreds of thousands of
iterations — no time-out word, no solver failure, finite state, the iteration count the host asked for.
   python scripts/soak.py [iterations in thousands]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for wl, k in (("config2", K), ("config1", 2 * K), ("config3-share", K // 2), ("config4", K // 8)):
    pp = bench.make_problem(synth, wl); A = pp.G + pp.S
    e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / A, w_flow=10.0 if pp.L == 0 else 0.3 / A, eps=0.0), **pp.engine_kwargs())
    t0 = time.perf_counter(); done = 0
    for chunk in (1, 7, 1000 * k - 8):
        d, conv = e.iterate(chunk); done += d
    dt = time.perf_counter() - t0
    lam = e.get_duals()[0]; cost = e.get_consensus()[4]
    assert done == 1000 * k and not conv, (done, conv)
    assert np.isfinite(lam).all() and np.isfinite(cost) and e.solver_failures() == 0
    print(f"{wl}: {done} iterations in {dt:.2f} s = {1e6 * dt / done:.2f} us/iteration, residuals {e.get_residuals()[:3]}, cost {cost:.6e}", flush=True)
    e.close()
