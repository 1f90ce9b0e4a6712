"""How many kinks of Psi_{n,t} fall inside the node's window (table sizes the agent kernels search) on the network share."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
pp = synth.baseline_config(3, scale=0.125)
A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, w_flow=0.3 / A, eps=0.0), **pp.engine_kwargs())
f = api.lib.dopf_debug_table
M2 = 2 * pp.L
beta = (C.c_double * M2)(); psi = (C.c_double * M2)(); slope = (C.c_double * (M2 + 1))(); psi0 = C.c_double(); m = C.c_int32()
for its in (5, 50, 400):
    e.iterate(its)
    ms = []
    for n in range(0, pp.N, 3):
        for t in range(0, pp.T, 7):
            f(e._ctx, n, t, beta, psi, slope, C.byref(psi0), C.byref(m)); ms.append(m.value)
    ms = np.array(ms)
    print(f"after {e.get_residuals()[3]-1} iterations: table sizes over {len(ms)} sampled (n,t): mean {ms.mean():.2f}, max {ms.max()}, zero {np.mean(ms == 0):.2f}, <=2 {np.mean(ms <= 2):.2f}, <=4 {np.mean(ms <= 4):.2f}", flush=True)
