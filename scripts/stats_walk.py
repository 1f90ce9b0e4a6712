"""Share of (node, line, timestep) entries whose slack sum needs the agent walk (switch point inside the node's window)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
pp = synth.baseline_config(3, scale=float(sys.argv[1]) if len(sys.argv) > 1 else 0.125)
A = pp.G + pp.S
g, wf = 1.0 / A, 0.3 / A
e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=g, w_flow=wf, eps=0.0), **pp.engine_kwargs())
W = np.zeros(pp.N)
np.maximum.at(W, pp.gen_node, pp.gen_pmax); np.maximum.at(W, pp.sto_node, 2 * pp.sto_pmax)
for it in (1, 10, 50, 200, 500):
    e.iterate(it - (e.get_residuals()[3] - 1))
    inj, aU, aK, fl, cost = e.get_consensus()
    w2 = 2 * wf; inv = 1 / (w2 + g)
    a_u = (g * aU - w2 * (fl - pp.f_max[:, None])) * inv          # (L,T)
    a_k = (g * aK + w2 * (fl + pp.f_max[:, None])) * inv
    reach = np.abs(w2 * pp.ptdf * inv) * W[None, :]               # (L,N)
    wu = (a_u[:, None, :] - reach[:, :, None] < 0) & (a_u[:, None, :] + reach[:, :, None] > 0)
    wk = (a_k[:, None, :] - reach[:, :, None] < 0) & (a_k[:, None, :] + reach[:, :, None] > 0)
    on_u = (a_u[:, None, :] - reach[:, :, None] >= 0)
    print(f"after {it} it: walked U {wu.mean():.4f} K {wk.mean():.4f} | all-on U {on_u.mean():.4f} | (l,t) pairs with any walked node: {(wu.any(axis=1) | wk.any(axis=1)).mean():.4f}", flush=True)
