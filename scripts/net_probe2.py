import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
pp = synth.baseline_config(3, scale=0.125)
A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0), **pp.engine_kwargs())
for n in (1, 1, 1, 7, 40, 250):
    e.iterate(n)
    P, D, C, E = e.get_primal()
    lam, mu, rho = e.get_duals()
    inj, aU, aK, flow, cost = e.get_consensus()
    gen = P.sum(axis=0); dem = pp.demand.sum(axis=0)
    s = inj.sum(axis=0)
    t = int(np.argmax(np.abs(s)))
    over = np.abs(flow) - pp.f_max[:, None]
    print(f"it {e.get_residuals()[3]-1}: max|imbalance| {np.abs(s).max():.4g} at t={t} (demand {dem[t]:.4g}, gen {gen[t]:.4g}, sto {(D-C).sum(axis=0)[t]:.4g}) "
          f"lam[t] {lam[t]:.4g} lam range [{lam.min():.4g},{lam.max():.4g}] mu max {mu.max():.4g} rho max {rho.max():.4g} "
          f"flow over max {over.max():.4g} avgU max {aU.max():.4g} avgK max {aK.max():.4g} gens at pmax {(P >= pp.gen_pmax[:,None]).mean():.3f} at 0 {(P<=0).mean():.3f}", flush=True)
print("f_max", pp.f_max.min(), pp.f_max.max(), "ptdf absmax", np.abs(pp.ptdf).max())
