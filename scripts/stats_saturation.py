import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.hip_api()
for idx in (1, 2, 4):
    pp = synth.baseline_config(idx); A = pp.G + pp.S
    e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0), **pp.engine_kwargs())
    for n in (10, 40, 150, 400):
        e.iterate(n)
        P = e.get_primal()[0]
        z = (P == 0).all(axis=1); f = (P == pp.gen_pmax[:, None]).all(axis=1)
        print(f"config{idx} after {e.get_residuals()[3]-1} it: all-zero {z.mean():.3f} all-pmax {f.mean():.3f} mixed {1 - z.mean() - f.mean():.3f}", flush=True)
