import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
hipapi = _capi.hip_api()
oapi = _capi.CApi("oracle/libdopf_oracle.so", "oracle_")
def compare(pp, iters, mode=1, tol=1e-8, **kw):
    h = _capi.Engine(hipapi, params=_capi.default_params(eps=0.0, **kw), **pp.engine_kwargs())
    o = _capi.Engine(oapi, params=_capi.default_params(eps=0.0, **kw), mode=mode, **pp.engine_kwargs())
    worst = 0.0
    for k in range(iters):
        h.iterate(1); o.iterate(1)
        errs = [np.abs(a - b).max() if a.size else 0.0 for a, b in zip(h.get_primal() + h.get_duals() + h.get_consensus()[:4], o.get_primal() + o.get_duals() + o.get_consensus()[:4])]
        worst = max(worst, max(errs))
        if max(errs) > tol:
            print("  iter", k + 1, "errs", ["%.1e" % e for e in errs]); break
    print("  worst %.2e fails %d" % (worst, h.solver_failures()))
nodes, lines, gens, stos = pkg.three_node_case()
print("three-node"); compare(pkg.pack(nodes, gens, stos, lines), 60, mode=0, tol=1e-6)
print("copper 20/5/6"); compare(synth.synthetic_case(20, 5, 6), 40, gamma=0.05)
print("copper 100/30/24"); compare(synth.synthetic_case(100, 30, 24, seed=2), 40, gamma=0.01)
print("copper 50/20/96"); compare(synth.synthetic_case(50, 20, 96, seed=3), 20, gamma=0.02)
print("copper 50/20/168"); compare(synth.synthetic_case(50, 20, 168, seed=4), 10, gamma=0.02)
print("copper 50/20/40"); compare(synth.synthetic_case(50, 20, 40, seed=5), 10, gamma=0.02)
print("net 12/4/5"); compare(synth.synthetic_case(12, 4, 5, N=4, L=5, seed=5, fmax_factor=0.7, fmax_min=5), 40, gamma=0.1)
print("net 60/15/24 N6 L9"); compare(synth.synthetic_case(60, 15, 24, N=6, L=9, seed=7, fmax_factor=0.6, fmax_min=5), 30, gamma=0.05)
import __graft_entry__ as ge
ge.smoke()
