import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
for idx in (1, 2):
    pp = synth.baseline_config(idx); A = pp.G + pp.S
    e = _capi.Engine(api, params=_capi.default_params(gamma=1.0/A, eps=0.0), **pp.engine_kwargs())
    def stats():
        out = (C.c_uint64 * 9)(); api.lib.dopf_debug_stats(e._ctx, out); return np.array(list(out), dtype=np.float64)
    for (n0, n1) in [(0, 1), (1, 2), (2, 10), (10, 100), (100, 400)]:
        s0 = stats(); e.iterate(n1 - n0); s1 = stats(); d = (s1 - s0) / (n1 - n0) / pp.S
        print(f"config{idx} iters {n0}-{n1}: warm ok {d[3]:.3f} fail {d[4]:.3f} | noprices {d[5]:.3f} newton {d[6]:.3f} level {d[7]:.3f} sign {d[8]:.3f} | scans/storage {d[0]:.1f}")
