import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
LPS = {1: 8, 2: 32, 4: 8}
for idx, gam in [(1, None), (2, None), (4, None), (1, 1.0)]:
    pp = synth.baseline_config(idx)
    A = pp.G + pp.S
    g = gam if gam else 1.0 / A
    e = _capi.Engine(api, params=_capi.default_params(gamma=g, eps=0.0), **pp.engine_kwargs())
    def stats():
        out = (C.c_uint64 * 3)(); api.lib.dopf_debug_stats(e._ctx, out); return np.array(list(out), dtype=np.float64)
    for (n0, n1) in [(0, 1), (1, 50), (50, 400)]:
        s0 = stats(); e.iterate(n1 - n0); s1 = stats(); d = (s1 - s0) / (n1 - n0)
        waves = pp.S / (64 / LPS[idx])
        print(f"config{idx} gamma={g:.2e} iters {n0}-{n1}: scans/storage {d[0]/pp.S:.1f} events/storage {d[2]/pp.S:.2f} wave-loop trips/wave {d[1]/waves:.1f}")
