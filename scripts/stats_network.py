import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
pp = synth.baseline_config(3, scale=0.125); A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0), **pp.engine_kwargs())
def stats():
    out = (C.c_uint64 * 9)(); api.lib.dopf_debug_stats(e._ctx, out); return np.array(list(out), dtype=np.float64)
for (n0, n1) in [(0, 1), (1, 5), (5, 30)]:
    s0 = stats(); t0 = time.perf_counter(); e.iterate(n1 - n0); dt = time.perf_counter() - t0; s1 = stats(); d = (s1 - s0) / (n1 - n0)
    print(f"iters {n0}-{n1}: scans/storage {d[0]/pp.S:.1f} events/storage {d[2]/pp.S:.1f} wave-trips {d[1]:.0f} | reasons noprice {d[5]/pp.S:.2f} newton {d[6]/pp.S:.2f} level {d[7]/pp.S:.2f} sign {d[8]/pp.S:.2f} | {1e3*dt/(n1-n0):.1f} ms/it", flush=True)
