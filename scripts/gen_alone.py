import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stamps.so", "dopf_")
api.lib.dopf_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
S = int(sys.argv[1])
pp = synth.synthetic_case(45455, S, 96, seed=1); A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0, flags=_capi.F_NO_GRAPH), **pp.engine_kwargs())
e.iterate(150)
n = 8192 * 16
buf = (C.c_uint64 * n)()
assert api.lib.dopf_debug_timeline(e._ctx, buf, n) == 0
tl = np.array(list(buf), dtype=np.float64)[32768:].reshape(-1, 2)
tl = tl[(tl[:, 0] > 0) & (tl[:, 1] > 0)]
t0 = tl[:, 0].min()
d = (tl[:, 1] - tl[:, 0]) / 100
print(S, "blocks", len(tl), "kernel span %.2f us" % ((tl[:, 1].max() - t0) / 100), "durations: p5 %.2f p50 %.2f p95 %.2f max %.2f" % tuple(np.percentile(d, [5, 50, 95, 100])))
tm = e.iterate_timed(16)
print("k_agents %.2f us" % (1e3 * (tm["gen_ms"] - tm["empty_ms"])), "fused", tm["agents_fused"])
