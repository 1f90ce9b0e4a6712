"""Where a wave of the storage body spends its time in a settled iteration (counters build): stamps at entry, after the
loads + contact kinds, after the segment set-up, after Newton, after the certificate, after the stores, at the end."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
_capi._pin_hip_runtime()
api = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
api.lib.dopf_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
its = int(sys.argv[2]) if len(sys.argv) > 2 else 120
flags = int(sys.argv[3]) if len(sys.argv) > 3 else _capi.F_NO_FUSE
pp = synth.baseline_config(idx); A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0, flags=flags | _capi.F_NO_GRAPH), **pp.engine_kwargs())
e.iterate(its)
n = 8192 * 8
buf = (C.c_uint64 * n)()
assert api.lib.dopf_debug_timeline(e._ctx, buf, n) == 0
tl = np.array(list(buf), dtype=np.float64).reshape(8192, 8)
tl = tl[(tl[:, 0] > 0) & (tl[:, 6] > 0)]
t0 = tl[:, 0].min()
names = ["entry", "loads+kinds", "segments", "newton", "certificate", "stores", "end"]
print(f"{tl.shape[0]} waves; kernel span {10 * (tl[:, 6].max() - t0) / 1e3:.2f} us (10 ns ticks)")
print("stage            mean start   p50 dur   p95 dur   max end")
for k in range(6):
    d = (tl[:, k + 1] - tl[:, k]) * 10 / 1e3
    print(f"{names[k + 1]:14s} {10 * (tl[:, k].mean() - t0) / 1e3:10.2f} {np.percentile(d, 50):9.2f} {np.percentile(d, 95):9.2f} {10 * (tl[:, k + 1].max() - t0) / 1e3:9.2f}")
print("wave entry times: p5 %.2f p50 %.2f p95 %.2f us" % tuple(10 * (np.percentile(tl[:, 0], q) - t0) / 1e3 for q in (5, 50, 95)))
