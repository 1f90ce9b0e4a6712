"""Per-iteration duration of the x-update launch (HIP events, product build) beside the storage solve's statistics
(counters build): python scripts/stats_iter_time.py [config] [iterations] [gamma*A]"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gmul = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
pp = synth.baseline_config(idx); A = pp.G + pp.S
_capi._pin_hip_runtime()
sapi = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
es = _capi.Engine(sapi, params=_capi.default_params(gamma=gmul/A, eps=0.0), **pp.engine_kwargs())
e = _capi.Engine(_capi.CApi(os.environ["DOPF_LIB"], "dopf_") if os.environ.get("DOPF_LIB") else _capi.hip_api(), params=_capi.default_params(gamma=gmul/A, eps=0.0), **pp.engine_kwargs())
def stats():
    out = (C.c_uint64 * 15)(); sapi.lib.dopf_debug_stats(es._ctx, out); return np.array(list(out), dtype=np.float64)
s0 = stats()
for it in range(n):
    t = e.iterate_timed(1)
    es.iterate(1); s1 = stats(); d = s1 - s0; s0 = s1
    print(f"it {it+1}: x-update {1e3*(t['gen_ms']-t['empty_ms']):.1f} us sto {1e3*(t['sto_ms']-t['empty_ms']):.1f} us iter {1e3*t['iter_ms']:.1f} us | left {int(s1[4])} rounds/sto {d[5]/pp.S:.2f} newton/sto {d[6]/pp.S:.2f} (cumulative max per lane group: rounds {int(s1[7])} newton {int(s1[8])}) failed rounds: level {int(d[0])} sign {int(d[1])} newton {int(d[2])} | Mcycles A {d[9]/1e6:.2f} B {d[10]/1e6:.2f} C {d[11]/1e6:.2f} D {d[12]/1e6:.2f}", flush=True)
