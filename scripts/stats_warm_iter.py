"""Per-iteration warm-start failures and scan counts (needs the counters build, see scripts/README.md)."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 140
pp = synth.baseline_config(idx); A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0/A, eps=0.0), **pp.engine_kwargs())
def stats():
    out = (C.c_uint64 * 9)(); api.lib.dopf_debug_stats(e._ctx, out); return np.array(list(out), dtype=np.float64)
s0 = stats()
for it in range(n):
    e.iterate(1); s1 = stats(); d = s1 - s0; s0 = s1
    fail = s1[4]
    if fail > 0:
        print(f"it {it+1}: fail {int(fail)} of {pp.S} | noprices {int(d[5])} newton {int(d[6])} level {int(d[7])} sign {int(d[8])} | scans {int(d[0])} = {d[0]/max(fail,1):.1f}/failed storage, wave loops {int(d[1])}", flush=True)
