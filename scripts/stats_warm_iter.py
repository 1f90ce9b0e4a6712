"""Per-iteration statistics of the storage solve (needs the counters build, see scripts/README.md):
storages left to the scan kernel, contact-set rounds and Newton iterations of the active-set body, scans of the scan kernel."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stats.so", "dopf_")
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 140
gmul = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
pp = synth.baseline_config(idx); A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=gmul/A, eps=0.0), **pp.engine_kwargs())
def stats():
    out = (C.c_uint64 * 15)(); api.lib.dopf_debug_stats(e._ctx, out); return np.array(list(out), dtype=np.float64)
s0 = stats()
for it in range(n):
    e.iterate(1); s1 = stats(); d = s1 - s0; s0 = s1
    fail = s1[4]
    print(f"it {it+1}: left to the scan {int(fail)} of {pp.S} | rounds/storage {d[5]/pp.S:.2f} newton/storage {d[6]/pp.S:.2f} | scans {int(d[0])} = {d[0]/max(fail,1):.1f}/failed storage", flush=True)
print("solver failures", e.solver_failures())
