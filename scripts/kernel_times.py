"""Per-kernel HIP-event times (dopf_iterate_timed, medians of 5 x 50 iterations after 300) of one workload for one or more builds of the
library: usage: python scripts/kernel_times.py <workload> <lib.so> [<lib.so> ...]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
_capi._pin_hip_runtime()
wl = sys.argv[1]
pp = bench.make_problem(synth, wl); A = pp.G + pp.S
for path in sys.argv[2:]:
    api = _capi.CApi(path, "dopf_")
    e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, w_flow=0.3 / A, eps=0.0), **pp.engine_kwargs())
    e.iterate(300)
    r = []
    for k in range(5):
        t = e.iterate_timed(50); r.append(t)
    med = {k: sorted(x[k] for x in r)[2] for k in ("gen_ms", "sto_ms", "slack_ms", "reduce_ms", "dual_ms", "iter_ms")}
    print(wl, path, {k: round(v * 1e3, 2) for k, v in med.items()}, "quiet", r[0]["quiet"], flush=True)
    e.close()
