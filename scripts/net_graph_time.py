"""Wall time per iteration of the graph path over windows (python scripts/net_graph_time.py [workload] [windows] [len] [flags])"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "config3-share"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wlen = int(sys.argv[3]) if len(sys.argv) > 3 else 48
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if wl.startswith("config3"):
    pp = synth.baseline_config(3, scale=0.125 if wl == "config3-share" else 1.0)
else:
    pp = synth.baseline_config(int(wl[-1]))
A = pp.G + pp.S
e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / A, w_flow=10.0 if pp.L == 0 else 0.3 / A, eps=0.0, flags=flags), **pp.engine_kwargs())
e.iterate(0)
it = 0
for w in range(nw):
    t0 = time.perf_counter(); e.iterate(wlen); dt = time.perf_counter() - t0
    it += wlen
    print(f"its {it - wlen + 1}..{it}: {1e6 * dt / wlen:.1f} us/iteration", flush=True)
for pause in (0.0, 0.2, 1.0):
    time.sleep(pause)
    t0 = time.perf_counter(); e.iterate(200); dt = time.perf_counter() - t0
    print(f"after a {pause:.1f} s pause: 200 its at {1e6 * dt / 200:.1f} us/iteration", flush=True)
