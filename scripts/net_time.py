"""Per-kernel times of the network workloads over windows of the run (HIP events, eager launches):
python scripts/net_time.py [config3-share|config3] [n windows] [window length]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "config3-share"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wlen = int(sys.argv[3]) if len(sys.argv) > 3 else 50
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pp = synth.baseline_config(3, scale=0.125 if wl == "config3-share" else 1.0)
A = pp.G + pp.S
e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / A, w_flow=0.3 / A, eps=0.0, flags=flags), **pp.engine_kwargs())
it = 0
for w in range(nw):
    t = e.iterate_timed(wlen)
    it += wlen
    em = t["empty_ms"]
    print(f"its {it - wlen + 1}..{it}: iter {1e3 * t['iter_ms']:.0f} us | tables {1e3 * (t['tables_ms'] - em):.0f} gen {1e3 * (t['gen_ms'] - em):.0f} "
          f"sto {1e3 * (t['sto_ms'] - em):.0f} slack {1e3 * (t['slack_ms'] - em):.0f} reduce {1e3 * (t['reduce_ms'] - em):.0f} dual {1e3 * (t['dual_ms'] - em):.0f} | "
          f"left to scan {e.warm_start_stats()[1]} res {e.get_residuals()[0]:.2e}", flush=True)
