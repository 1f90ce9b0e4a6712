"""Per-kernel HIP-event times (settled state) of the network workloads: python scripts/net_kernels.py config3-share config3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
for wl in sys.argv[1:] or ["config3-share"]:
    pp = bench.make_problem(synth, wl); A = pp.G + pp.S
    e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / A, w_flow=0.3 / A, eps=0.0), **pp.engine_kwargs())
    a = torch.randn(2048, 2048, device="cuda")
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        b = a @ a
        torch.cuda.synchronize()
    e.iterate(250)
    tm = e.iterate_timed(32)
    print(wl, {k: round(1e3 * (v - tm["empty_ms"]), 2) for k, v in tm.items() if k.endswith("_ms") and k != "empty_ms"}, "items", flush=True)
    e.close()
