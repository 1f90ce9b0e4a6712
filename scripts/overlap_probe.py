"""Settled per-iteration time with and without the storage kernel on a side stream (DOPF_F_OVERLAP_AGENTS):
python scripts/overlap_probe.py config3-share config3 config4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
for wl in sys.argv[1:] or ["config3-share"]:
    pp = bench.make_problem(synth, wl); A = pp.G + pp.S
    for flags in (0, _capi.F_OVERLAP_AGENTS):
        e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / A, w_flow=10.0 if pp.L == 0 else 0.3 / A, eps=0.0, flags=flags), **pp.engine_kwargs())
        a = torch.randn(2048, 2048, device="cuda")
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            b = a @ a
            torch.cuda.synchronize()
        e.iterate(250)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); e.iterate(400); best = min(best, (time.perf_counter() - t0) / 400)
        print(f"{wl} flags={flags}: {best*1e6:.2f} us/iteration", flush=True)
        e.close()
