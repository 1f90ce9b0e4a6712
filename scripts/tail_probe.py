"""Settled per-iteration time of a workload (graph path) — quick A/B of library variants / environment switches.
   python scripts/tail_probe.py config2 [config1 ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench

LIB = os.environ.get("DOPF_LIB")
api = _capi.CApi(LIB, "dopf_") if LIB else _capi.hip_api()
for wl in sys.argv[1:] or ["config2"]:
    pp = bench.make_problem(synth, wl)
    A = pp.G + pp.S
    for flags in ([int(x) for x in os.environ["FLAGS"].split(",")] if os.environ.get("FLAGS") else (0, _capi.F_NO_TAIL_FUSE)):
        e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, w_flow=10.0 if pp.L == 0 else 0.3 / A, eps=0.0, flags=flags),
                         **pp.engine_kwargs())
        a = torch.randn(2048, 2048, device="cuda")
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            for _ in range(20):
                b = a @ a
            torch.cuda.synchronize()
        e.iterate(200)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            e.iterate(400)
            best = min(best, (time.perf_counter() - t0) / 400)
        tm = e.iterate_timed(32)
        print(f"{os.path.basename(LIB or 'product')} {wl} flags={flags}: {best*1e6:.2f} us/iteration; x-update launch {1e3*(tm['gen_ms']-tm['empty_ms']):.2f} us "
              f"(sto {1e3*(tm['sto_ms']-tm['empty_ms']):.2f}) tail_fused={tm['tail_fused']}", flush=True)
        e.close()
