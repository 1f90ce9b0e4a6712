"""Settled and cold per-iteration wall time of one workload for one or more builds of the library, in ONE process on one box
(boxes differ by a few per cent: only numbers from the same call compare).
usage: python scripts/lib_time.py <workload> <lib.so>[:flags] [<lib.so>[:flags] ...]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
wl = sys.argv[1]
pp = bench.make_problem(synth, wl); A = pp.G + pp.S
_capi._pin_hip_runtime()
import torch
def condition():
    x = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 0.25: x = (x @ x).clamp_(-1, 1)
    torch.cuda.synchronize()
res = {}
apis = {}
for rnd in range(3):
    for spec in sys.argv[2:]:
        path, _, fl = spec.partition(":")
        flags = int(fl) if fl else 0
        api = apis.setdefault(path, _capi.CApi(path, "dopf_"))
        e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, w_flow=10.0 if pp.L == 0 else 0.3 / A, eps=0.0, flags=flags), **pp.engine_kwargs())
        condition()
        e.iterate(5); torch.cuda.synchronize()
        t0 = time.perf_counter(); e.iterate(20); cold = (time.perf_counter() - t0) / 20
        e.iterate(175)
        condition()
        e.iterate(48)
        t0 = time.perf_counter(); e.iterate(400); warm = (time.perf_counter() - t0) / 400
        res.setdefault(spec, []).append((cold * 1e6, warm * 1e6))
        e.close()
for spec, v in res.items():
    c = sorted(x[0] for x in v); w = sorted(x[1] for x in v)
    print(f"{wl} {spec}: cold window (iterations 6..25) {c[len(c)//2]:.2f} us [{c[0]:.2f}..{c[-1]:.2f}], settled {w[len(w)//2]:.2f} us [{w[0]:.2f}..{w[-1]:.2f}]")
