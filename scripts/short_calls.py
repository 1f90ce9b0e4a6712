"""What a short dopf_iterate call costs beyond its kernels: per-iteration time of calls of 20 and of 400 iterations (config2,
settled state), i.e. the fixed cost of a call (graph launches + status read-back + sync) spread over its iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
LIB = os.environ.get("DOPF_LIB")
api = _capi.CApi(LIB, "dopf_") if LIB else _capi.hip_api()
pp = bench.make_problem(synth, "config2"); A = pp.G + pp.S
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0), **pp.engine_kwargs())
a = torch.randn(2048, 2048, device="cuda")
t_end = time.perf_counter() + 0.25
while time.perf_counter() < t_end:
    b = a @ a
    torch.cuda.synchronize()
e.iterate(200)
for n in (20, 400):
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter(); e.iterate(n); best = min(best, (time.perf_counter() - t0) / n)
    print(f"{os.path.basename(LIB or 'product')}: calls of {n} iterations: {best*1e6:.2f} us/iteration", flush=True)
