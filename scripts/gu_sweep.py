"""Experiment: generator rows in flight per lane (build-time DOPF_GU) x generator blocks (DOPF_GEN_TARGET_ITEMS) in the fused
x-update launch of config2: python scripts/gu_sweep.py lib1.so[,lib2.so...] items1,items2,...   (settled state, graph replay).
The libraries must be built with -DDOPF_EXPERIMENTS (DOPF_HIPCC_FLAGS=-DDOPF_EXPERIMENTS scripts/build_lib.sh out.so): the shipped
library does not read tuning knobs from the environment."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
libs = sys.argv[1].split(",")
items = [int(x) for x in sys.argv[2].split(",")]
pp = synth.baseline_config(2); A = pp.G + pp.S
a = torch.randn(2048, 2048, device="cuda")
def warm(sec=0.25):
    t = time.perf_counter() + sec
    while time.perf_counter() < t:
        for _ in range(20): b = a @ a
        torch.cuda.synchronize()
for lib in libs:
    api = _capi.CApi(lib, "dopf_")
    for it in items:
        os.environ["DOPF_GEN_TARGET_ITEMS"] = str(it)
        e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0), **pp.engine_kwargs())
        warm()
        e.iterate(250)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); e.iterate(400); dt = time.perf_counter() - t0
            best = min(best, dt / 400)
        tm = e.iterate_timed(32)
        print(f"{os.path.basename(lib)} items {it}: {1e6*best:.2f} us/iter  k_agents {1e3*(tm['gen_ms']-tm['empty_ms']):.2f} us", flush=True)
        e.close()
