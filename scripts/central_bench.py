"""The device central reference (dopf_central_solve) on the BASELINE configurations: objective against the HiGHS fixtures
(tests/golden/synthetic_optima.json), iterations, seconds.  python scripts/central_bench.py [tol] [workloads...]"""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
names = sys.argv[2:] or ["config1", "config2", "config4", "config3-share", "config3"]
opt = json.load(open("tests/golden/synthetic_optima.json"))
for name in names:
    pp = synth.baseline_config(3, scale=0.125) if name == "config3-share" else synth.baseline_config(int(name[-1]))
    t0 = time.perf_counter()
    r = _capi.central_solve(_capi.hip_api(), tol=tol, max_iters=200000, **pp.engine_kwargs())
    dt = time.perf_counter() - t0
    want = opt[name]["objective"]
    print(f"{name}: objective {r['objective']:.6f} (HiGHS {want:.1f}, rel diff {abs(r['objective'] - want) / want:.2e}) dual {r['dual_objective']:.6f} "
          f"infeas {r['primal_infeasibility']:.2e} gap {r['gap']:.2e} iterations {r['iterations']} converged {r['converged']} "
          f"{dt:.2f} s ({1e6 * dt / max(r['iterations'], 1):.0f} us/iteration incl. set-up and checks)", flush=True)
