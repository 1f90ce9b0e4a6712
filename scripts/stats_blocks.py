"""When do the blocks of the fused x-update launch run (counters build)? Start / end of every block of k_agents in a settled
iteration: storage blocks first in the grid, generator blocks after."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
api = _capi.CApi("scripts/tmp/libdopf_stamps.so", "dopf_")      # product code + two stamps per block (-DDOPF_BLOCK_STAMPS)
api.lib.dopf_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"         # config2 (k_agents) or a network workload (k_net_agents)
pp = bench.make_problem(synth, wl); A = pp.G + pp.S
_capi._pin_hip_runtime()
e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, w_flow=10.0 if pp.L == 0 else 0.3 / A, eps=0.0, flags=_capi.F_NO_GRAPH | (int(sys.argv[3]) if len(sys.argv) > 3 else 0)), **pp.engine_kwargs())
e.iterate(300 if pp.L else 150)
n = 8192 * 16
buf = (C.c_uint64 * n)()
assert api.lib.dopf_debug_timeline(e._ctx, buf, n) == 0
tl = np.array(list(buf), dtype=np.float64)[32768:].reshape(-1, 2)
tl = tl[(tl[:, 0] > 0) & (tl[:, 1] > 0)]
nS = int(sys.argv[2]) if len(sys.argv) > 2 else 569        # storage blocks in front of the grid
t0 = tl[:, 0].min()
us = lambda x: 10 * (x - t0) / 1e3
sto, gen = tl[:nS], tl[nS:]
print(f"{len(sto)} storage blocks: start p50 {np.median(us(sto[:, 0])):.2f} max {us(sto[:, 0]).max():.2f}; end p50 {np.median(us(sto[:, 1])):.2f} max {us(sto[:, 1]).max():.2f} us; duration p50 {np.median(sto[:, 1] - sto[:, 0]) / 100:.2f} us")
print(f"{len(gen)} generator blocks: start p5 {np.percentile(us(gen[:, 0]), 5):.2f} p50 {np.median(us(gen[:, 0])):.2f} p95 {np.percentile(us(gen[:, 0]), 95):.2f}; end max {us(gen[:, 1]).max():.2f} us; duration p50 {np.median(gen[:, 1] - gen[:, 0]) / 100:.2f} p95 {np.percentile(gen[:, 1] - gen[:, 0], 95) / 100:.2f} us")
h, edges = np.histogram(us(gen[:, 0]), bins=12)
print("generator block starts per time bin:", [(round(float(a), 1), int(b)) for a, b in zip(edges[:-1], h)])
# how many generator blocks are resident over time, and when the bytes get moved
ts = np.arange(0, us(tl[:, 1]).max() + 0.5, 1.0)
gs, ge = us(gen[:, 0]), us(gen[:, 1])
ss, se = us(sto[:, 0]), us(sto[:, 1])
print("t [us]: storage blocks resident / generator blocks resident / generator blocks finished")
for t in ts:
    print(f"  {t:5.1f}: {int(((ss <= t) & (se > t)).sum()):4d} {int(((gs <= t) & (ge > t)).sum()):4d} {int((ge <= t).sum()):5d}")
