"""Timeline of the tail block of a one-launch iteration (stamps build, scripts/build_lib.sh with -DDOPF_BLOCK_STAMPS):
when do the producer blocks end, when does the tail block see the last contribution, when does it end."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
api = _capi.CApi("scripts/tmp/libdopf_stamps.so", "dopf_")
api.lib.dopf_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
for wl in sys.argv[1:] or ["config2"]:
    pp = bench.make_problem(synth, wl); A = pp.G + pp.S
    e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, eps=0.0, flags=_capi.F_NO_GRAPH), **pp.engine_kwargs())
    e.iterate(150)
    n = 8192 * 16
    buf = (C.c_uint64 * n)()
    assert api.lib.dopf_debug_timeline(e._ctx, buf, n) == 0
    raw = np.array(list(buf), dtype=np.float64)
    tail = raw[:4]
    tl = raw[32768:].reshape(-1, 2)
    tl = tl[(tl[:, 0] > 0) & (tl[:, 1] > 0)]
    t0 = tl[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    prod = tl[:-1]
    print(f"{wl}: {len(tl)} blocks; producers start max {us(prod[:,0]).max():.2f}, end p50 {np.median(us(prod[:,1])):.2f} p95 {np.percentile(us(prod[:,1]),95):.2f} max {us(prod[:,1]).max():.2f} us")
    print(f"   tail block (grid's last): start {us(tl[-1,0]):.2f} end {us(tl[-1,1]):.2f}; inside: entered {us(tail[0]):.2f}, loads done {us(tail[1]):.2f}, slot 0 complete {us(tail[2]):.2f}, after stores+barrier {us(tail[3]):.2f} us")
    e.close()
