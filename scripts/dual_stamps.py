"""Phase stamps of the one-launch dual/price kernel (block T/2; DOPF_DUAL_STAMPS build in scripts/tmp/libdopf_dstamps.so):
   python scripts/dual_stamps.py [config3-share|config3]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, dopf_pkg
dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
import bench
_capi._pin_hip_runtime()
api = _capi.CApi(os.environ.get("DOPF_STAMPS_LIB", "scripts/tmp/libdopf_dstamps.so"), "dopf_")
names = ["entry", "first barrier (sums, demand, line state in)", "flow + slack dot products", "barrier", "line update, barriers", "price dot products",
         "barrier", "price stores", "ticket back"]
for wl in sys.argv[1:] or ["config3-share"]:
    pp = synth.baseline_config(3, scale=float(os.environ["DOPF_STAMPS_SCALE"])) if os.environ.get("DOPF_STAMPS_SCALE") else bench.make_problem(synth, wl)
    A = pp.G + pp.S
    e = _capi.Engine(api, params=_capi.default_params(gamma=1.0 / A, w_flow=0.3 / A, eps=0.0, flags=int(os.environ.get("DOPF_STAMPS_FLAGS", "0"))), **pp.engine_kwargs())
    e.iterate(300)
    acc = np.zeros(9)
    n = 50
    for _ in range(n):
        e.iterate(1)
        out = (C.c_uint64 * 15)(); api.lib.dopf_debug_stats(e._ctx, out)
        s = np.array([out[5 + i] for i in range(9)], dtype=np.float64)
        acc += (s - s[0]) / 100.0        # 100 MHz constant clock -> us
    acc /= n
    print(wl, "us since kernel entry (block T/2, thread 0):")
    for k in range(1, 9):
        print(f"  {names[k]:48s} {acc[k]:7.2f}  (+{acc[k]-acc[k-1]:.2f})")
