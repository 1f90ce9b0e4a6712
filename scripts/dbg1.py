import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, state_of
pp = synth.synthetic_case(n_gen=25, n_sto=6, T=8, N=30, L=50, seed=17, fmax_factor=0.6, fmax_min=5)
api = _capi.hip_api()
h = make_engine(api, pp, eps=0.0, gamma=0.05)
h.iterate(1)
L = pp.L; M2 = 2*L
dp = C.POINTER(C.c_double)
def table(n, t):
    beta = np.zeros(M2); psi = np.zeros(M2); slope = np.zeros(M2+1); psi0 = C.c_double(0); m = C.c_int32(0)
    f = api.lib.dopf_debug_table; f.restype = C.c_int
    rc = f(h._ctx, C.c_int32(n), C.c_int32(t), beta.ctypes.data_as(dp), psi.ctypes.data_as(dp), slope.ctypes.data_as(dp), C.byref(psi0), C.byref(m))
    assert rc == 0
    return beta, psi, slope, psi0.value, m.value
# numpy Psi at iteration 1 state: avgU=avgK=0, flow = ptdf @ (-d), price=0, s = sum(-d)
g = 0.05; w2 = 20.0
def psi_np(n, t, dl):
    c = -pp.demand[:, t]; s = c.sum(); f0 = pp.ptdf @ c
    v = 0 + g*(s + dl)
    for l in range(L):
        hh = pp.ptdf[l, n]
        if hh == 0: continue
        f = f0[l] + hh*dl; F = pp.f_max[l]
        U = max(0, (0 - w2*(f - F))/(w2+g)); K = max(0, (0 + w2*(f+F))/(w2+g))
        v += w2*hh*((f + U - F) - (K - f - F))
    return v
for n in (4, 3):
    beta, psi, slope, psi0, m = table(n, 0)
    print("node", n, "m", m, "nnz h", np.count_nonzero(pp.ptdf[:, n]))
    print(" sorted?", np.all(np.diff(beta[:m]) >= 0), "psi0 err", psi0 - psi_np(n, 0, 0.0))
    err = [psi[j] - psi_np(n, 0, beta[j]) for j in range(m)]
    print(" max psi err", np.abs(err).max() if m else 0, "argmax", int(np.argmax(np.abs(err))) if m else -1)
    print(" beta[:6]", beta[:6], "psi[:6]", psi[:6]); print(" err[:8]", np.array(err[:8]))
    sl = [(psi_np(n,0,beta[j]+1e-6)-psi_np(n,0,beta[j]-1e-6))/2e-6 for j in range(min(m,5))]
    print(" slopes tab", slope[:6]);
