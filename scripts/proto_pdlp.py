"""Prototype (numpy): PDHG with diagonal preconditioning, averaging and restarts on the central LP of a copper-plate or
network case, against the HiGHS optimum. Decides whether a device version (central reference) is worth writing.
variables x = (P, D, C) in boxes;  rows: balance_t (=), flows (<=, both signs), level E = cumsum(C - D) in [0, emax]."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import synth
from decentralopf_jl_amd.central import solve_central_packed
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
PLAIN = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pp = synth.baseline_config(idx, scale=scale)
G, S, T, N, L = pp.G, pp.S, pp.T, pp.N, pp.L
opt = solve_central_packed(pp, duals=False).objective
print("HiGHS optimum", opt, "G S T N L", G, S, T, N, L, flush=True)
gn, sn = pp.gen_node, pp.sto_node
H = pp.ptdf                                   # (L, N)
dem = pp.demand                               # (N, T)
cP = np.repeat(pp.gen_mc[:, None], T, 1); cS = np.repeat(pp.sto_mc[:, None], T, 1)
uP = np.repeat(pp.gen_pmax[:, None], T, 1); uS = np.repeat(pp.sto_pmax[:, None], T, 1)
em = pp.sto_emax[:, None]
# cost scaling: objective in units of a typical price, injections in units of typical power
def Kx(P, D, C):
    inj = np.zeros((N, T)); np.add.at(inj, gn, P); np.add.at(inj, sn, D - C); inj -= dem
    return inj.sum(0), (H @ inj if L else np.zeros((0, T))), np.cumsum(C - D, axis=1)
def KTy(yb, yf, yE):
    pi = yb[None, :] + (H.T @ yf if L else 0.0)          # (N, T): d/dinj
    suf = np.cumsum(yE[:, ::-1], axis=1)[:, ::-1]       # sum_{tau >= t} yE
    return pi[gn], pi[sn] - suf, -pi[sn] + suf
# diagonal preconditioners (Pock-Chambolle, alpha = 1): tau_j = 1 / sum_i |K_ij|, sigma_i = 1 / sum_j |K_ij|
absH = np.abs(H) if L else np.zeros((0, N))
colP = 1.0 + absH.sum(0)[gn][:, None] * np.ones((1, T))
tt = np.arange(T)[None, :]
colS = 1.0 + absH.sum(0)[sn][:, None] + (T - tt)        # balance + flows + the level rows tau >= t
tauP, tauS = 1.0 / colP, 1.0 / colS
cntn = np.bincount(gn, minlength=N) + 2 * np.bincount(sn, minlength=N)
sig_b = 1.0 / max(1, G + 2 * S)
sig_f = 1.0 / np.maximum(1e-12, absH @ cntn)[:, None] if L else np.zeros((0, 1))
sig_E = 1.0 / (2.0 * (tt + 1))
w = 1.0                                                 # primal weight: tau *= 1/w, sigma *= w
P = np.zeros((G, T)); D = np.zeros((S, T)); C = np.zeros((S, T))
yb = np.zeros(T); yf = np.zeros((L, T)); yE = np.zeros((S, T))
fmax = pp.f_max[:, None] if L else np.zeros((0, 1))
def step(P, D, C, yb, yf, yE, w):
    gP, gD, gC = KTy(yb, yf, yE)
    Pn = np.clip(P - tauP / w * (cP + gP), 0, uP)
    Dn = np.clip(D - tauS / w * (cS + gD), 0, uS)
    Cn = np.clip(C - tauS / w * (cS + gC), 0, uS)
    b, f, E = Kx(2 * Pn - P, 2 * Dn - D, 2 * Cn - C)
    ybn = yb + w * sig_b * b
    # flows: -fmax <= f <= fmax ; multiplier yf >= 0 for f > fmax, <= 0 for f < -fmax: prox of the interval indicator's conjugate
    z = yf + w * sig_f * f
    yfn = z - w * sig_f * np.clip(z / (w * sig_f + 1e-300), -fmax, fmax) if L else yf
    zE = yE + w * sig_E * E
    yEn = zE - w * sig_E * np.clip(zE / (w * sig_E), 0.0, em)
    return Pn, Dn, Cn, ybn, yfn, yEn
def metrics(P, D, C, yb, yf, yE):
    b, f, E = Kx(P, D, C)
    pobj = (cP * P).sum() + (cS * (D + C)).sum()
    pinf = max(np.abs(b).max(), (np.abs(f) - fmax).max() if L else 0.0, (-E).max(), (E - em).max(), 0.0)
    gP, gD, gC = KTy(yb, yf, yE)
    rP, rD, rC = cP + gP, cS + gD, cS + gC               # reduced costs
    dobj = (np.minimum(rP, 0) * uP).sum() + (np.minimum(rD, 0) * uS).sum() + (np.minimum(rC, 0) * uS).sum() \
        - (yb * dem.sum(0)).sum() - (np.abs(yf) * fmax).sum() - ((yf * (H @ dem)).sum() if L else 0.0) - (np.maximum(yE, 0) * em).sum()
    return pobj, dobj, pinf
t0 = time.time()
avg = [np.zeros_like(a) for a in (P, D, C, yb, yf, yE)]; navg = 0
last_gap = np.inf
for k in range(1, iters + 1):
    P, D, C, yb, yf, yE = step(P, D, C, yb, yf, yE, w)
    for a, b_ in zip(avg, (P, D, C, yb, yf, yE)): a += b_
    navg += 1
    if k % 200 == 0:
        cand = [a / navg for a in avg]
        pa, da, ia = metrics(*cand)
        pc, dc, ic = metrics(P, D, C, yb, yf, yE)
        ga, gc = abs(pa - da) / (1 + abs(pa)) + ia / (1 + np.abs(dem).max()), abs(pc - dc) / (1 + abs(pc)) + ic / (1 + np.abs(dem).max())
        use_avg = ga < gc
        if PLAIN: use_avg = False
        if not PLAIN and (min(ga, gc) < 0.5 * last_gap or navg >= 4000):   # restart (to the better of current / average)
            if use_avg: P, D, C, yb, yf, yE = cand
            last_gap = min(ga, gc)
            avg = [np.zeros_like(a) for a in avg]; navg = 0
        if k % 2000 == 0:
            p_, d_, i_ = (pa, da, ia) if use_avg else (pc, dc, ic)
            print(f"it {k}: primal {p_:.6e} (rel to opt {abs(p_ - opt) / opt:.2e}) dual {d_:.6e} infeas {i_:.3e} gap {min(ga, gc):.2e} {time.time() - t0:.0f}s", flush=True)
