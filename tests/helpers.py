"""Shared helpers of the test suite (CPU and GPU)."""
import numpy as np

from decentralopf_jl_amd import _capi


def make_engine(api, pp, mode=None, **params):
    return _capi.Engine(api, params=_capi.default_params(**params), mode=mode, **pp.engine_kwargs())


def state_of(e):
    """Everything the C ABI exposes after an iteration, as a flat dict of arrays."""
    P, D, C, E = e.get_primal()
    lam, mu, rho = e.get_duals()
    inj, aU, aK, flow, cost = e.get_consensus()
    return dict(P=P, D=D, C=C, E=E, lam=lam, mu=mu, rho=rho, inj=inj, avg_U=aU, avg_K=aK, flow=flow,
                cost=np.asarray([cost]))


def max_diff(a, b, keys=None):
    worst, where = 0.0, None
    for k in (keys or a.keys()):
        if a[k].size == 0:
            continue
        d = float(np.abs(a[k] - b[k]).max())
        if d > worst:
            worst, where = d, k
    return worst, where


def golden_arrays(rec):
    return dict(P=np.asarray(rec["P"]), C=np.asarray(rec["C"])[None, :], D=np.asarray(rec["D"])[None, :],
                lam=np.asarray(rec["lam"]), mu=np.asarray(rec["mu"]), rho=np.asarray(rec["rho"]))


def follow_golden(engine, gold, atol_primal, atol_dual):
    """Run the iteration from zeros and compare with the reference's dump at every kept iteration.
    Dual row k of the dump is the dual USED by iteration k (src/helpers/output.jl:14-43)."""
    kept = sorted(int(k) for k in gold["iterations"])
    worst_p = worst_d = 0.0
    it = 0
    for k in kept:
        if k - 1 > it:
            engine.iterate(k - 1 - it)
            it = k - 1
        g = golden_arrays(gold["iterations"][str(k)])
        lam, mu, rho = engine.get_duals()          # duals that iteration k will use
        worst_d = max(worst_d, np.abs(lam - g["lam"]).max(), np.abs(mu - g["mu"]).max(), np.abs(rho - g["rho"]).max())
        engine.iterate(1)
        it = k
        P, D, C, _ = engine.get_primal()
        worst_p = max(worst_p, np.abs(P - g["P"]).max(), np.abs(D - g["D"]).max(), np.abs(C - g["C"]).max())
    assert worst_p <= atol_primal, worst_p
    assert worst_d <= atol_dual, worst_d
    return worst_p, worst_d


def storage_kkt_violation(pp, s_idx, D0, C0, D, C, E, theta, gamma, w=1.0, tol=1e-7):
    """Optimality certificate of the copper-plate storage QP, vectorised over storages.

    theta[s, t] = price_t + gamma * (s_t - (D0 - C0)) is the linear coefficient of q = D - C.
    KKT: there are prices nu_t with   gD + nu_t  >= 0 at D = 0, = 0 inside, <= 0 at D = pmax,
    gC - nu_t likewise, nu_{t+1} - nu_t >= 0 where E_t = emax, <= 0 where E_t = 0, = 0 inside,
    nu_{T+1} = 0. The set of feasible nu_t is an interval, propagated backwards exactly.
    Returns the largest amount by which any interval is empty (0 = optimal within tol)."""
    mc = pp.sto_mc[s_idx][:, None]
    pm = pp.sto_pmax[s_idx][:, None]
    em = pp.sto_emax[s_idx][:, None]
    q = D - C
    gD = mc + theta + gamma * q + w * (D - D0)
    gC = mc - theta - gamma * q + w * (C - C0)
    inf = np.inf
    # nu in [lo, hi] from D:  gD + nu >= 0 (D at 0) -> nu >= -gD ; D free -> nu = -gD ; D at pm -> nu <= -gD
    lo = np.where(D <= tol, -gD, np.where(D >= pm - tol, -inf, -gD))
    hi = np.where(D <= tol, inf, np.where(D >= pm - tol, -gD, -gD))
    # from C: gC - nu >= 0 (C at 0) -> nu <= gC ; free -> nu = gC ; at pm -> nu >= gC
    lo = np.maximum(lo, np.where(C <= tol, -inf, np.where(C >= pm - tol, gC, gC)))
    hi = np.minimum(hi, np.where(C <= tol, gC, np.where(C >= pm - tol, inf, gC)))
    # pmax = 0: both at both bounds -> free
    degenerate = pm <= tol
    lo = np.where(degenerate, -inf, lo)
    hi = np.where(degenerate, inf, hi)
    T = D.shape[1]
    flo = np.zeros(D.shape[0])
    fhi = np.zeros(D.shape[0])
    worst = 0.0
    for t in range(T - 1, -1, -1):
        at_hi = E[:, t] >= em[:, 0] - tol
        at_lo = E[:, t] <= tol
        both = at_hi & at_lo          # emax = 0: any jump allowed
        nlo = np.where(both, -inf, np.where(at_hi, -inf, flo))      # E at emax: nu_t <= nu_{t+1}
        nhi = np.where(both, inf, np.where(at_lo, inf, fhi))        # E at 0:    nu_t >= nu_{t+1}
        nlo = np.where(at_lo & ~both, flo, nlo)
        nhi = np.where(at_hi & ~both, fhi, nhi)
        flo = np.maximum(nlo, lo[:, t])
        fhi = np.minimum(nhi, hi[:, t])
        worst = max(worst, float(np.max(flo - fhi)))
        # keep going with a non-empty interval
        mid = 0.5 * (flo + fhi)
        bad = flo > fhi
        flo = np.where(bad, mid, flo)
        fhi = np.where(bad, mid, fhi)
    return max(worst, 0.0)
