"""The central reference: the whole multi-period DC-OPF as ONE LP (host side, SciPy/HiGHS).

Mirrors the reference's second script, src/opf_central_reference.jl:1-81 — there a JuMP model handed to Gurobi, here the
same LP handed to HiGHS (the LP solver that ships with SciPy; no licence):

    variables     P[g,t] in [0, max_generation], D[s,t], C[s,t] in [0, max_power], E[s,t] in [0, max_level],
                  U[l,t], K[l,t] >= 0                                                            (:21-31)
    injection     I[n,t] = sum of the node's P + D - C - demand                                  (:34-38)
    objective     sum mc P + sum mc (D + C)                                                      (:41-44)
    EB[t]         sum_n I[n,t] = 0                                                               (:47)
    FlowUpper     ptdf I + U = f_max,   FlowLower   K - ptdf I = f_max                           (:49-51)
    StorageBalance  E[s,t] = E[s,t-1] - D + C, E[s,0] = 0                                        (:53)
    outputs       objective, P, D, C, line utilisation ptdf I, system price lambda = dual(EB),
                  nodal price = lambda + sum_l (dual(FlowUpper) + dual(FlowLower))[l,t] ptdf[l,:]  (:57-81)

It is the parity target of the decentral ADMM ("converged objective within 1e-3 of opf_central_reference.jl") and, like in
the reference, NOT part of the hot path: a one-off host solve. The injections are explicit LP variables, so a flow row
has N non-zeros instead of one per unit (what makes the 118-node cases tractable); duals come back in the reference's
sign convention (d objective / d right-hand side).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np

from .network import Generator, Line, Node, PackedProblem, Storage, pack


@dataclass
class CentralResult:
    objective: float
    generation: np.ndarray          # (G, T)  value.(P)
    discharge: np.ndarray           # (S, T)  value.(D)
    charge: np.ndarray              # (S, T)  value.(C)
    level: np.ndarray               # (S, T)  value.(E)
    injection: np.ndarray           # (N, T)  value.(I)
    line_utilization: np.ndarray    # (L, T)  ptdf * I
    system_price: np.ndarray        # (T,)    dual.(EB)
    flow_upper_dual: np.ndarray     # (L, T)  dual.(FlowUpper)
    flow_lower_dual: np.ndarray     # (L, T)  dual.(FlowLower)
    nodal_price: np.ndarray         # (N, T)


def solve_central_packed(pp: PackedProblem, *, duals: bool = True) -> CentralResult:
    """The LP on a packed case (any size HiGHS can take; synthetic cases with 1e5 agents go through
    tests/central_lp.aggregate_* first)."""
    from scipy import sparse
    from scipy.optimize import linprog
    N, L, T, G, S = pp.N, pp.L, pp.T, pp.G, pp.S
    nP, nS, nI, nL = G * T, S * T, N * T, L * T
    oD, oC, oE, oI, oU, oK = nP, nP + nS, nP + 2 * nS, nP + 3 * nS, nP + 3 * nS + nI, nP + 3 * nS + nI + nL
    nv = oK + nL
    c = np.zeros(nv)
    c[:nP] = np.repeat(pp.gen_mc, T)
    c[oD:oD + nS] = np.repeat(pp.sto_mc, T)
    c[oC:oC + nS] = np.repeat(pp.sto_mc, T)
    lb = np.zeros(nv)
    ub = np.concatenate([np.repeat(pp.gen_pmax, T), np.repeat(pp.sto_pmax, T), np.repeat(pp.sto_pmax, T),
                         np.repeat(pp.sto_emax, T), np.full(nI, np.inf), np.full(2 * nL, np.inf)])
    lb[oI:oI + nI] = -np.inf
    tt = np.arange(T)
    rows, cols, vals, beq = [], [], [], []
    # I[n,t] - sum of the node's units = -demand[n,t]          rows n*T + t
    gi = np.repeat(np.asarray(pp.gen_node, dtype=np.int64), T) * T + np.tile(tt, G)
    si = np.repeat(np.asarray(pp.sto_node, dtype=np.int64), T) * T + np.tile(tt, S)
    rows += [gi, si, si, np.arange(nI)]
    cols += [np.arange(nP), oD + np.arange(nS), oC + np.arange(nS), oI + np.arange(nI)]
    vals += [-np.ones(nP), -np.ones(nS), np.ones(nS), np.ones(nI)]
    beq.append(-np.asarray(pp.demand, dtype=np.float64).reshape(N, T).reshape(-1))
    r0 = nI
    rEB = r0                                                   # EB[t]: sum_n I[n,t] = 0
    rows.append(r0 + np.tile(tt, N)); cols.append(oI + np.arange(nI)); vals.append(np.ones(nI))
    beq.append(np.zeros(T)); r0 += T
    k = np.arange(nS)                                          # E[t] - E[t-1] + D - C = 0
    rows += [r0 + k, r0 + k, r0 + k]; cols += [oE + k, oD + k, oC + k]; vals += [np.ones(nS), np.ones(nS), -np.ones(nS)]
    k1 = k[(k % T) > 0]
    rows.append(r0 + k1); cols.append(oE + k1 - 1); vals.append(-np.ones(k1.size))
    beq.append(np.zeros(nS)); r0 += nS
    Aeq = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(r0, nv))
    rUp = rLo = r0
    if L > 0:
        flow = sparse.kron(sparse.csr_matrix(np.asarray(pp.ptdf, dtype=np.float64)), sparse.identity(T, format="csr"), format="csr")
        eyeL = sparse.identity(nL, format="csr")
        z = lambda n_: sparse.csr_matrix((nL, n_))
        up = sparse.hstack([z(oI), flow, eyeL, z(nL)], format="csr")          # ptdf I + U = f_max      rows l*T + t
        lo = sparse.hstack([z(oI), -flow, z(nL), eyeL], format="csr")         # K - ptdf I = f_max
        rUp, rLo = r0, r0 + nL
        Aeq = sparse.vstack([Aeq, up, lo], format="csr")
        beq += [np.repeat(pp.f_max, T), np.repeat(pp.f_max, T)]
    res = linprog(c, A_eq=Aeq, b_eq=np.concatenate(beq), bounds=np.stack([lb, ub], axis=1), method="highs")
    if res.status != 0:
        raise RuntimeError(f"central LP: {res.message}")
    x = res.x
    inj = x[oI:oI + nI].reshape(N, T)
    lam = np.zeros(T)
    fu = np.zeros((L, T))
    fl = np.zeros((L, T))
    if duals:
        m = res.eqlin.marginals
        lam = m[rEB:rEB + T].copy()
        if L > 0:
            fu = m[rUp:rUp + nL].reshape(L, T)
            fl = m[rLo:rLo + nL].reshape(L, T)
    ptdf = np.asarray(pp.ptdf, dtype=np.float64).reshape(L, N)
    nodal = lam[None, :] + ptdf.T @ (fu + fl) if L > 0 else np.tile(lam, (N, 1))      # opf_central_reference.jl:71-79
    return CentralResult(objective=float(res.fun), generation=x[:nP].reshape(G, T), discharge=x[oD:oD + nS].reshape(S, T),
                         charge=x[oC:oC + nS].reshape(S, T), level=x[oE:oE + nS].reshape(S, T), injection=inj,
                         line_utilization=ptdf @ inj if L > 0 else np.zeros((0, T)), system_price=lam,
                         flow_upper_dual=fu, flow_lower_dual=fl, nodal_price=nodal)


def central_reference(nodes: Sequence[Node], generators: Sequence[Generator], storages: Sequence[Storage],
                      lines: Sequence[Line], *, verbose: bool = False) -> CentralResult:
    """src/opf_central_reference.jl for a case given as the reference's element vectors; `verbose` prints what the
    script prints (:60-81)."""
    r = solve_central_packed(pack(nodes, generators, storages, lines))
    if verbose:
        print(f"Objective value: {r.objective}\n")
        print(f"Generator results:\n{r.generation}\n")
        print(f"Discharge results:\n{r.discharge}\n")
        print(f"Charge results:\n{r.charge}\n")
        print(f"Line utilization:\n{r.line_utilization}\n")
        print(f"System price:\n{r.system_price}\n")
        print(f"Nodal price:\n{r.nodal_price}\n")
    return r


def central_reference_on_device(nodes: Sequence[Node], generators: Sequence[Generator], storages: Sequence[Storage],
                                lines: Sequence[Line], *, tol: float = 1e-9, max_iters: int = 200000, device: int = -1) -> CentralResult:
    """The same LP solved on the GPU by libdopf_hip (dopf_central_solve: first-order primal-dual method, csrc/kernels_central.hip)
    — for cases beyond a host LP solver, and as a cross-check that shares no code with HiGHS."""
    from . import _capi
    pp = pack(nodes, generators, storages, lines)
    r = _capi.central_solve(_capi.hip_api(), tol=tol, max_iters=max_iters, params=_capi.default_params(device=device),
                            **pp.engine_kwargs())
    if not r["converged"]:
        raise RuntimeError(f"central LP on the device: gap {r['gap']:.2e} after {r['iterations']} iterations (infeasible case?)")
    inj = -np.asarray(pp.demand, dtype=np.float64).copy()
    np.add.at(inj, pp.gen_node, r["P"])
    np.add.at(inj, pp.sto_node, r["D"] - r["C"])
    return CentralResult(objective=r["objective"], generation=r["P"], discharge=r["D"], charge=r["C"], level=r["E"], injection=inj,
                         line_utilization=r["line_utilization"], system_price=r["system_price"], flow_upper_dual=r["flow_upper_dual"],
                         flow_lower_dual=r["flow_lower_dual"], nodal_price=r["nodal_price"])
