"""Host-side mirror of the reference's ADMM driver API, on top of the C ABI (include/dopf.h).

Julia is not available in this image, so this Python module is the runnable host; the thin Julia
shim with the same calls lives in julia/DecentralOPFHip.jl. Names, argument meaning and stopping
behaviour follow the reference:

  ADMM(gamma, nodes, generators, storages, lines)   src/structures/admm.jl:1-63
  run(admm)                  ~ run!(admm)            src/optimization/run.jl:1-5
  calculate_iteration(admm)  ~ calculate_iteration!  src/optimization/run.jl:7-16
  Result / ResultGenerator / ResultStorage           src/structures/results.jl:1-117
  Convergence                                        src/structures/convergence.jl:1-20
  get_nodal_price(admm, iteration)                   src/helpers/network_elements.jl:16-25
  export_results(admm, filename)                     src/helpers/output.jl:1-85

Differences, all deliberate: no console printing per iteration; an optional
iteration cap (the reference loops forever on a non-convergent case); history recording can be
switched off (`record=False`) because the reference's whole-history vectors are O(iterations x
agents) and would dominate at 1e6 agents — then only the last state is fetched.
All compute happens in libdopf_hip; this file only marshals.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _capi
from .network import Generator, Line, Node, Storage, pack


@dataclass
class PenaltyTerm:
    """src/structures/penalty_terms.jl:1-5 (diagnostics: never read back by the algorithm)"""
    energy_balance: np.ndarray
    upper_flow: np.ndarray
    lower_flow: np.ndarray


@dataclass
class ResultGenerator:
    generator: Generator
    generation: np.ndarray
    penalty_term: Optional[PenaltyTerm] = None      # filled with ADMM(..., record_slacks=True)
    U: Optional[np.ndarray] = None                  # (L, T)
    K: Optional[np.ndarray] = None


@dataclass
class ResultStorage:
    storage: Storage
    discharge: np.ndarray
    charge: np.ndarray
    level: np.ndarray
    penalty_term: Optional[PenaltyTerm] = None
    U: Optional[np.ndarray] = None
    K: Optional[np.ndarray] = None


@dataclass
class ResultNode:
    """src/structures/results.jl:19-35: what the units of one node produce, discharge and charge per timestep."""
    node: Node
    generation: np.ndarray
    discharge: np.ndarray
    charge: np.ndarray


@dataclass
class Result:
    unit_to_result: Dict[int, object]
    generation: np.ndarray
    discharge: np.ndarray
    charge: np.ndarray
    avg_U: np.ndarray
    avg_K: np.ndarray
    total_costs: float
    injection: np.ndarray
    line_utilization: np.ndarray
    node_to_result: Dict[int, ResultNode] = field(default_factory=dict)
    penalty_term: Optional[PenaltyTerm] = None      # results.jl:66-70 (sum over the units); filled with record_slacks=True

    def of(self, unit):
        return self.unit_to_result[id(unit)]

    def of_node(self, node):
        return self.node_to_result[id(node)]


@dataclass
class Convergence:
    """src/structures/convergence.jl:1-20. With history recording the *_res lists hold the reference's vectors /
    matrices |dual change| per entry; without it their inf-norms (floats)."""
    lambda_: bool = False
    lambda_res: list = field(default_factory=list)
    mue: bool = False
    mue_res: list = field(default_factory=list)
    rho: bool = False
    rho_res: list = field(default_factory=list)
    all: bool = False


class ADMM:
    """State of one decentral OPF run. `backend` is a loaded C-ABI library (default: the HIP one)."""

    def __init__(self, gamma: float, nodes: Sequence[Node], generators: Sequence[Generator],
                 storages: Sequence[Storage], lines: Sequence[Line], *, backend: Optional[_capi.CApi] = None,
                 record: bool = True, max_iters: int = 0, backend_mode: Optional[int] = None,
                 record_slacks: bool = False, **params):
        self.iteration = 1
        self.gamma = float(gamma)
        self.nodes, self.generators = list(nodes), list(generators)
        self.storages, self.lines = list(storages), list(lines)
        self.packed = pack(self.nodes, self.generators, self.storages, self.lines)
        p = self.packed
        self.T = list(range(1, p.T + 1))
        self.N = list(range(1, p.N + 1))
        self.L = list(range(1, p.L + 1))
        self.lambdas = [np.zeros(p.T)]
        self.mues = [np.zeros((p.L, p.T))]
        self.rhos = [np.zeros((p.L, p.T))]
        self.results: List[Result] = []
        self.convergence = Convergence()
        self.ptdf = p.ptdf
        self.f_max = p.f_max
        self.total_demand = p.demand.sum(axis=0)
        self.node_id_to_demand = {i + 1: list(n.demand) for i, n in enumerate(self.nodes)}
        self.node_to_id = {id(n): i + 1 for i, n in enumerate(self.nodes)}
        self.node_to_units: Dict[int, list] = {}
        for u in self.generators + self.storages:
            self.node_to_units.setdefault(id(u.node), []).append(u)
        self.record = record
        self.record_slacks = record_slacks        # also fetch every unit's U, K and PenaltyTerm (diagnostics, HIP backend)
        if record_slacks:
            params["flags"] = int(params.get("flags", 0)) | _capi.F_KEEP_DELTAS
        self.params = _capi.default_params(gamma=self.gamma, max_iters=max_iters, **params)
        self.engine = _capi.Engine(backend if backend is not None else _capi.hip_api(),
                                   params=self.params, mode=backend_mode, **p.engine_kwargs())

    # -- one iteration -------------------------------------------------------------------------
    def _fetch_result(self) -> Result:
        P, D, C, E = self.engine.get_primal()
        inj, aU, aK, flow, cost = self.engine.get_consensus()
        u2r: Dict[int, object] = {}
        prev = self.results[-1] if self.results else None
        G = len(self.generators)

        def extras(a, delta):
            if not self.record_slacks:
                return {}
            U, K = self.engine.get_agent_slacks(a)
            return dict(U=U, K=K, penalty_term=PenaltyTerm(*self.engine.get_agent_penalty(a, delta=delta)))
        for i, g in enumerate(self.generators):
            d = P[i] - (prev.of(g).generation if prev else 0.0)
            u2r[id(g)] = ResultGenerator(g, P[i].copy(), **extras(i, d))
        for i, s in enumerate(self.storages):
            q0 = (prev.of(s).discharge - prev.of(s).charge) if prev else 0.0
            u2r[id(s)] = ResultStorage(s, D[i].copy(), C[i].copy(), E[i].copy(), **extras(G + i, (D[i] - C[i]) - q0))
        T = self.packed.T
        if hasattr(self.engine.api, "get_node_results"):        # dopf_get_node_results (the device adds the units of a node)
            ng, nd, nc = self.engine.get_node_results()
        else:                                                   # a backend without it: the same sums here
            ng, nd, nc = (np.zeros((self.packed.N, T)) for _ in range(3))
            np.add.at(ng, np.asarray(self.packed.gen_node, dtype=np.int64), P)
            np.add.at(nd, np.asarray(self.packed.sto_node, dtype=np.int64), D)
            np.add.at(nc, np.asarray(self.packed.sto_node, dtype=np.int64), C)
        n2r = {id(n): ResultNode(n, ng[i].copy(), nd[i].copy(), nc[i].copy()) for i, n in enumerate(self.nodes)}
        pen = None
        if self.record_slacks:
            # Result.penalty_term = sum_up over the units (results.jl:66-70, helpers/penalty_terms.jl:1-6): one pass on the device
            # where it holds the injection changes (networks), else the sum of the per-unit terms fetched above
            if self.packed.L > 0 and hasattr(self.engine.api, "get_penalty_sums"):
                pen = PenaltyTerm(*self.engine.get_penalty_sums())
            else:
                terms = [r.penalty_term for r in u2r.values()]
                pen = PenaltyTerm(*(sum(getattr(t_, f) for t_ in terms) for f in ("energy_balance", "upper_flow", "lower_flow")))
        return Result(u2r, P.sum(axis=0) if P.size else np.zeros(T), D.sum(axis=0) if D.size else np.zeros(T),
                      C.sum(axis=0) if C.size else np.zeros(T), aU, aK, cost, inj, flow, n2r, pen)

    def _after(self, done: int):
        lam_res, mu_res, rho_res, it = self.engine.get_residuals()
        _, conv = self.engine.sync()
        if done and (self.iteration != 1 or done > 1):
            c = self.convergence
            if self.record and len(self.lambdas) >= 2:       # the reference's per-entry residuals (convergence.jl:5-12)
                c.lambda_res.append(np.abs(self.lambdas[-1] - self.lambdas[-2]))
                c.mue_res.append(np.abs(self.mues[-1] - self.mues[-2]))
                c.rho_res.append(np.abs(self.rhos[-1] - self.rhos[-2]))
            else:
                c.lambda_res.append(lam_res)
                c.mue_res.append(mu_res)
                c.rho_res.append(rho_res)
            eps = self.params.eps
            c.lambda_, c.mue, c.rho = lam_res < eps, mu_res < eps, rho_res < eps
        self.convergence.all = bool(conv)
        self.iteration = it


def calculate_iteration(admm: ADMM) -> None:
    """One ADMM iteration: all sub-problems, dual update, stop test."""
    done, _ = admm.engine.iterate(1)
    if done and admm.record:
        admm.results.append(admm._fetch_result())
        lam, mu, rho = admm.engine.get_duals()
        admm.lambdas.append(lam)
        admm.mues.append(mu)
        admm.rhos.append(rho)
    admm._after(done)


def run(admm: ADMM, chunk: int = 64) -> ADMM:
    """run!(admm): iterate until Convergence.all (or the optional cap `max_iters`)."""
    cap = admm.params.max_iters
    while not admm.convergence.all and not (cap > 0 and admm.iteration > cap):
        if admm.record:
            calculate_iteration(admm)
        else:
            done, _ = admm.engine.iterate(chunk)
            admm._after(done)
            if done == 0:
                break
    if not admm.record:
        admm.results = [admm._fetch_result()]
        admm.lambdas = [admm.engine.get_duals_used()[0], admm.engine.get_duals()[0]]
        admm.mues = [admm.engine.get_duals_used()[1], admm.engine.get_duals()[1]]
        admm.rhos = [admm.engine.get_duals_used()[2], admm.engine.get_duals()[2]]
    return admm


def get_nodal_price(admm: ADMM, iteration: Optional[int] = None) -> np.ndarray:
    """lambda_t + sum_l (mu + rho)[l,t] ptdf[l,:] with the duals of `iteration` (1-based, default
    admm.iteration = the duals the last solve used, as src/opf_admm_decentral.jl:9 does)."""
    if iteration is None or not admm.record:
        return admm.engine.get_nodal_price(0)
    lam, mu, rho = admm.lambdas[iteration - 1], admm.mues[iteration - 1], admm.rhos[iteration - 1]
    return lam[None, :] + admm.ptdf.T @ (mu + rho)


def export_results(admm: ADMM, filename: str, parent_dir: str = "results/") -> None:
    """The three long-format CSVs of src/helpers/output.jl:1-85, byte-compatible in layout:
    <name>_duals.csv (iteration,dual,timestep,line,value; duals lambda, rho, mue; row i = the dual
    USED in iteration i), <name>_generators.csv, <name>_storages.csv (charge before discharge)."""
    if not admm.record:
        raise ValueError("export_results needs the iteration history: build ADMM(..., record=True)")
    os.makedirs(parent_dir, exist_ok=True)
    n_it = min(admm.iteration, len(admm.results))
    T, L = len(admm.T), len(admm.L)
    with open(os.path.join(parent_dir, filename + "_duals.csv"), "w") as f:
        f.write("iteration,dual,timestep,line,value\n")
        for name, hist in (("lambda", admm.lambdas), ("rho", admm.rhos), ("mue", admm.mues)):
            for i in range(1, n_it + 1):
                for t in range(T):
                    if name == "lambda":
                        f.write(f"{i},lambda,{t + 1},,{float(hist[i - 1][t])!r}\n")
                    else:
                        for l in range(L):
                            f.write(f"{i},{name},{t + 1},{l + 1},{float(hist[i - 1][l, t])!r}\n")
    with open(os.path.join(parent_dir, filename + "_generators.csv"), "w") as f:
        f.write("iteration,generator,timestep,generation\n")
        for g in admm.generators:
            for i in range(1, n_it + 1):
                r = admm.results[i - 1].of(g)
                for t in range(T):
                    f.write(f"{i},{g.name},{t + 1},{float(r.generation[t])!r}\n")
    with open(os.path.join(parent_dir, filename + "_storages.csv"), "w") as f:
        f.write("iteration,storage,timestep,charge,discharge\n")
        for s in admm.storages:
            for i in range(1, n_it + 1):
                r = admm.results[i - 1].of(s)
                for t in range(T):
                    f.write(f"{i},{s.name},{t + 1},{float(r.charge[t])!r},{float(r.discharge[t])!r}\n")
