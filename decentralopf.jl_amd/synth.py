"""Seed-stable synthetic N-agent x T-timestep cases (BASELINE.json configs 2-5).

The reference ships one case (src/cases/three_node.jl); the synthetic grids keep its value ranges:
generators mc ~ U_int[1,60], pmax ~ U_int[10,300] (three_node.jl:13-16); storages mc ~ U_int[1,3],
pmax ~ U_int[5,20], emax = 2*pmax (ratio of three_node.jl:20); integer demand
d_tot[t] = round(0.55 * sum(pmax) * (1 + 0.3 sin(2 pi t / 24))) split over the nodes by Dirichlet(1)
weights; agents are assigned to nodes uniformly at random and sorted by node.
Copper plate = one node, no lines. Network cases use a seeded random connected graph (NOT IEEE
data, which is not available offline — labelled "synthetic-118" where BASELINE.json says IEEE-118).
"""
from __future__ import annotations

import numpy as np

from .network import PackedProblem

SEED = 20220720


def _ptdf_from_edges(N, frm, to, sus, slack=0):
    L = len(frm)
    inc = np.zeros((L, N))
    inc[np.arange(L), frm] = 1.0
    inc[np.arange(L), to] = -1.0
    B = np.diag(sus.astype(np.float64))
    Bn = inc.T @ B @ inc
    keep = [i for i in range(N) if i != slack]
    Binv = np.zeros((N, N))
    Binv[np.ix_(keep, keep)] = np.linalg.inv(Bn[np.ix_(keep, keep)])
    return (B @ inc) @ Binv


def synthetic_case(n_gen: int, n_sto: int, T: int, *, N: int = 1, L: int = 0, seed: int = SEED,
                   fmax_factor: float = 1.5, fmax_min: float = 50.0) -> PackedProblem:
    rng = np.random.default_rng(seed)
    gen_mc = rng.integers(1, 61, size=n_gen).astype(np.float64)
    gen_pmax = rng.integers(10, 301, size=n_gen).astype(np.float64)
    sto_mc = rng.integers(1, 4, size=n_sto).astype(np.float64)
    sto_pmax = rng.integers(5, 21, size=n_sto).astype(np.float64)
    sto_emax = 2.0 * sto_pmax
    gen_node = np.sort(rng.integers(0, N, size=n_gen)).astype(np.int32)
    sto_node = np.sort(rng.integers(0, N, size=n_sto)).astype(np.int32)
    t = np.arange(1, T + 1)
    d_tot = np.round(0.55 * gen_pmax.sum() * (1.0 + 0.3 * np.sin(2.0 * np.pi * t / 24.0)))
    w = rng.dirichlet(np.ones(N)) if N > 1 else np.ones(1)
    demand = np.round(np.outer(w, d_tot))
    if L > 0:
        if N < 2 or L < N - 1:
            raise ValueError("a connected graph needs L >= N-1 >= 1")
        # random spanning tree + extra distinct edges
        perm = rng.permutation(N)
        edges = set()
        frm, to = [], []
        for i in range(1, N):
            a, b = int(perm[i]), int(perm[rng.integers(0, i)])
            edges.add((min(a, b), max(a, b)))
            frm.append(a)
            to.append(b)
        guard = 0
        while len(frm) < L and guard < 100 * L:
            guard += 1
            a, b = (int(v) for v in rng.integers(0, N, size=2))
            if a == b or (min(a, b), max(a, b)) in edges:
                continue
            edges.add((min(a, b), max(a, b)))
            frm.append(a)
            to.append(b)
        if len(frm) < L:
            raise ValueError("graph too dense for distinct edges")
        frm, to = np.asarray(frm), np.asarray(to)
        sus = rng.integers(1, 6, size=L)
        ptdf = _ptdf_from_edges(N, frm, to, sus, slack=0)
        # merit-order copper-plate dispatch -> flows -> capacities
        order = np.argsort(gen_mc, kind="stable")
        pm_sorted = gen_pmax[order]
        before = np.cumsum(pm_sorted) - pm_sorted                     # capacity cheaper than each generator
        take = np.clip(d_tot[None, :] - before[:, None], 0.0, pm_sorted[:, None])       # (G, T)
        inj = -demand.copy()
        np.add.at(inj, gen_node[order], take)
        flow = ptdf @ inj
        f_max = np.maximum(fmax_min, np.ceil(fmax_factor * np.abs(flow).max(axis=1)))
    else:
        ptdf = np.zeros((0, N))
        f_max = np.zeros(0)
    return PackedProblem(
        N=N, L=L, T=T, demand=demand, ptdf=ptdf, f_max=f_max, gen_mc=gen_mc, gen_pmax=gen_pmax,
        gen_node=gen_node, sto_mc=sto_mc, sto_pmax=sto_pmax, sto_emax=sto_emax, sto_node=sto_node,
        meta=dict(kind="synthetic", seed=seed, n_gen=n_gen, n_sto=n_sto, T=T, N=N, L=L))


# BASELINE.json `configs`, by index (0 is the shipped three-node case, see network.three_node_case)
def baseline_config(i: int, *, scale: float = 1.0) -> PackedProblem:
    if i == 1:      # "synthetic 1k generators + 100 storages, 24 timesteps"
        return synthetic_case(int(1000 * scale), int(100 * scale), 24)
    if i == 2:      # "synthetic 50k agents, 96 timesteps with storage SoC coupling" (10:1 split)
        a = int(50000 * scale)
        return synthetic_case(a - a // 11, a // 11, 96)
    if i == 3:      # "IEEE-118-bus PTDF topology scaled to 100k agents x 168 timesteps" (synthetic graph)
        a = int(100000 * scale)
        return synthetic_case(a - a // 11, a // 11, 168, N=118, L=186)
    if i == 4:      # "1M agents x 24 timesteps"
        a = int(1000000 * scale)
        return synthetic_case(a - a // 11, a // 11, 24)
    raise ValueError("config index 1..4")
