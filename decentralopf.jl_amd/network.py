"""Network element types, PTDF set-up and the shipped three-node case (host side, CPU by design).

Mirrors, for the Python host (Julia is not in this image; the Julia shim is julia/DecentralOPFHip.jl):
  Node / Generator / Storage / Line      src/structures/network_elements.jl:1-30
  calculate_ptdf(nodes, lines)           src/helpers/ptdf.jl:1-41   (one-off O(N^3) set-up, stays on host)
  three_node_case()                      src/cases/three_node.jl:1-21
Numeric struct fields are Int in the reference and are promoted to Float64 when packed.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np


@dataclass(eq=False)
class Node:
    name: str
    demand: List[int]
    slack: bool


@dataclass(eq=False)
class Generator:
    name: str
    marginal_costs: int
    max_generation: int
    plot_color: str
    node: Node


@dataclass(eq=False)
class Storage:
    name: str
    marginal_costs: int
    max_power: int
    max_level: int
    plot_color: str
    node: Node


@dataclass(eq=False)
class Line:
    name: str
    from_: Node          # `from` is a Python keyword; Julia field: from
    to: Node
    max_capacity: int
    susceptance: int


def calculate_ptdf(nodes: Sequence[Node], lines: Sequence[Line]) -> np.ndarray:
    """PTDF (L x N). Incidence from=+1, to=-1; Bl = B A; Bn = A' B A; invert without the first
    slack node's row/column; PTDF = Bl * B_inv (src/helpers/ptdf.jl:1-41)."""
    N, L = len(nodes), len(lines)
    if L == 0:
        return np.zeros((0, N))
    slack = next((i for i, n in enumerate(nodes) if n.slack), None)
    if slack is None:
        raise ValueError("no slack node")
    idx = {id(n): i for i, n in enumerate(nodes)}
    inc = np.zeros((L, N))
    for l, line in enumerate(lines):
        inc[l, idx[id(line.from_)]] = 1.0
        inc[l, idx[id(line.to)]] = -1.0
    B = np.diag([float(line.susceptance) for line in lines])
    Bl = B @ inc
    Bn = inc.T @ B @ inc
    keep = [i for i in range(N) if i != slack]
    B_inv = np.zeros((N, N))
    B_inv[np.ix_(keep, keep)] = np.linalg.inv(Bn[np.ix_(keep, keep)])
    return Bl @ B_inv


def three_node_case():
    """The shipped case study: 3 nodes, 3 lines, 4 generators, 1 battery, T = 2."""
    n1 = Node("N1", [10, 250], False)
    n2 = Node("N2", [50, 70], False)
    n3 = Node("N3", [120, 200], True)
    nodes = [n1, n2, n3]
    lines = [Line("L1", n2, n1, 20, 1), Line("L2", n3, n1, 45, 1), Line("L3", n2, n3, 70, 2)]
    generators = [
        Generator("pv", 3, 80, "yellow", n1),
        Generator("wind", 4, 120, "lightblue", n2),
        Generator("coal", 30, 300, "brown", n3),
        Generator("gas", 50, 120, "grey", n1),
    ]
    storages = [Storage("battery", 1, 10, 20, "purple", n1)]
    return nodes, lines, generators, storages


@dataclass
class PackedProblem:
    """Flat SoA view of a case in the layouts include/dopf.h documents."""
    N: int
    L: int
    T: int
    demand: np.ndarray      # (N, T)
    ptdf: np.ndarray        # (L, N)
    f_max: np.ndarray       # (L,)
    gen_mc: np.ndarray
    gen_pmax: np.ndarray
    gen_node: np.ndarray    # int32, 0-based
    sto_mc: np.ndarray
    sto_pmax: np.ndarray
    sto_emax: np.ndarray
    sto_node: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def G(self):
        return int(self.gen_mc.size)

    @property
    def S(self):
        return int(self.sto_mc.size)

    def engine_kwargs(self):
        """Arguments of _capi.Engine in the C ABI's memory order (column-major matrices)."""
        return dict(
            N=self.N, L=self.L, T=self.T,
            demand=np.asarray(self.demand, dtype=np.float64).reshape(self.N, self.T).T.ravel(),
            ptdf=np.asarray(self.ptdf, dtype=np.float64).reshape(self.L, self.N).T.ravel(),
            f_max=self.f_max, gen_mc=self.gen_mc, gen_pmax=self.gen_pmax, gen_node=self.gen_node,
            sto_mc=self.sto_mc, sto_pmax=self.sto_pmax, sto_emax=self.sto_emax, sto_node=self.sto_node)

    def shard(self, rank: int, world: int) -> "PackedProblem":
        """Contiguous slice of the agent lists for one rank (network data replicated)."""
        def cut(n):
            base, rem = divmod(n, world)
            lo = rank * base + min(rank, rem)
            return lo, lo + base + (1 if rank < rem else 0)
        g0, g1 = cut(self.G)
        s0, s1 = cut(self.S)
        return PackedProblem(
            N=self.N, L=self.L, T=self.T, demand=self.demand, ptdf=self.ptdf, f_max=self.f_max,
            gen_mc=self.gen_mc[g0:g1], gen_pmax=self.gen_pmax[g0:g1], gen_node=self.gen_node[g0:g1],
            sto_mc=self.sto_mc[s0:s1], sto_pmax=self.sto_pmax[s0:s1], sto_emax=self.sto_emax[s0:s1],
            sto_node=self.sto_node[s0:s1],
            meta=dict(self.meta, rank=rank, world=world, gen_range=(g0, g1), sto_range=(s0, s1),
                      n_agents_global=self.G + self.S))


def pack(nodes: Sequence[Node], generators: Sequence[Generator], storages: Sequence[Storage],
         lines: Sequence[Line]) -> PackedProblem:
    """What ADMM(...) derives from the element vectors (src/structures/admm.jl:28-60)."""
    idx = {id(n): i for i, n in enumerate(nodes)}
    T = len(nodes[0].demand)
    for n in nodes:
        if len(n.demand) != T:
            raise ValueError("all nodes need a demand series of the same length")
    f64 = lambda xs: np.asarray(list(xs), dtype=np.float64)
    i32 = lambda xs: np.asarray(list(xs), dtype=np.int32)
    return PackedProblem(
        N=len(nodes), L=len(lines), T=T,
        demand=np.asarray([n.demand for n in nodes], dtype=np.float64).reshape(len(nodes), T),
        ptdf=calculate_ptdf(nodes, lines),
        f_max=f64(l.max_capacity for l in lines),
        gen_mc=f64(g.marginal_costs for g in generators),
        gen_pmax=f64(g.max_generation for g in generators),
        gen_node=i32(idx[id(g.node)] for g in generators),
        sto_mc=f64(s.marginal_costs for s in storages),
        sto_pmax=f64(s.max_power for s in storages),
        sto_emax=f64(s.max_level for s in storages),
        sto_node=i32(idx[id(s.node)] for s in storages))
